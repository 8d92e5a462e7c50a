run() { echo -n "$* : "; "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fps %7.1f  blend %.4f ms' % (d['value'], d['roofline']['avg_launch_ms']))"; }
for i in 1 2; do
run python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5
run env GSR_BENCH_PREHEAT=1 python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5
done
rocm-smi --showclocks 2>/dev/null | head -20

run() { echo -n "$* : "; "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fps %7.1f  blend %.4f ms  k1 %.4f' % (d['value'], d['roofline']['avg_launch_ms'], (d.get('roofline_preprocess') or {}).get('avg_launch_ms', 0)))"; }
for i in 1 2; do
run python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5
run python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 40
run python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 200
run python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 5
run env GSR_ORDER_KEEP=4 python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5
run python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --time-every 1000
done

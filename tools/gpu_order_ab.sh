for rep in 1 2; do for cfg in C4 C3 C5 C2 C1; do for o in 1 2 3; do
python bench.py --no-cpu-baseline --config $cfg --tile-order $o --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('stages_ms_last_frame',{})
print('$cfg order $o fps %7.1f  blend %.3f ' % (d['value'], d['roofline']['avg_launch_ms']))"
done; done; done

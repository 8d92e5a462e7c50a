#!/bin/bash
# Usage (the first node with >= 2 MI355X this repository meets; repo root):   bash tools/first_multi_gpu_node.sh [max_gpus]
# One command for the whole scaling curve: the RCCL test, the N = 1 point through the multi-GPU door, then N = 2 / 4 / 8 on BASELINE C4
# (1080p) and C5 (4K) in both shard layouts, through ONE process (gsr_multi_*: a worker thread per rank, grouped ncclSend / ncclRecv) and
# through one process per GPU (torch.distributed.run, gsr_comm_*).  Every bench line checks its last stitched frame bit for bit against the
# unsharded frame and refuses a value when the communicator does not span the N ranks.  Output: gpurun_out/multi/*.json + one table.
set -u
MAXG=${1:-8}
NG=$(python -c "import torch; print(torch.cuda.device_count())")
[ "$NG" -lt "$MAXG" ] && MAXG=$NG
OUT=gpurun_out/multi
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $NG GPU(s) visible, running up to $MAXG"
python -m pytest tests -q -m gpu -k "rccl or multi_gpu" 2>&1 | tail -3 | tee $OUT/pytest_rccl.txt
python bench.py --gpus 1 --via-multi --no-cpu-baseline --no-extra-legs --steps 200 --warmup 20 > $OUT/c4_n1_via_multi.json 2> $OUT/c4_n1_via_multi.err
python bench.py --gpus 1 --no-cpu-baseline --no-extra-legs --steps 200 --warmup 20 > $OUT/c4_n1.json 2> $OUT/c4_n1.err
for cfg in C4 C5; do
  python bench.py --config $cfg --gpus 1 --no-cpu-baseline --no-extra-legs --steps 200 --warmup 20 > $OUT/${cfg}_n1_single.json 2>/dev/null
  for n in 2 4 8; do
    [ $n -gt $MAXG ] && continue
    for layout in 1 0; do
      python bench.py --config $cfg --gpus $n --shard-layout $layout --no-cpu-baseline --steps 200 --warmup 20 \
          > $OUT/${cfg}_n${n}_layout${layout}_one_process.json 2> $OUT/${cfg}_n${n}_layout${layout}_one_process.err
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n + 10 * layout)) \
          bench.py --config $cfg --gpus $n --shard-layout $layout --no-cpu-baseline --no-extra-legs --steps 200 --warmup 20 \
          > $OUT/${cfg}_n${n}_layout${layout}_per_gpu.json 2> $OUT/${cfg}_n${n}_layout${layout}_per_gpu.err
    done
  done
done
python - $OUT <<'PY'
import glob, json, os, sys
rows = []
for p in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        rows.append((os.path.basename(p), "no line")); continue
    link = d.get("gather_links") or {}
    rows.append((os.path.basename(p), "n %s  fps %s  ms %.4f  gather_ms %s  GB/s per link %s  bit-identical %s  rccl ranks %s%s" % (
        d.get("n_gpus"), ("%.0f" % d["value"]) if d.get("value") else "NONE", d.get("ms_per_step") or 0.0,
        ("%.4f" % d["gather_ms"]) if d.get("gather_ms") else "-", ("%.1f" % link["GBps_per_link"]) if link.get("GBps_per_link") else "-",
        d.get("sharded_frame_bit_identical", d.get("timed_frame_bit_identical")), d.get("rccl_comm_count", "-"),
        ("  ERROR " + d["error"]) if d.get("error") else "")))
w = max(len(r[0]) for r in rows)
for name, text in rows:
    print(name.ljust(w), text)
PY

#!/bin/bash
# round-5 batch: EGL probe; ASan-on-GPU repro with output; R1 timeline + tile stats; adaptive-batch variants of k_blend on T1 / S1 / R1 / C4 / C5
set -u
mkdir -p gpurun_out
python tools/probe_egl.py > gpurun_out/egl_probe.txt 2>&1
V=houdini-gsplat-renderer_amd/variants
( export LD_PRELOAD=$(cat $V/asan.preload) GSR_LIBRARY=$PWD/$V/libgsplat_hip_asan.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
  timeout 600 python -m pytest -x -q -p no:cacheprovider tests/test_gpu_parity.py -m gpu -k "test_renderer_shim_frame_protocol" > gpurun_out/asan_gpu_repro.txt 2>&1; echo "rc=$?" >> gpurun_out/asan_gpu_repro.txt )
bash tools/gpu_timeline.sh median --no-extra-legs --config R1 > gpurun_out/tl_r1.txt 2>&1
python tools/tile_hits.py R1 > gpurun_out/tile_hits_r1.txt 2>&1
for cfg in "--config T1" "--config S1" "--config R1" "" "--config C5"; do
  bash tools/gpu_ab_blend.sh "$cfg" orig grow1 grow2 grow4 orig >> gpurun_out/ab_grow.txt 2>&1
done
tail -5 gpurun_out/egl_probe.txt; tail -30 gpurun_out/asan_gpu_repro.txt; cat gpurun_out/tl_r1.txt gpurun_out/tile_hits_r1.txt gpurun_out/ab_grow.txt

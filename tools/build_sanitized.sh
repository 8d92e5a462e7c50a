#!/bin/bash
# Usage: bash tools/build_sanitized.sh [asan|tsan|all]      (build container or GPU box; no GPU needed to build)
# Sanitizer builds of libgsplat_hip.so's HOST code (the registry borrows raw pointers exactly as the reference does,
# /root/reference/gsplat_plugin/src/GSplatRenderer.C:277-284, lifetime src/GR_GSplat.C:63-70; gsr_multi.cpp runs a worker thread per rank):
#   variants/libgsplat_hip_asan.so   every host translation unit with -fsanitize=address,undefined (device code untouched)
#   variants/libgsplat_hip_tsan.so   the same with -fsanitize=thread
# Run a test against one:  LD_PRELOAD=$(cat variants/asan.preload) ASAN_OPTIONS=detect_leaks=0 GSR_LIBRARY=.../libgsplat_hip_asan.so python -m pytest ...
# (tests/test_sanitizers.py does exactly that).
set -e
cd "$(dirname "$0")/.."
P=houdini-gsplat-renderer_amd
V=$P/variants
mkdir -p $V
WHAT=${1:-all}
CLANG_RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)")
SRC="$P/csrc/gsr_api.hip $P/csrc/gsr_multi.cpp $P/csrc/GSplatRenderer.cpp $P/csrc/gsplat_ingest.cpp"
COMMON="--offload-arch=gfx950 -O1 -g -fno-omit-frame-pointer -std=c++17 -ffp-contract=off -fPIC -shared -fno-gpu-sanitize -shared-libsan -Wno-unused-result"
if [ "$WHAT" = asan ] || [ "$WHAT" = all ]; then
  hipcc $COMMON -fsanitize=address,undefined -fno-sanitize-recover=undefined -o $V/libgsplat_hip_asan.so $SRC -ldl -pthread -Wl,-rpath,$CLANG_RT
  echo "$CLANG_RT/libclang_rt.asan-x86_64.so" > $V/asan.preload
fi
if [ "$WHAT" = tsan ] || [ "$WHAT" = all ]; then
  hipcc $COMMON -fsanitize=thread -o $V/libgsplat_hip_tsan.so $SRC -ldl -pthread -Wl,-rpath,$CLANG_RT
  echo "$CLANG_RT/libclang_rt.tsan-x86_64.so" > $V/tsan.preload
fi
ls -la $V/*san*

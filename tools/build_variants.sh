#!/bin/bash
# Usage (build container): bash tools/build_variants.sh name1:"-DFOO=1 -DBAR=2" name2:"" ...
# Builds houdini-gsplat-renderer_amd/variants/libgsplat_hip_<name>.so for A/B runs (tools/gpu_ab.sh).
set -e
cd "$(dirname "$0")/.."
P=houdini-gsplat-renderer_amd
mkdir -p $P/variants
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  [ "$defs" = "$spec" ] && defs=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $defs \
      -o $P/variants/libgsplat_hip_$name.so $P/csrc/gsr_api.hip $P/csrc/gsr_multi.cpp $P/csrc/GSplatRenderer.cpp $P/csrc/gsplat_ingest.cpp -ldl -pthread \
      > /tmp/variant_$name.log 2>&1 &
done
wait
grep -l "error" /tmp/variant_*.log 2>/dev/null | xargs -r tail -5
ls -la $P/variants/

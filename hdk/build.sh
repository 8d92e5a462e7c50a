#!/bin/sh
# Builds the Houdini DSO of the GSplat plugin over libgsplat_hip.  Needs a Houdini installation: `source houdini_setup`
# first (sets $HFS and puts hcustom on the PATH) -- this repository's build and test machines have none, so the files in
# this directory have never been BUILT (they are type-checked against mock HDK headers: tests/test_hdk_glue.py); they are the glue
# a maintainer drops into the reference tree.
#   REF  = checkout of rubendhz/houdini-gsplat-renderer (its gsplat_plugin/ directory)
#   REPO = this repository (include/, houdini-gsplat-renderer_amd/libgsplat_hip.so built by __graft_entry__.build())
set -e
if [ -z "$HFS" ] || ! command -v hcustom >/dev/null 2>&1; then
    echo "hdk/build.sh: \$HFS is not set / hcustom not found -- source houdini_setup first" >&2
    exit 2
fi
REPO="$(cd "$(dirname "$0")/.." && pwd)"
REF="${REF:?set REF to the reference's gsplat_plugin directory}"
# the reference's own line is  hcustom -I include -I shaders gsplat_plugin.C  (scripts/houdini_env.sh, .vscode/tasks.json)
cd "$REF"
# (this repo's directories FIRST: hdk/GR_GSplat.h and include/GSplatRenderer.h shadow the reference's headers of the same names)
hcustom -I "$REPO/hdk" -I "$REPO/include" -I include -I . \
        -L "$REPO/houdini-gsplat-renderer_amd" -l gsplat_hip -l amdhip64 -l GL \
        "$REPO/hdk/gsplat_plugin_hip.C"
echo "at run time: LD_LIBRARY_PATH must reach $REPO/houdini-gsplat-renderer_amd (libgsplat_hip.so) and ROCm's lib directory"

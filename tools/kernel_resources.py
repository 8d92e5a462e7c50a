#!/usr/bin/env python3
"""Compile csrc/gsr_api.hip for gfx950 with -save-temps (into a scratch directory) and print, per kernel, the
registers / LDS / scratch the compiler allocated.  `--isa NAME` also dumps that kernel's assembly; `-D...` flags are
passed through.  No GPU needed (hipcc cross-compiles).   python tools/kernel_resources.py [--isa k_blend] [-DBL_ROUND=64]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "houdini-gsplat-renderer_amd", "csrc", "gsr_api.hip")


def main():
    isa = None
    extra = []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--isa":
            isa = args.pop(0)
        else:
            extra.append(a)
    tmp = tempfile.mkdtemp(prefix="gsr_res_")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c",
           "-save-temps=obj", "-o", os.path.join(tmp, "gsr_api.o"), SRC] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
    if r.returncode:
        sys.exit(r.stdout + r.stderr)
    asm = open(os.path.join(tmp, "gsr_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for blk in re.findall(r"- \.agpr_count.*?\.wavefront_size:\s+\d+", asm, re.S):
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} "
              f"scratch {g('private_segment_fixed_size'):>4s}")
    if isa:
        m = re.search(r"^(_Z\w*" + re.escape(isa) + r"\w*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", asm, re.S | re.M)   # (a kernel may hold several s_endpgm)
        if m:
            out = os.path.join(tmp, isa + ".s")
            open(out, "w").write(m.group(0))
            print("ISA of", m.group(1), "->", out)


if __name__ == "__main__":
    main()

"""CPU checks of the tools the GPU tests lean on (no GPU: nothing is rendered)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fuzz_scripts_are_drawn_up_front_and_reproducible():
    """tools/fuzz_parity.py draws an iteration's whole script before anything runs: the same seed gives the same scripts whatever
    the GPU answers, in every mode; and a script only holds what the engine accepts"""
    fz = _load("fuzz_parity")
    E = fz.E
    for mode in ("0", "2"):
        a = [fz.make_script(np.random.default_rng(5), it, False, mode == "2")["desc"] for it in range(1)]
        b = [fz.make_script(np.random.default_rng(5), it, False, mode == "2")["desc"] for it in range(1)]
        assert a == b
    rng = np.random.default_rng(9)
    kinds = set()
    for it in range(60):
        sc = fz.make_script(rng, it, False, it % 2 == 1)
        d = sc["desc"]
        kinds.add((d["proj"], bool(d["multi"]), d["depth"], d["dev_target"], bool(d["parts"])))
        assert len(sc["cams"]) == 8 and all(c.width == d["w"] and c.height == d["h"] for c in sc["cams"])
        assert 0 <= d["shard"][0] < d["shard"][1] and d["shard"][2] in (0, 1)
        assert set(sc["opts"]) <= set(fz.DEFAULTS)
        if d["multi"]:
            assert E.OPT_DEFERRED_CHECK not in sc["opts"] and not d["dev_target"] and d["shard"][:2] == (0, 1)
        if d["parts"]:
            assert all(0 < c < d["n"] for c in d["parts"]) and d["parts"] == sorted(d["parts"])
        for k, fl in sc["flips"].items():
            assert 0 <= k < 8 and fl[0] in ("opt", "shard")
    assert len(kinds) >= 10          # (the generator really varies what it is meant to vary)


def test_frame_timeline_picks_the_median_frame(tmp_path):
    """tools/frame_timeline.py on a synthetic kernel trace: 'median' = the frame of median period, not one with a hiccup"""
    import subprocess, sys
    rows = ["Kernel_Name,Start_Timestamp,End_Timestamp"]
    t = 0
    periods = [250, 250, 900, 250, 260, 250, 250, 255, 250]     # ns x 1000: one frame carries a hiccup
    for p in periods:
        rows.append(f"k_preprocess(unsigned int),{t},{t + 40000}")
        rows.append(f"k_blend<false>(GsrBlendArgs),{t + 40000},{t + 170000}")
        t += p * 1000
    rows.append(f"k_preprocess(unsigned int),{t},{t + 40000}")
    f = tmp_path / "trace.csv"
    f.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "frame_timeline.py"), str(f), "median"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "frame period 250.0 us" in out.stdout


def test_pmc_summary_drops_cold_dispatches(tmp_path):
    """tools/pmc_summary.py on a synthetic counter CSV: four warm-up dispatches five times as long (and as heavy) as the steady ones
    must not reach the per-kernel mean, and the summary says how many rows it dropped"""
    import json, subprocess, sys
    hdr = "Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"
    for name, steady, cold in (("pass1.csv", 1000.0, 5000.0), ("pass2.csv", 200.0, 900.0)):
        rows, t = [hdr], 0
        counter = "FETCH_SIZE" if name == "pass1.csv" else "WRITE_SIZE"
        for i in range(24):
            dur, val = (200000, cold) if i < 4 else (40000 + (i % 3) * 500, steady)
            rows.append(f"k_preprocess(unsigned int),{counter},{val},{t},{t + dur}")
            t += dur + 1000
        (tmp_path / name).write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "dropped 8 of 48 counter rows" in out.stdout
    t = json.load(open(tmp_path / "pmc_traffic.json"))["k_preprocess"]
    assert t["dispatches"] == 20 and t["cold_rows_dropped"] == 8
    assert abs(t["hbm_bytes_per_launch"] - (2 * 1000.0 + 200.0) * 1024.0) < 1e-6


def test_frame_kernels_keep_their_register_and_scratch_budgets():
    """DESIGN.md section 4 states, per frame kernel, the registers / LDS / scratch the compiler allocates (tools/kernel_resources.py
    prints them from a cross-compile: no GPU).  A checked claim: no frame kernel uses scratch except the shading K1 (two dwords, stored
    once and reloaded per workgroup-iteration: 12 bytes at its 80-register, six-waves-per-SIMD budget); the blend kernels keep the
    occupancy they were tuned for (plain <= 72 registers, depth-tested <= 80: six waves per SIMD either way)."""
    import re
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    rows = {}
    for ln in res.stdout.splitlines():
        m = re.match(r"(\S+)\s+vgpr\s+(\d+)\s+sgpr\s+(\d+)\s+lds\s+(\d+)\s+scratch\s+(\d+)", ln)
        if m:
            rows[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
    def find(prefix):
        hit = [v for k, v in rows.items() if k.startswith(prefix)]
        assert len(hit) == 1, (prefix, [k for k in rows if k.startswith(prefix)])
        return hit[0]
    frame_kernels = ["_Z14k_cluster_cullILb0E", "_Z14k_cluster_cullILb1E", "_Z17k_preprocess_lazyjj", "_Z18k_preprocess_depth", "_Z23k_preprocess_lazy_depth", "_Z15k_depth_pyramid",
                     "_Z13k_radix_localI15HIP", "_Z11k_bin_countILi2E", "_Z11k_bin_placeILi2ELb0E", "_Z11k_bin_placeILi2ELb1E", "_Z11k_scan_rows", "_Z7k_blendILb0E", "_Z7k_blendILb1E",
                     "_Z11k_tile_pass", "_Z11k_frame_end", "_Z17k_frame_end_order", "_Z10k_slab_mid", "_Z15k_colour_prefix", "_Z6k_packILb1E"]
    for k in frame_kernels:
        vg, sg, lds, scratch = find(k)
        assert scratch == 0, (k, scratch)
    vg, sg, lds, scratch = find("_Z12k_preprocessjj")
    assert scratch <= 12 and vg <= 80, (vg, scratch)
    assert find("_Z7k_blendILb0E")[0] <= 72 and find("_Z7k_blendILb1E")[0] <= 80
    assert find("_Z7k_blendILb0E")[2] <= 20 * 1024 and find("_Z7k_blendILb1E")[2] <= 22 * 1024       # LDS: seven workgroups of either fit a CU's 160 KB

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

"""GPU box: replay ONE fuzz_parity iteration (seed, index) with the round-6 depth machinery switched off piece by piece
   python tools/depth_bisect.py SEED IT"""
import importlib.util, os, sys
import numpy as np
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join("tools", "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
E = fz.E
seed, it = int(sys.argv[1]), int(sys.argv[2])
def script():
    rng = np.random.default_rng(seed)         # (the same random sequence as fuzz_parity.py's main loop)
    for k in range(it + 1):
        sc = fz.make_script(rng, k, False, False)
    return sc
for name, flags in (("all on", 0), ("no blend classification", 32)):
    sc = script()
    sc["opts"][E.OPT_DEBUG_FLAGS] = flags
    dut = E.MultiEngine([0] * sc["multi"], E.TRANSPORT_COPY) if sc["multi"] else E.Engine(0)
    try:
        if not sc["multi"]:
            pass
        frames, bad = fz.execute(sc, dut)
        print(f"{name:28s}", "ok" if bad is None else bad, flush=True)
    finally:
        dut.close()

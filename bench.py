#!/usr/bin/env python3
"""bench.py -- frames/sec of the GSplat hot path on MI355X (BASELINE.json metric).

A "step" is one full frame of the hot path (SH -> projection -> depth sort -> tile
binning -> front-to-back compositing [-> RCCL gather + stitch when N>1]) with a new
camera position every step (3 deg orbit), so every frame re-sorts -- the reference
re-sorts on any camera translation (src/GSplatRenderer.C:165-186).  Inputs are
resident in HBM before the timed region (uploaded once, as the reference stages
textures once).

  python bench.py --gpus 1 --steps 60 --warmup 5
  python bench.py --gpus N ...          (no launcher: ONE process drives the N GPUs through gsr_multi_* -- the form the
                                         reference's single draw thread implies, src/DM_GSplatHook.C:30-39; RCCL gather)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W          (one process per GPU, gsr_comm_*)
Fewer than N GPUs visible: one JSON line with an "error" field and a non-zero exit code (GSR_BENCH_ALLOW_DUP=1 lets the
single-process form put several ranks on one GPU over the COPY transport: a functional test, flagged in the line).

Workload at every N: BASELINE config C4 -- 6,000,000 synthetic splats (SH degree 3,
seed 1004), 1920x1080 -- the scene the ">= 60 fps on one MI355X" target is quoted on.
N>1 shards tile rows (row r -> rank r % N) and gathers the band images to rank 0:
total work is fixed, so "scaling" is "strong".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def measured_hbm_peak():
    """what this GPU's HBM delivers to a 16-byte-per-lane streaming kernel (tools/ubench_hbm.hip, built by __graft_entry__.build()):
    the second denominator of the rooflines.  None when the binary is missing."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench_hbm")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1]
        return json.loads(out)
    except Exception:
        return None


def kernel_source_sha() -> str:
    """fingerprint of the kernel sources: ties profiles/pmc_traffic.json to the kernels it was collected with"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "houdini-gsplat-renderer_amd", "csrc")
    import re
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".h", ".hip", ".cpp")):
            text = open(os.path.join(csrc, name), "r", encoding="utf-8", errors="replace").read()
            # comments and white space do not make a different kernel (the sources hold no string literal with "//" or "/*" in it
            # that matters to the device code)
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = re.sub(r"//[^\n]*", "", text)
            text = re.sub(r"\s+", " ", text)
            h.update(name.encode())
            h.update(text.encode())
    return h.hexdigest()[:16]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C4", help="BASELINE config (C1..C5); C4 is the headline workload.  T1 (a landscape under "
                    "open sky) and S1 (a thin wall) are extra scenes with silhouettes")
    ap.add_argument("--splats", type=int, default=None, help="override the splat count (debug only)")
    ap.add_argument("--ply", default=None, help="an INRIA 3D-Gaussian-Splatting PLY (the kind of file the reference's example scene imports, "
                    "hip/GSplatPlugin_simpleScene_v001.hip): rendered instead of --config, through the scene's activations (ply.py), 1920x1080, "
                    "on an orbit fitted to the cloud's bounds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="wall budget of the cpu_baseline leg (23 frames of a sample sized to fit)")
    ap.add_argument("--no-swizzle", action="store_true", help="disable the XCD-aware tile mapping (A/B)")
    ap.add_argument("--morton", action="store_true", help="experiment: hand the splats over in Morton order of their positions")
    ap.add_argument("--time-every", type=int, default=0, help="HIP events around the blend kernel of every N-th frame of the timed region; "
                    "0 = 4 (8 from 64 steps on), at least five launches timed "
                    "(each pair idles the queue ~12 us; the roofline's launch time is the mean over the sampled launches)")
    ap.add_argument("--cull", type=int, default=1, help="GSR_OPT_OCCLUSION_CULL: 0 = no occlusion culling, 1 = the library's policy (default), 2 = against the previous "
                    "frame whenever it left horizons, 3 = inside the frame only (every frame a front-slab frame: nothing depends on the previous frame)")
    ap.add_argument("--front-slab", type=int, default=1, help="GSR_OPT_FRONT_SLAB (A/B): 0 = frames that cannot use the previous frame's horizons are plain unculled frames")
    ap.add_argument("--jump-every", type=int, default=0, help="the orbit jumps by 111 degrees every N steps (0 = a steady orbit): N = 1 makes every frame a cold frame")
    ap.add_argument("--cluster-cull", type=int, default=1, help="0 = no cluster culling in front of K1 (A/B)")
    ap.add_argument("--storage-order", type=int, default=1, help="0 = splats stored in upload order instead of Morton order (A/B)")
    ap.add_argument("--dilate", type=int, default=-1, help="GSR_OPT_CULL_DILATE (A/B; -1 = library default)")
    ap.add_argument("--local-sort", type=int, default=1, help="GSR_OPT_LOCAL_SORT (A/B)")
    ap.add_argument("--tile-order", type=int, default=2, help="1 = XCD-aware static tile order, 2 = + heaviest tiles first (A/B)")
    ap.add_argument("--super-tile", type=int, default=0, help="super-tile edge in tiles (0 = auto) (A/B)")
    ap.add_argument("--flags", type=int, default=0, help="GSR_OPT_DEBUG_FLAGS (A/B)")
    ap.add_argument("--stage-timing", type=int, default=1, choices=(1, 2),
                    help="HIP events per frame in the TIMED region: 1 (default) = only around the blend kernel (what the roofline "
                         "needs; every event costs the GPU ~6 us of idle queue), 2 = around every stage.  The per-stage "
                         "breakdown is always taken from a short extra leg with level 2 after the timed region")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="timed region: 1 (default) = strictly serial frames, what an interactive viewport does and "
                         "what gives clean per-kernel durations for the roofline; 2 = frame f+1's front end overlaps "
                         "frame f's blend kernel (reported separately as 'pipelined')")
    ap.add_argument("--pipelined", action="store_true",
                    help="after the timed region, time the same K steps again with two frames in flight and report "
                         "them as 'pipelined' (off by default so that a rocprofv3 run of the default command sees "
                         "only the serial frames the roofline is computed from)")
    ap.add_argument("--shard-layout", type=int, default=1, choices=(0, 1),
                    help="N>1 (and --emulate-shard): 1 (default) = contiguous bands of tile rows -- a rank keeps ~1/N of the splats and "
                         "K1 drops the rest before the covariance chain; 0 = interleaved tile rows (balances any scene)")
    ap.add_argument("--emulate-rank", type=int, default=0, help="which shard --emulate-shard renders")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="single-GPU diagnostic: render only tile-row shard 0 of N (per-rank cost of an N-GPU run, no gather)")
    ap.add_argument("--lazy", type=int, default=1, choices=(0, 1, 2),
                    help="GSR_OPT_LAZY_COLOUR: 1 (library default) = SH colours only for the splats a frame can composite, when the "
                         "kernels find that it pays; 2 = always; 0 = eager (A/B)")
    ap.add_argument("--torch-gather", action="store_true",
                    help="N>1: gather with torch.distributed (multigpu.FrameGatherer) instead of the in-library RCCL gather")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="run ONLY the warm-up and the timed region (no per-stage leg, no culling-off leg, no pipelined leg): what a "
                         "rocprofv3 run should see, so that its per-kernel averages describe one regime")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the informational leg that renders BASELINE C1, C2, C3 and C5 after the timed region")
    ap.add_argument("--via-multi", action="store_true",
                    help="--gpus 1 through the multi-GPU entry (gsr_multi with ONE rank): what the N = 1 point of a scaling curve measured with "
                         "`bench.py --gpus N` runs, so that it can be checked against the single-context line (they must agree within ~3 %)")
    ap.add_argument("--depth", default="none", choices=("none", "far", "occluder"),
                    help="the TIMED region renders depth-tested frames (gsr_render_depth, what the viewport hook issues on every redraw: the reference "
                         "draws with the depth test on, src/GSplatRenderer.C:595-610): far = an opaque-pass depth buffer cleared to the far plane, "
                         "occluder = with an opaque sphere among the splats covering ~30 %% of the screen (scenes.sphere_occluder_depth).  "
                         "The default line reports both as `depth_tested` beside the plain frame; this switch is for rocprofv3 runs")
    ap.add_argument("--no-verify", action="store_true",
                    help="N>1: skip the check (after the timed region) that the last stitched frame is bit-identical to the same frame "
                         "rendered unsharded on rank 0")
    return ap.parse_args()


def cpu_baseline(oracle, splats, cam0, pkg, budget_s: float) -> dict:
    """The oracle (a from-scratch port of the reference's semantics; the reference's own
    HDK/GL path cannot run without Houdini) timed on the host cores, bounded in wall time."""
    threads = oracle.max_threads()
    n = splats.n
    # BASELINE.md asks for the median of >= 20 frames after 3 warm-ups; the contract bounds the leg to tens of seconds.  So the
    # SAMPLE is a prefix of the scene sized (from a probe on 1/32 of it) so that 23 frames fit the budget.
    probe_n = max(1, n // 32)
    sub = splats.subset(slice(0, probe_n))
    oracle.render(sub, cam0, threads=threads)
    t0 = time.perf_counter()
    oracle.render(sub, cam0, threads=threads)
    t_probe = time.perf_counter() - t0
    frames_wanted, warm = 20, 3
    frac = min(1.0, max(1.0 / 64.0, budget_s / ((frames_wanted + warm) * t_probe * 32.0)))
    sample_n = max(1, int(n * frac))
    sub = splats if sample_n == n else splats.subset(slice(0, sample_n))
    times = []
    for frame in range(warm + frames_wanted):
        cam = pkg.camera.make_camera(cam0.width, cam0.height, sh_order=cam0.sh_order, frame=frame)
        t0 = time.perf_counter()
        oracle.render(sub, cam, threads=threads)
        if frame >= warm:
            times.append(time.perf_counter() - t0)
    t_med = float(np.median(times))
    fps_sample = 1.0 / t_med
    # the reference's own per-camera-move HOST stage (squared distances + parallel comparison argsort of the indices,
    # src/GSplatRenderer.C:188-208; oracle/host_stage_ref.cpp) on all splats: what the GPU depth sort replaces
    hs = []
    for k in range(4):
        cam = pkg.camera.make_camera(cam0.width, cam0.height, sh_order=cam0.sh_order, frame=k)
        t0 = time.perf_counter()
        oracle.reference_host_stage(splats.P, cam.cam_pos, threads)
        hs.append(time.perf_counter() - t0)
    host_stage_ms = float(np.median(hs[1:])) * 1e3
    # ... and ONE un-scaled frame of the whole scene, so that the extrapolation above can be checked against a measurement
    full_ms = None
    if sample_n != n:
        oracle.render(splats, cam0, threads=threads) if n <= 2_000_000 else None      # (a warm-up only where it is cheap)
        t0 = time.perf_counter()
        oracle.render(splats, pkg.camera.make_camera(cam0.width, cam0.height, sh_order=cam0.sh_order, frame=warm), threads=threads)
        full_ms = (time.perf_counter() - t0) * 1e3
    else:
        full_ms = t_med * 1e3
    return {
        "value": fps_sample * (sample_n / n),
        "unit": "frames/sec",
        "cores": threads,
        "kind": "port",
        "sample": (f"oracle (C/OpenMP port of the reference semantics: per-splat SH+projection, argsort, per-pixel "
                   f"gaussian + under-blend) on the first {sample_n} of {n} splats, {cam0.width}x{cam0.height}, "
                   f"median of {len(times)} frames after 3 warm-ups: {t_med * 1e3:.1f} ms/frame; value = sample fps x {sample_n}/{n} "
                   f"(per-splat work is linear in splats; per-pixel work grows more slowly, so this flatters the CPU)"),
        "sample_fps": fps_sample,
        "frames": len(times),
        "full_scene_ms": full_ms,
        "full_scene_fps": (1e3 / full_ms) if full_ms else None,
        "full_scene": f"one frame of all {n} splats (not scaled), {threads} threads",
        "reference_host_stage_ms": host_stage_ms,
        "reference_host_stage": (f"argsortByDistance restated (distance^2 + __gnu_parallel::sort of int indices by indirect float "
                                 f"compare, standing in for tbb::parallel_sort) on all {n} splats, {threads} threads: median of 3"),
    }


def workload_name(args, splats, cfg, order, W, H) -> str:
    if cfg.get("kind") == "ply":
        return (f"PLY {os.path.basename(cfg['path'])}: {splats.n} splats of an INRIA capture (SH deg {order}), {W}x{H}, camera orbiting the cloud's "
                f"median at 1.6 x its 80 % radius (re-sort every frame)")
    return f"{args.config}: {splats.n} synthetic splats (SH deg {order}, seed {cfg['seed']}), {W}x{H}, orbiting camera (re-sort every frame)"


# the opaque sphere of the depth-tested legs: centred on the view axis 3.0 units in front of the camera, radius 0.566: 30 % of a
# 1920x1080 frame.  C4: its front pokes 0.19 units OUT of the cloud (the middle of the frame shows bare geometry behind a thin veil:
# tiles that never saturate), its rim lies 0.2 units under the cloud's surface (tiles that saturate just in front of it), and in
# between sits the ring where the geometry is AT the depth the tiles saturate at -- the case that decides whether the culling copes
OCCLUDER = {"centre_distance": 3.0, "radius": 0.566}


def depth_buffer(kind: str, pkg, torch, cam, scale: float = 1.0):
    """device tensor [H, W] float32: the opaque pass's window depth (row 0 = bottom); scale = the orbit distance relative to C4's"""
    if kind == "far":
        d = np.ones((cam.height, cam.width), np.float32)
    else:
        d = pkg.scenes.sphere_occluder_depth(cam, OCCLUDER["centre_distance"] * scale, OCCLUDER["radius"] * scale)
    return torch.from_numpy(d).to("cuda")


def orbit_frame(i: int, jump_every: int) -> int:
    """the orbit frame of step i: 3 degrees per step, plus a jump of 37 frames (111 degrees) every `jump_every` steps"""
    return i + (37 * (i // jump_every) if jump_every > 0 else 0)


def error_line(args, msg: str, code: int = 2):
    """a parsable line instead of a bare exit string: the driver records what went wrong"""
    print(json.dumps({"metric": "frames/sec at 1920x1080 + achieved HBM GB/s (blend kernel)", "value": None, "unit": "frames/sec",
                      "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": msg}))
    sys.stdout.flush()
    sys.exit(code)


def blend_roofline(st, own_px: float, regime: str, hbm, tj, traffic_from, kernel="k_blend") -> dict:
    """blend-kernel roofline of ONE context from its gsr_stats: HIP events on the kernel's own stream, counts from the kernel"""
    launches = max(1, st["blend_launches"])                    # launches bracketed by events (every --time-every-th frame)
    frames_done = max(1, st["frames"])                         # launches in all: the kernel's own counters cover every one
    blend_ms = st["blend_ms_total"] / launches
    d_eff = st["blend_pairs_consumed_total"] / frames_done     # (tile, splat) pairs CONSUMED per launch = records gathered
    scanned = st["blend_entries_scanned_total"] / frames_done  # list entries (idx + mask) read per launch
    rec_b, pair_b = st["record_bytes"], st["pair_bytes"]
    # unit of work = one consumed (tile, splat) pair = its list entry (8 B) + its projected record (48 B; this
    # build's true sizes), plus one RGBA-f32 store per pixel.  The entries a tile merely SCANS in its super-tile's list to
    # find its own (the price of coarse lists) are NOT algorithmic bytes: they are reported beside it.
    bytes_blend = (pair_b + rec_b) * d_eff + 16.0 * own_px
    achieved = bytes_blend / (blend_ms * 1e-3) / 1e9 if blend_ms > 0 else 0.0
    traffic = None
    if tj is not None:
        try:
            traffic = float(next(v for k, v in tj.items() if kernel in k)["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
    peak_meas = hbm["peak_GBps"] if hbm else None
    roofline = {
        "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS, "peak_measured": peak_meas, "frac_of_measured": (achieved / peak_meas) if peak_meas else None,
        "peak_measured_by": "tools/ubench_hbm.hip: best of the streaming copies over 1 GiB arrays on this GPU (read + write bytes)" if hbm else None, "hbm_ubench": hbm,
        "regime": regime, "traffic": traffic, "traffic_replayed_from": traffic_from if traffic is not None else None,
        "avg_launch_ms": blend_ms, "launches_timed": int(st["blend_launches"]), "launches": int(st["frames"]),
        "algorithmic_bytes_per_launch": bytes_blend,
        "pairs_consumed_per_launch": d_eff, "bytes_per_consumed_pair": pair_b + rec_b,
        "entries_scanned_per_launch": scanned, "list_scan_bytes": pair_b * scanned,
        "scan_amplification": scanned / d_eff if d_eff > 0 else None,
        "pairs_sorted_last_frame": st["pairs_total"], "record_bytes": rec_b, "entry_bytes": pair_b,
        "super_tile": st["super_tile"],
        "note": "k_blend is FP32-vector-issue bound at this arithmetic intensity (DESIGN.md); the HBM fraction is reported "
                "as the metric asks, on consumed pairs only (SURVEY 8d)",
    }
    # the bound that actually binds k_blend: FP32 vector issue.  One pixel evaluation of one record is 22 FLOP
    # (fma = 2: affine forms 8, power 3, log-domain alpha 2 = one subtraction + one v_exp_f32 (contract v3; the software
    # 2^x + multiply of contract v2 were 15), quad test 1, under-blend 8), counted
    # per 64-lane wave evaluation by the kernel itself (lanes outside the quad execute the same instructions)
    FLOP_PER_EVAL = 22.0
    wave_evals = st["blend_wave_evals_total"] / frames_done
    valu_tflops = wave_evals * 64 * FLOP_PER_EVAL / (blend_ms * 1e-3) / 1e12 if blend_ms > 0 else 0.0
    # ... and against the ISSUE RATES measured on this GPU (tools/ubench_valu.hip -> profiles/ubench_valu_mi355x.txt).
    # One inner-loop iteration = two wave-record evaluations (ISA of k_blend<false>, tools/kernel_resources.py --isa)
    n_simd = 256 * 4
    iter_ns = BLEND_ITER_NS
    issue_ms = (wave_evals / 2.0) * iter_ns / n_simd * 1e-6
    roofline["valu"] = {"bound": "fp32 vector", "achieved": valu_tflops, "peak": 157.3, "unit": "TFLOP/s",
                        "frac": valu_tflops / 157.3, "wave_record_evals_per_launch": wave_evals,
                        "flop_per_pixel_eval": FLOP_PER_EVAL, "inner_loop_iteration_ns": iter_ns,
                        "inner_loop_issue_bound_ms": issue_ms, "inner_loop_issue_frac": issue_ms / blend_ms if blend_ms > 0 else 0.0,
                        "frac_at_contract_v2_flop_count": (wave_evals * 64 * 34.0 / (blend_ms * 1e-3) / 1e12 / 157.3) if blend_ms > 0 else 0.0,
                        "note": "issue bound = inner-loop instruction mix x issue intervals measured with tools/ubench_valu.hip; "
                                "contract v3 removed 12 of the 34 FLOP of a pixel evaluation (the software 2^x), so the same evaluations "
                                "count for fewer FLOP: frac_at_contract_v2_flop_count prices them as rounds 1-2 did"}
    return roofline


# one inner-loop iteration of k_blend<false> = two wave-record evaluations (ISA, tools/kernel_resources.py --isa):
# 5 v_pk_fma_f32 with three full operands (2.02 ns per SIMD), 2 with a broadcast operand (1.77), 1 v_pk_mul (1.76),
# 2 v_max + 6 v_cmp (1.72), 2 v_cndmask (1.64), 2 v_exp_f32 (3.40), 9 full-rate mul/fmac/sub/add (1.0)
BLEND_ITER_NS = 5 * 2.02 + 2 * 1.77 + 1 * 1.76 + 8 * 1.72 + 2 * 1.64 + 2 * 3.40 + 9 * 1.0


def load_traffic(regime: str, usable: bool):
    """HBM traffic per launch: PMC counters cannot be read from inside this process, so the figure is REPLAYED from the
    rocprofv3 --pmc passes of this same command (tools/gpu_pmc.sh -> profiles/pmc_traffic.json; 2*FETCH_SIZE + WRITE_SIZE as
    the MI355X guide prescribes) -- and only when that file was produced by the kernels being run now"""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not (usable and os.path.exists(tpath)):
        return None, None
    try:
        tall = json.load(open(tpath))
        if tall.get("_kernel_source_sha") != kernel_source_sha():   # (stale: collected with other kernels)
            return None, None
        return tall.get(regime), f"profiles/pmc_traffic.json[{regime}] (" + str(tall.get("_collected", "rocprofv3 --pmc passes of this command")) + ")"
    except Exception:
        return None, None


def time_every(args) -> int:
    """HIP events around the blend kernel of every N-th timed frame: 4 by default (8 for long runs) -- an event pair idles the queue
    ~12 us, so bracketing EVERY launch makes the frames it measures slower than the ones rocprofv3 sees; at least five launches are timed"""
    if args.time_every > 0:
        return args.time_every
    return max(1, min(8 if args.steps >= 64 else 4, args.steps // 5))


def set_common_options(target, pkg, args):
    """the options both forms (one context / gsr_multi) take"""
    E = pkg.engine
    target.set_option(E.OPT_OCCLUSION_CULL, args.cull)
    target.set_option(E.OPT_CLUSTER_CULL, args.cluster_cull)
    target.set_option(E.OPT_FRONT_SLAB, args.front_slab)
    target.set_option(E.OPT_LOCAL_SORT, args.local_sort)
    if args.dilate >= 0:
        target.set_option(E.OPT_CULL_DILATE, args.dilate)
    target.set_option(E.OPT_STORAGE_ORDER, args.storage_order)
    target.set_option(E.OPT_TIMING_EVERY, time_every(args))
    target.set_option(E.OPT_XCD_SWIZZLE, 0 if args.no_swizzle else args.tile_order)
    target.set_option(E.OPT_SUPER_TILE, args.super_tile)
    target.set_option(E.OPT_DEBUG_FLAGS, args.flags)
    target.set_option(E.OPT_FRAMES_IN_FLIGHT, args.frames_in_flight)
    target.set_option(E.OPT_STAGE_TIMING, args.stage_timing)
    target.set_option(E.OPT_LAZY_COLOUR, args.lazy)
    target.set_option(E.OPT_SHARD_LAYOUT, args.shard_layout)


def reference_frame(pkg, torch, dev_index, stream, splats, cam_struct, W, H, exact: bool, depth=None):
    """the same frame from a second, unsharded context on GPU `dev_index`; exact = occlusion AND cluster culling off (every
    clip-visible splat is projected, sorted, binned: what the timed frames must be bit-identical to)"""
    ref_eng = pkg.Engine(dev_index)
    try:
        ref_eng.set_stream(stream.cuda_stream)
        if exact:
            ref_eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
            ref_eng.set_option(pkg.engine.OPT_CLUSTER_CULL, 0)
        ref_eng.upload(splats)
        ref = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{dev_index}")
        if depth is not None:
            ref_eng.render_struct_depth_to_device(cam_struct, depth.data_ptr(), ref.data_ptr())
        else:
            ref_eng.render_struct_to_device(cam_struct, ref.data_ptr())
        torch.cuda.synchronize(dev_index)
        return ref
    finally:
        ref_eng.close()


def main_single_process(args):
    """python bench.py --gpus N, no launcher: ONE process, gsr_multi_* (a worker thread per rank inside the library, the
    frame's one collective over RCCL).  The reference draws from one thread of one process (src/DM_GSplatHook.C:30-39)."""
    import torch

    N = args.gpus
    if not torch.cuda.is_available():
        error_line(args, "bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    allow_dup = os.environ.get("GSR_BENCH_ALLOW_DUP", "0") == "1"
    if ndev < N and not allow_dup:
        error_line(args, f"--gpus {N} but only {ndev} GPU(s) visible (GSR_BENCH_ALLOW_DUP=1 runs the ranks as contexts on the "
                         f"GPUs there are, over the COPY transport: a functional test, not a measurement)")
    devices = [g % ndev for g in range(N)]
    functional_only = ndev < N
    pkg = ge.load_package()
    E = pkg.engine
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.ply:
        args.config = pkg.scenes.register_ply_config(args.ply, pkg.ply)
    splats, cfg = pkg.scenes.make_config(args.config, args.splats)
    W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]
    torch.cuda.set_device(devices[0])
    try:
        M = pkg.MultiEngine(devices, E.TRANSPORT_COPY if functional_only else E.TRANSPORT_RCCL)
    except pkg.engine.GsrError as e:
        error_line(args, f"gsr_multi_create over devices {devices}: {e}")
    stream = torch.cuda.Stream(device=devices[0])
    torch.cuda.set_stream(stream)
    M.set_stream(stream.cuda_stream)
    set_common_options(M, pkg, args)
    M.upload(splats)
    cams = [E.camera_struct(pkg.scenes.config_camera(args.config, pkg.camera, W, H, order, orbit_frame(i, args.jump_every))) for i in range(args.warmup + args.steps)]
    final = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{devices[0]}")

    def sync_all():
        M.synchronize()
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    def step(i):
        M.render_struct_to_device(cams[i], final.data_ptr())

    for i in range(args.warmup):
        step(i)
    sync_all()
    for g in range(N):
        E._check(M.L.gsr_stats_reset(M.L.gsr_multi_context(M.h, g)))
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync_all()
    elapsed = time.perf_counter() - t0

    verified = None
    if not args.no_verify:
        ref = reference_frame(pkg, torch, devices[0], stream, splats, cams[args.warmup + args.steps - 1], W, H, exact=True)
        verified = bool(torch.equal(final, ref)) and bool(ref[..., 3].max() > 0)
        if not verified:
            print("[bench] the sharded frame DIFFERS from the unsharded frame: max |diff| %.3e" % float((final - ref).abs().max()), file=sys.stderr)
        del ref
    stats = [M.stats(g) for g in range(N)]
    # extra leg (untimed): per-rank stage times (HIP events around every stage) and the gather on the root's transfer stream
    stage = None
    gather_ms = None
    if not args.no_extra_legs:
        M.set_option(E.OPT_STAGE_TIMING, 2)
        last = args.warmup + args.steps
        for i in range(max(0, last - 12), max(0, last - 10)):
            step(i)
        sync_all()
        for g in range(N):
            E._check(M.L.gsr_stats_reset(M.L.gsr_multi_context(M.h, g)))
        M.gather_stats(1)
        for i in range(max(0, last - 10), last):
            step(i)
        sync_all()
        ms, k = M.gather_stats(0)
        gather_ms = ms / k if k else None
        stage = []
        for g in range(N):
            sg = M.stats(g)
            fr = max(1, sg["stage_frames"])
            stage.append({"rank": g, "device": devices[g], "ms_preprocess": sg["stage_ms_total"][0] / fr, "ms_depth_sort": sg["stage_ms_total"][1] / fr,
                          "ms_binning": (sg["stage_ms_total"][2] + sg["stage_ms_total"][3]) / fr, "ms_blend": sg["stage_ms_total"][4] / fr,
                          "ms_total": sg["frame_ms_total"] / fr, "n_visible": sg["n_visible"], "clusters_kept": sg["clusters_kept"],
                          "frames_culled": sg["frames_culled"], "frames_repaired": sg["frames_repaired"]})
        M.set_option(E.OPT_STAGE_TIMING, args.stage_timing)
    rccl_ranks, rccl_n = M.comm_info()
    # a communicator that does not span the N ranks means the frame never crossed xGMI the way the line says: no number then
    comm_ok = (M.transport != E.TRANSPORT_RCCL) or N == 1 or rccl_n == N
    if not comm_ok:
        print(f"[bench] RCCL reports {rccl_n} ranks in the communicator, {N} were asked for", file=sys.stderr)
    # the gather, per link: every peer sends its band over its own xGMI link to the root (SURVEY 8e assumes ~153 GB/s per link)
    band_bytes = [int(sum(min(16, H - r * 16) for r in pkg.multigpu.owned_tile_rows(H, g, N, args.shard_layout)) * W * 16) for g in range(N)]
    link = None
    if N > 1:
        peer_bytes = max(band_bytes[1:])
        link = {"bytes_per_peer": band_bytes[1:], "bytes_total_into_root": int(sum(band_bytes[1:])), "gather_ms": gather_ms,
                "GBps_per_link": (peer_bytes / (gather_ms * 1e-3) / 1e9) if gather_ms else None, "GBps_assumed_per_link": 153.0,
                "GBps_into_root": (sum(band_bytes[1:]) / (gather_ms * 1e-3) / 1e9) if gather_ms else None,
                "note": "gather_ms = HIP events on the root's transfer stream around the grouped ncclRecv x (N-1) (+ the root's own band copy); the "
                        "peers send concurrently, each over its own link: per-link rate = the largest peer band / gather_ms"
                        + ("" if M.transport == E.TRANSPORT_RCCL else " -- COPY transport on shared GPUs: NOT a link measurement")}
    hbm = measured_hbm_peak()
    per_rank = []
    for g in range(N):
        own_px = sum(min(16, H - r * 16) for r in pkg.multigpu.owned_tile_rows(H, g, N, args.shard_layout)) * W
        regime = "culled" if stats[g]["frames_culled"] * 2 > stats[g]["frames"] else "unculled"
        per_rank.append(blend_roofline(stats[g], own_px, regime, hbm if g == 0 else None, None, None))
        per_rank[-1]["rank"] = g
    heavy = max(range(N), key=lambda g: per_rank[g]["avg_launch_ms"])
    line = {
        "metric": "frames/sec at 1920x1080 + achieved HBM GB/s (blend kernel)" if (W, H) == (1920, 1080)
        else f"frames/sec at {W}x{H} + achieved HBM GB/s (blend kernel)",
        "value": (args.steps / elapsed) if (verified is not False and comm_ok) else None,
        "unit": "frames/sec", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.ply else "INRIA PLY capture (--ply)",
        "config": {"workload": workload_name(args, splats, cfg, order, W, H),
                   "parallelism": f"tile-row shard x{N}, " + ("contiguous bands" if args.shard_layout else "interleaved rows") +
                                  ", one process (gsr_multi: a worker thread per rank)",
                   "gather": ("in-library RCCL: ncclSend / ncclRecv in one group on per-rank transfer streams, " +
                              ("received straight into the framebuffer (zero copy)" if args.shard_layout else "k_stitch_bands on the root") +
                              ", overlapping the next frame's kernels") if M.transport == E.TRANSPORT_RCCL else
                             "COPY transport (device-to-device copies ordered by events): FUNCTIONAL TEST, several ranks share a GPU",
                   "devices": devices, "functional_only": functional_only,
                   "n_splats": splats.n, "width": W, "height": H, "frames_in_flight": args.frames_in_flight},
        "rccl_ranks": rccl_ranks, "rccl_comm_count": rccl_n,
        "sharded_frame_bit_identical": verified,
        "gather_ms": gather_ms,
        "gather_links": link,
        "per_rank_ms_per_step": [sg["ms_total"] for sg in stage] if stage else None,
        "per_rank_stages_ms": stage,
        "roofline": per_rank[heavy],
        "roofline_per_rank": [{k: r[k] for k in ("rank", "achieved", "frac", "avg_launch_ms", "pairs_consumed_per_launch", "algorithmic_bytes_per_launch", "regime")} for r in per_rank],
        "n_visible_per_rank": [st["n_visible"] for st in stats],
        "occlusion_culling": {"enabled": bool(args.cull), "frames_culled": [st["frames_culled"] for st in stats],
                              "frames_repaired": [st["frames_repaired"] for st in stats], "frames": [st["frames"] for st in stats]},
    }
    print(json.dumps(line))
    sys.stdout.flush()
    M.close()
    if verified is False:
        sys.exit(3)
    if not comm_ok:
        sys.exit(4)


def host_target_leg(eng, cam_structs, W, H):
    """frames/s with a HOST target (pageable, re-used): the library composites in bands of tile rows and copies each band back while the
    next ones composite (GSR_HOST_BANDS, default 4)"""
    import numpy as _np
    buf = _np.zeros((H, W, 4), _np.float32)
    n = len(cam_structs)
    warm = min(10, n // 3)
    for c in cam_structs[:warm]:
        eng.render_struct_to_host(c, buf.ctypes.data)
    t0 = time.perf_counter()
    for c in cam_structs[warm:]:
        eng.render_struct_to_host(c, buf.ctypes.data)
    dt = (time.perf_counter() - t0) / (n - warm)
    mb = W * H * 16 / 1e6
    return {"value": 1.0 / dt, "unit": "frames/sec", "ms_per_step": dt * 1e3, "steps": n - warm, "bytes_back_per_frame": W * H * 16,
            "link_GBps_incl_render": mb / 1e3 / dt, "bands": int(os.environ.get("GSR_HOST_BANDS", "4")),
            "note": "gsr_render with a host pointer (pageable numpy array, re-used): PCIe-inclusive, never `value`; the floor is the link: "
                    "%.1f MB per frame" % mb}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.via_multi):
        return main_single_process(args)
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            error_line(args, f"--gpus {args.gpus} but the launcher started {world} rank(s)")
        sys.exit(2)
    if not torch.cuda.is_available():
        if rank == 0:
            error_line(args, "bench.py needs a GPU (the HIP path has no CPU fallback)")
        sys.exit(2)
    # one rank per GPU; GSR_BENCH_BACKEND=gloo + fewer GPUs than ranks is a functional test mode only
    # (lets the sharded path run end-to-end on a 1-GPU box), never a performance configuration
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        if rank == 0:
            error_line(args, f"{world} ranks but only {ndev} GPU(s) visible")
        sys.exit(2)
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    pkg = ge.load_package()
    if args.ply:
        args.config = pkg.scenes.register_ply_config(args.ply, pkg.ply)
    splats, cfg = pkg.scenes.make_config(args.config, args.splats)
    if args.morton:
        P = splats.P.astype(np.float64)
        q = ((P - P.min(axis=0)) / np.maximum(np.ptp(P, axis=0), 1e-30) * 1023.0).astype(np.uint64)
        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            v = (v | (v << 2)) & 0x09249249
            return v
        code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        splats = splats.subset(np.argsort(code, kind="stable"))
    W, H, order = cfg["width"], cfg["height"], cfg["sh_order"]

    eng = pkg.Engine(dev_index)
    # one explicit (non-null) HIP stream: frames are ordered on it (N > 1: the kernels and the RCCL gather run on streams of the library)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    set_common_options(eng, pkg, args)
    if world > 1:
        eng.set_row_shard(rank, world)
    elif args.emulate_shard > 1:
        eng.set_row_shard(args.emulate_rank % args.emulate_shard, args.emulate_shard)
    eng.upload(splats)  # once: geometry stays resident in HBM

    cams = [pkg.engine.camera_struct(pkg.scenes.config_camera(args.config, pkg.camera, W, H, order, orbit_frame(i, args.jump_every)))
            for i in range(args.warmup + args.steps)]
    # N>1: the frame's ONE collective -- band images -> rank 0 over xGMI -- lives INSIDE the library (gsr_comm_render:
    # render band -> ncclSend / ncclRecv x (N-1) in one group on a transfer stream; band layout: straight into the framebuffer).
    # torch.distributed only carries the 128-byte communicator id, the barriers and the max-over-ranks of the timing.  Every
    # rank must be able to load RCCL for that; otherwise (and in the gloo functional-test mode) the torch gather of multigpu.py is used.
    gather = "none (1 GPU)"
    fg = None
    final = None
    rccl_info = None
    if world > 1:
        use_lib = backend == "nccl" and not args.torch_gather
        if use_lib:
            # every rank must be able to load RCCL, rank 0 must get an id, and every rank's ncclCommInitRank must succeed --
            # each step agreed on collectively, so that a failure anywhere sends ALL ranks to the torch gather
            def all_ok(flag: bool) -> bool:
                t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return bool(t.item())
            use_lib = all_ok(bool(pkg.load_library().gsr_comm_available()))
            idbuf = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if use_lib:
                got = True
                if rank == 0:
                    try:
                        idbuf.copy_(torch.frombuffer(bytearray(eng.comm_unique_id()), dtype=torch.uint8))
                    except Exception:
                        got = False
                use_lib = all_ok(got)
            if use_lib:
                dist.broadcast(idbuf, src=0)
                inited = True
                try:
                    eng.comm_init(bytes(idbuf.cpu().numpy().tobytes()), rank, world)     # collective (ncclCommInitRank)
                except Exception as e:  # noqa: BLE001
                    inited = False
                    print(f"[rank {rank}] gsr_comm_init failed: {e}", file=sys.stderr)
                use_lib = all_ok(inited)
                if not use_lib and inited:
                    eng.comm_destroy()
                    eng.set_row_shard(rank, world)
            if use_lib:
                final = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") if rank == 0 else None
                gather = ("in-library RCCL: gsr_comm_render (ncclSend / ncclRecv group on a transfer stream, " +
                          ("received straight into the framebuffer" if args.shard_layout else "k_stitch_bands on the root") +
                          ", overlapping the next frame's kernels)")
                rccl_info = list(eng.comm_info())
        if not use_lib:
            fg = pkg.multigpu.FrameGatherer(dist, rank, world, W, H, "cuda", engine=eng, via_host=(backend != "nccl"),
                                            layout=args.shard_layout)
            gather = "torch.distributed.gather + gsr_stitch_bands"
    in_lib = world > 1 and fg is None
    band = None
    if not in_lib:
        band = torch.zeros((eng.band_rows(H), W, 4), dtype=torch.float32, device="cuda") if fg is None else fg.band
        assert band.shape[0] == eng.band_rows(H)

    # --depth: the timed region's frames are depth-tested against an opaque pass's depth buffer (device memory, full image)
    cam0_py = pkg.scenes.config_camera(args.config, pkg.camera, W, H, order, 0)
    depth_scale = float(pkg.scenes.CONFIGS[args.config].get("distance", 4.61995)) / 4.61995
    depth_t = depth_buffer(args.depth, pkg, torch, cam0_py, depth_scale) if args.depth != "none" else None

    def step(i):
        if in_lib:
            eng.comm_render(cams[i], final.data_ptr() if final is not None else 0, depth_t.data_ptr() if depth_t is not None else 0)
        else:
            if depth_t is not None:
                eng.render_struct_depth_to_device(cams[i], depth_t.data_ptr(), band.data_ptr())
            else:
                eng.render_struct_to_device(cams[i], band.data_ptr())
            if fg is not None:
                fg.gather_and_stitch()

    def current_frame():
        """rank 0: the full frame of the last step (device tensor)"""
        if final is not None:
            return final
        return fg.gather_and_stitch() if fg is not None else band

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    eng.stats_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    stamps = [0.0] * (args.steps + 1)        # host clock after every call (no synchronisation: gsr_render returns once the frame is queued and,
    t0 = time.perf_counter()                 # in the temporal regime, has checked itself -- the stamps trail the GPU by less than a frame)
    for i in range(args.steps):
        step(args.warmup + i)
        stamps[i + 1] = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stamps[args.steps] = elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # dispersion inside the timed region: frames/s of five equal windows of it (host stamps; the last window ends at the synchronised end)
    windows = None
    if args.steps >= 10:
        edges = [round(k * args.steps / 5) for k in range(6)]
        windows = [(edges[k + 1] - edges[k]) / (stamps[edges[k + 1]] - stamps[edges[k]]) for k in range(5) if stamps[edges[k + 1]] > stamps[edges[k]]]

    # Is the last TIMED frame the full frame?  The same camera from a second context on this GPU that culls nothing
    # (GSR_OPT_OCCLUSION_CULL = 0, GSR_OPT_CLUSTER_CULL = 0, unsharded): bit for bit.
    verified = None
    if not args.no_verify and (world > 1 or args.emulate_shard <= 1):
        last = cams[args.warmup + args.steps - 1]
        shown = current_frame()                                 # the last step's frame on rank 0
        if rank == 0:
            ref = reference_frame(pkg, torch, dev_index, stream, splats, last, W, H, exact=True, depth=depth_t)
            verified = bool(torch.equal(shown, ref)) and bool(ref[..., 3].max() > 0)
            if not verified:   # reported in the JSON line; the exit code follows after the line (the other ranks are at the barrier below)
                print("[bench] the timed frame DIFFERS from the frame rendered without culling / sharding: max |diff| %.3e" %
                      float((shown - ref).abs().max()), file=sys.stderr)
            del ref
            torch.cuda.empty_cache()
        if world > 1:
            dist.barrier()
    st = eng.stats()
    # extra leg: LATENCY -- BASELINE.md section 5 defines fps as 1 / median wall-clock of a synchronous gsr_render; `value` above is the
    # throughput of back-to-back frames (the queue never drains).  Here every frame is waited for before the next is issued.
    latency = None
    if not args.no_extra_legs:
        lat = []
        last = args.warmup + args.steps
        k_lat = min(30, last)
        for i in range(last - k_lat, last):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(i)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        lat = sorted(lat[2:]) if len(lat) > 4 else sorted(lat)       # (the first two re-enter the orbit: they see another frame's horizons)
        latency = {"latency_ms_median": lat[len(lat) // 2] * 1e3, "latency_ms_min": lat[0] * 1e3, "latency_ms_max": lat[-1] * 1e3, "frames": len(lat),
                   "fps_at_median_latency": 1.0 / lat[len(lat) // 2],
                   "note": "synchronous frames: issue, wait for the device, repeat (BASELINE.md section 5's definition of fps); `value` is back-to-back throughput"}
    # extra leg (untimed, informational): a few more frames with HIP events around EVERY stage -> per-stage breakdown and the
    # k_preprocess / k_colour_prefix durations.  Kept out of the timed region: six extra events per frame stall the queue ~35 us.
    st_stage = None
    if not args.no_extra_legs:
        # ... in the regime of the timed region: the orbit simply continues (re-rendering its last frames), and the leg's first
        # two frames -- which still see the horizons / hints of the frame before the leg -- are not counted
        eng.set_option(pkg.engine.OPT_STAGE_TIMING, 2)
        last = args.warmup + args.steps
        for i in range(max(0, last - 12), max(0, last - 10)):
            step(i)
        torch.cuda.synchronize()
        eng.stats_reset()
        for i in range(max(0, last - 10), last):
            step(i)
        torch.cuda.synchronize()
        st_stage = eng.stats()
        eng.set_option(pkg.engine.OPT_STAGE_TIMING, args.stage_timing)
    if world > 1:
        dist.barrier()
    # extra leg (informational, single GPU): the same frames with occlusion culling switched off
    unculled = None
    if world == 1 and args.cull and not args.no_extra_legs:
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, 0)
        for i in range(min(5, args.warmup + args.steps)):
            step(i)
        torch.cuda.synchronize()
        k2 = min(40, args.steps)
        t0 = time.perf_counter()
        for i in range(k2):
            step(args.warmup + i)
        torch.cuda.synchronize()
        unculled = {"value": k2 / (time.perf_counter() - t0), "unit": "frames/sec", "steps": k2,
                    "note": "GSR_OPT_OCCLUSION_CULL=0: every clip-visible splat is coloured (lazily), sorted and binned"}
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, args.cull)
    # extra leg (informational, single GPU): COLD frames -- every frame jumps 111 degrees, so no frame can use the previous one's
    # horizons.  Under the library's policy (a culled attempt that breaks is re-rendered as a front-slab frame; culling backs off) and
    # with occlusion culling inside the frame only (GSR_OPT_OCCLUSION_CULL = 3: every frame a front-slab frame)
    cold = None
    if world == 1 and args.cull and not args.no_extra_legs and args.emulate_shard <= 1:
        k3 = min(40, args.steps)
        # (a rotation about a ball-shaped cloud leaves its depth horizons valid; what makes a frame cold is a jump in DISTANCE:
        #  every other frame from 1.3 x as far, 111 degrees further round)
        base_d = pkg.scenes.CONFIGS[args.config].get("distance", 4.61995)
        if pkg.scenes.CONFIGS[args.config].get("kind") == "terrain":
            jcams = [pkg.engine.camera_struct(pkg.scenes.terrain_camera(pkg.camera, W, H, frame=orbit_frame(i, 1), sh_order=order, distance=4.2 * (1.3 if i % 2 else 1.0))) for i in range(k3 + 6)]
        else:
            jcams = [pkg.engine.camera_struct(pkg.camera.make_camera(W, H, sh_order=order, frame=orbit_frame(i, 1), distance=base_d * (1.3 if i % 2 else 1.0),
                                                                         pivot=pkg.scenes.CONFIGS[args.config].get("pivot", (0.0, 0.0, 0.0)))) for i in range(k3 + 6)]
        cold = {}
        for name, mode in (("policy", args.cull), ("intra_frame_only", 3), ("one_pass", 0)):
            eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, mode)
            for i in range(6):
                eng.render_struct_to_device(jcams[i], band.data_ptr())
            torch.cuda.synchronize()
            eng.stats_reset()
            t0 = time.perf_counter()
            for i in range(k3):
                eng.render_struct_to_device(jcams[6 + i], band.data_ptr())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            stc = eng.stats()
            cold[name] = {"value": k3 / dt, "unit": "frames/sec", "steps": k3, "frames_slab": stc["frames_slab"], "frames_culled": stc["frames_culled"],
                          "frames_repaired": stc["frames_repaired"], "frames_jumped": stc["frames_jumped"]}
        cold["note"] = ("every frame jumps 111 degrees round the cloud and to / from 1.3 x the distance: nothing of the previous frame applies.  policy = the default (GSR_OPT_OCCLUSION_CULL = 1, "
                        "GSR_OPT_FRONT_SLAB = 1); intra_frame_only = GSR_OPT_OCCLUSION_CULL = 3; one_pass = GSR_OPT_OCCLUSION_CULL = 0 on the same cameras")
        eng.set_option(pkg.engine.OPT_OCCLUSION_CULL, args.cull)
    # extra leg (single GPU): the DEPTH-TESTED frame -- the one the viewport hook issues on every redraw (hdk/DM_GSplatHook_hip.C binds the
    # opaque pass's depth; the reference draws with the depth test on, src/GSplatRenderer.C:595-610).  The same orbit three ways, back
    # to back in this leg's own loop: plain (no depth buffer), a depth buffer cleared to the far plane, and one with an opaque sphere
    # among the splats covering ~30 % of the screen.  Each variant's last frame is checked bit for bit against a context that culls nothing.
    depth_leg = None
    if world == 1 and not args.no_extra_legs and args.emulate_shard <= 1 and args.depth == "none":
        k4 = min(60, args.steps)
        depth_leg = {}
        for name in ("plain", "far_plane", "occluder"):
            dt_ = None if name == "plain" else depth_buffer("far" if name == "far_plane" else "occluder", pkg, torch, cam0_py, depth_scale)
            # (a context of its own per variant, like `other_configs`: the policies' state -- dilation radius, hold-offs, the 64-frame
            #  streak counter -- must not carry from one variant into the next)
            de = pkg.Engine(dev_index)
            de.set_stream(stream.cuda_stream)
            set_common_options(de, pkg, args)
            de.upload(splats)

            def dstep(i):
                if dt_ is None:
                    de.render_struct_to_device(cams[i], band.data_ptr())
                else:
                    de.render_struct_depth_to_device(cams[i], dt_.data_ptr(), band.data_ptr())
            for i in range(min(args.warmup + 5, args.warmup + args.steps)):
                dstep(i)
            torch.cuda.synchronize()
            de.stats_reset()
            t0 = time.perf_counter()
            for i in range(k4):
                dstep(args.warmup + i)
            torch.cuda.synchronize()
            ddt = time.perf_counter() - t0
            sd = de.stats()
            ok = None
            if not args.no_verify:
                ref = reference_frame(pkg, torch, dev_index, stream, splats, cams[args.warmup + k4 - 1], W, H, exact=True, depth=dt_)
                ok = bool(torch.equal(band, ref)) and bool(ref[..., 3].max() > 0)
                del ref
                torch.cuda.empty_cache()
            dreg = "temporal (culled)" if sd["frames_culled"] * 2 > sd["frames"] else ("front slab" if sd["frames_slab"] * 2 > sd["frames"] else "one pass")
            depth_leg[name] = {"value": k4 / ddt, "unit": "frames/sec", "ms_per_step": ddt / k4 * 1e3, "steps": k4, "regime": dreg,
                               "kernel": "k_blend<false>" if dt_ is None else "k_blend<true>",
                               "blend_launch_ms": sd["blend_ms_total"] / max(1, sd["blend_launches"]), "blend_launches_timed": sd["blend_launches"],
                               "frames_culled": sd["frames_culled"], "frames_repaired": sd["frames_repaired"], "frames_slab": sd["frames_slab"],
                               "n_visible": sd["n_visible"], "pairs": sd["pairs_total"], "clusters_kept": sd["clusters_kept"],
                               "pairs_consumed_per_frame": sd["blend_pairs_consumed_total"] / max(1, sd["frames"]),
                               "cull_dilate": sd["cull_dilate"], "policy_bits": sd["policy_bits"],
                               "depth_culling_active": bool(sd["policy_bits"] & 32),
                               "opaque_pixels_frac": float((dt_ < 1.0).float().mean()) if dt_ is not None else 0.0,
                               "last_frame_bit_identical_to_unculled": ok}
            de.close()
            del dt_, de
            torch.cuda.empty_cache()
        if depth_leg["plain"]["value"] > 0:
            for name in ("far_plane", "occluder"):
                depth_leg[name]["vs_plain"] = depth_leg[name]["value"] / depth_leg["plain"]["value"]
        depth_leg["occluder_sphere"] = dict(OCCLUDER, scale=depth_scale,
                                            note="an opaque sphere on the view axis (it follows the camera, so every frame of the orbit sees the same depth buffer)")
        depth_leg["note"] = ("gsr_render_depth with a device depth buffer: fragment survives iff its quad's window depth <= depth[pixel] (one depth per quad); "
                             "plain = gsr_render on the same cameras in the same loop")
    # extra leg (single GPU): the BOUNDARY as the reference drives it.  C4 as three registry entries (three details) behind the nine verbs
    # of GSplatRenderer: per redraw includeInRenderPass x 3 -> generateRenderGeometry -> render -> postRender (src/DM_GSplatHook.C:30-39,
    # src/GR_GSplat.C:485-492), one foreign call per redraw (gsplat_renderer_redraw).  `via_shim`: the set does not change -- what the
    # verbs (string ids, a staging plan rebuilt and compared every redraw) cost beside the direct line.  `restage`: every frame brings
    # a new cache version of the three details (an animated sequence: registerUpdate x 3 -> re-stage -> frame), with the upload's stages.
    shim_leg = None
    if world == 1 and not args.no_extra_legs and args.emulate_shard <= 1 and args.depth == "none":
        E = pkg.engine
        R = pkg.GSplatRenderer(dev_index)
        R_eng = pkg.load_library().gsplat_renderer_engine(R.h)
        pkg.engine._check(pkg.load_library().gsr_set_stream(R_eng, stream.cuda_stream))
        for opt, val in ((E.OPT_TIMING_EVERY, 1000), (E.OPT_OCCLUSION_CULL, args.cull), (E.OPT_FRONT_SLAB, args.front_slab)):
            pkg.engine._check(pkg.load_library().gsr_set_option(R_eng, opt, val))
        R.setSphericalHarmonicsOrder(order)
        cuts = [0, splats.n // 3, 2 * (splats.n // 3), splats.n]
        parts = [splats.subset(slice(cuts[k], cuts[k + 1])) for k in range(3)]
        origin0 = np.zeros(3, np.float32)
        def register(version):
            return [R.registerUpdate(0x1000 + 16 * k, (version, 0, 0, 0), 0, parts[k], splatOrigin=origin0) for k in range(3)]
        ids = register(1)
        cams_py = [pkg.scenes.config_camera(args.config, pkg.camera, W, H, order, orbit_frame(i, args.jump_every)) for i in range(args.warmup + args.steps)]
        ctxs = [R.context(cp, band.data_ptr(), True) for cp in cams_py]
        k5 = min(100, args.steps)
        for i in range(min(args.warmup + 5, len(ctxs))):
            R.redraw(ids, ctxs[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k5):
            R.redraw(ids, ctxs[args.warmup + i])
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t0) / k5
        ok_s = None
        if not args.no_verify:
            ref = reference_frame(pkg, torch, dev_index, stream, splats, cams[args.warmup + k5 - 1], W, H, exact=True)
            # (the shim finds the camera position itself -- the inverse of the view matrix in double, src/GSplatRenderer.C:556-562 -- so the
            #  sort keys may differ in the last bit from the ones of the harness's numpy inverse: compared to the reference within 1e-3)
            ok_s = bool((band - ref).abs().max() <= 1e-3) and bool(ref[..., 3].max() > 0)
            del ref
        # the same loop through the direct door, for the ratio
        for i in range(min(args.warmup + 5, len(cams))):
            eng.render_struct_to_device(cams[i], band.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k5):
            eng.render_struct_to_device(cams[args.warmup + i], band.data_ptr())
        torch.cuda.synchronize()
        dtd = (time.perf_counter() - t0) / k5
        stv = R.engine_stats()
        shim_leg = {"via_shim": {"value": 1.0 / dts, "unit": "frames/sec", "ms_per_step": dts * 1e3, "steps": k5, "direct_same_loop": 1.0 / dtd,
                                 "vs_direct": dtd / dts, "stagings": int(R.query(R.Q_STAGING_COUNT)), "frames_culled": stv["frames_culled"],
                                 "frame_within_1e-3_of_unculled": ok_s,
                                 "note": "GSplatRenderer verbs (three registry entries, a staging plan rebuilt and compared every redraw) over the same kernels; "
                                         "direct_same_loop = Engine.render_struct_to_device on the same cameras right behind it"}}
        # restage: a new cache version of every detail per frame
        k6 = min(12, args.steps)
        ups, t_reg, t_frame = [], 0.0, 0.0
        for i in range(2 + k6):
            torch.cuda.synchronize()
            ta = time.perf_counter()
            ids = register(2 + i)
            tb = time.perf_counter()
            R.redraw(ids, ctxs[args.warmup + (i % max(1, args.steps))])
            torch.cuda.synchronize()
            tc = time.perf_counter()
            if i >= 2:
                t_reg += tb - ta; t_frame += tc - tb
                ups.append(R.engine_stats()["upload_ms"])
        um = np.median(np.asarray(ups), axis=0)
        shim_leg["restage"] = {"value": k6 / (t_reg + t_frame), "unit": "frames/sec", "steps": k6, "n_splats": int(splats.n), "entries": 3,
                               "ms_per_step": (t_reg + t_frame) / k6 * 1e3, "register_update_ms": t_reg / k6 * 1e3,
                               "restage_and_frame_ms": t_frame / k6 * 1e3,
                               "upload_ms": {"total_begin_to_end": float(um[3]), "host_to_device": float(um[0]), "bbox_morton_sort": float(um[1]),
                                             "pack_to_storage_order_and_cluster_bounds": float(um[2]),
                                             "device_side": float(um[1] + um[2])},
                               "bytes_host_to_device": int(splats.n) * (12 + 4 + 6 + 6 + 8 + (96 if splats.shx is not None else 0)),
                               "H2D_GBps": int(splats.n) * (12 + 4 + 6 + 6 + 8 + (96 if splats.shx is not None else 0)) / (um[0] * 1e-3) / 1e9 if um[0] > 0 else None,
                               "stagings": int(R.query(R.Q_STAGING_COUNT)),
                               "note": "every frame: registerUpdate x 3 with a new cache version (the reference re-stages whenever a version changes, "
                                       "src/GSplatRenderer.C:246-265, 322-532) -> includeInRenderPass x 3 -> generateRenderGeometry (gsr_upload_begin / "
                                       "append x 3 / end) -> render -> postRender; the first frame of a new cloud has no horizons (a front-slab or one-pass frame). "
                                       "upload_ms: medians over the leg's uploads (gsr_stats.upload_ms)"}
        R.close()
    # extra leg (informational): the same K steps with two frames in flight
    pipelined = None
    if args.pipelined and args.frames_in_flight == 1 and not args.no_extra_legs:
        eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 2)
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el2], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        pipelined = {"frames_in_flight": 2, "value": args.steps / el2, "unit": "frames/sec",
                     "ms_per_step": el2 / args.steps * 1e3,
                     "note": "GSR_OPT_FRAMES_IN_FLIGHT=2: frame f+1's front end overlaps frame f's blend kernel"}
        eng.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, args.frames_in_flight)
    # extra leg (informational, single GPU, the default run of the headline config only): the other BASELINE configs, the same
    # library and options, serial frames -- so that their frame rates are on the driver's record too, not only in DESIGN.md
    other = None
    if world == 1 and args.config == "C4" and args.splats is None and not args.no_extra_legs and not args.no_other_configs and args.emulate_shard <= 1:
        other = {}
        for oc in ("C1", "C2", "C3", "C5", "T1", "S1", "R1"):
            osp, ocfg = pkg.scenes.make_config(oc)
            oW, oH, oord = ocfg["width"], ocfg["height"], ocfg["sh_order"]
            oe = pkg.Engine(dev_index)
            oe.set_stream(stream.cuda_stream)
            oe.set_option(pkg.engine.OPT_FRAMES_IN_FLIGHT, 1)
            oe.set_option(pkg.engine.OPT_TIMING_EVERY, 1000)
            oe.upload(osp)
            oband = torch.zeros((oH, oW, 4), dtype=torch.float32, device="cuda")
            ocams = [pkg.engine.camera_struct(pkg.scenes.config_camera(oc, pkg.camera, oW, oH, oord, i)) for i in range(110)]
            for i in range(10):
                oe.render_struct_to_device(ocams[i], oband.data_ptr())
            torch.cuda.synchronize()
            oe.stats_reset()
            t0 = time.perf_counter()
            for i in range(10, 110):
                oe.render_struct_to_device(ocams[i], oband.data_ptr())
            torch.cuda.synchronize()
            odt = (time.perf_counter() - t0) / 100
            ost = oe.stats()
            oreg = "temporal (culled)" if ost["frames_culled"] * 2 > ost["frames"] else ("front slab" if ost["frames_slab"] * 2 > ost["frames"] else "one pass")
            other[oc] = {"value": 1.0 / odt, "unit": "frames/sec", "ms_per_step": odt * 1e3, "steps": 100, "n_splats": int(osp.n),
                         "width": oW, "height": oH, "regime": oreg, "frames_culled": ost["frames_culled"], "frames_slab": ost["frames_slab"],
                         "frames_repaired": ost["frames_repaired"], "n_visible": ost["n_visible"], "pairs": ost["pairs_total"]}
            if oc == "C5":
                other[oc]["host_target"] = host_target_leg(oe, ocams, oW, oH)
            oe.close()
            del oband, osp
            torch.cuda.empty_cache()
    # extra leg: the PCIe-inclusive rate -- gsr_render into a HOST buffer, what a caller without GL interop gets (never `value`)
    host_leg = None
    if world == 1 and not args.no_extra_legs and args.emulate_shard <= 1 and not args.via_multi:
        host_leg = host_target_leg(eng, cams[:min(len(cams), 50)], W, H)
    # blend-kernel roofline, measured with HIP events on the kernel's own stream
    shards = args.emulate_shard if (world == 1 and args.emulate_shard > 1) else world
    srank = (args.emulate_rank % shards) if (world == 1 and args.emulate_shard > 1) else rank
    own_px = sum(min(16, H - r * 16) for r in pkg.multigpu.owned_tile_rows(H, srank, shards, args.shard_layout)) * W
    regime = "culled" if st["frames_culled"] * 2 > st["frames"] else ("slab" if st["frames_slab"] * 2 > st["frames"] else "unculled")
    tj, traffic_from = load_traffic(regime, world == 1 and args.config == "C4" and args.splats is None)
    hbm = measured_hbm_peak() if rank == 0 else None
    peak_meas = hbm["peak_GBps"] if hbm else None
    roofline = blend_roofline(st, own_px, regime, hbm, tj, traffic_from)
    stages = {k: st_stage[k] for k in ("ms_preprocess", "ms_depth_sort", "ms_emit", "ms_tile_sort", "ms_blend", "ms_total")} if st_stage else {}
    stages["note"] = ("one frame of the extra leg with events around every stage: ms_emit = binning count+scans, ms_tile_sort = "
                      "binning placement + lazy colour pass")
    # the kernel next to it: k_preprocess (stage 0 of the frame is exactly this one kernel).  Algorithmic bytes per splat at SH
    # order 3: 32 (geometry) read by every splat; one that stays also reads 96 (colour; eager mode only) and writes 48 (record)
    # + 12 (key, payload, compacted per workgroup); a dropped one writes nothing
    roofline_k1 = None
    if st_stage and st_stage["stage_frames"] > 0 and world == 1:
        k1_ms = st_stage["stage_ms_total"][0] / st_stage["stage_frames"]
        nvis = st_stage["n_visible"]
        # did K1 shade in the frames of THIS leg?  (frames_lazy counts the frames whose K1 left the colours pending; round 4 asked
        # lazy_colours_total, which is a running device counter and said "lazy" for a leg whose K1 shaded every frame)
        lazy_on = st_stage["frames_lazy"] * 2 > st_stage["frames"]
        k1_regime = "culled" if st_stage["frames_culled"] * 2 > st_stage["frames"] else ("slab" if st_stage["frames_slab"] * 2 > st_stage["frames"] else "unculled")
        # eager: the colour halves are read for every visible splat; lazy: K1 reads geometry only (colours: k_colour_prefix)
        col_b = 0 if lazy_on else {0: 16, 1: 32, 2: 64, 3: 96}[order if splats.shx is not None else 0]
        # stage 0 = k_cluster_cull + k_preprocess.  Algorithmic bytes: 32 B of bounds per cluster; 32 B of geometry for every splat of a
        # SURVIVING cluster (64 each); a splat that stays also reads its colour halves (eager mode only) and writes 48 (record) + 12
        # (key, payload); 4 B per cluster for the ordered survivor list (written and read)
        ckept, call = st_stage["clusters_kept"], st_stage["clusters_total"]
        # key + payload (12 B) go either to the head of the workgroup's 256 slots (large frames: the first radix pass gathers them) or
        # straight into the small-frame sort's bucket regions (culled frames, since round 4: + one 4-byte atomic on the bucket counter)
        k1_scatters = k1_regime != "unculled"
        k1_bytes = call * 32 + ckept * 64 * 32 + nvis * (col_b + 48 + 12 + (4 if k1_scatters else 0)) + ckept * 8
        k1_traffic = None
        try:
            k1_traffic = float(tj["k_preprocess"]["hbm_bytes_per_launch"]) if tj is not None else None
        except Exception:
            k1_traffic = None
        k1_gbps = k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        roofline_k1 = {"bound": "hbm", "kernel": "k_preprocess", "achieved": k1_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                       "frac": k1_gbps / HBM_PEAK_GBPS, "peak_measured": peak_meas, "frac_of_measured": (k1_gbps / peak_meas) if peak_meas else None,
                       "traffic": k1_traffic, "avg_launch_ms": k1_ms, "regime": k1_regime, "colour": "lazy" if lazy_on else "eager",
                       "algorithmic_bytes_per_launch": k1_bytes,
                       "bytes_per_kept_splat": col_b + 48 + 12 + (4 if k1_scatters else 0),
                       "splats_kept": int(nvis), "clusters_kept": int(ckept), "clusters": int(call),
                       # of the covariance chains K1 runs (one per splat of a surviving cluster), how many end in a splat that is dropped
                       # (off-screen rect, alpha support shrunk to nothing, behind every horizon its rect reaches): what an earlier test could save
                       "covariance_chains": int(ckept * 64), "chains_that_end_in_a_dropped_splat_frac": (1.0 - nvis / (ckept * 64.0)) if ckept else None,
                       "note": "k_cluster_cull + k_preprocess (stage 0 of the frame); " +
                               ("lazy colour: geometry only" if lazy_on else "eager colour: + the colour halves of every splat that stays") +
                               ("; key + payload dropped into the small-frame sort's buckets by K1 itself (12 B + a 4-byte atomic)" if k1_scatters else "") +
                               "; K1 is FP32-issue bound (the 250-instruction covariance chain for every clip-visible splat of a surviving cluster)"}

    if rank == 0:
        line = {
            "metric": "frames/sec at 1920x1080 + achieved HBM GB/s (blend kernel)" if (W, H) == (1920, 1080)
            else f"frames/sec at {W}x{H} + achieved HBM GB/s (blend kernel)",
            "value": (args.steps / elapsed) if verified is not False else None,   # (a frame rate of wrong pixels is not a number)
            "unit": "frames/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "value_min": min(windows) if windows else None, "value_max": max(windows) if windows else None,
            "value_windows": windows,
            "latency": latency,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.ply else "INRIA PLY capture (--ply)",
            "config": {"workload": workload_name(args, splats, cfg, order, W, H),
                       "parallelism": (f"tile-row shard x{world}, " + ("contiguous bands" if args.shard_layout else "interleaved rows") + ", one process per GPU")
                       if world > 1 else "single GPU", "gather": gather,
                       "n_splats": splats.n, "width": W, "height": H, "frames_in_flight": args.frames_in_flight},
            "roofline": roofline,
            "roofline_preprocess": roofline_k1,
            "stages_ms_last_frame": stages,
            "n_visible": st["n_visible"],
            "cluster_culling": {"clusters": st["clusters_total"], "kept_last_frame": st["clusters_kept"], "splats_per_cluster": 64},
            "lazy_colour": {"mode": args.lazy, "active": st["lazy_colours_total"] > 0, "colours_evaluated_per_frame": st["lazy_colours_total"] / max(1, st["frames"]),
                            "visible_splats": st["n_visible"],
                            "fallback_tiles_last_frame": st["lazy_redo_tiles"]},
            "occlusion_culling": {"enabled": bool(args.cull), "policy_bits": st["policy_bits"], "dilate_tiles": st["cull_dilate"], "holdoff_frames": st["cull_holdoff"], "frames_culled": st["frames_culled"], "frames_repaired": st["frames_repaired"], "frames_slab": st["frames_slab"],
                                  "frames": st["frames"], "without": unculled,
                                  "note": "splats whose tile rect lies wholly behind the previous frame's per-tile depth horizons get no colour, no record, and "
                                          "no place in the sort and the lists; every culled frame verifies itself and is rendered again without culling if a horizon "
                                          "broke (frames_repaired; those frames are inside the timed region)"},
        }
        if host_leg is not None:
            line["host_target"] = host_leg
        if depth_leg is not None:
            line["depth_tested"] = depth_leg
        if shim_leg is not None:
            line["boundary"] = shim_leg
        if args.depth != "none":
            line["config"]["depth_tested"] = args.depth
        if cold is not None:
            line["cold_frames"] = cold
        if pipelined is not None:
            line["pipelined"] = pipelined
        if other is not None:
            line["other_configs"] = other
        if verified is not None:
            # the last frame of the timed region against the same camera from a context that culls nothing (and is not sharded)
            line["timed_frame_bit_identical"] = verified
            if world > 1:
                line["sharded_frame_bit_identical"] = verified
        if rccl_info is not None:
            line["rccl_rank_of_root"], line["rccl_comm_count"] = rccl_info
        if world == 1 and not args.no_cpu_baseline:
            oracle = ge.load_oracle()
            cam0 = pkg.camera.make_camera(W, H, sh_order=order, frame=0)
            line["cpu_baseline"] = cpu_baseline(oracle, splats, cam0, pkg, args.cpu_seconds)
        print(json.dumps(line))
        sys.stdout.flush()
    eng.close()
    if world > 1:
        # every rank leaves with the same code: a run that rendered wrong pixels is a failed run
        t = torch.tensor([0 if verified is not False else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bad = bool(t.item())
        dist.destroy_process_group()
        if bad:
            sys.exit(3)
    elif verified is False:
        sys.exit(3)


if __name__ == "__main__":
    main()

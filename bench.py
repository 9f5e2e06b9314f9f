#!/usr/bin/env python3
"""bench.py — ray-surface intersections/s of the OpticalTable.ray_tracing hot path on MI355X.

Headline workload (BASELINE.json configs[1], SURVEY.md §8d cfg 2): 1e6 rays from a point source at the focus of
Lens([5,0,0], f=5, r=1) followed by MirrorPair([10,0,0], 4, 4); S = 3 leaf surfaces; cap 5 segments (every ray
uses exactly 5); fp64; inputs resident in HBM before the timed region.  A "step" is one trace of one whole batch
(one launch of k_trace_fused<double>); consecutive steps rotate through several distinct input / output batches
(> 256 MiB of inputs in total) so that no step can be served from the Infinity Cache.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5] [--rays R]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run`, before this process has touched the GPU); under a launcher (WORLD_SIZE set) the
process is one rank.  cfg 2 / cfg 3 scale weakly (every rank traces the config's full size); cfg 4 / cfg 5 name a
TOTAL (6.4e8 ray-wavelength pairs, 1e8 rays) that is sharded over the ranks (`"scaling": "strong"`).  No
collective in the data path; the single end-of-job gather of the per-ray final state over RCCL is timed
separately (`gather_ms`).  The default N = 1 run also measures the other BASELINE configs at the size one GPU sees
(`configs`), a >= 6 s sustained loop (`sustained`), the cold-start figure (`cold`), call latencies at the
reference's own sizes (`latency`) and the CPU baseline.

Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES = {"f64": 104, "f32": 56}  # 12 reals + two int32 per ray record and per segment record
PRELOAD_LAUNCHES = 500  # cfg 2 only: ~60 ms of load so the chip reaches its steady clocks before the W warmup steps
# size ONE GPU sees in the configuration BASELINE.json quotes (cfg 4: the 4-GPU shard; cfg 5: the 8-GPU shard)
QUOTED_SHARD = {"cfg3": 10_000_000, "cfg4": 160_000_000, "cfg5": 12_500_000}
# most rays one rank takes (HBM: rays + the full [k][ray] history must fit 288 GB with headroom)
MAX_PER_RANK = {"cfg2": 50_000_000, "cfg3": 50_000_000, "cfg4": 320_000_000, "cfg5": 25_000_000}
# the reference itself (pure Python), single thread, measured by importing /root/reference in the build container
# (it cannot travel to the GPU box): segments/s per config
REFERENCE_PYTHON_SEG_S = {"cfg2": 3.0e3, "cfg3": 434.0, "cfg4": 2.7e3, "cfg5": 593.0}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=None, help="rays per GPU (default: from the workload and the world size)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2 = the bench line; the others are the remaining BASELINE configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the cfg3/cfg4/cfg5 survey of the default run")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the small-N call latencies of the default run")
    ap.add_argument("--sustained-seconds", type=float, default=6.0, help="length of the back-to-back loop after the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU time spent on the oracle baseline (bounded sample)")
    ap.add_argument("--sharded-configs", default="auto",
                    help="with --gpus N > 1 on the default workload: the configs BASELINE quotes multi-GPU, at their quoted totals, as "
                         "`sharded_configs` of the line: auto (cfg4 on 2 / 4 ranks, cfg5 on 8), none, or a comma list")
    ap.add_argument("--sharded-rays", type=int, default=None, help="rays per rank for --sharded-configs (rehearsals on one card)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real thing) or gloo (rehearsal: ranks may share a card)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (rehearsal of the RCCL calls on a 1-GPU box)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / sharding / gather rehearsal without a GPU: no trace, fabricated per-ray state (tests)")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """Start `args.gpus` ranks of this script and wait for them.  Runs before anything here has initialised HIP
    (importing torch does not), as a CHILD process: a process that touched the GPU must never exec another."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def cpu_baseline(table, batch_host, max_seg, n_leaves, label, budget_s=10.0):
    """The oracle (CPU restatement, single thread) on the same workload, timed on this box's host."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as orc

    orc.build()
    scene = table.compile()
    n = len(batch_host["ox"])
    sample = {k: v[: min(n, 200_000 if max_seg <= 8 else 20_000)] for k, v in batch_host.items()}
    t0 = time.perf_counter()
    segs = 0
    rays = 0
    while True:
        out = orc.trace(scene, sample, max_trace_num=max_seg)
        segs += len(out["ray"])
        rays += len(sample["ox"])
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    base = {"value": segs * n_leaves / dt, "unit": "ray-surface intersections/s", "cores": 1, "kind": "port",
            "sample": f"{rays} rays of the same {label} batch ({segs} segments) in {dt:.2f} s, C oracle, 1 thread; "
                      f"host has {os.cpu_count()} cores",
            "segments_per_s": segs / dt}
    # the same restatement on several host cores (rays are independent: every thread traces its own slice of the
    # sample; the oracle is stateless C and ctypes releases the GIL) — BASELINE.md §4 asks for both figures
    threads = max(1, min(16, (os.cpu_count() or 1)))
    m = len(sample["ox"])
    per = max(1, min(25_000, m))
    shards = [{k: v[(i * per) % max(m - per + 1, 1):(i * per) % max(m - per + 1, 1) + per] for k, v in sample.items()}
              for i in range(threads)]
    deadline = [0.0]

    def work(shard):
        done = 0
        while True:
            done += len(orc.trace(scene, shard, max_trace_num=max_seg)["ray"])
            if time.perf_counter() > deadline[0]:
                return done

    with ThreadPoolExecutor(threads) as pool:
        # untimed warm-up: the first multi-threaded seconds of a process run nearly serialised (page faults of
        # fresh output arrays until glibc's mmap threshold has grown)
        deadline[0] = time.perf_counter() + min(2.0, budget_s / 4)
        list(pool.map(work, shards))
        deadline[0] = time.perf_counter() + budget_s / 3
        t1 = time.perf_counter()
        total = sum(pool.map(work, shards))
        dt_mt = time.perf_counter() - t1
    base["multithread"] = {"value": total * n_leaves / dt_mt, "cores": threads, "segments_per_s": total / dt_mt,
                           "sample": f"{total} segments in {dt_mt:.2f} s on {threads} threads"}
    ref = REFERENCE_PYTHON_SEG_S.get(label)
    if ref:
        base["reference_python"] = {
            "value": ref * n_leaves, "segments_per_s": ref, "cores": 1, "kind": "reference",
            "provenance": "the reference package itself (pure Python), imported from /root/reference in the build "
                          "container (8 host cores, Python 3.10, numpy 2.2), single thread, same scene and generator at "
                          "400-3200 rays; BASELINE.md §2 / SURVEY.md §6.  Not measured on this box: the reference does "
                          "not travel to it."}
    return base


def make_batch(oa, wl, n, rank, device, seed_shift=0):
    """This rank's rays of workload `wl` as a device-resident RayBatch."""
    import numpy as np
    from optable_amd import workloads as W
    from optable_amd.batch import RayBatch

    if wl.name == "cfg4":  # base rays on the host, the 64 wavelength copies made on the device (ray.py:428-445)
        nb = max(n // W.CFG4_WAVELENGTHS, 1)
        o, d, _ = W.cfg4_rays(nb, 4 + rank + seed_shift, n_wavelengths=1)
        base = RayBatch.from_arrays(o, d, wavelength=W.WL, q=1j * np.pi * W.W0**2 / W.WL, precision=wl.precision, device=device)
        return base.multiplexed_in_wavelength(np.linspace(400e-7, 1100e-7, W.CFG4_WAVELENGTHS))
    o, d, lam = wl.rays(n, rank + seed_shift)
    return RayBatch.from_arrays(o, d, wavelength=lam, q=1j * np.pi * W.W0**2 / lam, precision=wl.precision, device=device)


def sample_device_clocks(out):
    """`rocm-smi --showclocks --showpower --json` of the first card, from a child process, into `out` (best effort: a
    diagnostic next to the sustained figure, never a reason for the bench to fail)."""
    import subprocess
    try:
        time.sleep(0.3)  # let the loop get going
        res = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=2.5)
        text = res.stdout[res.stdout.index("{"):]
        card = next(iter(json.loads(text).values()))
        for key, val in card.items():
            k = key.lower()
            digits = "".join(ch for ch in str(val) if ch.isdigit() or ch == ".")
            if not digits:
                continue
            if k.startswith("sclk clock speed"):
                out["sclk_mhz"] = float(digits)
            elif k.startswith("mclk clock speed"):
                out["mclk_mhz"] = float(digits)
            elif k.startswith("fclk clock speed"):
                out["fclk_mhz"] = float(digits)
            elif "power (w)" in k:
                out["package_power_w"] = float(digits)
    except Exception:  # noqa: BLE001 - diagnostic only
        pass


def kernel_name(scene, wl):
    heavy = scene.n_nodes >= 24
    return ("k_trace_rolling" if heavy else "k_trace_fused") + ("<double>" if wl.precision == "f64" else "<float>")


def committed_counters(name):
    """SQ-counter summary of this workload committed under profiles/ (tools/profile_r04.sh; the newest round that has
    one), or None."""
    for rnd in ("r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}_counters.json")
        if os.path.exists(path):
            rec = json.load(open(path))
            rec["source"] = os.path.relpath(path, ROOT)
            return rec
    return None


def survey_config(oa, eng, name, device):
    """One of the other BASELINE configs at the size one GPU sees in the quoted configuration: device time per
    trace (library HIP events around every launch), segments/s, algorithmic GB/s."""
    import torch
    from optable_amd import workloads as W
    from optable_amd.batch import SegmentBatch

    wl = W.baseline_workloads(oa)[name]
    n = QUOTED_SHARD[name]
    table = oa.OpticalTable()
    table.add_components(wl.components())
    scene = table.compile()
    eng.upload(scene)
    batch = make_batch(oa, wl, n, 0, device)
    n = batch.n
    heavy = scene.n_nodes >= 24
    b = BYTES[wl.precision]
    reps = 5 if name == "cfg3" else 3

    default_call = {}

    def measure(layout):
        """device time per trace with the output in `layout`: [k][ray] slots (ot_trace_*) or the dense append-order list
        (ot_trace_append_*) in the block the DEFAULT call allocates: `table.trace_batch(batch, K)` with no layout and no
        capacity sizes it from a 1 % sample of the batch (Engine._append_estimate), not from a second full trace"""
        torch.cuda.empty_cache()
        if layout == "append":
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = table.trace_batch(batch, wl.max_segments, scene=scene)  # first call: sample trace + allocation + trace
            valid = int(out.n_valid)
            first_ms = (time.perf_counter() - t0) * 1e3
            assert out.layout == "append", out.layout
            records = int(out.count.abs().sum().item())
            del out
            t0 = time.perf_counter()
            out = table.trace_batch(batch, wl.max_segments, scene=scene)  # later calls: the estimate is remembered per scene
            valid = int(out.n_valid)
            next_ms = (time.perf_counter() - t0) * 1e3
            default_call.update(first_call_ms=first_ms, next_call_ms=next_ms, capacity_slots=out.capacity, records=records,
                                capacity_over_records=out.capacity / max(records, 1), slots_claimed=valid,
                                note="wall clock of table.trace_batch(batch, K, scene=compiled) with no layout and no capacity, read-back of "
                                     "the slot count included: first call = 1 % sample trace + block allocation + trace; next = trace into "
                                     "a fresh block of the remembered estimate")
        else:
            out = SegmentBatch(n * wl.max_segments, wl.precision, batch.device)
        t_load = time.perf_counter()
        while time.perf_counter() - t_load < 0.06:  # clocks up
            eng.trace(batch, wl.max_segments, out=out, layout=layout)
            torch.cuda.synchronize()
        eng.timing(True)
        for _ in range(reps):
            eng.trace(batch, wl.max_segments, out=out, layout=layout)
        ms, cnt = eng.timing_read()
        eng.timing(False)
        segs = int(out.count.abs().sum().item())
        slots = out.capacity if layout == "slots" else int(out.n_valid)
        launch = eng.last_launch()
        del out
        return ms / cnt / 1e3, cnt, segs, slots, launch

    # Every config in the layout its kernels write fastest (what trace_batch(layout="auto") gives: tiles for light scenes,
    # the dense append-order list for heavy ones) and, beside it, in the [k][ray] slot arrays of ot_trace_*.
    def measure_tiled():
        torch.cuda.empty_cache()
        out = SegmentBatch(n * wl.max_segments, wl.precision, batch.device, tiled=True)
        t_load = time.perf_counter()
        while time.perf_counter() - t_load < 0.06:  # clocks up, as for the other layouts
            eng.trace(batch, wl.max_segments, out=out, layout="tiled")
            torch.cuda.synchronize()
        eng.timing(True)
        for _ in range(reps):
            eng.trace(batch, wl.max_segments, out=out, layout="tiled")
        ms, cnt_t = eng.timing_read()
        eng.timing(False)
        segs_t = int(out.count.abs().sum().item())
        launch_t = eng.last_launch()
        del out
        return ms / cnt_t / 1e3, cnt_t, segs_t, n * wl.max_segments, launch_t

    ts, cnts, segs, slots_s, launch_s = measure("slots")
    alg = n * b + segs * b
    if heavy:
        t, cnt, segs_a, slots, launch = measure("append")
        pooled = bool(launch.get("pair_queue", 0) & 16)  # curved-surface scenes, fp32: the workgroup-wide block pool
        kern = ("k_trace_pool" if pooled else "k_trace_rolling") + ("<double>" if wl.precision == "f64" else "<float>")
        layout = "append: dense list in append order, a stable sort by ray is the reference's order (ot_trace_append_*)"
        assert segs_a == segs
    else:
        t, cnt, _, slots, launch = measure_tiled()
        kern = kernel_name(scene, wl)
        layout = "tiled: slot k * n_rays + i in tile / 64, lane % 64 (ot_trace_tiled_*)"
        chosen = "tiled" if t <= ts else "slots"  # (both were just measured on this very workload: the faster one is the config's figure)
        if chosen == "slots":  # this device streams the 14 arrays faster: they are the config's figure, the tiles the companion
            (t, cnt, slots, launch), (ts, cnts, slots_s, launch_s) = (ts, cnts, slots_s, launch_s), (t, cnt, slots, launch)
            layout = "slots: segment k of ray i at k * n_rays + i (ot_trace_*)"
    rec = {"workload": wl.label, "rays": n, "dtype": wl.precision, "kernel": kern, "leaf_surfaces": scene.n_leaves,
           "layout": layout, "output_slots": slots, "launches": cnt, "ms_per_trace": t * 1e3, "segments_per_ray": segs / n,
           "segments_per_s": segs / t, "intersections_per_s": segs * scene.n_leaves / t, "algorithmic_gbs": alg / t / 1e9,
           "hbm_frac": alg / t / 1e9 / HBM_PEAK_GBS, "launch": launch,
           "bound": "hbm" if not heavy else "valu (S >= 24, SURVEY.md §8d): the HBM fraction is for comparison"}
    if heavy:
        rec["holes"] = slots - segs
        rec["default_call"] = default_call
    other_name = ("tiled: 64-slot tiles (ot_trace_tiled_*)" if (not heavy and layout.startswith("slots")) else "slots: segment k of ray i at k * n_rays + i (ot_trace_*)")
    rec["other_layout"] = {"layout": other_name, "kernel": kernel_name(scene, wl),
                           "output_slots": slots_s, "launches": cnts, "ms_per_trace": ts * 1e3, "segments_per_s": segs / ts,
                           "intersections_per_s": segs * scene.n_leaves / ts, "algorithmic_gbs": alg / ts / 1e9,
                           "hbm_frac": alg / ts / 1e9 / HBM_PEAK_GBS, "launch": launch_s}
    counters = committed_counters(name)
    if counters:
        rec["sq_counters"] = counters
    del batch
    torch.cuda.empty_cache()
    return rec


def survey_branching(oa, eng, device):
    """cfg 4 with reflectivity 0.2 (SURVEY.md §8d branching variant): 1.28e7 ray trees x 12 segments through the default
    call (one lane-per-tree launch) and through the generation loop; device time from the library's HIP events."""
    import numpy as np
    import torch
    from optable_amd import workloads as W
    from optable_amd.batch import RayBatch

    table = oa.OpticalTable()
    table.add_components(W.cfg4_components(oa, reflectivity=0.2))
    scene = table.compile()
    eng.upload(scene)
    nb = 200_000
    o, d, _ = W.cfg4_rays(nb, 4, n_wavelengths=1)
    base = RayBatch.from_arrays(o, d, wavelength=W.WL, q=1j * np.pi * W.W0**2 / W.WL, precision="f64", device=device)
    batch = base.multiplexed_in_wavelength(np.linspace(400e-7, 1100e-7, W.CFG4_WAVELENGTHS))
    def timed(fn):
        for _ in range(2):  # warm: output arrays (the allocator's cache), scratch and generation buffers
            fn()
        torch.cuda.synchronize()
        eng.timing(True)
        t0 = time.perf_counter()
        segs = fn()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms, launches = eng.timing_read()
        eng.timing(False)
        return segs, ms, launches, wall

    # the default call (Engine.trace_branching: one launch, a lane per tree with its FIFO in LDS) ...
    segs, ms, launches, wall = timed(lambda: eng.trace_branching(batch, 12))
    per_tree = eng.last_launch()["kernel"] == 4  # (k_trace_trees; its output: [k][tree] slots or the dense append list, both with a count per tree)
    n_seg = int(segs.count.abs().sum()) if per_tree else int(segs.n_valid)
    segs_layout = segs.layout
    tree_launch = dict(eng.last_launch(), plan=eng.trees_plan("f64", 12, batch.n)) if per_tree else None  # workgroups per CU, queue entries in LDS / scratch, claim size
    del segs
    # ... and the generation loop (count / look-ahead, scan, emit per generation): what caps beyond the LDS queues take
    gsegs, gms, glaunches, gwall = timed(lambda: eng.trace_tree(batch, 12, out_capacity=batch.n * 13))
    assert int(gsegs.n_valid) == n_seg
    del gsegs
    # algorithmic bytes: every tree's first ray read once, every processed ray's segment written once; the generation path
    # additionally writes and reads back every child (its own algorithmic figure, as round 3 reported it)
    alg = batch.n * 104 + n_seg * 104
    alg_gen = n_seg * 104 * 2 + (n_seg - batch.n) * 104
    rec = {"workload": "cfg4 with reflectivity 0.2: 1.28e7 ray trees (2e5 rays x 64 wavelengths) x 12 segments, fp64",
           "rays": batch.n, "dtype": "f64", "leaf_surfaces": scene.n_leaves,
           "kernel": f"k_trace_trees<double> (a lane per tree, FIFO in LDS, one launch, {segs_layout} layout)" if per_tree else "k_gen_pass<double> per generation",
           "launches": int(launches), "ms_per_trace": ms, "wall_ms_per_trace": wall * 1e3, "segments_per_ray": n_seg / batch.n,
           "segments_per_s": n_seg / (ms / 1e3), "intersections_per_s": n_seg * scene.n_leaves / (ms / 1e3),
           "algorithmic_gbs": alg / (ms / 1e3) / 1e9, "hbm_frac": alg / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "bound": "hbm",
           "generation_loop": {"kernel": "k_gen_pass<double> (count or look-ahead recount + emit) per generation", "generations": int(glaunches),
                               "ms_per_trace": gms, "wall_ms_per_trace": gwall * 1e3, "algorithmic_gbs": alg_gen / (gms / 1e3) / 1e9,
                               "hbm_frac": alg_gen / (gms / 1e3) / 1e9 / HBM_PEAK_GBS},
           "generation_mismatches": eng.generation_mismatches()}
    if tree_launch:
        rec["launch"] = tree_launch
    counters = committed_counters("cfg4b")
    if counters:
        rec["sq_counters"] = counters
    del batch, base
    torch.cuda.empty_cache()
    return rec


def sharded_config(oa, eng, dist, name, world, rank, device, comm_dev, rays_override=None, min_region_s=0.05):
    """One of the configs BASELINE quotes on several GPUs (cfg 4 on 2 and 4, cfg 5 on 8), at its quoted TOTAL sharded over the
    ranks (contiguous ray shards, scene replicated, no collective in the trace).  Every rank traces its shard in the layout
    the default call uses; the timed region is at least `min_region_s` long (as many launches as that takes, the same on
    every rank) between barrier + synchronize on both sides, the maximum over the ranks counts.  Called by ALL ranks;
    returns the record on rank 0."""
    import math

    import torch
    from optable_amd import workloads as W
    from optable_amd.batch import SegmentBatch

    wl = W.baseline_workloads(oa)[name]
    n_wanted = wl.rays_per_rank(world, rays_override)
    n = min(n_wanted, MAX_PER_RANK[name])
    if name == "cfg4":
        n = max(n // W.CFG4_WAVELENGTHS, 1) * W.CFG4_WAVELENGTHS
    table = oa.OpticalTable()
    table.add_components(wl.components())
    scene = table.compile()
    eng.upload(scene)
    batch = make_batch(oa, wl, n, rank, device)
    n, K = batch.n, wl.max_segments
    first = table.trace_batch(batch, K, scene=scene)  # the default call: sizes an append block from a sample, picks the light layout by the device probe
    layout = first.layout
    segs_step = int(first.count.abs().sum().item())
    out = first if layout == "append" else SegmentBatch(n * K, wl.precision, batch.device, tiled=(layout == "tiled"))
    if out is not first:
        del first
    for _ in range(2):
        eng.trace(batch, K, out=out, layout=layout)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.trace(batch, K, out=out, layout=layout)
    torch.cuda.synchronize()
    mine = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=comm_dev)
    dist.all_reduce(mine, op=dist.ReduceOp.MAX)
    steps = max(3, int(math.ceil(min_region_s / max(float(mine.item()), 1e-6))))
    for attempt in range(4):  # (back-to-back launches run faster than the lone one the step count was guessed from: lengthen until the region is long enough)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.trace(batch, K, out=out, layout=layout)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        longest = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(longest, op=dist.ReduceOp.MAX)
        if float(longest.item()) >= min_region_s or attempt == 3:
            break
        steps = int(steps * min_region_s / max(float(longest.item()), 1e-6) * 1.25) + 1
    row = torch.tensor([dt, float(segs_step), float(n)], dtype=torch.float64, device=comm_dev)
    rows = [torch.zeros_like(row) for _ in range(world)]
    dist.all_gather(rows, row)
    per_rank = [[float(x) for x in r.tolist()] for r in rows]
    launch = eng.last_launch()
    del out, batch
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    dt_max = max(r[0] for r in per_rank)
    segs_total, rays_total = sum(r[1] for r in per_rank), sum(r[2] for r in per_rank)
    b = BYTES[wl.precision]
    return {"workload": wl.label, "n_gpus": world, "rays_total": int(rays_total), "total_requested": wl.total_rays,
            "clamped": bool(n < n_wanted), "rays_per_gpu": n, "dtype": wl.precision, "layout": layout, "steps": steps,
            "timed_region_ms": dt_max * 1e3, "ms_per_step": dt_max / steps * 1e3, "segments_per_s": segs_total * steps / dt_max,
            "value": segs_total * scene.n_leaves * steps / dt_max, "unit": "intersections/s (whole job, all ranks)",
            "algorithmic_gbs_per_gpu": (n * b + segs_step * b) * steps / dt_max / 1e9,
            "ranks": {"ms_per_step_by_rank": [r[0] / steps * 1e3 for r in per_rank], "segments_by_rank": [int(r[1]) for r in per_rank],
                      "rays_by_rank": [int(r[2]) for r in per_rank]},
            "launch": launch}


def survey_latency(oa):
    """Milliseconds per `table.ray_tracing(rays)` call at the sizes the reference is actually used at: its GUI re-runs the whole
    script on every slider event (interact.py:455-457, 20 frames per second: :202-208), with a handful of Ray objects.  Median
    over 200 calls after 20 warm-up calls; Ray objects in, Ray objects out (packing, launch(es), read-back, object rebuild)."""
    import statistics

    import numpy as np
    from optable_amd import workloads as W

    def cfg2_small(ns):
        o, d = W.cfg2_rays(100, 0)
        return W.cfg2_components(ns), [ns.Ray(o[i], d[i], wavelength=W.WL, w0=W.W0, id=i) for i in range(100)]

    # the reference's own time for the same call: cfg 1 measured (11.5 ms for 13 segments); the others from its
    # measured segment rates (2.7e3 / 3.0e3 segments per second), build container, single thread
    cases = [("cfg1 examples/gaussian_beam.py: 6 rays -> 13 segments (non-branching, one launch)", W.gaussian_beam_scene, None, 11.5),
             ("examples/chromatic_aberration.py: 3 rays, every hit branches -> 59 segments (ray trees)", W.chromatic_scene, None, 59 / 2.7e3 * 1e3),
             ("cfg2 scene at 100 rays, cap 5 -> 500 segments", cfg2_small, {"max_trace_num": 5}, 500 / 3.0e3 * 1e3),
             # user-defined parts (optical_component.py:235-240, surfaces.py:5-65): the reference's time measured in the build
             # container on the same scene (workloads.user_parts_scene with the reference's classes: 95-112 ms per call)
             ("user-defined parts: a grating whose interact_local is Python + a parabolic mirror whose surface is a user class, "
              "3 rays -> 120 segments (device search per generation, the user's method per grating hit)", W.user_parts_scene,
              {"max_trace_num": 40}, 105.0)]
    out = []
    for label, make, limit, ref_ms in cases:
        comps, rays = make(oa)
        table = oa.OpticalTable()
        table.add_components(comps)
        res = None
        ts = []
        with contextlib.redirect_stdout(io.StringIO()):  # (a capped trace prints the reference's message, optical_table.py:138-143)
            for _ in range(20):
                table.rays = []
                res = table.ray_tracing(rays, perfomance_limit=limit)
            for _ in range(200):
                table.rays = []
                t0 = time.perf_counter()
                res = table.ray_tracing(rays, perfomance_limit=limit)
                ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        out.append({"workload": label, "rays": len(rays), "segments": len(res), "calls": len(ts), "median_ms_per_call": statistics.median(ts),
                    "p10_ms": ts[len(ts) // 10], "p90_ms": ts[9 * len(ts) // 10], "reference_python_ms": ref_ms,
                    "speedup_vs_reference": ref_ms / statistics.median(ts)})
    real = real_example_latency(oa)
    if real:
        out.append(real)
    return out


def real_example_latency(oa):
    """The reference's LARGEST example as the reference itself ran it (examples/ripa_gen2_lensless.py: a multi-pass cavity of
    micro-mirror arrays, 7,689 leaf surfaces, ONE Gaussian ray, cap 1e5 -> 3,202 segments; 16.8 s there, scene construction
    included).  The reference's objects do not travel to the GPU box; the scene does, as the tables this package's compiler made of
    them, with the ray and every output segment (tests/golden/g27_real_example.npz, tools/make_golden.py real_example_fixture;
    tests/test_gpu_real_example.py compares segment by segment).  Timed: ray upload, the trace (ONE launch: the lane-per-tree
    kernel that reads the image from global memory) and the read-back of all segments in the reference's order."""
    import statistics

    import numpy as np
    import torch
    from optable_amd import abi
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine
    from optable_amd.scene import CompiledScene

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "g27_real_example.npz")
    if not os.path.exists(path):
        return None
    gold = dict(np.load(path))
    scene = CompiledScene.from_tables(gold)
    eng = get_engine()
    cap = int(gold["max_trace_num"][0])
    has_q = gold["in_has_q"]
    flags = np.where(has_q, abi.RAY_HAS_Q, 0).astype(np.int32)
    ts, segments = [], 0
    with eng.lock:
        eng.upload(scene)
        for it in range(8):
            t0 = time.perf_counter()
            batch = RayBatch.from_arrays(gold["in_origin"], gold["in_direction"], wavelength=gold["in_wavelength"], intensity=gold["in_intensity"],
                                         q=np.where(has_q, gold["in_q"], 0j), n_index=gold["in_n"], pathlength=gold["in_pathlength"],
                                         ids=np.arange(len(has_q), dtype=np.int32), device=eng.device, normalize=False)
            batch.flags.copy_(torch.from_numpy(flags))
            counts = torch.zeros((len(scene.limited), 1), dtype=torch.int32, device=eng.device)
            got = eng.trace_branching(batch, cap, counts=counts, distinct_ids=True).to_host(reference_order=True)
            ts.append((time.perf_counter() - t0) * 1e3)
            segments = len(got["ray"])
    ref_ms = float(gold["reference_seconds"][0]) * 1e3
    ok = segments == len(gold["seg_tree"]) and bool(np.allclose(got["length"], gold["seg_length"], rtol=1e-6, atol=1e-9))
    return {"workload": "the reference's largest example as it stands (examples/ripa_gen2_lensless.py: 7,689 leaf surfaces, ONE ray, cap 1e5): "
                        "scene tables + ray from fixture g27, one launch of the lane-per-tree kernel for scenes no LDS holds",
            "rays": int(len(has_q)), "segments": segments, "calls": len(ts) - 2, "median_ms_per_call": statistics.median(ts[2:]),
            "first_call_ms": ts[0], "reference_python_ms": ref_ms, "speedup_vs_reference": ref_ms / statistics.median(ts[2:]),
            "segments_match_the_reference": ok}


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    distributed = world > 1 or args.force_dist

    import numpy as np
    import torch

    if not args.dry_run:
        if args.backend == "gloo":  # rehearsal: several ranks may share one card
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist

        if args.backend == "nccl" and not args.dry_run:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    use_nccl = args.backend == "nccl" and not args.dry_run
    comm_dev = torch.device("cuda", local_rank) if use_nccl else torch.device("cpu")

    import optable_amd as oa
    from optable_amd import dist as odist
    from optable_amd import workloads as W

    wl = W.baseline_workloads(oa)[args.workload]
    n_wanted = wl.rays_per_rank(world, args.rays)
    n = min(n_wanted, MAX_PER_RANK[wl.name])
    clamped = n < n_wanted  # a rank takes at most what fits its HBM with the full history: the total then grows with N
    if wl.name == "cfg4":
        n = max(n // W.CFG4_WAVELENGTHS, 1) * W.CFG4_WAVELENGTHS
    MAX_SEG, prec, bytes_rec = wl.max_segments, wl.precision, BYTES[wl.precision]
    table = oa.OpticalTable()
    table.add_components(wl.components())
    scene = table.compile()
    S_LEAVES = scene.n_leaves
    extra = {}

    if args.dry_run:
        # no device: pretend every ray used its whole budget, keep the launcher, the shard sizes and the collective real
        t0 = time.perf_counter()
        time.sleep(0.01)
        dt = time.perf_counter() - t0
        segs_step, region_ms, kernel_ms, launches, ceil_ms, ceil_n = n * MAX_SEG, dt * 1e3, dt * 1e3, args.steps, 0.0, 0
        local_final = torch.full((12, min(n, 50_000)), float(rank), dtype=torch.float64)  # a bounded stand-in block
        n_inputs = 1
    else:
        from optable_amd.batch import SegmentBatch
        from optable_amd.engine import get_engine

        device = f"cuda:{local_rank}"
        eng = get_engine(local_rank)
        eng.upload(scene)
        # distinct batches to rotate through: > 256 MiB of inputs in total, so a step's reads cannot come from the
        # Infinity Cache (cfg 2: 104 MB each -> 4; the larger workloads exceed it with one)
        n_inputs = max(1, min(4, -(-300_000_000 // (n * bytes_rec))))
        batches = [make_batch(oa, wl, n, rank, device, seed_shift=1000 * k) for k in range(n_inputs)]
        n = batches[0].n
        # Output layout: light scenes (the lane-per-ray kernel: cfg 2, cfg 4) write their [k][ray] slots in 64-slot tiles
        # (ot_trace_tiled_*: one contiguous block per wave and segment; the same records, tests/test_gpu_append.py); heavy
        # scenes write the 14 slot arrays here (their dense output, ot_trace_append_*, is measured in `configs`).
        # Which of the two the DEVICE streams faster differs by box (tiles +12 % on some, -8 % on others, stable within a box:
        # tools/stream_layouts2.hip), so it is what the library recommends after measuring both once (ot_trace_plan /
        # ot_probe_layouts: what trace_batch's default layout="auto" uses); the other layout is timed beside it.
        heavy_wl = scene.n_nodes >= 24
        plan = eng.plan(prec, n, MAX_SEG)
        layout = "slots" if heavy_wl else plan["layout"]
        cand = {}  # output buffers per slot layout (light scenes: both, so that the choice is measured on the buffers that are timed)
        if not heavy_wl:
            # ... and since the real trace on the real buffers can come out the other way round than the generic stream probe
            # (seen: probe 119 vs 136 us for tiles, trace 126 vs 118 against them; and a tune on ONE buffer pair before the
            # clocks had settled chose tiles at 119 vs 130 us where the timed region then ran 125 vs 119 against them), the
            # workload itself is measured in both layouts ON THE BUFFERS OF THE TIMED REGION, after the pre-load, interleaved:
            # three rounds of 20 rotating launches per layout, the minimum of a layout's rounds counts, the faster one runs.
            out_bytes = n * MAX_SEG * bytes_rec * n_inputs
            free_b = torch.cuda.mem_get_info(batches[0].device)[0]
            both = 2 * out_bytes < 0.8 * free_b and plan["kernel"] == 1 and bool(plan["tiled_ok"])
            for lay in (("slots", "tiled") if both else (layout,)):
                cand[lay] = [SegmentBatch(n * MAX_SEG, prec, batches[0].device, tiled=(lay == "tiled")) for _ in range(n_inputs)]
        extra["output_layout"] = layout
        outs = cand[layout] if cand else [SegmentBatch(n * MAX_SEG, prec, batches[0].device) for _ in range(n_inputs)]

        def step(s):
            eng.trace(batches[s % n_inputs], MAX_SEG, out=outs[s % n_inputs], layout=layout)

        # cold: what a caller sees right after the upload — one launch to load the code object, then the next 20
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        step(0)
        torch.cuda.synchronize()
        ev0.record()
        for s in range(20):
            step(s)
        ev1.record()
        torch.cuda.synchronize()
        extra["cold"] = {"us_per_step": ev0.elapsed_time(ev1) / 20 * 1e3, "steps": 20,
                         "note": "first 20 launches after one code-object-loading launch, no pre-load: the chip is still "
                                 "below its steady clocks"}
        # The chip takes tens of milliseconds of sustained load to reach its steady clocks (kernel time 122 -> 109 us
        # between 5 and 300 launches of pre-load on the same device): load it for ~60 ms, then the W warmup steps.
        preload = 0
        if wl.name == "cfg2":
            for s in range(PRELOAD_LAUNCHES):
                step(s)
            preload = PRELOAD_LAUNCHES
        else:
            t_load = time.perf_counter()
            while time.perf_counter() - t_load < 0.06:
                step(preload)
                preload += 1
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        extra["preload_launches"] = preload
        if len(cand) == 2:
            tev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            rounds = {"slots": [], "tiled": []}
            for rnd in range(3):
                for lay in (("slots", "tiled") if rnd % 2 == 0 else ("tiled", "slots")):
                    for s in range(3):
                        eng.trace(batches[s % n_inputs], MAX_SEG, out=cand[lay][s % n_inputs], layout=lay)
                    tev[0].record()
                    for s in range(20):
                        eng.trace(batches[s % n_inputs], MAX_SEG, out=cand[lay][s % n_inputs], layout=lay)
                    tev[1].record()
                    torch.cuda.synchronize()
                    rounds[lay].append(tev[0].elapsed_time(tev[1]) / 20 * 1e3)
            probe_says = layout
            layout = "tiled" if min(rounds["tiled"]) < min(rounds["slots"]) else "slots"
            outs = cand[layout]
            extra["output_layout"] = layout
            extra["layout_probe"] = {"stream_probe_slots_us": plan["probe_us"][0], "stream_probe_tiled_us": plan["probe_us"][1],
                                     "stream_probe_says": probe_says, "trace_slots_us": rounds["slots"], "trace_tiled_us": rounds["tiled"],
                                     "chosen": layout,
                                     "note": "stream probe: cfg 2's streams with no tracing, 2^20 rays, both slot layouts (ot_probe_layouts, once per "
                                             "context: what layout=\"auto\" uses by default); trace: this workload on the buffers of the timed region, "
                                             "after the pre-load, three interleaved rounds of 20 rotating launches per layout (us per launch by round) "
                                             "— the layout with the faster best round is the layout of the timed region"}
            if 2 * n * MAX_SEG * bytes_rec * n_inputs > 8e9:  # (large workloads: the loser's buffers go back before the timed region)
                del cand["tiled" if layout == "slots" else "slots"]
                torch.cuda.empty_cache()
        for s in range(args.warmup):
            step(s)
        # Timed region: K launches back to back, bracketed by barrier + synchronize (wall clock -> `value`) and by ONE
        # pair of HIP events on the launch stream (the engine launches on torch's current stream, so
        # torch.cuda.Event records there) -> average launch duration for the roofline.
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for s in range(args.steps):
            step(s)
        ev1.record()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        region_ms = ev0.elapsed_time(ev1)
        # the same K launches again with a HIP event pair around every launch (library timing): the pure kernel
        # duration, the number rocprofv3 reports
        eng.timing(True)
        for s in range(args.steps):
            step(s)
        kernel_ms, launches = eng.timing_read()
        eng.timing_reset()
        ceil_ms, ceil_n = 0.0, 0
        if scene.n_nodes < 24:  # the HBM-bound workloads: the same streams with no tracing (every ray fills its K slots)
            for s in range(3):  # (untimed: the first launch of a kernel loads its code object)
                eng.stream_ceiling(batches[s % n_inputs], MAX_SEG, outs[s % n_inputs])
            eng.timing_reset()
            for s in range(10 if wl.name == "cfg2" else 3):
                eng.stream_ceiling(batches[s % n_inputs], MAX_SEG, outs[s % n_inputs])
            ceil_ms, ceil_n = eng.timing_read()
        eng.timing(False)
        if not heavy_wl and world == 1:
            # the same K launches into the OTHER slot layout (the one the probe did not choose), for comparison
            other = "slots" if layout == "tiled" else "tiled"
            outs_o = cand.get(other) or [SegmentBatch(n * MAX_SEG, prec, batches[0].device, tiled=(other == "tiled")) for _ in range(n_inputs)]
            for s in range(args.warmup + 3):
                eng.trace(batches[s % n_inputs], MAX_SEG, out=outs_o[s % n_inputs], layout=other)
            torch.cuda.synchronize()
            ev0.record()
            for s in range(args.steps):
                eng.trace(batches[s % n_inputs], MAX_SEG, out=outs_o[s % n_inputs], layout=other)
            ev1.record()
            torch.cuda.synchronize()
            other_us = ev0.elapsed_time(ev1) / args.steps * 1e3
            for s in range(3):
                eng.stream_ceiling(batches[s % n_inputs], MAX_SEG, outs_o[s % n_inputs])
            eng.timing(True)
            for s in range(10 if wl.name == "cfg2" else 3):
                eng.stream_ceiling(batches[s % n_inputs], MAX_SEG, outs_o[s % n_inputs])
            sc_ms, sc_n = eng.timing_read()
            eng.timing(False)
            extra["other_layout"] = {"layout": other, "kernel_us": other_us, "stream_ceiling_us": sc_ms / max(sc_n, 1) * 1e3,
                                     "note": "the same trace into the slot layout the measurement before the timed region did NOT choose"}
            cand.pop(other, None)
            del outs_o
            torch.cuda.empty_cache()
        if world == 1 and not args.no_sustained:
            # >= 6 s of back-to-back launches: long enough for any outside sampler (the driver polls the card every few seconds)
            # to see the GPU busy, and the figure a long job gets
            clocks = {}
            sampler = threading.Thread(target=sample_device_clocks, args=(clocks,), daemon=True)
            sampler.start()  # reads the clocks the chip runs at WHILE this loop keeps it busy (boxes of a pool differ)
            t_s = time.perf_counter()
            done = 0
            chunk = max(1, int(0.02 / max(dt / args.steps, 1e-6)))  # ~20 ms of launches between host syncs
            while time.perf_counter() - t_s < args.sustained_seconds:
                for s in range(chunk):
                    step(done + s)
                done += chunk
                torch.cuda.synchronize()
            dt_s = time.perf_counter() - t_s
            extra["sustained"] = {"seconds": dt_s, "steps": done, "ms_per_step": dt_s / done * 1e3}
            sampler.join(timeout=3.0)
            if clocks:
                extra["sustained"]["device_clocks_under_load"] = clocks
        step(0)  # leave real results in outs[0]
        torch.cuda.synchronize()
        segs_step = int(outs[0].count.abs().sum().item())
        local_final = None

    gather_ms = gather_error = None
    ranks_info = gather_split = None
    if distributed:
        # every rank's own time and work next to the maximum `value` is computed from: a straggler shows by rank
        mine = torch.tensor([dt, float(segs_step), float(n)], dtype=torch.float64, device=comm_dev)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        per_rank = [[float(x) for x in t.tolist()] for t in everyone]
        ms_steps = sorted(r[0] / args.steps * 1e3 for r in per_rank)
        ranks_info = {"ms_per_step": {"min": ms_steps[0], "median": ms_steps[len(ms_steps) // 2], "max": ms_steps[-1],
                                      "by_rank": [r[0] / args.steps * 1e3 for r in per_rank]},
                      "segments_per_step_by_rank": [int(r[1]) for r in per_rank], "rays_by_rank": [int(r[2]) for r in per_rank]}
        tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        n_gather = n if local_final is None else local_final.shape[1]
        tot = torch.tensor([segs_step, n, n_gather], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(tot)
        segs_total_step, rays_total, gather_total = (int(x) for x in tot.tolist())
        # the one collective of the job: per-ray final state to rank 0 (outside the timed steps).  It is not part
        # of `value`; if it fails the trace numbers are still reported, with the error beside them.
        try:
            local = local_final if local_final is not None else odist.final_state(outs[0])
            if not args.dry_run:
                torch.cuda.synchronize()
            dist.barrier()
            g0 = time.perf_counter()
            gather_split = {}
            gathered = odist.gather_final_state(local, dst=0, timings=gather_split)
            if not args.dry_run:
                torch.cuda.synchronize()
            dist.barrier()
            gather_ms = (time.perf_counter() - g0) * 1e3
            if rank == 0:
                extra["gathered_shape"] = list(gathered.shape)
                if tuple(gathered.shape) != (12, gather_total):
                    gather_error = f"gathered shape {tuple(gathered.shape)} != {(12, gather_total)}"
        except Exception as exc:  # noqa: BLE001 — reported in the JSON line
            gather_error = f"{type(exc).__name__}: {exc}"
    else:
        segs_total_step, rays_total = segs_step, n

    # The configs BASELINE quotes on several GPUs, at their quoted totals: cfg 4 on 2 and 4 ranks, cfg 5 on 8 (every rank takes part)
    sharded = []
    if distributed and world > 1 and not args.dry_run and wl.name == "cfg2" and args.sharded_configs != "none" \
            and (args.sharded_configs != "auto" or args.rays is None):
        names = (["cfg4"] if world in (2, 4) else []) + (["cfg5"] if world == 8 else []) if args.sharded_configs == "auto" \
            else [c for c in args.sharded_configs.split(",") if c]
        del batches, outs
        torch.cuda.empty_cache()
        for name in names:
            try:
                rec = sharded_config(oa, eng, dist, name, world, rank, device, comm_dev, rays_override=args.sharded_rays)
            except Exception as exc:  # noqa: BLE001 — reported; the other ranks raise alike (same code, same sizes)
                rec = {"workload": name, "error": f"{type(exc).__name__}: {exc}"}
            if rank == 0:
                sharded.append(rec)
        batches = [make_batch(oa, wl, min(n, 200_000), rank, device)]
        eng.upload(scene)

    if rank == 0:
        value = segs_total_step * S_LEAVES * args.steps / dt
        avg_kernel_s = region_ms / args.steps / 1e3             # HIP events over the timed region, incl. launch gaps
        per_launch_us = kernel_ms / max(launches, 1) * 1e3      # event pair per launch, companion loop
        alg_bytes = n * bytes_rec + segs_step * bytes_rec       # per launch, this rank
        achieved = alg_bytes / avg_kernel_s / 1e9
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic_cfg2_f64.json")
        r03 = committed_counters("cfg2") if wl.name == "cfg2" else None
        if r03 and "hbm_bytes" in r03.get("derived", {}):  # this round's PMC passes of the bench workload, in the layout the bench runs
            traffic = r03["derived"]["hbm_bytes"]
            traffic_source = (f"{r03['source']}: rocprofv3 --pmc passes of this workload (FETCH_SIZE x 2 + WRITE_SIZE per launch of "
                              f"{r03['kernel']}), committed — not re-measured by this run")
        elif wl.name == "cfg2" and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            traffic_source = ("profiles/traffic_cfg2_f64.json: rocprofv3 --pmc passes of this workload (FETCH_SIZE x 2 + "
                              "WRITE_SIZE, calibrated on k_stream_ceiling), committed — not re-measured by this run")
        heavy = scene.n_nodes >= 24
        line = {
            "metric": "ray-surface intersections/sec", "value": value, "unit": "intersections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if (clamped or args.rays) else wl.scaling, "vs_baseline": None, "dtype": prec,
            "data": "synthetic",
            "config": {"workload": wl.label, "rays_per_gpu": n, "rays_total": rays_total, "output_layout": extra.get("output_layout", "slots"),
                       "segments_per_ray": segs_step / n, "leaf_surfaces": S_LEAVES, "input_batches_rotated": n_inputs,
                       "parallelism": f"ray-shard x{world}, scene replicated"},
            "segments_per_s": segs_total_step * args.steps / dt,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name(scene, wl), "kernel_us": avg_kernel_s * 1e6,
                         "kernel_us_per_launch_events": per_launch_us, "algorithmic_bytes_per_launch": alg_bytes,
                         "stream_ceiling_gbs": (alg_bytes / (ceil_ms / ceil_n / 1e3) / 1e9) if ceil_n else None},
        }
        if heavy:
            line["roofline"]["note"] = ("this workload is VALU/latency-bound (S >= 24, SURVEY.md §8d): the HBM fraction is "
                                        "reported for comparison, not as its roof")
        if clamped:  # the config names a total that does not fit this few GPUs: every rank traces its HBM's worth instead
            line["clamped"] = True
            line["total_requested"] = wl.total_rays
            line["scaling_note"] = (f"{wl.name} names {wl.total_rays} rays in total; {world} rank(s) x {MAX_PER_RANK[wl.name]} (the per-rank HBM "
                                    "cap with the full history) is less, so the per-rank size is fixed and the total grows with N: weak scaling")
        if args.dry_run:
            line["dry_run"] = True
        if "sustained" in extra:
            sus = extra["sustained"]
            sus["value"] = segs_step * S_LEAVES / (sus["ms_per_step"] / 1e3)
            sus["hbm_frac"] = alg_bytes / (sus["ms_per_step"] / 1e3) / 1e9 / HBM_PEAK_GBS
        if "cold" in extra:
            extra["cold"]["hbm_frac"] = alg_bytes / (extra["cold"]["us_per_step"] / 1e6) / 1e9 / HBM_PEAK_GBS
        if "other_layout" in extra:
            sl = extra["other_layout"]
            sl["hbm_frac"] = alg_bytes / (sl["kernel_us"] / 1e6) / 1e9 / HBM_PEAK_GBS
            sl["stream_ceiling_gbs"] = alg_bytes / (sl["stream_ceiling_us"] / 1e6) / 1e9
        line.update(extra)
        if gather_error is not None:
            line["gather_error"] = gather_error
        elif gather_ms is not None:
            line["gather_ms"] = gather_ms
            line["value_incl_gather"] = segs_total_step * S_LEAVES * args.steps / (dt + gather_ms / 1e3)
            if gather_split:
                line["gather"] = {"sizes_ms": gather_split["sizes_ms"], "payload_ms": gather_split["payload_ms"],
                                  "payload_bytes": int(sum(gather_split["shard_sizes"])) * 12 * (8 if prec == "f64" else 4),
                                  "shard_sizes": gather_split["shard_sizes"]}
        if sharded:
            line["sharded_configs"] = sharded
        if ranks_info is not None:
            line["ranks"] = ranks_info
            line["comm"] = {"backend": "rccl" if use_nccl else "gloo",
                            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if use_nccl else None}
        if world == 1 and not args.dry_run and wl.name == "cfg2" and not args.no_configs and args.rays is None:
            # the other BASELINE configs, at the size one GPU sees in the configuration they are quoted on
            del batches, outs
            torch.cuda.empty_cache()
            line["configs"] = []
            for name in ("cfg3", "cfg4", "cfg5"):
                try:
                    line["configs"].append(survey_config(oa, eng, name, device))
                except Exception as exc:  # noqa: BLE001 — the headline is still worth printing
                    line["configs"].append({"workload": name, "error": f"{type(exc).__name__}: {exc}"})
            try:
                line["configs"].append(survey_branching(oa, eng, device))
            except Exception as exc:  # noqa: BLE001
                line["configs"].append({"workload": "cfg4 branching", "error": f"{type(exc).__name__}: {exc}"})
            batches = [make_batch(oa, wl, min(n, 200_000), rank, device)]
            eng.upload(scene)
        if world == 1 and not args.dry_run and wl.name == "cfg2" and not args.no_latency and args.rays is None:
            try:
                line["latency"] = survey_latency(oa)
            except Exception as exc:  # noqa: BLE001
                line["latency"] = [{"error": f"{type(exc).__name__}: {exc}"}]
            eng.upload(scene)
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            try:
                line["cpu_baseline"] = cpu_baseline(table, batches[0].slice(0, min(n, 200_000)).to_host(), MAX_SEG, S_LEAVES,
                                                    wl.name, budget_s=args.cpu_seconds)
            except Exception as exc:  # noqa: BLE001 — the GPU numbers above are still worth printing
                line["cpu_baseline"] = {"value": None, "unit": "ray-surface intersections/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {type(exc).__name__}: {exc}"}
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One BASELINE workload, a handful of device traces and nothing else: the program rocprofv3 wraps for the
profiles/ summaries (tools/profile_r02.sh).  Sizes are what one GPU sees in the quoted configuration.
    python3 tools/profile_workload.py cfg3|cfg4|cfg5|cfg4b|cfg3b|monitor [reps]
cfg4b = cfg 4 with reflectivity 0.2 (ray trees, generation kernels); monitor = Monitor.record over a cfg 2 history."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import optable_amd as oa
from optable_amd import workloads as W
from optable_amd.batch import RayBatch, SegmentBatch
from optable_amd.engine import get_engine

name = sys.argv[1]
GENLOOP = bool(os.environ.get("GENLOOP")) or name in ("cfg4bgen", "cfg3bgen")  # ...gen: the same trees through the generation loop instead of the default call
if name in ("cfg4bgen", "cfg3bgen"):
    name = name[:-3]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else {"cfg2": 400, "cfg3": 20}.get(name, 5)  # cfg3: enough launches that the clock ramp of the first ones does not carry the average
SIZES = {"cfg2": 1_000_000, "cfg3": 10_000_000, "cfg4": 160_000_000, "cfg5": 12_500_000, "cfg4b": 12_800_000, "cfg3b": 2_000_000, "monitor": 1_000_000, "allfeat64": 1_000_000, "allfeat32": 1_000_000}
n = int(os.environ.get("RAYS", SIZES[name]))
eng = get_engine()
from optable_amd import abi as _abi
for _env, _opt in (("FLAT", _abi.OPT_FLAT_QUEUE), ("RECLDS", _abi.OPT_LDS_RECORDS), ("CAP", _abi.OPT_LIST_CAP), ("MIX", _abi.OPT_MIX_GENERATIONS), ("KERNEL", _abi.OPT_KERNEL), ("REFILL", _abi.OPT_REFILL), ("ONEPASS", _abi.OPT_GEN_ONEPASS), ("AHEAD", _abi.OPT_GEN_AHEAD)):
    if os.environ.get(_env):
        eng.set_option(_opt, int(os.environ[_env]))
Q = lambda lam: 1j * np.pi * W.W0**2 / lam

# output layout of the non-branching traces: what the bench measures for this workload unless LAYOUT says otherwise
LAYOUT = os.environ.get("LAYOUT") or {"cfg2": "tiled", "cfg4": "tiled", "cfg3": "append", "cfg5": "append"}.get(name, "slots")


def output_for(batch, K, precision, records=None):
    if LAYOUT == "append":  # (the worst-case block: the default call's 1 % sample trace would show up in the profile as a ninth, tiny launch of the same kernel)
        return SegmentBatch(eng._append_worst_case(batch.n, K), precision, batch.device, block=True)
    return SegmentBatch(batch.n * K, precision, batch.device, tiled=(LAYOUT == "tiled"))


if name in ("cfg2", "cfg3", "cfg5"):
    wl = W.baseline_workloads(oa)[name]
    table = oa.OpticalTable()
    table.add_components(wl.components())
    eng.upload(table.compile())
    o, d, lam = wl.rays(n, 0)
    batch = RayBatch.from_arrays(o, d, wavelength=lam, q=Q(lam), precision=wl.precision)
    out = output_for(batch, wl.max_segments, wl.precision)
    for _ in range(reps + 2):
        eng.trace(batch, wl.max_segments, out=out, layout=LAYOUT)
    torch.cuda.synchronize()
    print(f"{name}: {n} rays, {int(out.count.abs().sum())} segments per trace, {reps + 2} traces, layout {LAYOUT}, launch {eng.last_launch()}")
elif name in ("cfg4", "cfg4b"):
    table = oa.OpticalTable()
    table.add_components(W.cfg4_components(oa, reflectivity=0.2 if name == "cfg4b" else 0))
    scene = table.compile()
    eng.upload(scene)
    nb = n // W.CFG4_WAVELENGTHS
    o, d, _ = W.cfg4_rays(nb, 4, n_wavelengths=1)
    base = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q(W.WL), precision="f64")
    batch = base.multiplexed_in_wavelength(np.linspace(400e-7, 1100e-7, W.CFG4_WAVELENGTHS))
    if name == "cfg4":
        out = output_for(batch, 3, "f64")
        # clocks up on ANOTHER kernel first (the stream companion on a slice of the batch, ~0.1 s): every launch of the traced
        # kernel in the profile is then a warm one, and its min / max are the call-to-call spread, not the clock ramp
        warm_in = batch.slice(0, 1 << 20)
        warm_out = SegmentBatch(3 << 20, "f64", batch.device, tiled=(LAYOUT == "tiled"))
        for _ in range(800):
            eng.stream_ceiling(warm_in, 3, warm_out)
        torch.cuda.synchronize()
        del warm_in, warm_out
        reps = max(reps, 12)
        for _ in range(reps):
            eng.trace(batch, 3, out=out, layout=LAYOUT)
        torch.cuda.synchronize()
        print(f"cfg4: {batch.n} ray-wavelength pairs, {int(out.count.abs().sum())} segments per trace, {reps} traces, layout {LAYOUT}")
    else:
        for _ in range(max(reps, 2)):
            t0 = time.perf_counter()
            if GENLOOP:  # the generation loop (count / recount + scan + emit per generation) instead of the default call
                segs = eng.trace_tree(batch, 12, out_capacity=batch.n * 13)
                n_seg = segs.n_valid
            else:  # one launch, a lane per tree (k_trace_trees)
                # (what Engine.trace_branching launches for this batch, into a block for every tree at its cap: the default call's 1 %
                # sample would show up in the profile as one more, tiny launch of the same kernel)
                segs = eng.trace_trees(batch, 12, layout="append")
                n_seg = int(segs.count.abs().sum())
                assert eng.last_launch()["kernel"] == 4
            torch.cuda.synchronize()
            print(f"cfg4b: {batch.n} trees, {n_seg} segments per trace, {1e3 * (time.perf_counter() - t0):.1f} ms wall, layout {segs.layout}")
            del segs
elif name == "cfg3b":  # heavy branching: cfg 3 with 10 % reflecting slab faces, trees capped at 20 segments, fp32
    table = oa.OpticalTable()
    table.add_components(W.cfg3_components(oa, slab_reflectivity=0.1))
    eng.upload(table.compile())
    o, d = W.cfg3_rays(n, 2)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q(W.WL), precision="f32")
    for _ in range(max(reps, 2)):
        t0 = time.perf_counter()
        if GENLOOP:
            segs = eng.trace_tree(batch, 20, out_capacity=batch.n * 21)
            n_seg = segs.n_valid
        else:  # what Engine.trace_branching launches for this batch (the lane-per-tree kernel with the pair-queue search), without its 1 % sample
            segs = eng.trace_trees(batch, 20, layout="append")
            n_seg = int(segs.count.abs().sum())
        torch.cuda.synchronize()
        print(f"cfg3b: {batch.n} trees, {n_seg} segments per trace, {1e3 * (time.perf_counter() - t0):.1f} ms wall, layout {segs.layout}")
        del segs
elif name in ("allfeat64", "allfeat32"):  # a light scene of the reference's example parts (tools/bench_allfeatures.py): the FM preset
    prec = "f64" if name.endswith("64") else "f32"
    table = oa.OpticalTable()
    table.add_components([oa.TriangularPrism([4, -0.6, 0], width=2.0, height=2.0, n1=1.0, n2=1.5),
                          oa.DovePrism([9, 0, 0], L=3.0, D=1.0, Ng=1.5).TY(-0.5),
                          oa.Block([13, 0, 0], hole=oa.Circle(0.6), width=3, height=3),
                          oa.BiConvexLens([16, 0, 0], CT=0.6, R1=12.0, R2=-12.0, diameter=3.0, n=1.5),
                          oa.Mirror([22, 0, 0], radius=3.0).RotZ(np.pi + 0.05)])
    scene = table.compile()
    rng = np.random.default_rng(11)
    o = np.stack([np.zeros(n), rng.uniform(-0.5, 0.5, n), rng.uniform(-0.4, 0.4, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.03, 0.03, n), rng.uniform(-0.02, 0.02, n)], 1)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q(W.WL), precision=prec)
    for _ in range(reps + 2):
        segs = table.trace_batch(batch, max_segments=16, scene=scene, layout="slots")
    torch.cuda.synchronize()
    print(f"{name}: {n} rays, {int(segs.count.abs().sum())} segments per trace, {reps + 2} traces, launch {eng.last_launch()}")
else:  # Monitor.record over the [k][ray] history of a cfg 2 trace
    wl = W.baseline_workloads(oa)["cfg2"]
    table = oa.OpticalTable()
    table.add_components(wl.components())
    mon = oa.Monitor([7.5, 0, 0], 5, 5)
    o, d, lam = wl.rays(n, 0)
    batch = RayBatch.from_arrays(o, d, wavelength=lam, q=Q(lam))
    segs = table.trace_batch(batch, max_segments=5)
    for _ in range(reps):
        hits = table.record_batch(mon, segs)
    torch.cuda.synchronize()
    print(f"monitor: {segs.capacity} slots, {len(hits)} hits per pass, {reps} passes")

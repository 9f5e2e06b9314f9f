#!/usr/bin/env python3
"""Batches whose ray trees differ widely in size: a beam-splitter lattice (9 components) in which a share AWAY (default 0.9) of
the 2e6 input rays leaves the table at once (a tree of one ray) and the rest grow bushy trees up to the cap (24 / 96).  The
lane-per-tree kernel into [k][tree] slots and into the append layout, and the generation loop, device ms each.
    AWAY=0.9 python tools/ab_tree_skew.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from optable_amd import abi
if os.environ.get("OT_LIB"):
    abi.LIB_PATH = os.path.abspath(os.environ["OT_LIB"])
import optable_amd as oa
from optable_amd import workloads as W
from optable_amd.batch import RayBatch, SegmentBatch
from optable_amd.engine import get_engine
eng = get_engine()
comps = []
for k in range(3):
    comps.append(oa.BeamSplitter([2.0 * (k + 1), 0, 0], width=6, height=2, eta=0.5).RotZ(np.pi / 4))
    comps.append(oa.Mirror([2.0 * (k + 1), 3.0 + 0.1 * k, 0], radius=2).RotZ(-np.pi / 2))
    comps.append(oa.BeamSplitter([2.0 * (k + 1) + 1.0, 1.5, 0], width=6, height=2, eta=0.3).RotZ(-np.pi / 4))
t = oa.OpticalTable(); t.add_components(comps); eng.upload(t.compile())
n = 2_000_000
rng = np.random.default_rng(5)
o = np.stack([np.zeros(n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.2, 0.2, n)], 1)
d = np.stack([np.ones(n), rng.uniform(-0.02, 0.02, n), rng.uniform(-0.01, 0.01, n)], 1)
away = rng.uniform(size=n) < float(os.environ.get("AWAY", 0.9))
d[away] = [-1.0, 0.0, 0.0]  # these leave the table at once: trees of one ray
for prec in ("f64", "f32"):
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=1j * np.pi * W.W0**2 / W.WL, precision=prec)
    for cap in (24, 96):
        for layout in ("slots", "append"):
            plan = eng.trees_plan(prec, cap, n)
            out = SegmentBatch(n * cap + plan["chunk"] * plan["waves"] if layout == "append" else n * cap, prec, batch.device, block=(layout == "append"))
            for rnd in range(3):
                eng.timing(True); torch.cuda.synchronize()
                segs = eng.trace_trees(batch, cap, out=out, layout=layout)
                torch.cuda.synchronize(); ms, _ = eng.timing_read(); eng.timing(False)
            print(f"{prec} cap {cap} {layout}: {ms:.3f} ms, {int(segs.count.abs().sum())} segments", flush=True)
            del out, segs
        for rnd in range(2):
            eng.timing(True); torch.cuda.synchronize(); t0 = time.perf_counter()
            g = eng.trace_tree(batch, cap, out_capacity=22_000_000)
            torch.cuda.synchronize(); wall = 1e3 * (time.perf_counter() - t0); ms, _ = eng.timing_read(); eng.timing(False)
        print(f"   generation loop: {ms:.3f} ms device, {wall:.3f} wall, {g.n_valid} segments", flush=True)
        del g

#!/usr/bin/env python3
"""The all-features preset on a scene made of the reference's own example parts: a TriangularPrism with
its `max_interact_count` faces (examples/prism_refl.py), a DovePrism (polygon faces in tilted planes, examples/dove_prism.py),
a Block with a round hole (boolean aperture), a BiConvexLens (spherical faces) and a mirror — 1e6 rays, cap 16, in both
precisions, non-branching and with 10 % reflecting lens faces (ray trees, count gates).  Library hipEvent time + launch shape.
    python tools/bench_allfeatures.py [n_rays]      env: ONLY=f32|f64, OT_LIB, GENLOOP=1 (ray trees through the generation loop)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import optable_amd as oa
from optable_amd import abi
from optable_amd import workloads as W
from optable_amd.batch import RayBatch
from optable_amd.engine import get_engine

if os.environ.get('OT_LIB'):
    abi.LIB_PATH = os.path.abspath(os.environ['OT_LIB'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = 16
eng = get_engine()
if os.environ.get("GENLOOP"):  # ray trees generation by generation instead of the lane-per-tree launch
    eng.LANE_PER_TREE = False


def scene(reflect):
    kw = {"reflectivity": 0.1, "transmission": 0.9} if reflect else {}
    return [oa.TriangularPrism([4, -0.6, 0], width=2.0, height=2.0, n1=1.0, n2=1.5),
            oa.DovePrism([9, 0, 0], L=3.0, D=1.0, Ng=1.5).TY(-0.5),
            oa.Block([13, 0, 0], hole=oa.Circle(0.6), width=3, height=3),
            oa.BiConvexLens([16, 0, 0], CT=0.6, R1=12.0, R2=-12.0, diameter=3.0, n=1.5, **kw),
            oa.Mirror([22, 0, 0], radius=3.0).RotZ(np.pi + 0.05)]


rng = np.random.default_rng(11)
o = np.stack([np.zeros(n), rng.uniform(-0.5, 0.5, n), rng.uniform(-0.4, 0.4, n)], 1)
d = np.stack([np.ones(n), rng.uniform(-0.03, 0.03, n), rng.uniform(-0.02, 0.02, n)], 1)
for prec in ("f64", "f32"):
    if os.environ.get("ONLY") and os.environ["ONLY"] != prec:
        continue
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=1j * np.pi * W.W0**2 / W.WL, precision=prec)
    for reflect in (False, True):
        table = oa.OpticalTable()
        table.add_components(scene(reflect))
        sc = table.compile()
        segs = table.trace_batch(batch, max_segments=K, scene=sc)  # warm (scratch, code objects)
        torch.cuda.synchronize()
        eng.timing(True)
        t0 = time.perf_counter()
        segs = table.trace_batch(batch, max_segments=K, scene=sc)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ms, launches = eng.timing_read()
        eng.timing(False)
        nseg = int(segs.n_valid) if segs.count is None else int(segs.count.abs().sum().item())
        print(f"{prec} {'ray trees (R = 0.1 lens faces)' if reflect else 'non-branching'}: {sc.n_nodes} nodes, {sc.n_leaves} leaves, {len(sc.limited)} limited; "
              f"{n} rays -> {nseg} segments; device {ms:.3f} ms in {launches} timed regions, wall {wall:.1f} ms; layout {segs.layout}; launch {eng.last_launch()}", flush=True)
        del segs

#!/usr/bin/env python3
"""Top-level grid of cfg 3 (pair-queue kernel, append layout): time of the full batch and of a 1 % strided sample for every
(g0, g1) in a range, in one process (same box, same buffers).  Tells (a) how much the choice of cells matters, (b) whether a
sample predicts the full batch (what an upload-time tuner would measure)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import optable_amd as oa
import optable_amd.scene as S
from optable_amd.batch import RayBatch
from optable_amd.engine import get_engine
from optable_amd import workloads as w

eng = get_engine()
n = int(os.environ.get("N3", 10_000_000))
o, d = w.cfg3_rays(n, 2)
q = 1j * np.pi * w.W0**2 / w.WL
batch = RayBatch.from_arrays(o, d, wavelength=w.WL, q=q, precision="f32")
idx = torch.arange(0, n, 100, device=batch.device)
sample = batch.take(idx) if hasattr(batch, "take") else None
comps = w.cfg3_components(oa)


def timed(b, reps):
    out = eng.trace(b, 20, layout="append")
    eng.timing(True)
    for _ in range(reps):
        eng.trace(b, 20, out=out, layout="append")
    ms, cnt = eng.timing_read()
    eng.timing(False)
    return ms / cnt


for g0 in [int(x) for x in os.environ.get("G0", "6,7,8,9,10,11,12,14,16").split(",")]:
    for g1 in [int(x) for x in os.environ.get("G1", "2,3,4,5,6,8").split(",")]:
        S.ROOT_GRID_DIMS = (g0, g1)
        t = oa.OpticalTable()
        t.add_components(comps)
        eng.upload(t.compile())
        full = timed(batch, 3)
        samp = timed(sample, 20) if sample is not None else float("nan")
        print(f"g0={g0:2d} g1={g1:2d} full={full:7.3f} ms  sample={samp:7.4f} ms", flush=True)

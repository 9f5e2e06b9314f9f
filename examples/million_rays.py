#!/usr/bin/env python3
"""BASELINE cfg 2 through the batch API: 1e6 rays resident in HBM, one kernel launch, results left on the GPU.
    python examples/million_rays.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optable_amd as oa  # noqa: E402
from optable_amd.batch import RayBatch  # noqa: E402

n = 1_000_000
rng = np.random.default_rng(0)
theta, phi = 0.15 * np.sqrt(rng.uniform(0, 1, n)), rng.uniform(0, 2 * np.pi, n)
directions = np.stack([np.cos(theta), np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi)], axis=1)

table = oa.OpticalTable()
table.add_components([oa.Lens([5, 0, 0], focal_length=5, radius=1.0), oa.MirrorPair([10, 0, 0], 4, 4)])
screen = oa.Monitor([7.5, 0, 0], 3, 3)

batch = RayBatch.from_arrays(np.zeros((n, 3)), directions, wavelength=780e-7, q=1j * np.pi * 61e-4**2 / 780e-7)
scene = table.compile()
table.trace_batch(batch, max_segments=5, scene=scene)          # first call: library load, upload
torch.cuda.synchronize()
t0 = time.perf_counter()
segs = table.trace_batch(batch, max_segments=5, scene=scene)    # SegmentBatch: [segment][ray] slots on the device
torch.cuda.synchronize()
dt = time.perf_counter() - t0
hits = table.record_batch(screen, segs)                         # Monitor.record without Python objects
total = int(segs.count.sum())
print(f"{n} rays, {total} segments in {dt * 1e3:.2f} ms ({total * 3 / dt:.3e} ray-surface intersections/s incl. launch)")
print(f"monitor at x = 7.5: {len(hits)} crossings, rms radius {float(torch.sqrt((hits.yList() ** 2 + hits.zList() ** 2).mean())):.4f}")

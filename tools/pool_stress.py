#!/usr/bin/env python3
"""Stress of k_trace_pool's lock-free scheduling (one process, sequential launches): cfg 5 with random batch sizes,
segment caps and append chunk sizes; every launch must hold exactly the records of the per-wave lists, in the
reference's order.    python tools/pool_stress.py [iterations] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import optable_amd as oa
from optable_amd import abi, workloads as W
from optable_amd.batch import RayBatch
from optable_amd.engine import get_engine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
wl = W.baseline_workloads(oa)["cfg5"]
table = oa.OpticalTable()
table.add_components(wl.components())
eng = get_engine()
eng.upload(table.compile())
bad = 0
for it in range(iters):
    n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 30_000, 100_003, 300_000]))
    K = int(rng.integers(2, 51))
    chunk = int(rng.choice([64, 128, 512, 2048]))
    o, d, lam = wl.rays(n, int(rng.integers(0, 1000)))
    batch = RayBatch.from_arrays(o, d, wavelength=lam, q=1j * np.pi * W.W0**2 / lam, precision="f32")
    eng.set_option(abi.OPT_BLOCK_POOL, 0)
    a = eng.trace(batch, K).to_host(reference_order=True)
    eng.set_option(abi.OPT_BLOCK_POOL, -1)
    eng.set_option(abi.OPT_APPEND_CHUNK, chunk)
    out = eng.trace(batch, K, layout="append")
    pooled = bool(eng.last_launch()["pair_queue"] & 16)
    b = out.to_host(reference_order=True)
    eng.set_option(abi.OPT_APPEND_CHUNK, 512)
    same = pooled and all(np.array_equal(a[f], b[f]) for f in abi.SEG_FIELDS + ("ray", "surface"))
    bad += not same
    print(f"{it:3d} n={n:7d} K={K:2d} chunk={chunk:4d} records={len(a['ray']):9d} slots={int(out.n_valid):9d} {'ok' if same else 'MISMATCH <<<<'}", flush=True)
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""cfg 2 in the tiled layout (the bench line's kernel) under the launch options, interleaved rounds in one process, next to
the stream ceiling of the same box: which setting is closest to the ceiling HERE (boxes of the pool differ)."""
import itertools
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import optable_amd as oa
from optable_amd import abi, workloads as W
from optable_amd.batch import RayBatch, SegmentBatch
from optable_amd.engine import get_engine

n, K = 1_000_000, 5
table = oa.OpticalTable()
table.add_components(W.cfg2_components(oa))
eng = get_engine()
eng.upload(table.compile())
batches = []
for seed in range(4):
    o, d = W.cfg2_rays(n, seed)
    batches.append(RayBatch.from_arrays(o, d, wavelength=W.WL, q=1j * np.pi * W.W0**2 / W.WL, precision="f64"))
out = SegmentBatch(n * K, "f64", batches[0].device, tiled=True)
turn = [0]


def step(fn):
    turn[0] += 1
    fn(batches[turn[0] % 4])


def timed(fn, reps=40):
    eng.timing(True)
    for _ in range(reps):
        step(fn)
    ms, cnt = eng.timing_read()
    eng.timing(False)
    return ms / cnt * 1e3


variants = list(itertools.product((0, 1), (0, 4), (0, 8, 16, 64)))  # NT stores, MINW, blocks per CU
res = {v: [] for v in variants}
ceil = []
for rnd in range(4):
    for v in variants:
        eng.set_option(abi.OPT_NT_STORES, v[0])
        eng.set_option(abi.OPT_MIN_WAVES, v[1])
        eng.set_option(abi.OPT_BLOCKS_PER_CU, v[2])
        for _ in range(5):
            step(lambda b: eng.trace(b, K, out=out, layout="tiled"))
        res[v].append(timed(lambda b: eng.trace(b, K, out=out, layout="tiled")))
    eng.set_option(abi.OPT_NT_STORES, 1); eng.set_option(abi.OPT_MIN_WAVES, 4); eng.set_option(abi.OPT_BLOCKS_PER_CU, 0)
    ceil.append(timed(lambda b: eng.stream_ceiling(b, K, out)))
alg = n * 104 * (1 + K)
print(f"stream ceiling: {statistics.median(ceil):.1f} us = {alg / statistics.median(ceil) / 1e3:.0f} GB/s")
for v in sorted(variants, key=lambda v: statistics.median(res[v])):
    t = statistics.median(res[v])
    print(f"nt={v[0]} minw={v[1]} blocks/CU={v[2]:3d}: {t:7.1f} us  {alg / t / 1e3:6.0f} GB/s  {alg / t / 8e6 * 1e3 / 1e3:.3f} of peak")

#!/usr/bin/env python3
"""A/B the launch options of k_trace_fused on the cfg-2 workload, interleaved rounds in ONE process
(cdna_hip_programming.md §5.4 rule 24); prints the median kernel time per variant and the
stream-ceiling of the same access pattern."""
import itertools
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import optable_amd as oa
from optable_amd import abi
from optable_amd.batch import RayBatch, SegmentBatch
from optable_amd.engine import get_engine
from optable_amd import workloads as scenes  # the BASELINE configs (scene + ray generators)

n, K = int(os.environ.get("N", 1_000_000)), 5
prec = os.environ.get("PREC", "f64")
table = oa.OpticalTable()
table.add_components(scenes.cfg2_components(oa))
o, d = scenes.cfg2_rays(n, 0)
batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j * np.pi * scenes.W0**2 / scenes.WL, precision=prec)
eng = get_engine()
eng.upload(table.compile())
out = SegmentBatch(n * K, prec, batch.device)
ROT = int(os.environ.get("ROT", 1))  # ROT=4: rotate four input batches as bench.py does (reads come from HBM, not the Infinity Cache)
batches = [batch]
for seed in range(1, ROT):
    o2, d2 = scenes.cfg2_rays(n, seed)
    batches.append(RayBatch.from_arrays(o2, d2, wavelength=scenes.WL, q=1j * np.pi * scenes.W0**2 / scenes.WL, precision=prec))
_turn = [0]


def next_batch():
    _turn[0] += 1
    return batches[_turn[0] % len(batches)]
bytes_alg = n * (104 if prec == "f64" else 56) * (1 + K)


REPS = int(os.environ.get('REPS', 10))


def timed(fn, reps=None):
    reps = reps or REPS
    eng.timing(True)
    for _ in range(reps):
        fn()
    ms, cnt = eng.timing_read()
    eng.timing(False)
    return ms / cnt * 1e3  # us


variants = list(itertools.product((0, 1), (0, 4), (0, 2, 4, 8, 16)))
if os.environ.get('QUICK'):
    variants = [(1, 4, 0), (1, 4, 8), (0, 0, 8)]
results = {v: [] for v in variants}
ceil = {0: [], 1: []}
for rnd in range(5):
    for v in variants:
        nt, mw, bpc = v
        eng.set_option(abi.OPT_NT_STORES, nt)
        eng.set_option(abi.OPT_MIN_WAVES, mw)
        eng.set_option(abi.OPT_BLOCKS_PER_CU, bpc)
        results[v].append(timed(lambda: eng.trace(next_batch(), K, out=out)))
    if prec == "f64":
        for nt in (0, 1):
            eng.set_option(abi.OPT_NT_STORES, nt)
            eng.set_option(abi.OPT_BLOCKS_PER_CU, 0)
            ceil[nt].append(timed(lambda: eng.stream_ceiling(next_batch(), K, out)))
print(f"workload cfg2 n={n} K={K} {prec}; algorithmic bytes/launch {bytes_alg}")
for v in variants:
    med = statistics.median(results[v])
    print(f"nt={v[0]} minw={v[1]} blocks/cu={v[2]:2d}: median {med:7.1f} us  min {min(results[v]):7.1f} us  -> {bytes_alg / med / 1e3:7.1f} GB/s")
for nt in (0, 1):
    if ceil[nt]:
        med = statistics.median(ceil[nt])
        print(f"stream ceiling nt={nt}: median {med:7.1f} us -> {bytes_alg / med / 1e3:7.1f} GB/s")

#!/usr/bin/env python3
"""Why does an fp32 trace leave the fp64 path?  For every ray whose surface sequence differs, look at the segment
where it first differs: both precisions start it from (nearly) the same point, and each names the leaf it hit by
the NEXT segment's origin.  The margin of those hit points to the aperture edge of their leaves (leaf frame) says
whether the disagreement is an edge case.  Prints the distribution; tests/test_gpu_fp32_contract.py asserts it.
    python tools/fp32_divergence.py cfg3|cfg5 [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import optable_amd as oa
from optable_amd import workloads as W
from optable_amd.batch import RayBatch
from optable_amd.fp32_audit import audit

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
wl = W.baseline_workloads(oa)[name]
table = oa.OpticalTable()
table.add_components(wl.components())
o, d, lam = wl.rays(n, 0)
q = 1j * np.pi * W.W0**2 / lam
K = wl.max_segments
s64 = table.trace_batch(RayBatch.from_arrays(o, d, wavelength=lam, q=q), max_segments=K)
s32 = table.trace_batch(RayBatch.from_arrays(o, d, wavelength=lam, q=q, precision="f32"), max_segments=K)
rep = audit(table.compile(), s64, s32, K)
print(f"{name}: {n} rays, same sequence {rep['same'].mean():.5f}, diverged {len(rep['margin'])}")
order = np.argsort(rep["margin"])
for w in order:
    print("   ray %6d k* %2d leaf64 %3d leaf32 %3d  hit-edge margin %.2e  start-edge margin %.2e  len64 %.3e len32 %.3e  pos err %.1e" % (
        rep["ray"][w], rep["kstar"][w], rep["leaf64"][w], rep["leaf32"][w], rep["margin"][w], rep["start_margin"][w], rep["len64"][w], rep["len32"][w], rep["pos_err"][w]))

#!/usr/bin/env python3
"""Diagnostic build only (make -C optable_amd/csrc liboptable_hip_stamp.so): what k_trace_pool's waves did in one cfg 5 launch.
    python tools/pool_stats.py [n_rays]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from optable_amd import abi

abi.LIB_PATH = os.path.join(ROOT, "optable_amd", "csrc", "liboptable_hip_stamp.so")
import optable_amd as oa
from optable_amd import workloads as W
from optable_amd.batch import RayBatch, SegmentBatch
from optable_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
wl = W.baseline_workloads(oa)["cfg5"]
eng = get_engine()
table = oa.OpticalTable()
table.add_components(wl.components())
eng.upload(table.compile())
o, d, lam = wl.rays(n, 0)
batch = RayBatch.from_arrays(o, d, wavelength=lam, q=1j * np.pi * W.W0**2 / lam, precision="f32")
out = None
for _ in range(2):
    out = eng.trace(batch, wl.max_segments, out=out, layout="append")
eng.timing(True)
out = eng.trace(batch, wl.max_segments, out=out, layout="append")
ms, cnt = eng.timing_read()
eng.timing(False)
print(eng.last_launch())
acc = (C.c_ulonglong * 12)()
eng.lib.ot_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
abi.check(eng.lib.ot_debug_stamps(eng._ctx, acc), eng.lib)
passes, rays, sleeps, lost, fills, filled, with_b, t_between, t_pass = [int(x) for x in acc][:9]
segs = int(out.count.abs().sum().item())
claim_waits = int(acc[9])
print(f"claim waits {claim_waits}")
print(f"n={n}: {ms / cnt:.3f} ms (diagnostic build); {segs} segments in {passes} passes = {rays / max(passes, 1):.1f} rays per pass "
      f"({with_b / max(passes, 1):.2f} with a second block); {sleeps / max(passes, 1):.2f} sleeps and {lost / max(passes, 1):.2f} lost locks per pass; "
      f"{fills} fills of {filled / max(fills, 1):.1f} blocks; {t_pass / max(passes, 1):.0f} cycles per pass, {t_between / max(passes, 1):.0f} between passes (s_memtime, 100 MHz ticks)")

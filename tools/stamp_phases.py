#!/usr/bin/env python3
"""Where a pass of k_trace_rolling spends its wave-cycles (diagnostic build, `make -C optable_amd/csrc
liboptable_hip_stamp.so`): s_memtime stamps around list read + loads / nearest hit / records + interaction /
compaction, with an s_waitcnt(0) at every stamp so that memory latency is charged to the phase that issued it.
    python tools/stamp_phases.py [cfg3|cfg5] [f32|f64] [n_rays]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3_000_000
wl = W.baseline_workloads(oa)[name]
eng = get_engine()
for k, v in (("CAP", abi.OPT_LIST_CAP), ("KERNEL", abi.OPT_KERNEL), ("RECLDS", abi.OPT_LDS_RECORDS), ("FLAT", abi.OPT_FLAT_QUEUE), ("INST", abi.OPT_INSTANCING), ("REFILL", abi.OPT_REFILL)):
    if os.environ.get(k):
        eng.set_option(v, int(os.environ[k]))
table = oa.OpticalTable()
table.add_components(wl.components())
eng.upload(table.compile())
o, d, lam = wl.rays(n, 0)
batch = RayBatch.from_arrays(o, d, wavelength=lam, q=1j * np.pi * W.W0**2 / lam, precision=prec)
LAYOUT = os.environ.get("LAYOUT", "slots")
out = SegmentBatch(n * wl.max_segments, prec, batch.device, block=(LAYOUT == "append"))
for _ in range(3):
    eng.trace(batch, wl.max_segments, out=out, layout=LAYOUT)
eng.timing(True)
eng.trace(batch, wl.max_segments, out=out, layout=LAYOUT)
ms, cnt = eng.timing_read()
eng.timing(False)
acc = (C.c_ulonglong * 12)()
eng.lib.ot_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
abi.check(eng.lib.ot_debug_stamps(eng._ctx, acc), eng.lib)
load, hit, inter, comp, passes = [int(x) for x in acc][:5]
walk, queue, test, verdict, slots, rounds = [int(x) for x in acc][5:11]
tot = load + hit + inter + comp
segs = int(out.count.abs().sum().item())
print(f"launch {eng.last_launch()}")
print(f"{name} {prec} n={n}: {ms / cnt:.3f} ms (stamped build), {passes} passes for {segs} segments = {segs / passes:.1f} lanes per pass")
for label, v in (("list + loads (waited)", load), ("nearest hit", hit), ("record + interact + state (waited)", inter), ("compaction", comp)):
    print(f"   {label:36s} {v / passes:9.0f} cycles per pass  {100 * v / tot:5.1f} %")
if not rounds and walk + queue + test + verdict:  # the linear pass (nearest_hit): its phases are part of "nearest hit" above
    for label, v in (("  group boxes + loop", walk), ("  planar leaves", queue), ("  gridded groups", test), ("  deferred curved leaves", verdict)):
        print(f"   {label:36s} {v / passes:9.0f} cycles per pass  {100 * v / tot:5.1f} %")
if rounds:  # the pair-queue walk (flat_grid_hit): its phases are part of none of the above ("nearest hit" holds only the setup)
    tot2 = tot + walk + queue + test + verdict
    print(f"   pair queue: {rounds / passes:.2f} rounds and {slots / passes:.2f} slots of 64 pairs per pass")
    fetch = int(acc[11])  # inside a slot: marker, scan, item, the ray by ds_bpermute and the node record, waited for
    tot2 += fetch
    for label, v in (("walk", walk), ("scan + queue", queue), ("pair tests: operands (waited)", fetch), ("pair tests: arithmetic + atomic", test), ("verdict", verdict)):
        print(f"   {label:36s} {v / passes:9.0f} cycles per pass  {100 * v / tot2:5.1f} % of {tot2 / passes:.0f}")

#!/usr/bin/env python3
"""PCIe-inclusive rate of BASELINE cfg 2 (for DESIGN.md §7; never the bench.py `value`): host numpy arrays in
(RayBatch.from_arrays = H2D), trace, full segment history back to host numpy (SegmentBatch.to_host = D2H)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); import numpy as np, torch
import optable_amd as oa
from optable_amd.batch import RayBatch
from optable_amd import dist as odist
from optable_amd import workloads as scenes  # the BASELINE configs (scene + ray generators)

n, K = 1_000_000, 5
table = oa.OpticalTable(); table.add_components(scenes.cfg2_components(oa))
o, d = scenes.cfg2_rays(n, 0)
q = 1j * np.pi * scenes.W0**2 / scenes.WL
scene = table.compile()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=q)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    segs = table.trace_batch(batch, max_segments=K, scene=scene)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    final = odist.final_state(segs).cpu()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    host = segs.to_host(reference_order=False)
    t4 = time.perf_counter()
    print(f"rep {rep}: H2D+pack {1e3*(t1-t0):7.2f} ms | trace {1e3*(t2-t1):6.2f} ms | final state D2H (96 MB) {1e3*(t3-t2):7.2f} ms | "
          f"full history D2H (520 MB) {1e3*(t4-t3):7.2f} ms")
    segs_n = int(np.sum(host["count"]))
    print(f"   intersections/s incl. H2D + final-state D2H: {segs_n*3/(t3-t0):.3e};  incl. full history D2H: {segs_n*3/((t2-t0)+(t4-t3)):.3e}")

# the same job through the streamed API: chunks on two HIP streams, copies overlapped with the trace, pinned results
for chunk in (1 << 18, 1 << 20):
    for hist in (False, True):
        res = table.trace_host(o, d, wavelength=scenes.WL, q=q, max_segments=K, chunk=chunk, history=hist)  # warm: pinned pools, engines
        del res
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = table.trace_host(o, d, wavelength=scenes.WL, q=q, max_segments=K, chunk=chunk, history=hist)
        dt = time.perf_counter() - t0
        print(f"trace_host chunk={chunk} history={hist}: {dt*1e3:7.2f} ms -> {int(res['count'].sum())*3/dt:.3e} intersections/s")

#!/usr/bin/env python3
"""A/B on one box: cfg 3 (fp32, 1e7 rays, append layout) through the per-wave lists (OT_OPT_REFILL = 0) and through
k_trace_refill, interleaved, library hipEvent timing.  env: N (rays), PREC, REPS, TICKET, LAYOUT, WL (cfg3 | cfg3b-style names)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import optable_amd as oa
from optable_amd import abi
from optable_amd import workloads as W
from optable_amd.batch import RayBatch, SegmentBatch
from optable_amd.engine import get_engine

if os.environ.get('OT_LIB'):
    abi.LIB_PATH = os.path.abspath(os.environ['OT_LIB'])
n = int(os.environ.get("N", 10_000_000))
prec = os.environ.get("PREC", "f32")
reps = int(os.environ.get("REPS", 5))
layout = os.environ.get("LAYOUT", "append")
eng = get_engine()
table = oa.OpticalTable()
table.add_components(W.cfg3_components(oa))
eng.upload(table.compile())
o, d = W.cfg3_rays(n, 2)
batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=1j * np.pi * W.W0**2 / W.WL, precision=prec)
K = 20
probe = eng.trace(batch, K, layout="append")
records = int(probe.count.abs().sum().item())
del probe
torch.cuda.empty_cache()
cap = eng.append_capacity(records) + 4_000_000
out = SegmentBatch(cap, prec, batch.device, block=True) if layout == "append" else SegmentBatch(n * K, prec, batch.device)


def timed(refill, ticket=0):
    eng.set_option(abi.OPT_REFILL, refill)
    eng.set_option(abi.OPT_REFILL_TICKET, ticket)
    for _ in range(2):
        eng.trace(batch, K, out=out, layout=layout)
    torch.cuda.synchronize()
    eng.timing(True)
    for _ in range(reps):
        eng.trace(batch, K, out=out, layout=layout)
    ms, cnt = eng.timing_read()
    eng.timing(False)
    return ms / cnt, eng.last_launch()


print(f"cfg3 {prec} {n} rays, {records} records, layout {layout}")
variants = [("lists", 0, 0), ("refill", 1, 0)] + [("refill t%d" % t, 1, t) for t in (64, 128, 512, 1024) if os.environ.get("TICKETS")]
for rnd in range(3):
    for name, r, t in variants:
        ms, info = timed(r, t)
        print(f"  round {rnd} {name:12s} {ms:7.3f} ms  {info}")
eng.set_option(abi.OPT_REFILL, 0)

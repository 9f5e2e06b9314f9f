#!/usr/bin/env python3
"""A/B on one box: ray trees through count + scan + emit (OT_OPT_GEN_ONEPASS = 0, OT_OPT_GEN_AHEAD = 0), the same with the
look-ahead emit pass (OT_OPT_GEN_AHEAD = 1: no count pass over the rays from the second generation on) and through the one-pass
generation kernel (k_gen_one), interleaved: cfg 4 with reflectivity 0.2 (1.28e7 trees x 12, fp64), cfg 3 with 10 % reflecting slabs (2e6
trees, fp32), a lattice of beam splitters (1e6 bushy trees, cap 12 / 24).  Library hipEvent time of the launches + wall."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import optable_amd as oa
from optable_amd import abi
from optable_amd import workloads as W
from optable_amd.batch import RayBatch
from optable_amd.engine import get_engine

if os.environ.get("OT_LIB"):  # kernel-variant experiments: an alternative build of the library
    abi.LIB_PATH = os.path.abspath(os.environ["OT_LIB"])
eng = get_engine()
if os.environ.get("CHUNK"):  # slots per claim of the append layouts
    eng.set_option(abi.OPT_APPEND_CHUNK, int(os.environ["CHUNK"]))
MODES = [int(m) for m in os.environ.get("MODES", "0,1,2").split(",")]
Q = 1j * np.pi * W.W0**2 / W.WL


def run(label, comps, batch, cap, out_cap):
    table = oa.OpticalTable()
    table.add_components(comps)
    eng.upload(table.compile())
    for rnd in range(3):
        for mode in MODES:
            eng.set_option(abi.OPT_GEN_ONEPASS, 1 if mode == 2 else 0)
            eng.set_option(abi.OPT_GEN_AHEAD, 1 if mode == 1 else 0)
            eng.trace_tree(batch, cap, out_capacity=out_cap)
            eng.timing(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            segs = eng.trace_tree(batch, cap, out_capacity=out_cap)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            ms, launches = eng.timing_read()
            eng.timing(False)
            print(f"{label:44s} round {rnd} {('two passes', 'look-ahead', 'one pass  ')[mode]}  device {ms:8.3f} ms  wall {wall:8.3f} ms  {launches:4d} timed regions  {segs.n_valid} segments", flush=True)
            del segs
    eng.set_option(abi.OPT_GEN_ONEPASS, -1)
    eng.set_option(abi.OPT_GEN_AHEAD, 1)
    for ql in [int(q) for q in os.environ.get("QL", "0").split(",")]:
        eng.set_option(abi.OPT_TREES_LDS_ENTRIES, ql)
        plan = eng.trees_plan(batch.precision, cap, batch.n)
        print(f"{label:44s} lane-per-tree plan {plan}")
        if plan["kernel"] and plan["full"]:
            from optable_amd.batch import SegmentBatch
            TL = os.environ.get("TREE_LAYOUT", "slots")
            slack = plan["chunk"] * plan["waves"] if TL == "append" else 0
            out = SegmentBatch(batch.n * cap + slack, batch.precision, batch.device, block=(TL == "append"))
            for rnd in range(4):
                eng.timing(True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                segs = eng.trace_trees(batch, cap, out=out, layout=TL)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) * 1e3
                ms, launches = eng.timing_read()
                eng.timing(False)
                print(f"{label:44s} round {rnd} lane per tree  device {ms:8.3f} ms  wall {wall:8.3f} ms  {launches:4d} timed regions  {int(segs.count.abs().sum())} segments  {eng.last_launch()}", flush=True)
            del out, segs
    eng.set_option(abi.OPT_TREES_LDS_ENTRIES, 0)


only = set(filter(None, os.environ.get("ONLY", "").split(",")))
if not only or "cfg4b" in only:
    o, d, _ = W.cfg4_rays(200_000, 4, n_wavelengths=1)
    base = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q, precision="f64")
    batch = base.multiplexed_in_wavelength(np.linspace(400e-7, 1100e-7, W.CFG4_WAVELENGTHS))
    run("cfg4 R=0.2: 1.28e7 trees x 12, fp64", W.cfg4_components(oa, reflectivity=0.2), batch, 12, batch.n * 13)
    del batch, base
if not only or "cfg3b" in only:
    o, d = W.cfg3_rays(2_000_000, 2)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q, precision="f32")
    run("cfg3 + 10 % reflecting slabs: 2e6 trees, fp32", W.cfg3_components(oa, slab_reflectivity=0.1), batch, 20, batch.n * 21)
    del batch
if not only or "lattice" in only:
    comps = []
    LK = int(os.environ.get("LATTICE_K", 5))  # 3: nine components, no grids: the planar preset (and the lane-per-tree kernel)
    for k in range(LK):
        comps.append(oa.BeamSplitter([2.0 * (k + 1), 0, 0], width=6, height=2, eta=0.5).RotZ(np.pi / 4))
        comps.append(oa.Mirror([2.0 * (k + 1), 3.0 + 0.1 * k, 0], radius=2).RotZ(-np.pi / 2))
        comps.append(oa.BeamSplitter([2.0 * (k + 1) + 1.0, 1.5, 0], width=6, height=2, eta=0.3).RotZ(-np.pi / 4))
    n = 1_000_000
    rng = np.random.default_rng(5)
    o = np.stack([np.zeros(n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.2, 0.2, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.02, 0.02, n), rng.uniform(-0.01, 0.01, n)], 1)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q)
    for prec in os.environ.get("PRECS", "f64").split(","):
        b = batch if prec == "f64" else batch.astype(prec)
        for cap in [int(c) for c in os.environ.get("CAPS", "12,24").split(",")]:
            run(f"beam-splitter lattice x{LK}: 1e6 trees, cap {cap}, {prec}", comps, b, cap, n * (cap + 1))

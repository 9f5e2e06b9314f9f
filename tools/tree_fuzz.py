#!/usr/bin/env python3
"""Differential fuzz of the lane-per-tree kernel (k_trace_trees): random branching scenes (tests/test_gpu_fuzz.py's generator:
lenses, mirrors, slabs, apertures + beam splitters, partially reflecting slabs and mirrors), random batch sizes (partial waves,
shares of uneven length), caps 1..60, 1..6 queue entries in LDS, lanes that refill one by one / in groups / 64 at a time, both output layouts, count-limited surfaces included — against the generation kernels
(ot_trace_tree_*), bit for bit in the reference's order, and (double precision, first 300 trees) against the oracle.
    python tools/tree_fuzz.py [n_cases] [first_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import optable_amd as oa
import scenes
from optable_amd import abi
from optable_amd.batch import RayBatch
from optable_amd.engine import get_engine
from oracle import oracle
from test_gpu_fuzz import random_branching_scene, random_large_scene, random_planar_scene

GLOBAL_IMAGE = bool(os.environ.get("GLOBAL_IMAGE"))  # every case through the global-image tree kernel
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = get_engine()
oracle.build()
bad = skipped = flat = 0
for seed in range(first, first + cases):
    rng = np.random.default_rng(770000 + seed)
    table = oa.OpticalTable()
    pick = rng.uniform()
    if pick < 0.15:  # planar components under a top-level grid (the pair-queue presets; with irises: polygon / boolean apertures) + splitters
        comps = random_planar_scene(oa, rng, irises=bool(rng.integers(0, 2)))
        for _ in range(int(rng.integers(1, 4))):
            comps.append(oa.BeamSplitter([rng.uniform(3, 35), rng.uniform(-5, 5), 0.0], width=2.5, height=2, eta=rng.uniform(0.3, 0.7)).RotZ(rng.uniform(-1.5, 1.5)))
        comps.append(oa.GlassSlab([rng.uniform(3, 35), rng.uniform(-4, 4), 0.0], width=2, height=2, thickness=0.4, n1=1, n2=1.5, reflectivity=0.15).RotZ(rng.uniform(-0.5, 0.5)))
        wide = True
    elif pick < 0.30:  # grids, lens arrays, dispersion (the all-features preset) + something that splits
        comps = random_large_scene(oa, rng)
        comps.append(oa.BeamSplitter([rng.uniform(2, 20), rng.uniform(-2, 2), 0.0], width=3, height=3, eta=rng.uniform(0.3, 0.7)).RotZ(rng.uniform(-1, 1)))
        comps.append(oa.Mirror([rng.uniform(2, 25), rng.uniform(-3, 3), 0.0], radius=1.5, reflectivity=0.6, transmission=0.4).RotZ(rng.uniform(-1, 1)))
        wide = True
    else:
        comps = random_branching_scene(oa, rng)
        wide = False
    table.add_components(comps)
    scene = table.compile()
    eng.upload(scene)
    n = int(rng.choice([1, 63, 64, 65, 200, 1000, 5000, 20000]))
    cap = int(rng.integers(1, 61))
    ql = int(rng.integers(1, 7))
    prec = "f64" if rng.uniform() < 0.6 else "f32"
    half = 6 if wide else 3
    o = np.stack([np.zeros(n), rng.uniform(-half, half, n), rng.uniform(-0.3, 0.3, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.12, 0.12, n), rng.uniform(-0.02, 0.02, n)], 1)
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j * np.pi * scenes.W0**2 / scenes.WL, precision=prec)
    eng.set_option(abi.OPT_TREES_LDS_ENTRIES, ql)
    eng.set_option(abi.OPT_TREES_REFILL_AT, int(rng.choice([1, 16, 16, 40, 64])))
    eng.set_option(abi.OPT_TREES_GLOBAL_IMAGE, int(GLOBAL_IMAGE or rng.uniform() < 0.15))  # (the kernel of scenes no LDS holds, on scenes it does)
    plan = eng.trees_plan(prec, cap, n)
    if not plan["kernel"]:
        skipped += 1
        continue
    gens = eng.trace_tree(batch, cap)  # (count-limited surfaces: a fresh column of the counts table per tree, in both paths)
    g = gens.to_host(reference_order=True)
    layouts = ["append"] + (["slots"] if plan["slots"] else [])
    ok = True
    for layout in layouts:
        t = eng.trace_trees(batch, cap, layout=layout)
        flat += eng.last_launch()["pair_queue"] & 1
        if not plan["full"] and bool((t.count < 0).any()):
            continue
        h = t.to_host(reference_order=True)
        same = torch.equal(t.capped, gens.capped) and all(np.array_equal(h[f], g[f]) for f in abi.SEG_FIELDS + ("ray", "surface"))
        if scene.limited:
            same = same and torch.equal(t.counts_table, gens.counts_table)
        if not same:
            ok = False
            print(f"seed {seed}: {layout} differs from the generations (n={n} cap={cap} ql={ql} {prec}, plan {plan})", flush=True)
    if ok and prec == "f64":
        m = min(n, 300)
        small = batch.slice(0, m)
        got = eng.trace_trees(small, cap, layout="append").to_host(reference_order=True)
        ref = oracle.trace(scene, small.to_host(), max_trace_num=cap)
        seq_g = [[] for _ in range(m)]
        seq_r = [[] for _ in range(m)]
        for r, s in zip(got["ray"], got["surface"]):
            seq_g[r].append(int(s))
        for r, s in zip(ref["ray"], ref["surface"]):
            seq_r[r].append(int(s))
        differ = sum(a != b for a, b in zip(seq_g, seq_r))
        if differ > max(1, 0.01 * m):  # (a hit within an ulp of an edge may differ between compilers; more is a defect)
            ok = False
            print(f"seed {seed}: {differ} of {m} trees differ from the oracle (cap={cap} ql={ql})", flush=True)
    bad += not ok
    if seed % 20 == 19:
        print(f"... {seed - first + 1} cases, {bad} bad, {skipped} skipped", flush=True)
eng.set_option(abi.OPT_TREES_LDS_ENTRIES, 0)
eng.set_option(abi.OPT_TREES_REFILL_AT, 16)
eng.set_option(abi.OPT_TREES_GLOBAL_IMAGE, 0)
print(f"{cases} cases: {bad} bad, {skipped} skipped (no lane-per-tree kernel for the scene); {flat} launches searched through the pair queue")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Fixture g27 (the reference's largest example: 7,689 leaves, one ray, 3,202 segments) through the tree kernels for scenes no LDS holds,
in both precisions, with the node records in LDS (OT_OPT_TREES_GLOBAL_IMAGE = 1) and with everything read from global memory (= 2).
Round 4: 84-88 ms / 96-99 ms in BOTH precisions — the launch is one lane's dependent chain (issue + L2 round trips), not arithmetic."""
import os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from optable_amd import abi
if os.environ.get('OT_LIB'):  # an A/B build of the library (make -C optable_amd/csrc variant V=... EXTRA=...)
    abi.LIB_PATH = os.path.abspath(os.environ['OT_LIB'])
from optable_amd.batch import RayBatch
from optable_amd.engine import get_engine
from optable_amd.scene import CompiledScene
gold = dict(np.load(os.path.join(ROOT, "tests/golden/g27_real_example.npz")))
scene = CompiledScene.from_tables(gold)
eng = get_engine(); eng.upload(scene)
for prec in ("f64", "f32"):
    for glob in (1, 2):
        eng.set_option(abi.OPT_TREES_GLOBAL_IMAGE, glob)
        ts=[]
        for it in range(4):
            b = RayBatch.from_arrays(gold["in_origin"], gold["in_direction"], wavelength=gold["in_wavelength"], q=gold["in_q"], device=eng.device, normalize=False, precision=prec)
            counts = torch.zeros((len(scene.limited), 1), dtype=torch.int32, device=eng.device)
            eng.timing(True)
            segs = eng.trace_branching(b, 100000, counts=counts, distinct_ids=True)
            n = len(segs.to_host(reference_order=True)["ray"])
            ms, cnt = eng.timing_read(); eng.timing(False)
            ts.append(ms)
        print(prec, "global_image", glob, "segments", n, "kernel ms", [round(t,1) for t in ts], eng.last_launch())

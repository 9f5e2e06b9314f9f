#!/usr/bin/env python3
"""Where cfg 3's time goes: the same 8x4 lattice of poses populated with one component kind at a time."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); import numpy as np, torch
import optable_amd as oa
from optable_amd.batch import RayBatch, SegmentBatch
from optable_amd.engine import get_engine
from optable_amd import workloads as scenes  # the BASELINE configs (scene + ray generators)

def lattice(kind_of):
    rng = np.random.default_rng(1)
    comps = []
    for ix in range(8):
        for iy in range(4):
            origin = [4 * (ix + 1), 3 * (iy - 1.5), 0]
            a = rng.uniform(-np.pi, np.pi)
            kind = kind_of(ix, iy)
            if kind == "Mirror": c = oa.Mirror(origin, radius=1).RotZ(a)
            elif kind == "Lens": c = oa.Lens(origin, focal_length=rng.uniform(4, 12), radius=1).RotZ(0.2 * a)
            elif kind == "GlassSlab": c = oa.GlassSlab(origin, width=2, height=2, thickness=0.5, n1=1, n2=1.5).RotZ(0.3 * a)
            elif kind == "Block": c = oa.Block(origin, width=2, height=2).RotZ(0.3 * a)
            else: c = oa.Prism(origin, width=1.5, height=2, n1=1, n2=1.5).RotZ(a)
            comps.append(c)
    return comps

eng = get_engine()
n, K = int(os.environ.get("N", 4_000_000)), 20
o, d = scenes.cfg3_rays(n, 2)
names = ["Mirror", "Lens", "GlassSlab", "Prism"]
variants = {"cfg3 (mixed)": lambda ix, iy: names[(ix + iy) % 4]}
for nm in names + ["Block"]:
    variants["all " + nm] = (lambda k: (lambda ix, iy: k))(nm)
for prec in os.environ.get("PREC", "f64,f32").split(","):
    for name, kind_of in variants.items():
        t = oa.OpticalTable(); t.add_components(lattice(kind_of))
        sc = t.compile(); eng.upload(sc)
        b = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j*np.pi*scenes.W0**2/scenes.WL, precision=prec)
        out = SegmentBatch(n*K, prec)
        eng.set_option(4, 2)
        eng.trace(b, K, out=out)
        eng.timing(True)
        for _ in range(3): eng.trace(b, K, out=out)
        ms, cnt = eng.timing_read(); eng.timing(False)
        segs = int(out.count.abs().sum().item())
        print(f"{prec} {name:16s} leaves {sc.n_leaves:3d} {ms/cnt:8.2f} ms  segs/ray {segs/n:5.2f}  {ms/cnt*1e6/segs:7.3f} ns/seg", flush=True)
        del out, b
        torch.cuda.empty_cache()

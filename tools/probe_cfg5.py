import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); import numpy as np, torch
import optable_amd as oa
from optable_amd.batch import RayBatch, SegmentBatch
from optable_amd.engine import get_engine
from optable_amd import workloads as scenes  # the BASELINE configs (scene + ray generators)
eng = get_engine()
n, K = int(os.environ.get('N', 4_000_000)), 50
o, d = scenes.cfg5_rays(n, 3)
full = scenes.cfg5_components(oa)
variants = {"full": full, "no lens": [full[0], full[2]], "no MMA (mirror+lens+flat mirror)": [full[0], full[1], oa.SquareMirror([15, 0, 0], 6, 6).RotZ(np.pi)],
            "two flat mirrors": [full[0], oa.SquareMirror([15, 0, 0], 6, 6).RotZ(np.pi)]}
only = os.environ.get("VARIANT")
for prec in os.environ.get("PREC", "f64,f32").split(","):
    for name, comps in variants.items():
        if only and not name.startswith(only):
            continue
        t = oa.OpticalTable(); t.add_components(comps)
        sc = t.compile(); eng.upload(sc)
        b = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j*np.pi*scenes.W0**2/scenes.WL, precision=prec)
        out = SegmentBatch(n*K, prec)
        eng.set_option(4, 2)  # blocked kernel for all, same machinery
        eng.trace(b, K, out=out)
        eng.timing(True)
        for _ in range(3): eng.trace(b, K, out=out)
        ms, cnt = eng.timing_read(); eng.timing(False)
        segs = int(out.count.abs().sum().item())
        print(f"{prec} {name:36s} {ms/cnt:8.2f} ms  segs/ray {segs/n:5.1f}  {ms/cnt*1e6/segs:7.3f} ns/seg", flush=True)
        del out, b
        torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Throughput of the BASELINE configs 2-5 on one GPU (device-resident batches, library hipEvent
timing).  Not the bench.py line: a survey table for DESIGN.md.  Sizes via env N2..N5."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import optable_amd as oa
from optable_amd import abi
from optable_amd.batch import RayBatch, SegmentBatch
from optable_amd.engine import get_engine
from optable_amd import workloads as scenes  # the BASELINE configs (scene + ray generators)

if os.environ.get('OT_LIB'):  # kernel-variant experiments: an alternative build of the library
    abi.LIB_PATH = os.path.abspath(os.environ['OT_LIB'])
eng = get_engine()
if os.environ.get('RGC'):  # experiment: top-level grid resolution (cells per component)
    import optable_amd.scene as _scene
    _scene.ROOT_GRID_CELLS_PER_COMPONENT = float(os.environ['RGC'])
if os.environ.get('RGA'):  # experiment: aspect of the top-level grid's cells
    import optable_amd.scene as _scene
    _scene.ROOT_GRID_ASPECT = float(os.environ['RGA'])
if os.environ.get('GENDROP'):  # generation kernels: children of trees whose budget ends with this generation (0: emit them)
    eng.set_option(abi.OPT_GEN_DROP_DOOMED, int(os.environ['GENDROP']))
if os.environ.get('LDSKB'):
    eng.set_option(abi.OPT_LDS_LIMIT_KB, int(os.environ['LDSKB']))
if os.environ.get('FLAT'):
    eng.set_option(abi.OPT_FLAT_QUEUE, int(os.environ['FLAT']))
if os.environ.get('RECLDS'):
    eng.set_option(abi.OPT_LDS_RECORDS, int(os.environ['RECLDS']))
if os.environ.get('MIX'):
    eng.set_option(abi.OPT_MIX_GENERATIONS, int(os.environ['MIX']))
if os.environ.get('NT'):
    eng.set_option(abi.OPT_NT_STORES, int(os.environ['NT']))
if os.environ.get('PAIR'):
    eng.set_option(abi.OPT_PAIR_STORES, int(os.environ['PAIR']))
if os.environ.get('CAP'):
    eng.set_option(abi.OPT_LIST_CAP, int(os.environ['CAP']))
if os.environ.get('KERNEL'):
    eng.set_option(abi.OPT_KERNEL, int(os.environ['KERNEL']))
if os.environ.get('MINW'):
    eng.set_option(abi.OPT_MIN_WAVES, int(os.environ['MINW']))
if os.environ.get('INST'):
    eng.set_option(abi.OPT_INSTANCING, int(os.environ['INST']))
if os.environ.get('CHUNK'):
    eng.set_option(abi.OPT_APPEND_CHUNK, int(os.environ['CHUNK']))
LAYOUTS = os.environ.get('LAYOUT', 'slots').split(',')  # slots, append or both ("slots,append")
if os.environ.get('BPC'):
    eng.set_option(abi.OPT_BLOCKS_PER_CU, int(os.environ['BPC']))
Q = lambda wl: 1j * np.pi * scenes.W0**2 / wl


def run(name, comps, o, d, wl, K, prec, reps=5):
    base = int(os.environ.get('KERNEL', 0))
    for kern in ((1, 2) if os.environ.get('BOTH') else (base,)):
        eng.set_option(abi.OPT_KERNEL, kern)
        _run(f'{name} k{kern}', comps, o, d, wl, K, prec, reps)
    eng.set_option(abi.OPT_KERNEL, base)


def _run(name, comps, o, d, wl, K, prec, reps=5):
    table = oa.OpticalTable()
    table.add_components(comps)
    scene = table.compile()
    eng.upload(scene)
    n = len(o)
    batch = RayBatch.from_arrays(o, d, wavelength=wl, q=Q(wl), precision=prec)
    if os.environ.get('SORTRAYS'):  # spatially coherent input order (RayBatch.sorted_spatially)
        batch, _ = batch.sorted_spatially()
    S = scene.n_leaves
    b = 104 if prec == "f64" else 56
    fused = scene.max_children <= 1
    if scene.max_children == 2 and not scene.limited:
        out = SegmentBatch(n * K, prec, batch.device)
        eng.trace(batch, K, out=out)
        fused = not bool((out.count < 0).any())
    if fused:
        for layout in LAYOUTS:
            out = None
            torch.cuda.empty_cache()
            if layout == "append":  # sized by the records of a first trace, like a caller who knows the job
                probe = eng.trace(batch, K, layout="append")
                cap = eng.append_capacity(int(probe.count.abs().sum().item()))
                del probe
                torch.cuda.empty_cache()
                out = SegmentBatch(cap, prec, batch.device, block=True)
            elif layout == "tiled":
                if scene.n_nodes >= 24:
                    continue
                out = SegmentBatch(n * K, prec, batch.device, tiled=True)
            else:
                out = SegmentBatch(n * K, prec, batch.device)
            for _ in range(200 if n * K < 2e7 else 1):  # light launches: clocks up first
                eng.trace(batch, K, out=out, layout=layout)
            eng.timing(True)
            for _ in range(reps * (20 if n * K < 2e7 else 1)):
                eng.trace(batch, K, out=out, layout=layout)
            ms, cnt = eng.timing_read()
            eng.timing(False)
            segs = int(out.count.abs().sum().item())
            t = ms / cnt / 1e3
            mode = {"slots": "fused", "append": "append", "tiled": "tiled"}[layout]
            if os.environ.get("SHAPE"):
                print("   launch:", eng.last_launch(), flush=True)
            if layout != LAYOUTS[-1]:
                gbs = (n * b + segs * b) / t / 1e9
                print(f"{name:28s} {prec} {mode:11s} n={n:9d} S={S:3d} K={K:2d} segs/ray={segs / n:5.2f} time={t * 1e3:9.3f} ms "
                      f"{segs / t:10.3e} seg/s {segs * S / t:10.3e} isect/s  {gbs:7.1f} GB/s ({gbs / 80:4.1f}% of 8 TB/s)", flush=True)
        if os.environ.get("CEILING"):  # the same streams with no tracing (fixed K records per ray)
            for _ in range(20):
                eng.stream_ceiling(batch, K, out)
            eng.timing(True)
            for _ in range(reps):
                eng.stream_ceiling(batch, K, out)
            cms, ccnt = eng.timing_read()
            eng.timing(False)
            cb = n * b + n * K * b
            print(f"   stream ceiling ({out.layout}) for n={n} K={K}: {cms / ccnt:.3f} ms = {cb / (cms / ccnt / 1e3) / 1e9:.0f} GB/s "
                  f"(trace moves {(n * b + segs * b) / 1e9:.2f} GB in {t * 1e3:.3f} ms)", flush=True)
    else:
        eng.timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.trace_tree(batch, K)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        kms, kcnt = eng.timing_read()
        eng.timing(False)
        segs = out.n_valid
        mode = "generations"
        print(f"   generations: {kcnt} launches sequences, device time {kms:.2f} ms of {t * 1e3:.2f} ms wall", flush=True)
    gbs = (n * b + segs * b) / t / 1e9
    print(f"{name:28s} {prec} {mode:11s} n={n:9d} S={S:3d} K={K:2d} segs/ray={segs / n:5.2f} time={t * 1e3:9.3f} ms "
          f"{segs / t:10.3e} seg/s {segs * S / t:10.3e} isect/s  {gbs:7.1f} GB/s ({gbs / 80:4.1f}% of 8 TB/s)", flush=True)
    del out, batch
    torch.cuda.empty_cache()


n2 = int(os.environ.get("N2", 1_000_000))
n3 = int(os.environ.get("N3", 2_000_000))
n4 = int(os.environ.get("N4", 100_000))
n5 = int(os.environ.get("N5", 200_000))
only = set(filter(None, os.environ.get("ONLY", "").split(",")))   # e.g. ONLY=cfg3,cfg5
want = lambda c: (not only and c not in ("cfg4b", "cfg3b")) or c in only
for prec in filter(None, os.environ.get("PREC", "f64,f32").split(",")):
    if want("cfg2"):
        o, d = scenes.cfg2_rays(n2, 0)
        run("cfg2 lens+mirrorpair", scenes.cfg2_components(oa), o, d, scenes.WL, 5, prec)
    if want("cfg3"):
        o, d = scenes.cfg3_rays(n3, 2)
        run("cfg3 32 mixed components", scenes.cfg3_components(oa), o, d, scenes.WL, 20, prec)
    if want("cfg4"):
        nwl = 64
        rng = np.random.default_rng(4)
        jit = rng.uniform(-0.3, 0.3, (n4, 2))
        ob = np.stack([np.full(n4, -3.0), 2 + jit[:, 0], jit[:, 1]], 1)
        db = np.tile([np.cos(np.pi / 6), -np.sin(np.pi / 6), 0.0], (n4, 1))
        wl = np.repeat(np.linspace(400e-7, 1100e-7, nwl), n4)
        slab = [oa.GlassSlab([0, 0, 0], width=2, height=2, thickness=0.5, n1=oa.Vacuum(), n2=oa.Glass_NBK7(), reflectivity=0)]
        run("cfg4 NBK7 slab x64 wl", slab, np.tile(ob, (nwl, 1)), np.tile(db, (nwl, 1)), wl, int(os.environ.get("K4", 8)), prec)
    if want("cfg4b") and prec == "f64":   # branching variant of cfg 4: reflectivity 0.2, ray trees
        nwl, nb = 64, int(os.environ.get("N4B", 20_000))
        rng = np.random.default_rng(4)
        jit = rng.uniform(-0.3, 0.3, (nb, 2))
        ob = np.stack([np.full(nb, -3.0), 2 + jit[:, 0], jit[:, 1]], 1)
        db = np.tile([np.cos(np.pi / 6), -np.sin(np.pi / 6), 0.0], (nb, 1))
        wl = np.repeat(np.linspace(400e-7, 1100e-7, nwl), nb)
        slab = [oa.GlassSlab([0, 0, 0], width=2, height=2, thickness=0.5, n1=oa.Vacuum(), n2=oa.Glass_NBK7(), reflectivity=0.2)]
        run("cfg4b NBK7 slab R=0.2 trees", slab, np.tile(ob, (nwl, 1)), np.tile(db, (nwl, 1)), wl, 12, prec)
    if want("cfg3b") and prec == "f32":   # heavy branching: cfg 3 with 10 % reflecting slab faces, trees capped at 20 segments
        o, d = scenes.cfg3_rays(int(os.environ.get("N3B", 2_000_000)), 2)
        for reuse in (0, 1):
            eng.set_option(abi.OPT_GEN_REUSE, reuse)
            run(f"cfg3b slabs R=0.1 reuse={reuse}", scenes.cfg3_components(oa, slab_reflectivity=0.1), o, d, scenes.WL, 20, prec)
        eng.set_option(abi.OPT_GEN_REUSE, -1)
    if want("cfg5"):
        o, d = scenes.cfg5_rays(n5, 3)
        run("cfg5 asphere+MMA16x16", scenes.cfg5_components(oa), o, d, scenes.WL, 50, prec)

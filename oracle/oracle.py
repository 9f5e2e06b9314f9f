"""ctypes front end of the CPU oracle (oracle/ot_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (optable_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from optable_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "ot_oracle.c")
LIB = os.path.join(_HERE, "libot_oracle.so")
_lib = None

_dpp = C.POINTER(C.POINTER(C.c_double))


def build(force=False):
    """gcc -O2 the restatement into oracle/libot_oracle.so (skipped when up to date)."""
    hdr = os.path.join(_HERE, "..", "include", "optable_hip.h")
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= max(os.path.getmtime(SRC), os.path.getmtime(hdr))):
        return LIB
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off",
                           "-o", LIB, SRC, "-lm"])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.ot_oracle_trace.restype = C.c_int64
        lib.ot_oracle_trace.argtypes = [C.POINTER(abi.OtSceneDesc), _dpp, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_int32, _dpp, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int32, C.c_void_p]
        lib.ot_oracle_monitor.restype = C.c_int64
        lib.ot_oracle_monitor.argtypes = [C.POINTER(abi.OtMonitor), _dpp, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ot_oracle_slab.restype = C.c_int
        lib.ot_oracle_slab.argtypes = [C.c_void_p] * 5
        _lib = lib
    return _lib


def _ptr_table(arrays):
    tab = (C.POINTER(C.c_double) * len(arrays))()
    for k, a in enumerate(arrays):
        tab[k] = a.ctypes.data_as(C.POINTER(C.c_double))
    return tab


def trace(scene, rays, max_trace_num=2000, counts=None, n_classes=None, capacity=None):
    """Trace host rays through a CompiledScene.

    rays: dict with the 12 `abi.RAY_FIELDS` (float64 arrays), `id`, `flags`, optional `length`.
    Returns a dict of the 12 `abi.SEG_FIELDS` + `ray`, `surface` in the reference's output order
    (input-ray-major, FIFO within a tree), `capped` per input ray and the updated `counts`.
    """
    lib = load()
    n = len(rays["ox"])
    rf = [np.ascontiguousarray(rays[f], dtype=np.float64) for f in abi.RAY_FIELDS]
    ids = np.ascontiguousarray(rays.get("id", np.arange(n)), dtype=np.int32)
    flags = np.ascontiguousarray(rays.get("flags", np.zeros(n)), dtype=np.int32)
    length = rays.get("length")
    if length is not None:
        length = np.ascontiguousarray(length, dtype=np.float64)
    n_slots = len(scene.limited)
    if n_classes is None:
        n_classes = int(ids.max()) + 1 if n else 1
    if counts is None:
        counts = np.zeros((max(n_slots, 1), n_classes), dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    if capacity is None:
        capacity = int(min(n * max_trace_num, max(n * 64, 1 << 16)))
    desc = scene.desc()
    while True:
        sf = [np.empty(capacity, dtype=np.float64) for _ in abi.SEG_FIELDS]
        seg_ray = np.empty(capacity, dtype=np.int32)
        seg_surface = np.empty(capacity, dtype=np.int32)
        capped = np.zeros(max(n, 1), dtype=np.int32)
        work = counts.copy()
        nseg = lib.ot_oracle_trace(C.byref(desc), _ptr_table(rf), ids.ctypes.data, flags.ctypes.data,
                                   None if length is None else length.ctypes.data, n, int(max_trace_num),
                                   _ptr_table(sf), seg_ray.ctypes.data, seg_surface.ctypes.data, capacity,
                                   work.ctypes.data, n_classes, capped.ctypes.data)
        if nseg >= 0:
            break
        capacity *= 4
    out = {f: a[:nseg] for f, a in zip(abi.SEG_FIELDS, sf)}
    out["ray"], out["surface"] = seg_ray[:nseg], seg_surface[:nseg]
    out["capped"], out["counts"] = capped[:n], work
    return out


def monitor_record(monitor_struct, segs):
    lib = load()
    nseg = len(segs["ox"])
    sf = [np.ascontiguousarray(segs[f], dtype=np.float64) for f in abi.SEG_FIELDS]
    idx = np.empty(nseg, dtype=np.int64)
    P = [np.empty(nseg) for _ in range(3)]
    t = np.empty(nseg)
    nh = lib.ot_oracle_monitor(C.byref(monitor_struct), _ptr_table(sf), None, nseg, idx.ctypes.data,
                               P[0].ctypes.data, P[1].ctypes.data, P[2].ctypes.data, t.ctypes.data)
    return idx[:nh], np.stack([p[:nh] for p in P], axis=1), t[:nh]


def slab(o, d, box):
    lib = load()
    o, d, box = (np.ascontiguousarray(x, dtype=np.float64) for x in (o, d, box))
    t1, t2 = np.zeros(1), np.zeros(1)
    hit = lib.ot_oracle_slab(o.ctypes.data, d.ctypes.data, box.ctypes.data, t1.ctypes.data, t2.ctypes.data)
    return float(t1[0]), float(t2[0]), bool(hit)


def q_tolerance(scene, rays, ref, max_trace_num, factor=50.0):
    """Per-segment tolerance for comparing Gaussian q against `ref = trace(scene, rays, max_trace_num)`.

    q is reproducible to 1e-9 until it has passed an aspheric interface.  Behind one, the REFERENCE ALGORITHM's own
    arithmetic is ill-conditioned: ASphere.roc is a 3-point finite difference with h = 1e-4 * radius (surfaces.py:
    355-369), which amplifies rounding by 1/h^2 ~ 1e7-1e8 — its own q moves by up to 1e-4 relative when the inputs move
    by one ulp (tests/test_oracle_golden.py::test_asphere_q_is_ill_conditioned).  So the tolerance there is what that
    algorithm itself does under a last-digit change of its input: the oracle is re-run on origins moved by +-2 ulp in
    y / z, and `factor` times the spread of ITS q per segment (+ 1e-6 relative) is the bound.  Returns (tol, clean):
    absolute tolerance on |q_got - q_ref| per segment, and the mask of segments whose q has not met an asphere yet
    (tol = 1e-9 relative there).  Segments of rays whose perturbed trace takes another path get an infinite tolerance
    (none on the BASELINE scenes)."""
    spread = np.zeros(len(ref["ray"]))
    unstable = np.zeros(len(ref["ray"]), dtype=bool)
    for sy, sz in ((1, 1), (-1, 1), (1, -1)):
        moved = dict(rays)
        moved["oy"] = np.asarray(rays["oy"], dtype=np.float64) * (1 + sy * 4.4e-16)
        moved["oz"] = np.asarray(rays["oz"], dtype=np.float64) * (1 + sz * 4.4e-16)
        alt = trace(scene, moved, max_trace_num=max_trace_num)
        if len(alt["ray"]) == len(ref["ray"]) and np.array_equal(alt["surface"], ref["surface"]):
            spread = np.maximum(spread, np.hypot(alt["q_re"] - ref["q_re"], alt["q_im"] - ref["q_im"]))
        else:  # some ray sits on an aperture edge: no statement about the segments of rays whose path changed
            def paths(x):
                seq = {}
                for ray, surf in zip(x["ray"].tolist(), x["surface"].tolist()):
                    seq.setdefault(ray, []).append(surf)
                return seq
            pa, pb = paths(ref), paths(alt)
            bad = [ray for ray in pa if pa[ray] != pb.get(ray)]
            unstable |= np.isin(ref["ray"], bad)
    asph = np.array([type(c.surface).__name__ == "ASphere" for c in scene.leaves] + [False])
    hit_asph = asph[np.where(ref["surface"] >= 0, ref["surface"], len(asph) - 1)]
    behind = np.zeros(len(hit_asph), dtype=bool)  # True once an earlier segment of the same ray ended on an asphere
    seen = {}
    for s, (ray, h) in enumerate(zip(ref["ray"].tolist(), hit_asph.tolist())):
        behind[s] = seen.get(ray, False)
        if h:
            seen[ray] = True
    mag = np.hypot(ref["q_re"], ref["q_im"])
    tol = np.where(behind, factor * spread + 1e-6 * mag, 1e-9 * mag + 1e-9)
    tol[unstable] = np.inf
    return tol, ~behind

"""Audit of a single-precision trace against the double-precision trace of the same rays.

Single precision cannot promise the fp64 surface sequence for every ray over 20-50 bounces: a ray that grazes an
aperture edge flips between hit and miss at 6e-8 relative, and every bounce amplifies the difference.  What it can
promise is that rays leave the fp64 path only THERE: `audit` finds, for every ray whose sequences differ, the first
segment that ends on different leaves and measures how far the two hit points lie from the aperture edge of their
leaves, in the leaf frame (circle / sphere cap / asphere: |r - radius|; rectangle: distance to the nearest side).
"""
import numpy as np


def _edge_margin(comp, P):
    """Distance of lab point P (on or near the leaf's surface) to the leaf's aperture edge; inf when the shape has
    no simple edge description (polygons, booleans: not audited)."""
    surf = comp.surface
    M = np.asarray(comp.transform_matrix, dtype=float)
    loc = M.T @ (np.asarray(P, dtype=float) - np.asarray(comp.origin, dtype=float))
    kind = type(surf).__name__
    if kind == "Circle":
        return abs(np.linalg.norm(loc) - surf.radius)
    if kind == "Rectangle":
        return min(abs(surf.width / 2 - abs(loc[1])), abs(surf.height / 2 - abs(loc[2])))
    if kind == "Sphere":  # cap of height h: aperture radius sqrt(R^2 - (R-h)^2) around the x axis
        a = np.sqrt(max(surf.radius**2 - (surf.radius - surf.height) ** 2, 0.0))
        return abs(np.hypot(loc[1], loc[2]) - a)
    if kind == "ASphere":
        return abs(np.hypot(loc[1], loc[2]) - surf.radius)
    return np.inf


def audit(scene, s64, s32, K):
    """s64 / s32: SegmentBatch of the same rays (any layout of the non-branching trace).  Returns arrays over the diverged rays:
    ray, kstar (first differing segment), leaf64 / leaf32 (leaf ids, -1 = escaped), margin (smallest edge margin of
    the hit points involved), pos_err (|origin64 - origin32| of segment kstar: how far apart the two traces were
    when they disagreed), plus `same` (bool per ray)."""
    s64, s32 = s64.as_kray_slots(K), s32.as_kray_slots(K)  # (whatever layout the traces were written in)
    n = s64.n_rays
    c64, c32 = np.abs(s64.count.cpu().numpy()), np.abs(s32.count.cpu().numpy())
    def f(s, name):
        with np.errstate(invalid="ignore"):  # unused slots are uninitialised memory; they are masked by the counts
            return s.field(name).cpu().numpy().reshape(K, n).astype(np.float64)

    surf64 = s64.surface.cpu().numpy().reshape(K, n)
    surf32 = s32.surface.cpu().numpy().reshape(K, n)
    O64 = np.stack([f(s64, "ox"), f(s64, "oy"), f(s64, "oz")], axis=-1)
    O32 = np.stack([f(s32, "ox"), f(s32, "oy"), f(s32, "oz")], axis=-1)
    valid = np.arange(K)[:, None] < np.minimum(c64, c32)[None, :]
    differ = (surf64 != surf32) & valid
    same = (c64 == c32) & ~differ.any(axis=0)
    rays = np.nonzero(~same)[0]
    L64, L32 = f(s64, "length"), f(s32, "length")
    out = {k: [] for k in ("ray", "kstar", "leaf64", "leaf32", "margin", "pos_err", "start_margin", "len64", "len32")}
    for i in rays:
        ks = int(np.argmax(differ[:, i])) if differ[:, i].any() else int(min(c64[i], c32[i])) - 1
        a, b = int(surf64[ks, i]), int(surf32[ks, i])
        margins = []
        # a trace names the point where it hit by the origin of its NEXT segment
        if a >= 0 and ks + 1 < c64[i]:
            margins.append(_edge_margin(scene.leaves[a], O64[ks + 1, i]))
            if b >= 0:  # ... and where was that point relative to the leaf the other precision chose?
                margins.append(_edge_margin(scene.leaves[b], O64[ks + 1, i]))
        if b >= 0 and ks + 1 < c32[i]:
            margins.append(_edge_margin(scene.leaves[b], O32[ks + 1, i]))
            if a >= 0:
                margins.append(_edge_margin(scene.leaves[a], O32[ks + 1, i]))
        out["ray"].append(i)
        out["kstar"].append(ks)
        out["leaf64"].append(a)
        out["leaf32"].append(b)
        out["margin"].append(min(margins) if margins else np.inf)
        out["pos_err"].append(float(np.linalg.norm(O64[ks, i] - O32[ks, i])))
        # where the deciding segment STARTS: a start near an edge of the leaf it leaves (a prism corner, the rim of a
        # lens) is where the next face lies within the self-hit guard (1e-9 in fp64, 1e-5 in fp32: DESIGN.md §5)
        prev = int(surf64[ks - 1, i]) if ks > 0 else -1
        out["start_margin"].append(_edge_margin(scene.leaves[prev], O64[ks, i]) if prev >= 0 else np.inf)
        out["len64"].append(float(L64[ks, i]))
        out["len32"].append(float(L32[ks, i]))
    rep = {k: np.array(v) for k, v in out.items()}
    rep["same"] = same
    return rep

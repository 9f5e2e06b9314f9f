"""Host-side ray/AABB slab test with the reference's tolerances (optable/solver.py:5-48).

Kept for user scripts that call it directly; the per-ray version lives in the HIP kernel
(csrc/trace_core.h `slab_hit`).  The two scene-construction helpers of solver.py are provided
because component factories use one of them (component_group.py:351).
"""
from typing import List, Tuple, Union

import numpy as np


def solve_ray_bboxes_intersections(ray_origin, ray_direction, bboxes: Union[List[Tuple], Tuple]):
    """Per box: entry/exit parameters (t1, t2) clipped to t >= 0 and a hit flag.

    An axis with |d| <= 1e-8 (np.isclose) is parallel: a miss iff the origin lies outside the
    slab, otherwise unconstrained.  hit = (t2 + 1e-12 >= t1) and (t2 >= 0).
    """
    if isinstance(bboxes, tuple):
        bboxes = [bboxes]
    box = np.array(bboxes, dtype=float).reshape(-1, 6)
    t1 = np.zeros(len(box))
    t2 = np.full(len(box), np.inf)
    for axis in range(3):
        o, d = ray_origin[axis], ray_direction[axis]
        lo, hi = box[:, 2 * axis], box[:, 2 * axis + 1]
        if np.isclose(d, 0.0):
            miss = (o < lo) | (o > hi)
            t1[miss], t2[miss] = 1.0, 0.0
            continue
        ta, tb = (lo - o) * (1.0 / d), (hi - o) * (1.0 / d)
        t1 = np.maximum(t1, np.minimum(ta, tb))
        t2 = np.minimum(t2, np.maximum(ta, tb))
    return t1, t2, (t2 + 1e-12 >= t1) & (t2 >= 0.0)


def solve_ray_ray_intersection(ray1_origin, ray1_direction, ray2_origin, ray2_direction):
    """Closest approach of two rays clamped to t >= 0, its midpoint, and the mirror normal that
    would send ray 1 into ray 2 (solver.py:51-107)."""
    p1, p2 = np.array(ray1_origin, dtype=float), np.array(ray2_origin, dtype=float)
    d1 = np.array(ray1_direction, dtype=float)
    d2 = np.array(ray2_direction, dtype=float)
    d1, d2 = d1 / np.linalg.norm(d1), d2 / np.linalg.norm(d2)
    w = p1 - p2
    b, d, e = d1 @ d2, d1 @ w, d2 @ w
    det = 1.0 - b * b
    if det < 1e-6:
        t1, t2 = 0.0, e
    else:
        t1, t2 = (b * e - d) / det, (e - b * d) / det
    t1, t2 = max(0.0, t1), max(0.0, t2)
    P = 0.5 * ((p1 + t1 * d1) + (p2 + t2 * d2))
    n = -(d1 + d2) * 0.5
    return t1, t2, P, n / np.linalg.norm(n)


def solve_normal_to_normal_rotation(n1, n2):
    """Axis and angle rotating n1 onto n2 (solver.py:110-132)."""
    a, b = n1 / np.linalg.norm(n1), n2 / np.linalg.norm(n2)
    axis = np.cross(a, b)
    size = np.linalg.norm(axis)
    if size < 1e-12:
        return np.array([1, 0, 0]), 0.0
    return axis / size, np.arccos(np.clip(a @ b, -1.0, 1.0))

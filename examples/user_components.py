#!/usr/bin/env python3
"""User-defined parts, written against the public API exactly as they would be for the reference package:
a transmission grating (a component whose `interact_local` is Python) and a parabolic mirror (a `Surface` subclass).
`table.ray_tracing` runs the nearest-hit search, the count gates and the built-in parts on the MI355X and calls the
grating's method for the rays that hit it; the paraboloid is measured when the scene is compiled and, being a surface
of revolution, gets the asphere kernels.
    python examples/user_components.py          (needs an MI355X and the built library)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optable_amd import BaseMirror, Circle, Lens, OpticalComponent, OpticalTable, Ray, Surface  # noqa: E402


class Grating(OpticalComponent):
    """Three diffraction orders in transmission (local frame: the grating plane is x = 0, lines along z)."""

    def __init__(self, origin, radius, pitch, **kwargs):
        super().__init__(origin, **kwargs)
        self.surface, self.pitch = Circle(radius), pitch

    def interact_local(self, ray):
        P, t = self.intersect_point_local(ray)
        out = []
        for order, share in ((-1, 0.25), (0, 0.5), (1, 0.25)):
            d = np.array(ray.direction, dtype=float)
            d[1] += order * ray.wavelength / self.pitch
            d[0] = np.sign(d[0]) * np.sqrt(max(1.0 - d[1] ** 2 - d[2] ** 2, 0.0))
            out.append(ray.copy(origin=P, direction=d, intensity=ray.intensity * share,
                                qo=None if ray.qo is None else ray.q_at_z(t), _pathlength=ray.pathlength(float(t))))
        return out


class Paraboloid(Surface):
    """x = -r^2 / (4 focal)."""

    def __init__(self, focal, radius):
        super().__init__()
        self.planar, self.focal, self.radius = False, focal, radius

    def f(self, P):
        return P[0] + (P[1] ** 2 + P[2] ** 2) / (4 * self.focal)

    def normal(self, P):
        n = np.array([1.0, P[1] / (2 * self.focal), P[2] / (2 * self.focal)])
        return n / np.linalg.norm(n)

    def within_boundary(self, P):
        return P[1] ** 2 + P[2] ** 2 <= self.radius**2

    def get_bbox_local(self):
        R, sag = self.radius, self.radius**2 / (4 * self.focal)
        return (-sag, 0.0, -R, R, -R, R)


class ParabolicMirror(BaseMirror):
    def __init__(self, origin, focal, radius, **kwargs):
        super().__init__(origin, **kwargs)
        self.surface = Paraboloid(focal, radius)


table = OpticalTable()
table.add_components([
    Grating([3, 0, 0], radius=1.0, pitch=4e-4),
    Lens([6, 0, 0], focal_length=6.0, radius=1.5),
    ParabolicMirror([12, 0, 0], focal=3.0, radius=2.0).RotZ(np.pi),   # concave side towards the light
])
rays = [Ray([0, y, 0], [1, 0, 0], wavelength=633e-7, w0=50e-4) for y in (-0.3, 0.0, 0.3)]
segments = table.ray_tracing(rays, perfomance_limit={"max_trace_num": 30})
print(f"{len(rays)} rays -> {len(segments)} segments, hooks: {sorted(type(c).__name__ for c in table.compile().hooks.values())}")
back = [s for s in segments if s.alive and s.direction[0] < 0]
print(f"{len(back)} rays leave towards -x; total intensity {sum(s.intensity for s in segments if s.alive):.6f} of {len(rays)}")

#!/usr/bin/env python3
"""The scene of the reference's examples/gaussian_beam.py (a mirror, three thin lenses, a glass slab, a mirror;
six Gaussian rays) traced through optable_amd — same classes, same call, no rendering.
    python examples/gaussian_beam.py            (needs an MI355X and the built library)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optable_amd import GlassSlab, Lens, Mirror, Monitor, OpticalTable, Ray  # noqa: E402

wl, w0 = 780e-9, 10e-6
rays = [Ray([-10, y, 0], [1, 0, 0], wavelength=wl, w0=w0) for y in (0, 2, 4, 6, 9)]
rays.append(Ray([-10, 21, 0], [1, 0, 0], wavelength=wl, w0=w0).RotZ(-np.pi / 4))

table = OpticalTable()
table.add_components([
    Mirror([0, 0, 0]).RotZ(np.pi / 6),
    Lens([0, 2, 0], radius=0.8, focal_length=5),
    Lens([0, 4, 0], radius=0.8, focal_length=10),
    Lens([0, 6.5, 0], radius=0.8, focal_length=10),
    GlassSlab([0, 9, 0], n1=1, n2=2, thickness=5),
    Mirror([0, 11, 0]).RotZ(-np.pi / 2),
])
screen = Monitor([8, 6.5, 0], 4, 4)
table.add_monitors(screen)

segments = table.ray_tracing(rays)          # List[Ray], one per traced segment, in the reference's order
print(f"{len(rays)} rays -> {len(segments)} segments")
for s in segments:
    end = "escapes" if s.length is None else f"length {s.length:8.4f}"
    print(f"  from ({s.origin[0]:8.4f}, {s.origin[1]:8.4f})  dir ({s.direction[0]:+.4f}, {s.direction[1]:+.4f})  {end}   q = {s.qo:.4g}")
print(f"monitor at x = 8: {screen.ndata} hit(s), y = {np.round(screen.yList, 6).tolist()}")

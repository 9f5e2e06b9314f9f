"""optable_amd — MI355X-native engine behind optable's Ray / OpticalComponent / OpticalTable API.

`from optable_amd import *` exposes the names `from optable import *` does
(reference optable/__init__.py:1-9): scene-building classes are plain Python data holders,
`OpticalTable.ray_tracing` runs on the HIP engine (csrc/, include/optable_hip.h).
"""
from .geometry import *  # noqa: F401,F403
from .geometry import Base, Vector, Color, base_merge_bboxs
from .materials import *  # noqa: F401,F403
from .slab import solve_ray_bboxes_intersections, solve_ray_ray_intersection, solve_normal_to_normal_rotation
from .shapes import (Surface, Point, Plane, Circle, Rectangle, Cylinder, Sphere, ASphere, Polygon,
                     sag_parametric, sag_exact)
from .rays import GaussianBeam, Ray, multiplex_rays_in_wavelength
from .components import (OpticalComponent, PointObj, Block, BaseMirror, BaseRefraciveSurface, Mirror,
                         SquareMirror, SquareRefractive, CircleRefractive, SphereRefractive, BeamSplitter,
                         Lens, CylMirror)
from .assemblies import (ComponentGroup, GlassSlab, CircleGlassSlab, MLA, MMA, MMADisordered, DMD, WedgePlate,
                         MirrorPair, Prism, TriangularPrism, MirrorPrism, MirrorCube, DovePrism,
                         PlanoConvexLens, BiConvexLens, Doublet, ASphericLens, ASphericExactSphericalLens,
                         ASphericParametricLens)
from .monitors import Monitor
from .table import OpticalTable
from .scene import compile_scene, CompiledScene, SceneError
from .adapter import install
import numpy as np  # noqa: F401  (the reference's star import leaks np; scripts rely on it)

__version__ = "0.1.0"

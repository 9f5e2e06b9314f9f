"""Implicit surfaces in a component's local frame, normal along +x (reference: optable/surfaces.py).

Each shape is a small data holder: the device evaluates `f`, `normal`, `within_boundary`
and the local box itself from the numbers `lower()` hands to the scene compiler
(shape kind + parameter vector + optional aux record, see include/optable_hip.h
`ot_shape_kind`).  The Python methods are kept so user scripts can still query a shape.
Plot outlines (`parametric_boundary`) are out of scope.
"""
from typing import Callable, Optional, Sequence

import numpy as np

from .geometry import Base, unit_vector
from .slab import solve_ray_bboxes_intersections

# shape kinds == ot_shape_kind
CIRCLE, RECT, POLYGON2D, POLYGON3D, SPHERE, ASPHERE_PARAM, ASPHERE_EXACT, CYLINDER, POINT, CSG, ASPHERE_CHEB = range(11)
# CSG postfix opcodes (aux program): operand kinds reuse CIRCLE/RECT/POLYGON2D
CSG_OR, CSG_ANDNOT = 100, 101


class Lowered:
    """What the scene compiler needs from a shape."""

    __slots__ = ("kind", "params", "aux", "planar")

    def __init__(self, kind, params=(), aux=None, planar=True):
        self.kind = kind
        self.params = [float(x) for x in params]
        self.aux = None if aux is None else [float(x) for x in aux]
        self.planar = planar


def _merge(b1, b2):
    return (min(b1[0], b2[0]), max(b1[1], b2[1]), min(b1[2], b2[2]),
            max(b1[3], b2[3]), min(b1[4], b2[4]), max(b1[5], b2[5]))


class Surface(Base):
    def __init__(self):
        super().__init__()
        self.planar = True

    def _normalize_vector(self, vector):
        return unit_vector(vector)

    def f(self, P):
        raise NotImplementedError("Method 'f' must be implemented in the derived class.")

    def normal(self, P):
        raise NotImplementedError("Method 'normal' must be implemented in the derived class.")

    def within_boundary(self, P):
        raise NotImplementedError("Method 'within_boundary' must be implemented in the derived class.")

    def lower(self) -> Lowered:
        raise NotImplementedError(f"{type(self).__name__} has no device form")

    def merge_bbox(self, bbox1, bbox2):
        return _merge(bbox1, bbox2)

    def merge_bboxs(self, bboxs):
        b = np.asarray([tuple(x) for x in bboxs], dtype=float)
        return (b[:, 0].min(), b[:, 1].max(), b[:, 2].min(), b[:, 3].max(), b[:, 4].min(), b[:, 5].max())

    def solve_crosssection_ray_bbox_local(self, ray_origin, ray_direction):
        return solve_ray_bboxes_intersections(ray_origin, ray_direction, self.get_bbox_local())


class Point(Surface):
    """Marker that no ray ever hits: f = |P| has no sign change (surfaces.py:68-86)."""

    def __init__(self):
        super().__init__()
        self.planar = False

    def f(self, P):
        return np.linalg.norm(P)

    def normal(self, P):
        return P / np.linalg.norm(P)

    def within_boundary(self, P):
        return np.linalg.norm(P) < 1e-12

    def get_bbox_local(self):
        return (0, 0, 0, 0, 0, 0)

    def lower(self):
        return Lowered(POINT, planar=False)


class Plane(Surface):
    """The x = 0 plane; subclasses bound it.  `union`/`subtract` build boolean apertures
    (surfaces.py:100-136)."""

    def __init__(self):
        super().__init__()
        self._normal = np.array([1, 0, 0])

    def f(self, P):
        return np.dot(self._normal, P)

    def normal(self, P):
        return self._normal

    def union(self, other):
        return BooleanPlane(CSG_OR, self, other)

    def subtract(self, other):
        return BooleanPlane(CSG_ANDNOT, self, other)


class BooleanPlane(Plane):
    """`a or b` / `a and not b` of two planar apertures, kept as a tree so it can be lowered
    to a postfix program for the device."""

    def __init__(self, op, left, right):
        super().__init__()
        self.op, self.left, self.right = op, left, right

    def within_boundary(self, P):
        if self.op == CSG_OR:
            return self.left.within_boundary(P) or self.right.within_boundary(P)
        return self.left.within_boundary(P) and not self.right.within_boundary(P)

    def get_bbox_local(self):
        return _merge(self.left.get_bbox_local(), self.right.get_bbox_local())

    def _postfix(self, out):
        for side in (self.left, self.right):
            if isinstance(side, BooleanPlane):
                side._postfix(out)
            else:
                low = side.lower()
                if low.kind not in (CIRCLE, RECT, POLYGON2D):
                    raise NotImplementedError(
                        f"boolean aperture operand {type(side).__name__} has no device form")
                body = low.aux if low.kind == POLYGON2D else low.params
                out.extend([float(low.kind), float(len(body))] + list(body))
        out.extend([float(self.op), 0.0])

    def lower(self):
        prog = []
        self._postfix(prog)
        n_tokens = sum(1 for _ in _walk_tokens(prog))
        return Lowered(CSG, aux=[float(n_tokens)] + prog)


def _walk_tokens(prog):
    i = 0
    while i < len(prog):
        yield i
        i += 2 + int(prog[i + 1])


class Circle(Plane):
    def __init__(self, radius):
        super().__init__()
        self.radius = radius

    def within_boundary(self, P):
        return np.linalg.norm(P) <= self.radius  # 3-norm, x ~ 0 (surfaces.py:144-145)

    def get_bbox_local(self):
        r = self.radius
        return (0, 0, -r, r, -r, r)

    def lower(self):
        return Lowered(CIRCLE, [self.radius])


class Rectangle(Plane):
    def __init__(self, width, height):
        super().__init__()
        self.width = width    # along local y
        self.height = height  # along local z

    def within_boundary(self, P):
        return np.abs(P[1]) <= self.width / 2 and np.abs(P[2]) <= self.height / 2

    def get_bbox_local(self):
        return (0, 0, -self.width / 2, self.width / 2, -self.height / 2, self.height / 2)

    def lower(self):
        return Lowered(RECT, [self.width / 2, self.height / 2])


class Cylinder(Surface):
    """Wall of a cylinder about local z (surfaces.py:212-281)."""

    def __init__(self, radius, height, theta_range=(-np.pi, np.pi)):
        super().__init__()
        self.radius, self.height, self.theta_range = radius, height, theta_range
        self.planar = False

    def f(self, P):
        return np.linalg.norm(P[:2]) - self.radius

    def normal(self, P):
        return np.array([P[0], P[1], 0]) / self.radius

    def within_boundary(self, P):
        theta = np.arctan2(P[1], P[0])
        return (self.theta_range[0] <= theta <= self.theta_range[1]) and (-self.height / 2 <= P[2] <= self.height / 2)

    def get_bbox_local(self):
        r, h = self.radius, self.height / 2
        return (-r, r, -r, r, -h, h)

    def lower(self):
        return Lowered(CYLINDER, [self.radius, self.height / 2, self.theta_range[0], self.theta_range[1]], planar=False)


class Sphere(Surface):
    """Spherical cap centred on the local origin, x in [R - height, R] (surfaces.py:284-336)."""

    def __init__(self, radius, height=None):
        super().__init__()
        self.radius = radius
        self.height = height if height is not None else 2 * radius
        self.planar = False
        self.diameter = np.sqrt(radius**2 - (radius - height) ** 2) * 2  # height=None raises, as upstream

    def f(self, P):
        return np.linalg.norm(P) - self.radius

    def normal(self, P):
        return P / self.radius

    def within_boundary(self, P):
        return self.radius - self.height - 1e-12 <= P[0] <= self.radius + 1e-12

    def get_bbox_local(self):
        h = self.diameter / 2
        return (self.radius - self.height, self.radius, -h, h, -h, h)

    def lower(self):
        return Lowered(SPHERE, [self.radius, self.height], planar=False)


class ASphere(Surface):
    """Surface of revolution x = -F(r) (surfaces.py:339-423).  Slope and curvature use the
    reference's central differences with h = 1e-4 * radius; the device repeats them."""

    def __init__(self, radius, f_asphere: Callable):
        super().__init__()
        self.radius = radius
        self.f_asphere = f_asphere
        self.planar = False
        x_axis, x_rim = -f_asphere(0), -f_asphere(radius)
        self.xmin, self.xmax = min(x_axis, x_rim), max(x_axis, x_rim)

    def _df_asphere_dr(self, r):
        h = 1e-4 * self.radius
        return (self.f_asphere(r + h) - self.f_asphere(r - h)) / (2 * h)

    def _df_asphere_dr2(self, r):
        h = 1e-4 * self.radius
        return (self.f_asphere(r + h) - 2 * self.f_asphere(r) + self.f_asphere(r - h)) / (h**2)

    def roc_r(self, r):
        slope = self._df_asphere_dr(r)
        return (1 + slope**2) ** 1.5 / self._df_asphere_dr2(r)

    def roc(self, P):
        return self.roc_r(np.linalg.norm(P[1:3]))

    def f(self, P):
        return P[0] + self.f_asphere(np.linalg.norm(P[1:3]))

    def normal(self, P):
        r = np.linalg.norm(P[1:3])
        if r < 1e-12:
            return np.array([1.0, 0.0, 0.0])
        slope = self._df_asphere_dr(r)
        return unit_vector([1, slope * P[1] / r, slope * P[2] / r])

    def within_boundary(self, P):
        return np.linalg.norm(P[1:3]) <= self.radius + 1e-12

    def get_bbox_local(self):
        r = self.radius
        return (self.xmin, self.xmax, -r, r, -r, r)

    def lower(self):
        spec = getattr(self.f_asphere, "device_spec", None)
        if spec is None:
            return lower_asphere_callable(self.radius, self.f_asphere)
        kind, coeffs = spec
        return Lowered(kind, [self.radius] + list(coeffs), planar=False)


def lower_asphere_callable(radius, f_asphere):
    """Device form of an ASphere whose sag is an arbitrary Python function F(r) (component_group.py:1014-1055): a
    verified Chebyshev series of F on every radius the kernels can ask for — the root scan looks at points of the
    local box (r up to radius * sqrt 2), the reference's finite differences step h = 1e-4 * radius to either side,
    also across the axis (surfaces.py:355-369) — with the series of F' and F'' beside it (Newton slope of the root
    polish; normals and curvature in single precision, where a finite difference with that h loses five digits).
    A function that no series reproduces to 5e-14 of its range is refused (cheb.FitError names the reason)."""
    from . import cheb

    h = 1e-4 * radius
    lo, hi = -2.0 * h, radius * (np.sqrt(2.0) * 1.001) + 2.0 * h
    try:
        f_asphere(lo)
        func = f_asphere
    except Exception:  # noqa: BLE001 - a sag that only takes r >= 0: the reference's stencil reaches across the axis as F(|r|) would
        func = lambda r: f_asphere(abs(r))  # noqa: E731
    try:
        coef, _ = cheb.fit(func, lo, hi, what="ASphere f_asphere(r)")
    except cheb.FitError as exc:
        raise NotImplementedError(str(exc)) from exc
    return Lowered(ASPHERE_CHEB, [radius], aux=cheb.record(coef, lo, hi, derivatives=2), planar=False)


def sag_parametric(R, kappa, a4=0.0, a6=0.0, a8=0.0):
    """Conic + even polynomial sag F(r) (component_group.py:1092-1097) with a device form."""

    def F(r):
        return (r**2 / (R * (1 + np.sqrt(1 - (1 + kappa) * (r**2) / (R**2))))
                + a4 * r**4 + a6 * r**6 + a8 * r**8)

    F.device_spec = (ASPHERE_PARAM, (R, kappa, a4, a6, a8))
    return F


def sag_exact(EFL, n):
    """Aberration-free plano-convex sag (component_group.py:1061-1064) with a device form."""

    def F(r):
        return (EFL / (n + 1)) * (-1 + np.sqrt(1 + (n + 1) / (n - 1) * (r**2) / (EFL**2)))

    F.device_spec = (ASPHERE_EXACT, (EFL, n))
    return F


class Polygon(Plane):
    """Planar polygon (surfaces.py:426-568).  2-D vertices, or 3-D vertices all at x = 0,
    lie in the component plane (`planar=True`); any other plane is solved as a general
    implicit surface."""

    def __init__(self, vertices: Sequence[Sequence[float]], normal: Optional[Sequence[float]] = None):
        super().__init__()
        self._tol = 1e-9
        verts = np.asarray(vertices, dtype=float)
        if verts.ndim != 2 or verts.shape[0] < 3:
            raise ValueError("Need at least three vertices (shape (N,2) or (N,3)).")
        self.planar = False
        if verts.shape[1] == 2:
            verts = np.column_stack((np.zeros(len(verts)), verts))
            if normal is None:
                normal, self.planar = (1.0, 0.0, 0.0), True
        if verts.shape[1] == 3 and normal is None and np.allclose(verts[:, 0], 0.0):
            normal, self.planar = (1.0, 0.0, 0.0), True
        self.vertices = verts
        if normal is None:
            for i in range(2, len(verts)):
                cand = np.cross(verts[i] - verts[0], verts[1] - verts[0])
                if np.linalg.norm(cand) > self._tol:
                    normal = cand
                    break
            else:
                raise ValueError("Vertices are colinear - cannot define a plane.")
        self._normal = unit_vector(normal)
        if np.any(np.abs((verts - verts[0]) @ self._normal) > self._tol):
            raise ValueError("Vertices are not coplanar with the supplied normal.")
        helper = np.array([1.0, 0.0, 0.0])
        if abs(helper @ self._normal) > 0.99:
            helper = np.array([0.0, 1.0, 0.0])
        u = unit_vector(np.cross(self._normal, helper))
        self._basis = (u, np.cross(self._normal, u))
        self._verts2d = self._project_to_2d(verts)
        self._bbox = tuple(float(f(verts[:, a])) for a in range(3) for f in (np.min, np.max))

    def _project_to_2d(self, pts):
        u, v = self._basis
        rel = pts - self.vertices[0]
        return np.column_stack((rel @ u, rel @ v))

    def f(self, P):
        return np.dot(self._normal, P - self.vertices[0])

    def within_boundary(self, P):
        px, py = self._project_to_2d(P[None, :])[0]
        v2, tol, inside = self._verts2d, self._tol, False
        for i in range(len(v2)):
            x1, y1 = v2[i]
            x2, y2 = v2[(i + 1) % len(v2)]
            twice_area = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1)
            if (abs(twice_area) <= tol and min(x1, x2) - tol <= px <= max(x1, x2) + tol
                    and min(y1, y2) - tol <= py <= max(y1, y2) + tol):
                return True  # on an edge counts as inside
            if (y1 > py) != (y2 > py) and x1 + (py - y1) * (x2 - x1) / (y2 - y1) >= px:
                inside = not inside
        return inside

    def get_bbox_local(self):
        return self._bbox

    def lower(self):
        """aux record: [nverts, nx,ny,nz, v0(3), u(3), v(3), (x,y)*nverts]."""
        u, v = self._basis
        rec = [float(len(self._verts2d))] + list(self._normal) + list(self.vertices[0]) + list(u) + list(v)
        rec += [c for xy in self._verts2d for c in xy]
        return Lowered(POLYGON2D if self.planar else POLYGON3D, aux=rec, planar=self.planar)

"""Leaf optical components (reference: optable/optical_component.py).

On the MI355X path a component is a *scene-description record*: a pose (`origin`,
`transform_matrix`), a shape (shapes.py) and an interaction with its coefficients.  The
per-ray work of the reference — `interact`, `intersect_point_local`, `*.interact_local`
(optical_component.py:151-233, 337-378, 536-570, 617-717, 930-948) — is not done here: the
scene compiler (scene.py) lowers every leaf through `lower_interaction()` into an `ot_node`
and the HIP kernel (csrc/trace_core.h) does the arithmetic for all rays at once.
Rendering and CSV metadata (`render`, `gather_components`) are out of scope.
"""
from typing import Union

import numpy as np

from .geometry import Vector, Color, _NO_BOX, pivot_origin, out_of_scope
from .materials import Material, RefractiveIndex
from .shapes import Surface, Plane, Point, Circle, Rectangle, Sphere, Cylinder

# interaction kinds == ot_interaction_kind
MIRROR, REFRACT, LENS, BLOCK = range(4)
# roc kinds == ot_roc_kind
ROC_INF, ROC_CONST, ROC_ASPHERE = range(3)

_BOX_CORNER_PICK = np.array([[i & 1, 2 + ((i >> 1) & 1), 4 + ((i >> 2) & 1)] for i in range(8)])


class OpticalComponent(Vector):
    """Pose + surface + bookkeeping common to every element (optical_component.py:8-149)."""
    render = out_of_scope("render")
    gather_components = out_of_scope("gather_components")


    def __init__(self, origin, **kwargs):
        super().__init__(origin, **kwargs)
        self.transform_matrix = np.identity(3)
        self.surface = Plane()
        self._bbox = _NO_BOX
        self.render_obj = kwargs.get("render_obj", True)
        self.render_comp_vec = kwargs.get("render_comp_vec", False)
        self.name = kwargs.get("name", None)
        self.label = kwargs.get("label", None)
        self.label_position = kwargs.get("label_position", [1, 0, 0])
        self._interact_count = {}
        self.max_interact_count = kwargs.get("max_interact_count", None)

    def __repr__(self):
        return f"OpticalComponent(origin={self.origin}, transform_matrix=\n{self.transform_matrix})"

    # -- frame -----------------------------------------------------------------------
    @property
    def normal(self):
        return self.transform_matrix @ np.array([1, 0, 0])

    @property
    def tangent_Y(self):
        return self.transform_matrix @ np.array([0, 1, 0])

    @property
    def tangent_Z(self):
        return self.transform_matrix @ np.array([0, 0, 1])

    def _RotAroundLocal(self, axis, localpoint, theta):
        rot = self.R(axis, theta)
        self.transform_matrix = rot @ self.transform_matrix
        self.origin = pivot_origin(self.origin, rot, localpoint)
        return self

    def point_to_lab_coordinates(self, point_local):
        return self.transform_matrix @ point_local + self.origin

    def ray_to_local_coordinates(self, ray):
        """The ray seen from this component's frame (optical_component.py:106-111); M is a rotation."""
        R = self.transform_matrix.T
        return ray.copy(origin=R @ (ray.origin - self.origin), direction=R @ ray.direction)

    def ray_to_lab_coordinates(self, ray):
        """A local-frame ray back in the lab frame (optical_component.py:119-124)."""
        R = self.transform_matrix
        return ray.copy(origin=R @ ray.origin + self.origin, direction=R @ ray.direction)

    # -- the per-object form of the hot path: one ray against this component, on the device ------------
    def interact(self, ray):
        """`(t, [truncated, *children])` or `(None, None)` (optical_component.py:337-378)."""
        from .table import interact_component

        return interact_component(self, ray)

    def interact_local(self, ray_local):
        """The rays a hit emits, in this leaf's frame (the per-class physics upstream: mirror :536-570,
        interface :617-717, thin lens :930-948; a base OpticalComponent has none and raises, as :240 does)."""
        from .table import interact_leaf_local

        return interact_leaf_local(self, ray_local)

    def intersect_point_local(self, ray_local):
        """`(P_local, t)` of the first valid hit of a LOCAL-frame ray, or `(None, None)`
        (optical_component.py:151-233)."""
        from .table import intersect_leaf_local

        return intersect_leaf_local(self, ray_local)

    # -- bounding boxes (cached on first use and never invalidated, as upstream :62-67) -----
    def get_bbox_local(self):
        return self.surface.get_bbox_local()

    @property
    def bbox(self):
        if self._bbox == _NO_BOX:
            self._bbox = tuple(self.get_bbox())
        return tuple(self._bbox)

    def get_bbox(self) -> tuple:
        """Lab AABB of the 8 rotated corners of the local box (optical_component.py:69-97)."""
        local = np.asarray(self.get_bbox_local(), dtype=float)
        corners = local[_BOX_CORNER_PICK].T  # (3, 8)
        lab = self.transform_matrix @ corners + self.origin.reshape(3, 1)
        return (lab[0].min(), lab[0].max(), lab[1].min(), lab[1].max(), lab[2].min(), lab[2].max())

    # -- interact-count gate (state lives on the host between traces; the kernel updates a
    #    device table that table.py loads from / stores back into this dict) -----------------
    def get_interact_count(self, ray_id):
        return self._interact_count.get(ray_id, 0)

    def should_interact(self, ray_id):
        return self.max_interact_count is None or self._interact_count.get(ray_id, 0) < self.max_interact_count

    def increase_interact_count(self, ray_id):
        self._interact_count[ray_id] = self._interact_count.get(ray_id, 0) + 1

    # -- device lowering -----------------------------------------------------------------
    def lower_interaction(self) -> dict:
        raise NotImplementedError(f"{type(self).__name__} has no device interaction")

    def patch_block(self, width, height):
        """A Block sharing this pose, with this aperture cut out (optical_component.py:380-384)."""
        blk = Block(self.origin, hole=self.surface, width=width, height=height)
        blk.transform_matrix = self.transform_matrix
        return blk


class PointObj(OpticalComponent):
    """Reference point; its Point surface is never hit (optical_component.py:429-469)."""

    def __init__(self, origin, **kwargs):
        super().__init__(origin, **kwargs)
        self.surface = Point()
        self._edge_color = "orange"

    def lower_interaction(self):
        return dict(kind=BLOCK)  # unreachable: OT_SHAPE_POINT never yields a hit


class Block(OpticalComponent):
    """Absorbing rectangle, optionally with a hole (optical_component.py:472-511)."""

    def __init__(self, origin, hole: Union[Surface, None] = None, width: float = 1.0, height: float = 1.0, **kwargs):
        super().__init__(origin, **kwargs)
        self.width, self.height = width, height
        plate = Rectangle(width, height)
        self.surface = plate.subtract(hole) if hole is not None else plate
        self._edge_color = "black"

    def lower_interaction(self):
        return dict(kind=BLOCK)


class BaseMirror(OpticalComponent):
    """Reflects `reflectivity` and passes `transmission` of the intensity (:514-578)."""

    def __init__(self, origin, reflectivity: float = 1.0, transmission: float = 0.0, **kwargs):
        super().__init__(origin, **kwargs)
        self.reflectivity = reflectivity
        self.transmission = transmission
        self._edge_color = "green"

    def lower_interaction(self):
        return dict(kind=MIRROR, reflectivity=self.reflectivity, transmission=self.transmission)


class BaseRefraciveSurface(OpticalComponent):
    """Interface between media `n1` (local x > 0) and `n2` (x < 0) (:581-725)."""

    _n1 = RefractiveIndex("_n1")
    _n2 = RefractiveIndex("_n2")

    def __init__(self, origin, n1: Union[float, Material] = 1.0, n2: Union[float, Material] = 1.0,
                 reflectivity: float = 0.0, transmission: float = 1.0, **kwargs):
        super().__init__(origin, **kwargs)
        self._n1, self._n2 = n1, n2
        self.reflectivity = reflectivity
        self.transmission = transmission
        self._edge_color = "gray"
        self.surface = kwargs.get("surface", Plane())
        self.roc = self.surface.roc if hasattr(self.surface, "roc") else np.inf

    def lower_interaction(self):
        roc = getattr(self, "roc", np.inf)
        if callable(roc):
            owner = getattr(roc, "__self__", None)
            if owner is not self.surface:
                raise NotImplementedError("callable roc that is not the surface's own ASphere.roc has no device form")
            roc_kind, roc_val = ROC_ASPHERE, 0.0
        elif np.isinf(roc):
            roc_kind, roc_val = ROC_INF, np.inf
        else:
            roc_kind, roc_val = ROC_CONST, float(roc)
        return dict(kind=REFRACT, reflectivity=self.reflectivity, transmission=self.transmission,
                    mat1=self.__dict__["_n1"], mat2=self.__dict__["_n2"], roc_kind=roc_kind, roc=roc_val)


class Mirror(BaseMirror):
    def __init__(self, origin, radius: float = 0.5, reflectivity: float = 1.0, transmission: float = 0.0, **kwargs):
        super().__init__(origin, reflectivity=reflectivity, transmission=transmission, **kwargs)
        self.radius = radius
        self.surface = Circle(radius)


class SquareMirror(BaseMirror):
    def __init__(self, origin, width: float = 1.0, height: float = 1.0, reflectivity: float = 1.0,
                 transmission: float = 0.0, **kwargs):
        super().__init__(origin, reflectivity=reflectivity, transmission=transmission, **kwargs)
        self.width, self.height = width, height
        self.surface = Rectangle(width, height)


class SquareRefractive(BaseRefraciveSurface):
    def __init__(self, origin, width: float = 1.0, height: float = 1.0, n1=1.0, n2=1.0,
                 reflectivity: float = 0.0, transmission: float = 1.0, **kwargs):
        super().__init__(origin, n1=n1, n2=n2, reflectivity=reflectivity, transmission=transmission, **kwargs)
        self.width, self.height = width, height
        self.surface = Rectangle(width, height)


class CircleRefractive(BaseRefraciveSurface):
    def __init__(self, origin, radius: float = 0.5, n1=1.0, n2=1.0, reflectivity: float = 0.0,
                 transmission: float = 1.0, **kwargs):
        super().__init__(origin, n1=n1, n2=n2, reflectivity=reflectivity, transmission=transmission, **kwargs)
        self.radius = radius
        self.surface = Circle(radius)


class SphereRefractive(BaseRefraciveSurface):
    """Spherical cap; the cap's centre of curvature is the component origin (:822-848)."""

    def __init__(self, origin, radius: float = 0.5, height: float = 0.5, n1=1.0, n2=1.0,
                 reflectivity: float = 0.0, transmission: float = 1.0, **kwargs):
        super().__init__(origin, n1=n1, n2=n2, reflectivity=reflectivity, transmission=transmission, **kwargs)
        self.radius, self.height = radius, height
        self.roc = radius
        self.surface = Sphere(radius, height)


class BeamSplitter(SquareMirror):
    """Amplitude-style splitter: R = sqrt(eta), T = sqrt(1 - eta) (:851-872)."""

    def __init__(self, origin, width=1.0, height=1.0, eta: float = 0.5, **kwargs):
        super().__init__(origin, width=width, height=height, reflectivity=np.sqrt(eta),
                         transmission=np.sqrt(1 - eta), **kwargs)
        self._edge_color = kwargs.get("edgecolor", Color.SCIENCE_BLUE_DARK)


class Lens(OpticalComponent):
    """Ideal thin lens with a circular aperture (:902-948)."""

    def __init__(self, origin, focal_length: float, radius: float = 0.5, transmission: float = 1.0, **kwargs):
        super().__init__(origin, **kwargs)
        self.focal_length = focal_length
        self.transmission = transmission
        self.radius = radius
        self.surface = Circle(radius)
        self._edge_color = "purple"

    def lower_interaction(self):
        return dict(kind=LENS, transmission=self.transmission, focal_length=self.focal_length)


class CylMirror(BaseMirror):
    def __init__(self, origin, radius: float = 0.5, height: float = 1.0, theta_range=(-np.pi, np.pi), **kwargs):
        super().__init__(origin, **kwargs)
        self.radius, self.height = radius, height
        self.surface = Cylinder(radius, height, theta_range)

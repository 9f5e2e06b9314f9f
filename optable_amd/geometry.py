"""Pose primitives shared by rays and components.

Host-side scene-construction helpers only; nothing here runs per ray.  Semantics follow
optable/base.py (reference): `_id` is inherited by copies (base.py:10-22), `Vector` owns an
`origin` and the fluent Rot*/T* transforms that return `self` (base.py:106-156).
"""
import copy as _copy
from dataclasses import dataclass
from typing import Callable, List, Sequence, Tuple, Union  # re-exported for `import *` users

import numpy as np

_NO_BOX = (None,) * 6


def out_of_scope(name):
    """A method of the reference that lives outside the accelerated path (rendering, component metadata):
    calling it says so instead of failing with a bare AttributeError."""
    def method(self, *args, **kwargs):
        raise NotImplementedError(
            f"{type(self).__name__}.{name} is outside optable_amd's scope (it rebuilds only the ray-tracing path); "
            "keep the original package for it and route its hot path here with optable_amd.install(optable)")
    method.__name__ = name
    return method


_PLAIN = (int, float, complex, bool, str, bytes, type(None), np.generic)


class Base:
    """Attribute bag with an identity that survives `copy()` (base.py:8-22)."""

    def __init__(self, **kwargs):
        if not hasattr(self, "_id"):
            self._id = kwargs.get("id", id(self))
        for name, value in kwargs.items():
            setattr(self, name, value)

    def copy(self, **kwargs):
        """Deep copy keeping `_id`; keyword arguments overwrite attributes of the copy (base.py:24-28).  An object whose
        attributes are all numbers, strings, arrays or (immutable) materials — every Ray — is copied attribute by attribute:
        the same object graph `copy.deepcopy` builds, ~8x cheaper (user `interact_local` methods copy a ray per child)."""
        fresh = {}
        for name, value in self.__dict__.items():
            if isinstance(value, np.ndarray):
                fresh[name] = value.copy()
            elif isinstance(value, _PLAIN) or getattr(type(value), "__module__", "") == "optable_amd.materials":
                fresh[name] = value
            else:
                fresh = None
                break
        if fresh is None:
            twin = _copy.deepcopy(self)
        else:
            twin = object.__new__(type(self))
            twin.__dict__.update(fresh)
        for name, value in kwargs.items():
            setattr(twin, name, value)
        return twin


def unit_vector(v) -> np.ndarray:
    a = np.array(v, dtype=float)
    return a / np.linalg.norm(a)


def rotation_matrix(axis, theta: float) -> np.ndarray:
    """Rodrigues rotation about `axis` by `theta` (same matrix as base.py:38-79)."""
    u = unit_vector(axis)
    c, s = np.cos(theta), np.sin(theta)
    skew = np.array([[0.0, -u[2], u[1]], [u[2], 0.0, -u[0]], [-u[1], u[0], 0.0]])
    return c * np.identity(3) + s * skew + (1.0 - c) * np.outer(u, u)


class Vector(Base):
    """Something with a lab-frame `origin` that can be rotated and translated in place."""

    def __init__(self, origin, **kwargs):
        super().__init__(**kwargs)
        self.origin = np.array(origin, dtype=float)
        self.unit = kwargs.get("unit", 1e-2)  # metres per model unit; default cm (base.py:31)

    # -- helpers ---------------------------------------------------------------------
    def _normalize_vector(self, vector) -> np.ndarray:
        return unit_vector(vector)

    def R(self, axis, theta: float) -> np.ndarray:
        return rotation_matrix(axis, theta)

    def _vector_to_R(self, t) -> np.ndarray:
        """Rotation taking +x onto `t` (base.py:81-104)."""
        v = unit_vector(t)
        if np.allclose(v, [1, 0, 0]):
            return np.identity(3)
        if np.allclose(v, [-1, 0, 0]):
            return np.diag([-1.0, -1.0, 1.0])
        k = unit_vector(np.cross([1.0, 0.0, 0.0], v))
        return rotation_matrix(k, np.arccos(v[0]))

    # -- rotations -------------------------------------------------------------------
    def _RotAroundLocal(self, axis, localpoint, theta):
        raise NotImplementedError("_RotAroundLocal method not implemented")

    def _RotAroundCenter(self, axis, theta):
        return self._RotAroundLocal(axis, [0, 0, 0], theta)

    def _RotAround(self, axis, point, theta):
        return self._RotAroundLocal(axis, np.array(point) - self.origin, theta)

    def RotX(self, theta):
        return self._RotAroundCenter([1, 0, 0], theta)

    def RotY(self, theta):
        return self._RotAroundCenter([0, 1, 0], theta)

    def RotZ(self, theta):
        return self._RotAroundCenter([0, 0, 1], theta)

    def RotXAroundLocal(self, localpoint, theta):
        return self._RotAroundLocal([1, 0, 0], localpoint, theta)

    def RotYAroundLocal(self, localpoint, theta):
        return self._RotAroundLocal([0, 1, 0], localpoint, theta)

    def RotZAroundLocal(self, localpoint, theta):
        return self._RotAroundLocal([0, 0, 1], localpoint, theta)

    # -- translations ----------------------------------------------------------------
    def _Translate(self, movement):
        self.origin += np.array(movement)
        return self

    def TX(self, dx):
        return self._Translate([dx, 0, 0])

    def TY(self, dy):
        return self._Translate([0, dy, 0])

    def TZ(self, dz):
        return self._Translate([0, 0, dz])


def pivot_origin(origin, R, localpoint):
    """New origin after rotating by R about `origin + localpoint` (optical_component.py:103)."""
    lp = np.array(localpoint, dtype=float)
    return origin + R @ (-lp) + lp


def base_merge_bboxs(bboxs):
    """Union of (xmin,xmax,ymin,ymax,zmin,zmax) boxes (base.py:262-270)."""
    b = np.asarray([tuple(x) for x in bboxs], dtype=float)
    return (b[:, 0].min(), b[:, 1].max(), b[:, 2].min(), b[:, 3].max(), b[:, 4].min(), b[:, 5].max())


@dataclass(frozen=True)
class Color:
    SCIENCE_RED_LIGHT: str = "#febfbe"
    SCIENCE_RED_DARK: str = "#fa331a"
    SCIENCE_BLUE_LIGHT: str = "#bdd3ec"
    SCIENCE_BLUE_DARK: str = "#2556ae"


def to_mathematical_str(text):
    """Python list / complex literals -> Mathematica syntax (base.py:248-251), used by the CSV export."""
    if text == "None":
        return "None"
    return text.replace("[", "{").replace("]", "}").replace("e", "*10^").replace("j", "I")


def get_attr_str(obj, attr_name, default=None):
    value = getattr(obj, attr_name, None)
    return value if value else default

"""Ray state object and Gaussian-beam q algebra (reference: optable/ray.py).

A `Ray` is the per-object view of one row of the device's SoA ray stream
(origin, direction, wavelength, q, intensity, n, pathlength, id, flags — include/optable_hip.h
`ot_rays`).  Lists of `Ray` are the drop-in API for small N; `RayBatch` (batch.py) is the
scalable container.  Plot-time helpers of the reference (`render`, beam sampling) are out of
scope.
"""
from typing import List

import numpy as np

from .geometry import Vector, pivot_origin, out_of_scope
from .materials import RefractiveIndex

_RAY_NONE_LENGTH = 100


class GaussianBeam:
    """Complex beam parameter helpers (ray.py:8-55)."""

    @staticmethod
    def q_at_waist(w0, wl: float, n: float = 1):
        return (1j * n * np.pi * w0**2) / wl

    @staticmethod
    def q_at_z(qo, z):
        return qo + z

    @staticmethod
    def distance_to_waist(q):
        return np.real(q)

    @staticmethod
    def waist(q, wl: float, n: float = 1):
        return np.sqrt((wl * np.imag(q)) / (n * np.pi))

    @staticmethod
    def rayleigh_range(q):
        return np.imag(q)

    @staticmethod
    def radius_of_curvature(q):
        return 1 / np.real(1 / q)

    @staticmethod
    def spot_size(qo, z, wl: float, n: float = 1):
        q = qo + z
        return np.sqrt(-wl / (n * np.pi * np.imag(1 / q)))


class Ray(Vector):
    """Geometric ray with optional Gaussian q (ray.py:58-200)."""
    render = out_of_scope("render")


    _n = RefractiveIndex("_n")

    def __init__(self, origin, direction, intensity: float = 1.0, wavelength=None, length=None,
                 alive=True, qo=None, w0=None, **kwargs):
        super().__init__(origin, **kwargs)
        self.length = float(length) if length else None
        self.direction = direction
        self.intensity = float(intensity)
        self.wavelength = float(wavelength) if wavelength else 0.0
        self.alive = alive
        self._n = 1.0
        self._pathlength = 0.0
        if qo is not None:
            self.qo = qo
        elif w0 is not None:
            self.qo = self.q_at_waist(w0)
        else:
            self.qo = None

    def __repr__(self):
        return (f"Ray(origin={self.origin}, direction={self.direction}, intensity={self.intensity}, "
                f"length={self.length}, alive={self.alive}, qo={self.qo})")

    @property
    def direction(self) -> np.ndarray:
        return self._direction

    @direction.setter
    def direction(self, value):
        self._direction = self._normalize_vector(value)

    @property
    def n(self) -> float:
        return self._n()

    @property
    def transform_matrix(self) -> np.ndarray:
        return self._vector_to_R(self.direction)

    @property
    def tangent_1(self) -> np.ndarray:
        d = self.direction
        if d[0] == 0 and d[1] == 0:
            return np.array([1, 0, 0])
        return self._normalize_vector(np.cross(d, np.array([0, 0, 1])))

    @property
    def tangent_2(self) -> np.ndarray:
        return self._normalize_vector(np.cross(self.direction, self.tangent_1))

    def pathlength(self, t: float = 0) -> float:
        return float(self._pathlength + t * self.n)

    def phase(self, t: float = 0) -> float:
        return np.mod((2 * np.pi / self.wavelength) * self.pathlength(t), 2 * np.pi)

    def _RotAroundLocal(self, axis, localpoint, theta) -> "Ray":
        rot = self.R(axis, theta)
        self.direction = rot @ self.direction
        self.origin = pivot_origin(self.origin, rot, localpoint)
        return self

    # -- Gaussian beam ---------------------------------------------------------------
    def q_at_waist(self, w0):
        return GaussianBeam.q_at_waist(w0, self.wavelength, self.n)

    def q_at_z(self, z):
        return GaussianBeam.q_at_z(self.qo, z)

    def distance_to_waist(self, q):
        return GaussianBeam.distance_to_waist(q)

    def waist(self, q):
        return GaussianBeam.waist(q, self.wavelength, self.n)

    def rayleigh_range(self, q):
        return GaussianBeam.rayleigh_range(q)

    def radius_of_curvature(self, q):
        return GaussianBeam.radius_of_curvature(q)

    def spot_size(self, z):
        return GaussianBeam.spot_size(self.qo, z, self.wavelength, self.n)

    def Propagate(self, z) -> "Ray":
        return self.copy(qo=self.q_at_z(z))


def multiplex_rays_in_wavelength(rays: List[Ray], wavelength_list: List[float]) -> List[Ray]:
    """Wavelength-major copies sharing each source ray's `_id` (ray.py:428-445)."""
    return [ray.copy(wavelength=wl) for wl in wavelength_list for ray in rays]

"""Refractive-index models (reference: optable/material.py).

The device evaluates only two closed forms — a constant and the 3-term Sellmeier equation —
so every Material carries a `device_spec()` the scene compiler can lower to an `ot_material`
row.  A Material built from an arbitrary Python callable has no device form and makes the
scene compiler raise instead of silently evaluating on the host.
"""
from typing import Callable, List, Union

import numpy as np


WAVELENGTH_RANGE = (0.2e-6, 2.5e-6)  # metres: where a dispersion callable is fitted unless the Material names its own range


def callable_spec(n_func, wavelength_range=None):
    """Device form of n(wavelength in metres) given as a Python function (material.py:4-21): constant if it is one,
    else a verified Chebyshev series over `wavelength_range` (cheb.py) — ('cheb', lo, hi, coefficients) — or None
    with the reason in `callable_spec.why` when no series reproduces it.  Rays whose wavelength (in metres) lies
    outside the range are refused at trace time (engine.py): the series says nothing there."""
    from . import cheb

    lo, hi = wavelength_range or WAVELENGTH_RANGE
    try:
        vals = [float(n_func(w)) for w in (lo, 0.5 * (lo + hi), hi, 0.37 * lo + 0.63 * hi)]
        if max(vals) == min(vals):
            return ("const", vals[0])
        coef, _ = cheb.fit(n_func, lo, hi, what="Material n(wavelength)")
    except (cheb.FitError, TypeError, ValueError, ZeroDivisionError) as exc:
        callable_spec.why = str(exc)
        return None
    return ("cheb", float(lo), float(hi), [float(c) for c in coef])


callable_spec.why = ""


class Material:
    """`n(wavelength in metres)` (material.py:4-21).  `wavelength_range` (metres) is where a callable n is valid and
    gets its device form; the reference has no such argument (it calls the function wherever a ray asks)."""

    def __init__(self, name: str, n: Union[Callable, float], wavelength_range=None):
        self.name = name
        self.wavelength_range = wavelength_range
        self._spec = False  # device form of a callable: computed once, on first use
        if isinstance(n, (int, float)):
            self._const = float(n)
            self.n_func = lambda wavelength_m, _c=n: _c
        else:
            self._const = None
            self.n_func = n

    def n(self, wavelength_m: float) -> float:
        return self.n_func(wavelength_m)

    def device_spec(self):
        """('const', n) | ('sellmeier', Bs, Cs) | ('cheb', lo, hi, coefficients) | None when not representable on the
        device (a callable that no Chebyshev series reproduces: the scene compiler raises)."""
        if self._const is not None:
            return ("const", self._const)
        if self._spec is False:
            self._spec = callable_spec(self.n_func, self.wavelength_range)
        return self._spec


class ConstMaterial(Material):
    def __init__(self, name: str = "", n: float = 1.0):
        super().__init__(name, n)


class Vacuum(Material):
    def __init__(self):
        super().__init__("Vacuum", n=1.0)


class RefractiveIndex:
    """Descriptor: assigning a float or a Material stores a Material; reading returns a
    function `get_n(wavelength_m=None)` (material.py:48-85)."""

    def __init__(self, storage_name: str):
        self.storage_name = storage_name

    def __get__(self, instance, owner):
        if instance is None:
            return self
        material = instance.__dict__.get(self.storage_name)
        if material is None:
            raise AttributeError(f"Material for {self.storage_name} not initialized.")

        def get_n(wavelength_m=None):
            if wavelength_m is None:
                wavelength_m = 0.0
                if getattr(instance, "wavelength", None) is not None:
                    wavelength_m = instance.wavelength * instance.unit
            return float(material.n(wavelength_m))

        return get_n

    def __set__(self, instance, value):
        if not isinstance(value, Material):
            value = Material("Constant", n=float(value))
        instance.__dict__[self.storage_name] = value


class SellmeierMaterial(Material):
    """n^2 = 1 + sum_i B_i L^2 / (L^2 - C_i), L in microns (material.py:106-120)."""

    def __init__(self, name: str, Bs: List[float], Cs: List[float]):
        self.Bs = Bs
        self.Cs = Cs
        super().__init__(name, self.sellmeier_n)

    def sellmeier_n(self, wavelength_m: float) -> float:
        lam2 = (wavelength_m / 1e-6) ** 2
        acc = 1.0
        for b, c in zip(self.Bs, self.Cs):
            acc += b * lam2 / (lam2 - c)
        return np.sqrt(acc)

    def device_spec(self):
        if len(self.Bs) > 3 or len(self.Bs) != len(self.Cs):
            return Material.device_spec(self)  # more terms than the closed form holds: a series of the same function
        # missing terms: B = 0 over a denominator that cannot vanish (lam^2 - C with C = -1 is > 0 for every wavelength;
        # the device forms one common denominator of the three terms, so a padded C = 1 would give 0/0 at exactly 1 um)
        pad = 3 - len(self.Bs)
        return ("sellmeier", list(self.Bs) + [0.0] * pad, list(self.Cs) + [-1.0] * pad)


# Glass catalogue: coefficient tables are data (material.py:123-168).
_GLASSES = {
    "Glass_NBK7": ("BK7", [1.03961212, 0.231792344, 1.01046945], [0.00600069867, 0.0200179144, 103.560653]),
    "Glass_UVFS": ("UV Fused Silica", [0.6961663, 0.4079426, 0.8974794], [0.0684043**2, 0.1162414**2, 9.896161**2]),
    "Glass_NSF5": ("N_SF5", [1.52481889, 0.187085527, 1.42729015], [0.011254756, 0.0588995392, 129.141675]),
    "Glass_NSF11": ("N_SF11", [1.73759695, 0.313747346, 1.89878101], [0.013188707, 0.0623068142, 155.23629]),
    "Glass_NSK2": ("N_SK2", [1.28189012, 0.257738258, 0.96818604], [0.0072719164, 0.0242823527, 110.377773]),
    "Glass_NSF57": ("N_SF57", [1.87543481, 0.37375749, 2.30001797], [0.0141749518, 0.0640509927, 177.389795]),
}


def _glass_class(cls_name):
    label, Bs, Cs = _GLASSES[cls_name]

    def __init__(self):
        SellmeierMaterial.__init__(self, label, list(Bs), list(Cs))

    return type(cls_name, (SellmeierMaterial,), {"__init__": __init__, "__module__": __name__})


Glass_NBK7 = _glass_class("Glass_NBK7")
Glass_UVFS = _glass_class("Glass_UVFS")
Glass_NSF5 = _glass_class("Glass_NSF5")
Glass_NSF11 = _glass_class("Glass_NSF11")
Glass_NSK2 = _glass_class("Glass_NSK2")
Glass_NSF57 = _glass_class("Glass_NSF57")

"""CPU: device forms of user callables (optable_amd/cheb.py) — the fit reproduces smooth functions to the last digits,
refuses what it cannot reproduce, and the scene compiler lowers ASphere(f_asphere = callable) / Material(n = callable)
to series records (component_group.py:1014-1055, material.py:4-21).  The traced numbers are pinned against the
reference by fixture g24 (tests/test_oracle_golden.py on the CPU, tests/test_gpu_parity.py on the MI355X)."""
import numpy as np
import pytest

import optable_amd as oa
from optable_amd import abi, cheb
from optable_amd.scene import SceneError


@pytest.mark.parametrize("func,lo,hi", [
    (lambda r: 0.5 * (np.cosh(0.3 * r) - 1.0) + 1e-3 * r**4, -0.01, 3.7),
    (lambda w: 1.6 + 8e-15 / w**2, 0.2e-6, 2.5e-6),
    (lambda w: np.sqrt(2.2 + 6e-15 / w**2 - 1e9 * w**2), 0.3e-6, 2.0e-6),
    (lambda r: r**2 / (10 * (1 + np.sqrt(1 - 0.3 * r**2 / 100))), -0.001, 3.0),
])
def test_series_reproduces_the_function(func, lo, hi):
    coef, err = cheb.fit(func, lo, hi)
    x = np.random.default_rng(1).uniform(lo, hi, 5000)
    exact = np.array([func(v) for v in x])
    scale = np.abs(exact).max()
    assert np.abs(cheb.evaluate(coef, lo, hi, x) - exact).max() <= 1e-13 * scale
    assert err <= cheb.REL_TOL * scale and len(coef) <= 128
    # derivative series against a central difference of the function
    d1 = cheb.derivative(coef, lo, hi)
    h = 1e-5 * (hi - lo)
    xi = x[(x > lo + h) & (x < hi - h)][:200]
    fd = np.array([(func(v + h) - func(v - h)) / (2 * h) for v in xi])
    assert np.abs(cheb.evaluate(d1, lo, hi, xi) - fd).max() <= 1e-6 * (np.abs(fd).max() + scale / (hi - lo))


@pytest.mark.parametrize("func", [lambda r: abs(r - 1.0), lambda r: np.sign(r - 1.0), lambda r: 1.0 / (r - 1.0), lambda r: float("nan")])
def test_unsmooth_functions_are_refused(func):
    with pytest.raises(cheb.FitError):
        cheb.fit(func, 0.0, 3.0)


def test_scene_compiler_lowers_callables_to_series_records():
    sag = lambda r: 0.5 * (np.cosh(0.3 * r) - 1.0)  # noqa: E731
    glass = oa.Material("cauchy", n=lambda w: 1.6 + 8e-15 / w**2, wavelength_range=(0.35e-6, 1.2e-6))
    scene = oa.compile_scene([oa.ASphericLens([5, 0, 0], CT=0.6, f_asphere_1=sag, f_asphere_2=None, diameter=2.4, n=glass)])
    nodes = scene.node_table()
    assert abi.MAT_CHEB in [m.kind for m in scene.materials[: scene.n_materials]]
    assert scene.wavelength_range == (0.35e-6, 1.2e-6)
    asph = nodes[nodes["shape"] == 10][0]
    aux = np.ctypeslib.as_array(scene.aux)[: scene.n_aux]
    rec = aux[asph["aux"]:]
    n, lo, hi = int(rec[0]), rec[1], rec[2]
    assert lo < 0 < 1.2 * np.sqrt(2) < hi  # the finite-difference stencil reaches across the axis, the root scan into the box corners
    r = np.linspace(0, 1.2, 50)
    np.testing.assert_allclose(cheb.evaluate(rec[3:3 + n], lo, hi, r), [sag(v) for v in r], rtol=0, atol=1e-15)
    slope = cheb.evaluate(rec[3 + n:3 + 2 * n], lo, hi, r)
    np.testing.assert_allclose(slope, 0.15 * np.sinh(0.3 * r), rtol=0, atol=1e-12)
    # a constant function is a constant material, not a series
    flat = oa.compile_scene([oa.GlassSlab([0, 0, 0], n1=1.0, n2=oa.Material("flat", n=lambda w: 1.45))])
    assert [m.kind for m in flat.materials[: flat.n_materials]] == [abi.MAT_CONST, abi.MAT_CONST]


def test_callables_without_a_device_form_are_refused_with_the_reason():
    with pytest.raises(SceneError, match="Chebyshev"):
        oa.compile_scene([oa.ASphericLens([5, 0, 0], CT=0.6, f_asphere_1=lambda r: abs(r - 0.5), f_asphere_2=None, diameter=2.4, n=1.5)])
    with pytest.raises(SceneError, match="cauchy kink"):
        oa.compile_scene([oa.GlassSlab([0, 0, 0], n1=1.0, n2=oa.Material("cauchy kink", n=lambda w: 1.5 + abs(w - 8e-7)))])

"""GPU (MI355X): scenes that carry user functions — ASphericLens(f_asphere = callable), Material(n = callable)
(component_group.py:1014-1055, material.py:4-21) — traced through their series form (optable_amd/cheb.py) against the
oracle, which reads the same series, on batches; the reference run on the callables themselves is fixture g24
(tests/test_gpu_parity.py::test_ray_tracing_matches_reference_fixture[g24_callables])."""
import numpy as np
import pytest

import scenes
from optable_amd import abi

pytestmark = pytest.mark.gpu


def _case(n, seed=11):
    import optable_amd as oa

    comps = scenes.g24_callables(oa)["components"]
    table = oa.OpticalTable()
    table.add_components(comps)
    rng = np.random.default_rng(seed)
    o = np.stack([np.zeros(n), rng.uniform(-1.0, 1.0, n), rng.uniform(-0.6, 0.6, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.02, 0.02, n), rng.uniform(-0.02, 0.02, n)], 1)
    wl = rng.uniform(420e-7, 1000e-7, n)
    return table, o, d, wl


@pytest.mark.parametrize("kernel", [0, 2])
def test_callable_scene_matches_oracle(kernel, oracle):
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine

    n, K = 4000, 24
    table, o, d, wl = _case(n)
    batch = RayBatch.from_arrays(o, d, wavelength=wl, q=1j * np.pi * scenes.W0**2 / wl)
    eng = get_engine()
    eng.set_option(abi.OPT_KERNEL, kernel)
    try:
        got = table.trace_batch(batch, max_segments=K).to_host(reference_order=True)
    finally:
        eng.set_option(abi.OPT_KERNEL, 0)
    scene, host = table.compile(), batch.to_host()
    ref = oracle.trace(scene, host, max_trace_num=K)
    assert len(got["ray"]) == len(ref["ray"]) > 5 * n
    np.testing.assert_array_equal(got["surface"], ref["surface"])
    for f in abi.SEG_FIELDS:
        if f not in ("q_re", "q_im"):
            np.testing.assert_allclose(got[f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)
    qtol, _ = oracle.q_tolerance(scene, host, ref, K)
    qerr = np.hypot(got["q_re"] - ref["q_re"], got["q_im"] - ref["q_im"])
    assert np.all(qerr <= qtol), float((qerr / qtol).max())
    assert len(np.unique(np.round(ref["n"], 9))) > 100  # the two glasses really disperse: an index per wavelength


def test_callable_scene_fp32_follows_fp64():
    from optable_amd.batch import RayBatch

    n, K = 4000, 24
    table, o, d, wl = _case(n)
    out = {}
    for prec in ("f64", "f32"):
        b = RayBatch.from_arrays(o, d, wavelength=wl, q=1j * np.pi * scenes.W0**2 / wl, precision=prec)
        out[prec] = table.trace_batch(b, max_segments=K).to_host(reference_order=True)
    seq = lambda x: [tuple(x["surface"][x["ray"] == r]) for r in range(0, n, 7)]  # noqa: E731
    a, b = seq(out["f64"]), seq(out["f32"])
    same = np.mean([x == y for x, y in zip(a, b)])
    assert same >= 0.99, same
    if len(out["f64"]["ray"]) == len(out["f32"]["ray"]) and np.array_equal(out["f64"]["surface"], out["f32"]["surface"]):
        for f in ("ox", "oy", "oz"):
            assert np.abs(out["f64"][f] - out["f32"][f]).max() < 2e-3


def test_wavelengths_outside_the_fitted_interval_are_refused():
    import optable_amd as oa
    from optable_amd.batch import RayBatch

    glass = oa.Material("cauchy", n=lambda w: 1.6 + 8e-15 / w**2, wavelength_range=(0.4e-6, 0.9e-6))
    table = oa.OpticalTable()
    table.add_components([oa.GlassSlab([3, 0, 0], width=2, height=2, thickness=0.5, n1=1.0, n2=glass)])
    o, d = np.zeros((4, 3)), np.tile([1.0, 0.0, 0.0], (4, 1))
    ok = RayBatch.from_arrays(o, d, wavelength=[450e-7, 600e-7, 700e-7, 880e-7])
    assert int(table.trace_batch(ok, max_segments=4).count.abs().sum()) == 12
    with pytest.raises(ValueError, match="wavelength_range"):
        table.trace_batch(RayBatch.from_arrays(o, d, wavelength=[450e-7, 600e-7, 700e-7, 1100e-7]), max_segments=4)

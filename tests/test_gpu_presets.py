"""GPU (MI355X): the FE / FM presets (tables.h) — light scenes made of the reference's everyday parts (TriangularPrism with its
count-limited faces, Block with a hole, BiConvexLens; + DovePrism's tilted polygon faces) no longer fall into the all-features
instantiation: lane per ray in BOTH precisions (double precision used to have no such kernel for them), smaller generation
kernels.  Which instantiation traces a scene must not show in the results: lane per ray == rolling lists (the all-features
preset) bit for bit, both against the oracle."""
import numpy as np
import pytest
import torch

import optable_amd as oa
import scenes
from optable_amd import abi
from optable_amd.batch import RayBatch
from optable_amd.engine import get_engine

pytestmark = pytest.mark.gpu
Q = 1j * np.pi * scenes.W0**2 / scenes.WL


def _parts(dove, split=0.0):
    comps = [oa.TriangularPrism([4, -0.6, 0], width=2.0, height=2.0, n1=1.0, n2=1.5, reflectivity_1=split, transmission_1=1.0 - split),
             oa.Block([8, 0.4, 0], hole=oa.Circle(0.6), width=3, height=3),
             oa.BiConvexLens([11, 0.4, 0], CT=0.6, R1=12.0, R2=-12.0, diameter=3.0, n=1.5),
             oa.Mirror([16, 0.4, 0], radius=3.0).RotZ(np.pi + 0.05)]
    if dove:
        comps.insert(2, oa.DovePrism([9.5, 0.1, 0], L=1.2, D=0.5, Ng=1.5))
    return comps


def _rays(n, precision):
    rng = np.random.default_rng(11)
    o = np.stack([np.zeros(n), rng.uniform(-0.5, 0.5, n), rng.uniform(-0.4, 0.4, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.03, 0.03, n), rng.uniform(-0.02, 0.02, n)], 1)
    return RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=Q, precision=precision)


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("dove", [False, True], ids=["FE", "FM"])
def test_everyday_parts_lane_per_ray_equals_all_features_lists(dove, precision, oracle):
    table = oa.OpticalTable()
    table.add_components(_parts(dove))
    n, K = 30_000, 16
    batch = _rays(n, precision)
    eng = get_engine()
    scene = table.compile()
    try:
        fused = table.trace_batch(batch, max_segments=K, layout="slots", scene=scene)
        assert eng.last_launch()["kernel"] == 1, eng.last_launch()  # lane per ray, in double precision too
        eng.set_option(abi.OPT_KERNEL, 2)
        lists = table.trace_batch(batch, max_segments=K, layout="slots", scene=scene)
        assert eng.last_launch()["kernel"] == 2
    finally:
        eng.set_option(abi.OPT_KERNEL, 0)
    a, b = fused.to_host(reference_order=True), lists.to_host(reference_order=True)
    assert len(a["ray"]) > 2 * n
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    if precision == "f64":
        small = batch.slice(0, 2000)
        got = table.trace_batch(small, max_segments=K, scene=scene).to_host(reference_order=True)
        ref = oracle.trace(scene, small.to_host(), max_trace_num=K)
        np.testing.assert_array_equal(got["ray"], ref["ray"])
        np.testing.assert_array_equal(got["surface"], ref["surface"])
        for f in abi.SEG_FIELDS:
            np.testing.assert_allclose(got[f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)


@pytest.mark.parametrize("dove", [False, True], ids=["FE", "FM"])
def test_everyday_parts_ray_trees_match_the_oracle(dove, oracle):
    """The generation kernels of the same presets (count, emit and the probe pass of the count gates): a prism whose entrance
    face splits every ray, fp64, against the oracle."""
    table = oa.OpticalTable()
    table.add_components(_parts(dove, split=0.3))
    scene = table.compile()
    assert scene.max_children == 2 and len(scene.limited) == 2
    batch = _rays(1500, "f64")
    got = table.trace_batch(batch, max_segments=14, scene=scene).to_host(reference_order=True)
    ref = oracle.trace(scene, batch.to_host(), max_trace_num=14)
    assert len(ref["ray"]) > 3 * batch.n
    np.testing.assert_array_equal(got["ray"], ref["ray"])
    np.testing.assert_array_equal(got["surface"], ref["surface"])
    for f in abi.SEG_FIELDS:
        np.testing.assert_allclose(got[f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)

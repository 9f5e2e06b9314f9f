"""GPU (MI355X): k_gen_one — a generation of the breadth-first ray-tree trace in ONE pass (every workgroup traces its tile of
rays once, keeps the children in registers, takes its output offsets from a decoupled look-back over per-tile descriptors;
per-ray budgets instead of a per-tree table that is read and written in the same pass) — against the two-pass kernels
(count + scan + emit, OT_OPT_GEN_ONEPASS = 0): the flat list in generation order must be IDENTICAL, element by element, with
the same trees capped and the same budgets left; and against the oracle."""
import numpy as np
import pytest
import torch

import optable_amd as oa
import scenes
from optable_amd import abi
from optable_amd import workloads as W
from optable_amd.batch import RayBatch
from optable_amd.engine import get_engine

pytestmark = pytest.mark.gpu
Q = 1j * np.pi * W.W0**2 / W.WL


def _both(scene, batch, cap, **kw):
    eng = get_engine()
    eng.upload(scene)
    try:
        eng.set_option(abi.OPT_GEN_ONEPASS, 0)
        two = eng.trace_tree(batch, cap, **kw)
        eng.set_option(abi.OPT_GEN_ONEPASS, 1)
        one = eng.trace_tree(batch, cap, **kw)
        eng.set_option(abi.OPT_GEN_ONEPASS, -1)  # the default: one pass for generations of up to 65536 rays, two above — the
        mixed = eng.trace_tree(batch, cap, **kw)  # per-ray budgets are re-seeded from the tree table at every change-over
    finally:
        eng.set_option(abi.OPT_GEN_ONEPASS, -1)
    assert one.n_valid == two.n_valid == mixed.n_valid and one.n_valid > 0
    m = one.n_valid
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        assert torch.equal(one.field(f)[:m], two.field(f)[:m]), f
        assert torch.equal(mixed.field(f)[:m], two.field(f)[:m]), f
    assert torch.equal(one.capped, two.capped) and torch.equal(mixed.capped, two.capped)
    return one


def _lattice():
    comps = []
    for k in range(5):
        comps.append(oa.BeamSplitter([2.0 * (k + 1), 0, 0], width=6, height=2, eta=0.5).RotZ(np.pi / 4))
        comps.append(oa.Mirror([2.0 * (k + 1), 3.0 + 0.1 * k, 0], radius=2).RotZ(-np.pi / 2))
        comps.append(oa.BeamSplitter([2.0 * (k + 1) + 1.0, 1.5, 0], width=6, height=2, eta=0.3).RotZ(-np.pi / 4))
    t = oa.OpticalTable()
    t.add_components(comps)
    return t.compile()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_one_pass_equals_two_pass_on_cfg4_with_reflectivity(precision, oracle):
    table = oa.OpticalTable()
    table.add_components(W.cfg4_components(oa, reflectivity=0.2))
    scene = table.compile()
    o, d, wl = W.cfg4_rays(20_000, 4)  # x 64 wavelengths = 1.28e6 trees
    batch = RayBatch.from_arrays(o, d, wavelength=wl, q=1j * np.pi * W.W0**2 / wl, precision=precision)
    segs = _both(scene, batch, 12, out_capacity=batch.n * 13)
    assert segs.n_valid == 12 * batch.n and bool(segs.capped.all())
    if precision == "f64":  # ... and a slice of it against the oracle
        small = batch.slice(0, 400)
        get_engine().set_option(abi.OPT_GEN_ONEPASS, 1)
        try:
            got = get_engine().trace_tree(small, 12).to_host(reference_order=True)
        finally:
            get_engine().set_option(abi.OPT_GEN_ONEPASS, -1)
        ref = oracle.trace(scene, small.to_host(), max_trace_num=12)
        np.testing.assert_array_equal(got["ray"], ref["ray"])
        np.testing.assert_array_equal(got["surface"], ref["surface"])
        for f in abi.SEG_FIELDS:
            np.testing.assert_allclose(got[f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)


@pytest.mark.parametrize("cap", [3, 7, 12, 40, 300])
@pytest.mark.parametrize("drop", [0, 1])
def test_one_pass_equals_two_pass_on_bushy_trees(cap, drop):
    """A lattice of beam splitters: trees that double every generation, spanning many tiles of a generation, capped in their
    largest generation — the per-ray budgets, the doomed-children rule and the look-back across hundreds of tiles."""
    scene = _lattice()
    n = 3000 if cap != 40 else 12_000  # (cap 40 at 12 000 trees: generations grow past 65 536 rays and shrink again — both change-overs of the default mode)
    rng = np.random.default_rng(5)
    o = np.stack([np.zeros(n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.2, 0.2, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.02, 0.02, n), rng.uniform(-0.01, 0.01, n)], 1)
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=Q)
    eng = get_engine()
    try:
        eng.set_option(abi.OPT_GEN_DROP_DOOMED, drop)
        _both(scene, batch, cap)
    finally:
        eng.set_option(abi.OPT_GEN_DROP_DOOMED, 1)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_one_pass_equals_two_pass_on_the_cavity(precision):
    """examples/cavity_4mir.py: four R = 0.9 mirrors — deep trees (300 generations of one or two rays each)."""
    L, D, R = 10 * 4 / 3, 4, 0.9
    comps = [oa.Mirror([0, 0, 0], radius=D, reflectivity=R, transmission=1 - R).RotZ(-np.pi / 4),
             oa.Mirror([L, 0, 0], radius=D, reflectivity=R, transmission=1 - R).RotZ(+np.pi / 4 + 0.02),
             oa.Mirror([L, -L, 0], radius=D, reflectivity=R, transmission=1 - R).RotZ(-np.pi / 4 + 0.02),
             oa.Mirror([0, -L, 0], radius=D, reflectivity=R, transmission=1 - R).RotZ(+np.pi / 4)]
    table = oa.OpticalTable()
    table.add_components(comps)
    n = 4096
    o = np.tile([2.0, 0, 0], (n, 1)) + np.linspace(0, 1e-3, n)[:, None] * np.array([0, 1, 0])
    batch = RayBatch.from_arrays(o, np.tile([1.0, 0, 0], (n, 1)), precision=precision)
    _both(table.compile(), batch, 300)


def test_one_pass_equals_two_pass_on_a_heavy_planar_scene():
    """cfg 3 with 10 % reflecting slab faces (32 components under a top-level grid, the planar preset with grids), fp32."""
    table = oa.OpticalTable()
    table.add_components(W.cfg3_components(oa, slab_reflectivity=0.1))
    o, d = W.cfg3_rays(200_000, 2)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q, precision="f32")
    _both(table.compile(), batch, 20)


def test_one_pass_resumes_after_its_buffers_grow():
    """Segment arrays and generation buffers that are too small at first: the library hands the pending generation back, the
    engine grows the buffers and calls again — the per-ray budgets of the one-pass kernels are re-seeded from the tree
    table on re-entry and the trace continues where it stopped."""
    scene = _lattice()
    n = 2000
    rng = np.random.default_rng(6)
    o = np.stack([np.zeros(n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.2, 0.2, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.02, 0.02, n), rng.uniform(-0.01, 0.01, n)], 1)
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=Q)
    eng = get_engine()
    eng.upload(scene)
    eng.set_option(abi.OPT_GEN_ONEPASS, 1)
    try:
        big = eng.trace_tree(batch, 40, out_capacity=80 * n)
        small = eng.trace_tree(batch, 40, out_capacity=1024)  # grows the segment arrays (and, the trees doubling, the buffers) several times
    finally:
        eng.set_option(abi.OPT_GEN_ONEPASS, -1)
    assert small.n_valid == big.n_valid
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        assert torch.equal(small.field(f)[: small.n_valid], big.field(f)[: big.n_valid]), f

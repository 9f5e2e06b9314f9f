"""GPU (MI355X): the look-ahead of the two-pass generation kernels (k_gen_pass MODE 2 + k_gen_recount, OT_OPT_GEN_AHEAD) — the
emit pass of a generation counts the children of the children it writes, and the next generation replaces its count pass
over the ray records by a pass over one byte per ray — against the plain count + scan + emit: the flat list in generation
order must be IDENTICAL, element by element, with the same trees capped, and the two traces of a ray (as a child, as a
parent) must never disagree (ot_debug_generation_mismatches stays 0)."""
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


def _both(scene, batch, cap, onepass=0, **kw):
    eng = get_engine()
    eng.upload(scene)
    before = eng.generation_mismatches()
    try:
        eng.set_option(abi.OPT_GEN_ONEPASS, onepass)
        eng.set_option(abi.OPT_GEN_AHEAD, 0)
        plain = eng.trace_tree(batch, cap, **kw)
        eng.set_option(abi.OPT_GEN_AHEAD, 1)
        ahead = eng.trace_tree(batch, cap, **kw)
    finally:
        eng.set_option(abi.OPT_GEN_ONEPASS, -1)
        eng.set_option(abi.OPT_GEN_AHEAD, 1)
    assert ahead.n_valid == plain.n_valid and ahead.n_valid > 0
    m = ahead.n_valid
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        assert torch.equal(ahead.field(f)[:m], plain.field(f)[:m]), f
    assert torch.equal(ahead.capped, plain.capped)
    assert eng.generation_mismatches() == before
    return ahead


def _lattice():
    comps = []
    for k in range(5):
        comps.append(oa.BeamSplitter([2.0 * (k + 1), 0, 0], width=6, height=2, eta=0.5).RotZ(np.pi / 4))
        comps.append(oa.Mirror([2.0 * (k + 1), 3.0 + 0.1 * k, 0], radius=2).RotZ(-np.pi / 2))
        comps.append(oa.BeamSplitter([2.0 * (k + 1) + 1.0, 1.5, 0], width=6, height=2, eta=0.3).RotZ(-np.pi / 4))
    t = oa.OpticalTable()
    t.add_components(comps)
    return t.compile()


def _lattice_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = np.stack([np.zeros(n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.2, 0.2, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.02, 0.02, n), rng.uniform(-0.01, 0.01, n)], 1)
    return RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=Q)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_look_ahead_equals_count_pass_on_cfg4_with_reflectivity(precision, oracle):
    table = oa.OpticalTable()
    table.add_components(W.cfg4_components(oa, reflectivity=0.2))
    scene = table.compile()
    o, d, wl = W.cfg4_rays(20_000, 4)  # x 64 wavelengths = 1.28e6 trees
    batch = RayBatch.from_arrays(o, d, wavelength=wl, q=1j * np.pi * W.W0**2 / wl, precision=precision)
    segs = _both(scene, batch, 12, out_capacity=batch.n * 13)
    assert segs.n_valid == 12 * batch.n and bool(segs.capped.all())
    if precision == "f64":  # ... and a slice of it against the oracle (two passes with look-ahead only: no one-pass generation)
        small = batch.slice(0, 400)
        get_engine().set_option(abi.OPT_GEN_ONEPASS, 0)
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
def test_look_ahead_equals_count_pass_on_bushy_trees(cap, drop):
    """A lattice of beam splitters: trees that double every generation and are capped in their largest one — budget cut and
    doomed-children rule are k_gen_recount's, over bytes written by the generation before."""
    eng = get_engine()
    try:
        eng.set_option(abi.OPT_GEN_DROP_DOOMED, drop)
        _both(_lattice(), _lattice_rays(3000 if cap != 40 else 12_000, 5), cap)
        _both(_lattice(), _lattice_rays(3000, 6), cap, onepass=-1)  # small generations in one pass, large ones with look-ahead: both change-overs
    finally:
        eng.set_option(abi.OPT_GEN_DROP_DOOMED, 1)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_look_ahead_equals_count_pass_on_the_cavity(precision):
    """examples/cavity_4mir.py: four R = 0.9 mirrors — 300 generations of one or two rays per tree."""
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


def test_look_ahead_resumes_after_its_buffers_grow():
    """Segment arrays and generation buffers that are too small at first: the pending generation comes back as the caller's
    input, and a caller's rays always get a count pass of their own."""
    scene = _lattice()
    batch = _lattice_rays(2000, 6)
    eng = get_engine()
    eng.upload(scene)
    eng.set_option(abi.OPT_GEN_ONEPASS, 0)
    try:
        big = eng.trace_tree(batch, 40, out_capacity=80 * batch.n)
        small = eng.trace_tree(batch, 40, out_capacity=1024)
    finally:
        eng.set_option(abi.OPT_GEN_ONEPASS, -1)
    assert small.n_valid == big.n_valid
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        assert torch.equal(small.field(f)[: small.n_valid], big.field(f)[: big.n_valid]), f


def test_scenes_with_count_gates_or_heavy_searches_keep_their_count_pass():
    """Count-limited leaves (probe pass + per-slot scans) and scenes whose emit pass reuses the count pass's decision have no
    look-ahead: the option changes nothing for them."""
    table = oa.OpticalTable()
    table.add_components(W.cfg3_components(oa, slab_reflectivity=0.1))
    o, d = W.cfg3_rays(50_000, 2)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q, precision="f32")
    _both(table.compile(), batch, 20)

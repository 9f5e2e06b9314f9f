"""GPU (MI355X): k_trace_refill — the live rays of a mixed scene in registers, every lane taking its next fresh ray in place
(kernels.h) — against the per-wave lists of k_trace_rolling: which lane or pass carries a ray does not enter its
arithmetic, so every record must agree bit for bit, in both precisions and both output layouts, from one ray to a batch
that drains many tickets per wave; and against the oracle."""
import numpy as np
import pytest
import torch

import scenes
from optable_amd import abi

pytestmark = pytest.mark.gpu
REFILL = 32  # bit 5 of last_launch()["pair_queue"]: rays in registers


def _setup(n, precision="f32", seed=2):
    import optable_amd as oa
    from optable_amd.batch import RayBatch

    table = oa.OpticalTable()
    table.add_components(scenes.cfg3_components(oa))
    o, d = scenes.cfg3_rays(n, seed)
    q = 1j * np.pi * scenes.W0**2 / scenes.WL
    return table, RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=q, precision=precision, device="cuda")


def _both(table, batch, K, ticket=0, chunk=512):
    from optable_amd.engine import get_engine

    eng = get_engine()
    try:
        eng.set_option(abi.OPT_REFILL, 0)
        lists = table.trace_batch(batch, max_segments=K, layout="slots")
        assert not eng.last_launch()["pair_queue"] & REFILL
        eng.set_option(abi.OPT_REFILL, 1)
        eng.set_option(abi.OPT_REFILL_TICKET, ticket)
        eng.set_option(abi.OPT_APPEND_CHUNK, chunk)
        slots = table.trace_batch(batch, max_segments=K, layout="slots")
        assert not eng.last_launch()["pair_queue"] & REFILL  # (the [k][ray] slots stay with the lists: they write them faster)
        app = table.trace_batch(batch, max_segments=K, layout="append")
        info = eng.last_launch()
        assert info["kernel"] == 2 and info["pair_queue"] & REFILL and info["pair_queue"] & 4, info
    finally:
        eng.set_option(abi.OPT_REFILL, 0)
        eng.set_option(abi.OPT_REFILL_TICKET, 0)
        eng.set_option(abi.OPT_APPEND_CHUNK, 512)
    return lists, slots, app


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("n,K,ticket", [(1, 20, 0), (63, 20, 0), (64, 3, 64), (65, 20, 0), (1000, 20, 128), (20_011, 20, 0), (400_003, 6, 0), (1_200_000, 20, 0)])
def test_refill_equals_per_wave_lists(precision, n, K, ticket):
    table, batch = _setup(n, precision)
    lists, slots, app = _both(table, batch, K, ticket)
    assert torch.equal(lists.count, slots.count)
    valid = lists.valid_mask()
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        assert torch.equal(lists.field(f)[valid], slots.field(f)[valid]), f
    assert torch.equal(lists.count, app.count)
    if n <= 400_003:
        a, b = slots.to_host(reference_order=True), app.to_host(reference_order=True)
        assert len(a["ray"]) == int(lists.count.abs().sum().item())
        for f in abi.SEG_FIELDS + ("ray", "surface", "count"):
            np.testing.assert_array_equal(a[f], b[f], err_msg=f)


def test_refill_against_the_oracle(oracle):
    """fp64 through k_trace_refill against the oracle at 1e-9; every fourth ray dead on arrival, a cap that cuts rays."""
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine
    import optable_amd as oa

    n, K = 6000, 7
    table = oa.OpticalTable()
    table.add_components(scenes.cfg3_components(oa))
    o, d = scenes.cfg3_rays(n, 11)
    q = 1j * np.pi * scenes.W0**2 / scenes.WL
    flags = np.full(n, abi.RAY_HAS_Q, dtype=np.int32)
    flags[::4] |= abi.RAY_DEAD
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=q, device="cuda")
    batch.flags.copy_(torch.from_numpy(flags).to(batch.device))
    get_engine().set_option(abi.OPT_REFILL, 1)
    try:
        segs = table.trace_batch(batch, max_segments=K, layout="append")
        assert get_engine().last_launch()["pair_queue"] & REFILL
    finally:
        get_engine().set_option(abi.OPT_REFILL, 0)
    ref = oracle.trace(table.compile(), batch.to_host(), max_trace_num=K)
    got = segs.to_host(reference_order=True)
    np.testing.assert_array_equal(got["ray"], ref["ray"])
    np.testing.assert_array_equal(got["surface"], ref["surface"])
    for f in abi.SEG_FIELDS:
        np.testing.assert_allclose(got[f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)


def _planar_scene(oa, rng, polygons):
    """Random planar scenes big enough for a top-level grid (>= 12 components): mirrors, thin lenses, slabs, prisms, blocks."""
    comps = []
    for ix in range(6):
        for iy in range(3):
            pos = [4.0 * (ix + 1) + rng.uniform(-0.5, 0.5), 3.0 * (iy - 1) + rng.uniform(-0.4, 0.4), rng.uniform(-0.1, 0.1)]
            ang = rng.uniform(-np.pi, np.pi)
            kind = int(rng.integers(0, 6 if polygons else 5))
            if kind == 0:
                comps.append(oa.Mirror(pos, radius=rng.uniform(0.6, 1.2)).RotZ(ang))
            elif kind == 1:
                comps.append(oa.Lens(pos, focal_length=rng.uniform(4, 12), radius=1.0).RotZ(0.2 * ang))
            elif kind == 2:
                comps.append(oa.GlassSlab(pos, width=2, height=2, thickness=rng.uniform(0.2, 0.7), n1=1, n2=rng.uniform(1.3, 1.7)).RotZ(0.3 * ang))
            elif kind == 3:
                comps.append(oa.Prism(pos, width=1.5, height=2, n1=1, n2=1.5).RotZ(ang))
            elif kind == 4:
                comps.append(oa.SquareMirror(pos, width=1.4, height=1.4).RotZ(ang).RotY(rng.uniform(-0.1, 0.1)))
            elif rng.uniform() < 0.5:  # a pentagonal mirror: polygon aperture in the component plane
                m = oa.Mirror(pos, radius=1.0).RotZ(ang)
                m.surface = oa.Polygon(np.array([[-0.9, -0.7], [0.8, -0.9], [1.0, 0.3], [0.1, 1.0], [-0.8, 0.6]]))
                comps.append(m)
            else:  # an absorbing plate with a round hole: boolean aperture
                comps.append(oa.Block(pos, hole=oa.Circle(0.35), width=1.6, height=1.6).RotZ(0.3 * ang))
    return comps


@pytest.mark.parametrize("polygons", [False, True])
@pytest.mark.parametrize("seed", range(6))
def test_refill_on_random_planar_scenes(seed, polygons):
    """Random planar scenes under a top-level grid, 30 011 rays, tickets of 64 and append chunks of 64 (a wave crosses into
    a new chunk every pass): refill through the append layout against the lists through the slots, fp32."""
    import optable_amd as oa
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine

    rng = np.random.default_rng(9100 + seed + (50 if polygons else 0))
    table = oa.OpticalTable()
    table.add_components(_planar_scene(oa, rng, polygons))
    n, K = 30_011, 16
    o = np.stack([np.zeros(n), rng.uniform(-4.5, 4.5, n), rng.uniform(-0.4, 0.4, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.08, 0.08, n), rng.uniform(-0.02, 0.02, n)], 1)
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j * np.pi * scenes.W0**2 / scenes.WL, precision="f32")
    eng = get_engine()
    eng.upload(table.compile())
    try:  # (engine level: a ray tree that branches is marked in `count` and left to the caller, by both kernels alike)
        eng.set_option(abi.OPT_KERNEL, 2)
        eng.set_option(abi.OPT_REFILL, 0)
        lists = eng.trace(batch, K)
        eng.set_option(abi.OPT_REFILL, 1)
        eng.set_option(abi.OPT_REFILL_TICKET, 64)
        eng.set_option(abi.OPT_APPEND_CHUNK, 64)
        app = eng.trace(batch, K, layout="append")
        refilled = bool(eng.last_launch()["pair_queue"] & REFILL)
    finally:
        eng.set_option(abi.OPT_KERNEL, 0)
        eng.set_option(abi.OPT_REFILL, 0)
        eng.set_option(abi.OPT_REFILL_TICKET, 0)
        eng.set_option(abi.OPT_APPEND_CHUNK, 512)
    if not refilled:
        pytest.skip("this scene got no top-level grid: it is not a mixed scene")
    assert torch.equal(lists.count, app.count)
    a, b = lists.to_host(reference_order=True), app.to_host(reference_order=True)
    assert len(a["ray"]) == int(lists.count.abs().sum().item())
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)


def test_refill_block_too_small_loses_records_not_memory():
    from optable_amd.engine import get_engine

    table, batch = _setup(30_000)
    K = 20
    get_engine().set_option(abi.OPT_REFILL, 1)
    try:
        full = table.trace_batch(batch, max_segments=K, layout="append")
        assert get_engine().last_launch()["pair_queue"] & REFILL
        records = int(full.count.abs().sum().item())
        small = table.trace_batch(batch, max_segments=K, layout="append", capacity=(records // 3) // 64 * 64)
        with pytest.raises(RuntimeError, match="capacity >= "):
            _ = small.n_valid
        assert torch.equal(small.count, full.count)
        fits = table.trace_batch(batch, max_segments=K, layout="append", capacity=get_engine().append_capacity(records))
        assert fits.n_valid >= records
    finally:
        get_engine().set_option(abi.OPT_REFILL, 0)

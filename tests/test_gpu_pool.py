"""GPU (MI355X): k_trace_pool — the live rays of a workgroup in one pool of 64-ray blocks shared by its sixteen waves
(kernels.h) — against the per-wave lists of k_trace_rolling: which wave carries a ray, and next to which other rays, does
not enter its arithmetic, so every record must agree bit for bit, in both output layouts, from a launch smaller than one
block to one that refills every pool many times; and against the oracle."""
import numpy as np
import pytest
import torch

import scenes
from optable_amd import abi

pytestmark = pytest.mark.gpu


def _setup(n, precision="f32", seed=3):
    import optable_amd as oa
    from optable_amd.batch import RayBatch

    table = oa.OpticalTable()
    table.add_components(scenes.cfg5_components(oa))
    o, d = scenes.cfg5_rays(n, seed)
    q = 1j * np.pi * scenes.W0**2 / scenes.WL
    return table, RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=q, precision=precision, device="cuda")


@pytest.mark.parametrize("n,K", [(1, 50), (63, 50), (64, 7), (1000, 50), (20_011, 50), (400_003, 12), (1_500_000, 50)])
def test_block_pool_equals_per_wave_lists(n, K):
    from optable_amd.engine import get_engine

    table, batch = _setup(n)
    eng = get_engine()
    try:
        eng.set_option(abi.OPT_BLOCK_POOL, 0)
        lists = table.trace_batch(batch, max_segments=K, layout="slots")
        assert not eng.last_launch()["pair_queue"] & 16
        eng.set_option(abi.OPT_BLOCK_POOL, 1)  # (the slots take the pool only when asked to: scattered stores)
        pool = table.trace_batch(batch, max_segments=K, layout="slots")
        info = eng.last_launch()
        assert info["kernel"] == 2 and info["pair_queue"] & 16 and info["pair_queue"] & 2, info
        app = table.trace_batch(batch, max_segments=K, layout="append")
        assert eng.last_launch()["pair_queue"] & 16 and eng.last_launch()["pair_queue"] & 4
    finally:
        eng.set_option(abi.OPT_BLOCK_POOL, -1)
    assert torch.equal(lists.count, pool.count)
    valid = lists.valid_mask()
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        assert torch.equal(lists.field(f)[valid], pool.field(f)[valid]), f
    if n <= 400_003:
        a, b = pool.to_host(reference_order=True), app.to_host(reference_order=True)
        for f in abi.SEG_FIELDS + ("ray", "surface", "count"):
            np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    else:
        np.testing.assert_array_equal(app.count.cpu().numpy(), pool.count.cpu().numpy())


def test_block_pool_against_the_oracle(oracle):
    """fp32 through the pool against the fp64 oracle: the contract of tests/test_gpu_fp32_oracle.py holds for the default
    launch (which is the pool); here the dead-on-arrival and capped rays: a batch in which every fourth ray is dead and
    the cap cuts most trees."""
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine
    import optable_amd as oa

    n, K = 5000, 9
    table = oa.OpticalTable()
    table.add_components(scenes.cfg5_components(oa))
    o, d = scenes.cfg5_rays(n, 5)
    q = 1j * np.pi * scenes.W0**2 / scenes.WL
    inten = np.ones(n)
    inten[::4] = 0.0
    b64 = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=q, intensity=inten, device="cuda")
    b32 = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=q, intensity=inten, precision="f32", device="cuda")
    s32 = table.trace_batch(b32, max_segments=K, layout="append")
    assert get_engine().last_launch()["pair_queue"] & 16
    ref = oracle.trace(table.compile(), b64.to_host(), max_trace_num=K)
    got = s32.to_host(reference_order=True)
    np.testing.assert_array_equal(got["ray"], ref["ray"])
    same = got["surface"] == ref["surface"]
    assert same.mean() > 0.999
    np.testing.assert_allclose(got["ox"][same], ref["ox"][same], atol=2e-3)


def _curved_scene(oa, rng):
    """Random scenes of the curved-surface preset (no thin lenses): what k_trace_pool is instantiated for."""
    comps = []
    for _ in range(int(rng.integers(4, 10))):
        pos = [rng.uniform(2, 26), rng.uniform(-4, 4), rng.uniform(-0.5, 0.5)]
        ang = rng.uniform(-np.pi, np.pi)
        kind = int(rng.integers(0, 7))
        if kind == 0:
            c = oa.Mirror(pos, radius=rng.uniform(0.5, 1.5)).RotZ(ang)
        elif kind == 1:
            c = oa.GlassSlab(pos, width=2, height=2, thickness=rng.uniform(0.2, 0.8), n1=1, n2=rng.uniform(1.3, 1.8)).RotZ(0.4 * ang)
        elif kind == 2:
            c = oa.BiConvexLens(pos, CT=0.5, R1=rng.uniform(6, 15), R2=-rng.uniform(6, 15), diameter=2.4, n=1.52).RotZ(0.2 * ang)
        elif kind == 3:
            c = oa.SquareMirror(pos, width=1.6, height=1.2).RotZ(ang).RotY(rng.uniform(-0.2, 0.2))
        elif kind == 4:
            c = oa.SphereRefractive(pos, radius=rng.uniform(4.0, 9.0), height=1.0, n1=1.0, n2=1.5).RotZ(0.2 * ang)
        elif kind == 5:
            c = oa.ASphericParametricLens(pos, CT=0.6, diameter=2.4, n=1.5, R=rng.uniform(5, 12), kappa=-1, a4=1e-4).RotZ(0.15 * ang)
        else:
            c = oa.MMA(origin=pos, N=(5, 6), pitch=0.4, roc=rng.uniform(15, 40), n=1.5, thickness=0.1, reflectivity=1,
                       transmission=0).RotZ(np.pi + 0.3 * ang)
        comps.append(c)
    comps.append(oa.BiConvexLens([28, 0, 0], CT=0.5, R1=9.0, R2=-9.0, diameter=9.0, n=1.5))  # (every scene has a curved surface)
    return comps


@pytest.mark.parametrize("seed", range(10))
def test_block_pool_on_random_curved_scenes(seed):
    """Random curved scenes, 40 011 rays, small append chunks (a workgroup crosses into a new chunk every 16 passes):
    the pool through the append layout against the per-wave lists through the slots."""
    import optable_amd as oa
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine

    rng = np.random.default_rng(7000 + seed)
    table = oa.OpticalTable()
    table.add_components(_curved_scene(oa, rng))
    n, K = 40_011, 14
    o = np.stack([np.zeros(n), rng.uniform(-4, 4, n), rng.uniform(-0.4, 0.4, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.15, 0.15, n), rng.uniform(-0.03, 0.03, n)], 1)
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j * np.pi * scenes.W0**2 / scenes.WL, precision="f32")
    eng = get_engine()
    eng.upload(table.compile())
    try:  # (engine level: a ray tree that branches is marked in `count` and left to the caller, by both kernels alike)
        eng.set_option(abi.OPT_KERNEL, 2)
        eng.set_option(abi.OPT_BLOCK_POOL, 0)
        lists = eng.trace(batch, K)
        eng.set_option(abi.OPT_BLOCK_POOL, -1)
        eng.set_option(abi.OPT_APPEND_CHUNK, 64)
        app = eng.trace(batch, K, layout="append")
        pooled = bool(eng.last_launch()["pair_queue"] & 16)
    finally:
        eng.set_option(abi.OPT_KERNEL, 0)
        eng.set_option(abi.OPT_BLOCK_POOL, -1)
        eng.set_option(abi.OPT_APPEND_CHUNK, 512)
    if not pooled:  # (a scene that happens to draw no curved surface but the closing lens can fall into another preset)
        pytest.skip("this scene is outside the preset k_trace_pool is built for")
    assert torch.equal(lists.count, app.count)
    a, b = lists.to_host(reference_order=True), app.to_host(reference_order=True)
    assert len(a["ray"]) == int(lists.count.abs().sum().item())
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)


def test_block_pool_block_too_small_loses_records_not_memory():
    """An append block that is too small: the trace completes, counts are right, the cursor names the size that fits, and
    nothing is written beyond the block (a guard band behind it stays as it was)."""
    from optable_amd.batch import SegmentBatch
    from optable_amd.engine import get_engine

    table, batch = _setup(30_000)
    K = 50
    full = table.trace_batch(batch, max_segments=K, layout="append")
    assert get_engine().last_launch()["pair_queue"] & 16
    records = int(full.count.abs().sum().item())
    small = table.trace_batch(batch, max_segments=K, layout="append", capacity=(records // 3) // 64 * 64)
    with pytest.raises(RuntimeError, match="capacity >= "):
        _ = small.n_valid
    assert torch.equal(small.count, full.count)
    fits = table.trace_batch(batch, max_segments=K, layout="append", capacity=get_engine().append_capacity(records))
    assert fits.n_valid >= records


def test_block_pool_stress_random_sizes_caps_and_chunks():
    """The pool's lock-free scheduling under sixty launches of random batch size, segment cap and append chunk size (what
    tools/pool_stress.py runs by the hundred): every launch holds exactly the records of the per-wave lists, in the
    reference's order."""
    import optable_amd as oa
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine

    rng = np.random.default_rng(20260405)
    table = oa.OpticalTable()
    table.add_components(scenes.cfg5_components(oa))
    eng = get_engine()
    eng.upload(table.compile())
    q = 1j * np.pi * scenes.W0**2 / scenes.WL
    try:
        for it in range(60):
            n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 30_000, 100_003]))
            K = int(rng.integers(2, 51))
            chunk = int(rng.choice([64, 128, 512, 2048]))
            o, d = scenes.cfg5_rays(n, int(rng.integers(0, 1000)))
            batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=q, precision="f32", device="cuda")
            eng.set_option(abi.OPT_BLOCK_POOL, 0)
            a = eng.trace(batch, K).to_host(reference_order=True)
            eng.set_option(abi.OPT_BLOCK_POOL, -1)
            eng.set_option(abi.OPT_APPEND_CHUNK, chunk)
            out = eng.trace(batch, K, layout="append")
            assert eng.last_launch()["pair_queue"] & 16
            b = out.to_host(reference_order=True)
            for f in abi.SEG_FIELDS + ("ray", "surface"):
                np.testing.assert_array_equal(a[f], b[f], err_msg=f"launch {it}: n={n} K={K} chunk={chunk}: {f}")
    finally:
        eng.set_option(abi.OPT_BLOCK_POOL, -1)
        eng.set_option(abi.OPT_APPEND_CHUNK, 512)


@pytest.mark.parametrize("period", [1000, 37])
def test_block_pool_protocol_under_delayed_publications(period):
    """The cross-wave protocol of the pool rests on ORDER (one wave's LDS instructions execute in program order on the CU's one
    LDS pipeline), never on timing: with one in `period` publications of a state or control word held back ~8000 cycles after
    the records it announces (OT_OPT_POOL_JITTER) the records must not change — nor the slot order contract."""
    from optable_amd.engine import get_engine

    table, batch = _setup(150_001)
    K = 30
    eng = get_engine()
    try:
        eng.set_option(abi.OPT_BLOCK_POOL, 0)
        lists = table.trace_batch(batch, max_segments=K, layout="slots").to_host(reference_order=True)
        eng.set_option(abi.OPT_BLOCK_POOL, -1)
        eng.set_option(abi.OPT_APPEND_CHUNK, 64)      # a workgroup opens a new chunk every 16 passes: the claim path is busy too
        eng.set_option(abi.OPT_POOL_JITTER, period)
        app = table.trace_batch(batch, max_segments=K, layout="append")
        assert eng.last_launch()["pair_queue"] & 16
        got = app.to_host(reference_order=True)
    finally:
        eng.set_option(abi.OPT_POOL_JITTER, 0)
        eng.set_option(abi.OPT_APPEND_CHUNK, 512)
        eng.set_option(abi.OPT_BLOCK_POOL, -1)
    for f in abi.SEG_FIELDS + ("ray", "surface", "count"):
        np.testing.assert_array_equal(lists[f], got[f], err_msg=f)

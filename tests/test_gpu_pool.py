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
        lists = table.trace_batch(batch, max_segments=K)
        assert not eng.last_launch()["pair_queue"] & 8
        eng.set_option(abi.OPT_BLOCK_POOL, 1)  # (the slots take the pool only when asked to: scattered stores)
        pool = table.trace_batch(batch, max_segments=K)
        info = eng.last_launch()
        assert info["kernel"] == 2 and info["pair_queue"] & 8 and info["pair_queue"] & 2, info
        app = table.trace_batch(batch, max_segments=K, layout="append")
        assert eng.last_launch()["pair_queue"] & 8 and eng.last_launch()["pair_queue"] & 4
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
    assert get_engine().last_launch()["pair_queue"] & 8
    ref = oracle.trace(table.compile(), b64.to_host(), max_trace_num=K)
    got = s32.to_host(reference_order=True)
    np.testing.assert_array_equal(got["ray"], ref["ray"])
    same = got["surface"] == ref["surface"]
    assert same.mean() > 0.999
    np.testing.assert_allclose(got["ox"][same], ref["ox"][same], atol=2e-3)

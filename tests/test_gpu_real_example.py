"""GPU (MI355X): the reference's largest example as the reference itself ran it (fixture g27, tools/make_golden.py
real_example_fixture: examples/ripa_gen2_lensless.py — a multi-pass cavity of micro-mirror arrays, 7,689 leaf surfaces in
nested groups, count-limited prism faces, ONE Gaussian ray reflected about three thousand times; 16.8 s in the reference).
The reference's objects do not travel to the GPU box; the scene does, as the tables this package's compiler made of them."""
import time

import numpy as np
import pytest

import helpers
from optable_amd import abi

pytestmark = pytest.mark.gpu


def test_the_reference_s_largest_example_matches_segment_by_segment():
    import torch
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine

    gold = helpers.golden("g27_real_example")
    scene = helpers.stored_scene(gold)
    host = helpers.fixture_rays_host(gold)
    eng = get_engine()
    cap = int(gold["max_trace_num"][0])
    q = host["q_re"] + 1j * host["q_im"]
    o = np.stack([host["ox"], host["oy"], host["oz"]], 1)
    d = np.stack([host["dx"], host["dy"], host["dz"]], 1)

    def run():
        batch = RayBatch.from_arrays(o, d, wavelength=host["wavelength"], intensity=host["intensity"], q=q, n_index=host["n"],
                                     pathlength=host["pathlength"], ids=host["id"], device=eng.device, normalize=False)
        batch.flags.copy_(torch.from_numpy(host["flags"]))
        counts = torch.zeros((len(scene.limited), 1), dtype=torch.int32, device=eng.device)
        with eng.lock:
            eng.upload(scene)
            segs = eng.trace_branching(batch, cap, counts=counts, distinct_ids=True)
            got = segs.to_host(reference_order=True)
        return got, counts.cpu().numpy()

    got, counts = run()
    assert len(got["ray"]) == len(gold["seg_tree"]) == 3202
    helpers.assert_segments_match(got, gold, gold["in_has_q"])
    np.testing.assert_array_equal(counts[:, 0], gold["counts"])
    eng.timing(True)
    t0 = time.perf_counter()
    run()
    seconds = time.perf_counter() - t0
    kernel_ms, launches = eng.timing_read()
    eng.timing(False)
    print(f"g27: 3202 segments through 7689 leaves in {seconds * 1e3:.1f} ms, {kernel_ms:.1f} ms of it in {launches} launch(es) "
          f"(reference {float(gold['reference_seconds'][0]):.1f} s); launch: {eng.last_launch()}")
    assert seconds < 0.1 * float(gold["reference_seconds"][0])

"""GPU (MI355X): every script under the reference's examples/ that traces, as it stands, at its first `OpticalTable.ray_tracing`
call (tests/golden/examples/, tools/make_golden.py all_examples_fixture: the scene as this package's compiler flattens the
reference's objects, the input rays, every output segment, the interact counts before and after).  The reference's objects do not
travel to the GPU box; tables, rays and segments do.  The largest example is fixture g27 (tests/test_gpu_real_example.py)."""
import numpy as np
import pytest

import helpers
from optable_amd import abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", helpers.example_names())
def test_every_example_of_the_reference_matches_segment_by_segment(name):
    import torch
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine

    gold = dict(np.load(helpers.os.path.join(helpers.EXAMPLES, name + ".npz")))
    scene = helpers.stored_scene(gold)
    host = helpers.example_rays_host(gold)
    eng = get_engine()
    cap = int(gold["max_trace_num"][0])
    n = len(host["id"])
    n_classes = int(host["id"].max()) + 1
    counts = torch.from_numpy(np.ascontiguousarray(gold["counts_before"])).to(eng.device) if len(scene.limited) else None
    # rays that share an id share interact counters and see each other's updates in input order (optical_component.py:140-149):
    # round r = the r-th ray of every id, as table.py does
    rounds = np.zeros(n, dtype=np.int64)
    if len(scene.limited) and n_classes < n:
        seen = {}
        for k, c in enumerate(host["id"]):
            rounds[k] = seen.get(int(c), 0)
            seen[int(c)] = rounds[k] + 1
    parts = []
    with eng.lock:
        eng.upload(scene)
        for rnd in range(int(rounds.max()) + 1):
            pick = np.nonzero(rounds == rnd)[0]
            q = host["q_re"][pick] + 1j * host["q_im"][pick]
            o = np.stack([host["ox"], host["oy"], host["oz"]], 1)[pick]
            d = np.stack([host["dx"], host["dy"], host["dz"]], 1)[pick]
            batch = RayBatch.from_arrays(o, d, wavelength=host["wavelength"][pick], intensity=host["intensity"][pick], q=q,
                                         n_index=host["n"][pick], pathlength=host["pathlength"][pick], ids=host["id"][pick],
                                         device=eng.device, normalize=False)
            batch.flags.copy_(torch.from_numpy(host["flags"][pick]))
            segs = eng.trace_branching(batch, cap, counts=counts, distinct_ids=True)
            part = segs.to_host(reference_order=True)
            part["ray"] = pick[part["ray"]].astype(np.int32)
            parts.append(part)
    got = {k: np.concatenate([p[k] for p in parts]) for k in parts[0] if k != "count"}
    order = np.argsort(got["ray"], kind="stable")
    got = {k: v[order] for k, v in got.items()}
    helpers.assert_segments_match(got, gold, gold["in_has_q"])
    if counts is not None:
        np.testing.assert_array_equal(counts.cpu().numpy(), gold["counts"])

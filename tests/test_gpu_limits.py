"""Limits and guards around the trace (round-2 correctness debts):
  * `perfomance_limit["max_trace_time"]` (optical_table.py:84-97) cuts a branching trace between generations;
  * count tables: ids outside the table and ids shared inside one launch never corrupt memory or a counter
    (optical_component.py:140-149 keeps one counter per ray id);
  * Sellmeier materials with fewer than three terms (material.py:106-120 sums only the terms given)."""
import numpy as np
import pytest
import torch

import optable_amd as oa
import scenes
from optable_amd.batch import RayBatch
from optable_amd.engine import get_engine
from optable_amd.materials import SellmeierMaterial

pytestmark = pytest.mark.gpu


def _cavity_table():
    sc = scenes.g05_cavity(oa)
    table = oa.OpticalTable()
    table.add_components(sc["components"])
    return table, sc


def test_max_trace_time_cuts_between_generations(capsys):
    table, sc = _cavity_table()
    full = table.ray_tracing(sc["rays"], perfomance_limit={"max_trace_num": 300})
    assert len(full) > 10  # the ray circulates a few round trips before it walks out of the cavity
    table2, sc2 = _cavity_table()
    cut = table2.ray_tracing(sc2["rays"], perfomance_limit={"max_trace_num": 300, "max_trace_time": 1e-9})
    out = capsys.readouterr().out
    assert 1 <= len(cut) < 10  # the clock had run out after the first generation: only it was archived
    assert "exceeds the maximum tracing time" in out
    # what was traced before the cut is the same prefix of the same tree
    for a, b in zip(cut, full):
        np.testing.assert_allclose(a.origin, b.origin, atol=1e-12)
    # engine level: the cut is reported per tree
    eng = get_engine()
    scene = table2.compile()
    eng.upload(scene)
    batch = RayBatch.from_arrays([[2, 0, 0]], [[1, 0, 0]])
    segs = eng.trace_tree(batch, 300, max_trace_time=1e-9)
    assert segs.timed_out and bool(segs.capped[0])
    segs = eng.trace_tree(batch, 300, max_trace_time=600.0)
    assert not segs.timed_out and segs.n_valid == len(full)


def _limited_scene():
    far = oa.Mirror([4, 0, 0], radius=1, max_interact_count=3)   # leaf 0: reflects three times per ray id, then transparent
    back = oa.Mirror([8, 0, 0], radius=1)                        # leaf 1
    table = oa.OpticalTable()
    table.add_components([far, back])
    return table


@pytest.mark.parametrize("path", ["fused", "tree"])
def test_ids_outside_the_count_table_are_not_counted(path):
    """Direct engine callers pass their own ids: the table has `n_classes` columns and an id beyond them must not be
    used as an index anywhere (k_gen_counts wrote out of range in round 1; count_gate already guarded)."""
    table = _limited_scene()
    eng = get_engine()
    scene = table.compile()
    eng.upload(scene)
    n = 512
    o = np.tile([0.0, 0.0, 0.0], (n, 1)) + np.linspace(-0.5, 0.5, n)[:, None] * np.array([0, 1, 0])
    batch = RayBatch.from_arrays(o, np.tile([1.0, 0, 0], (n, 1)))
    batch.id = torch.full((n,), 2_000_000_000, dtype=torch.int32, device=batch.device)
    batch.id[::2] = -7
    guard = torch.zeros((1, 8), dtype=torch.int32, device=batch.device)  # n_classes = 8: every id is outside
    sentinel = guard.clone()
    segs = eng.trace(batch, 12, counts=guard) if path == "fused" else eng.trace_tree(batch, 12, counts=guard)
    torch.cuda.synchronize()
    assert torch.equal(guard, sentinel)  # nothing counted, nothing written
    host = segs.to_host(reference_order=True)
    # an uncounted ray always finds the limited mirror open: origin -> far (reflected) -> escape, for every ray
    per_ray = np.bincount(host["ray"], minlength=n)
    assert per_ray.min() == per_ray.max() == 2
    assert int(np.sum(host["surface"] == 0)) == n


@pytest.mark.parametrize("path", ["fused", "tree"])
def test_shared_ids_in_one_launch_never_overrun_a_counter(path):
    """All rays of a launch in ONE class (a caller that did not split them into rounds, as `table.py` does).  One-launch
    kernel: the gate is an atomic increment-below-cap, so exactly `max_interact_count` rays are reflected (which ones is
    unspecified) and the counter stops at the cap.  Generation kernels gate per TREE (FIFO inside a tree): rays of
    different trees that share an id all see the counter as it stood before the generation, so more may pass — but the
    table still ends at the cap, never beyond it (include/optable_hip.h states the precondition)."""
    table = _limited_scene()
    eng = get_engine()
    scene = table.compile()
    eng.upload(scene)
    n = 4096
    o = np.tile([0.0, 0.0, 0.0], (n, 1)) + np.linspace(-0.5, 0.5, n)[:, None] * np.array([0, 1, 0])
    batch = RayBatch.from_arrays(o, np.tile([1.0, 0, 0], (n, 1)))
    batch.id = torch.zeros(n, dtype=torch.int32, device=batch.device)
    counts = torch.zeros((1, 1), dtype=torch.int32, device=batch.device)
    segs = eng.trace(batch, 12, counts=counts) if path == "fused" else eng.trace_tree(batch, 12, counts=counts)
    host = segs.to_host(reference_order=True)
    reflected_at_far = int(np.sum(host["surface"] == 0))
    if path == "fused":
        assert int(counts[0, 0]) == 3 and reflected_at_far == 3
    else:
        assert 1 <= int(counts[0, 0]) <= 3 and reflected_at_far >= 3


def test_two_term_sellmeier_at_one_micron():
    """A padded third term must not poison the common denominator at lambda^2 == C (0/0 in round 1 with C = 1)."""
    glass = SellmeierMaterial("two-term", [1.03961212, 0.231792344], [0.00600069867, 0.0200179144])
    kind, B, C = glass.device_spec()
    assert kind == "sellmeier" and B[2] == 0.0 and C[2] < 0.0
    slab = oa.GlassSlab([0, 0, 0], width=2, height=2, thickness=0.5, n1=oa.Vacuum(), n2=glass, reflectivity=0)
    table = oa.OpticalTable()  # unit 1e-2 m: wavelength 1e-4 model units == exactly 1 um
    table.add_components([slab])
    ray = oa.Ray([-3, 0.1, 0], [1, 0.05, 0], wavelength=1e-4, w0=61e-4)
    out = table.ray_tracing([ray])
    inside = [r for r in out if abs(r.n - 1.0) > 1e-6]
    assert inside and all(np.isfinite(r.n) for r in out)
    assert inside[0].n == pytest.approx(glass.sellmeier_n(1e-6), rel=1e-12)
    for prec in ("f64", "f32"):
        b = RayBatch.from_arrays([[-3, 0.1, 0]], [[1, 0.05, 0]], wavelength=1e-4, precision=prec)
        segs = table.trace_batch(b, max_segments=4).to_host(reference_order=True)
        assert np.all(np.isfinite(segs["n"])) and len(segs["n"]) == 3


def test_two_pass_generations_agree_with_themselves():
    """A generation is traced twice (count, then emit: kernels.h k_gen_pass); both passes must take the same decisions
    for every ray.  2e6 trees of the branching cfg 4 variant (every hit splits, TIR inside the slab) and the cavity."""
    from optable_amd import workloads as W

    from optable_amd import abi

    eng = get_engine()
    eng.set_option(abi.OPT_GEN_ONEPASS, 0)  # (the two-pass kernels: the one-pass kernel has no second pass to disagree with)
    before = eng.generation_mismatches()
    table = oa.OpticalTable()
    table.add_components(W.cfg4_components(oa, reflectivity=0.2))
    o, d, wl = W.cfg4_rays(30_000, 4)
    batch = RayBatch.from_arrays(o, d, wavelength=wl, q=1j * np.pi * W.W0**2 / wl)
    scene = table.compile()
    eng.upload(scene)
    segs = eng.trace_tree(batch, 12, out_capacity=batch.n * 13)
    assert segs.n_valid == 12 * batch.n
    for prec in ("f64", "f32"):
        t2, sc = _cavity_table()
        b = RayBatch.from_arrays(np.tile([2.0, 0, 0], (4096, 1)) + np.linspace(0, 1e-3, 4096)[:, None] * np.array([0, 1, 0]),
                                 np.tile([1.0, 0, 0], (4096, 1)), precision=prec)
        eng.upload(t2.compile())
        eng.trace_tree(b, 300)
    eng.set_option(abi.OPT_GEN_ONEPASS, -1)
    assert eng.generation_mismatches() == before


@pytest.mark.parametrize("cap", [3, 7, 12, 40])
def test_trees_capped_in_a_bushy_generation_drop_their_children(cap, oracle):
    """A Mach-Zehnder-like lattice of beam splitters and mirrors: every splitter doubles the rays, so the generation in
    which `max_trace_num` runs out is the tree's largest.  The kernels do not emit the children of that generation
    (they could never be processed: optical_table.py:138-144; OT_OPT_GEN_DROP_DOOMED) — the segments must be exactly the
    oracle's and exactly what the trace gives with the children emitted and dropped a generation later."""
    from optable_amd import abi

    comps = []
    for k in range(5):
        comps.append(oa.BeamSplitter([2.0 * (k + 1), 0, 0], width=6, height=2, eta=0.5).RotZ(np.pi / 4))
        comps.append(oa.Mirror([2.0 * (k + 1), 3.0 + 0.1 * k, 0], radius=2).RotZ(-np.pi / 2))
        comps.append(oa.BeamSplitter([2.0 * (k + 1) + 1.0, 1.5, 0], width=6, height=2, eta=0.3).RotZ(-np.pi / 4))
    table = oa.OpticalTable()
    table.add_components(comps)
    scene = table.compile()
    n = 3000
    rng = np.random.default_rng(5)
    o = np.stack([np.zeros(n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.2, 0.2, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.02, 0.02, n), rng.uniform(-0.01, 0.01, n)], 1)
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j * np.pi * scenes.W0**2 / scenes.WL)
    eng = get_engine()
    eng.upload(scene)
    try:
        eng.set_option(abi.OPT_GEN_DROP_DOOMED, 0)
        kept = eng.trace_tree(batch, cap)
        eng.set_option(abi.OPT_GEN_DROP_DOOMED, 1)
        dropped = eng.trace_tree(batch, cap)
    finally:
        eng.set_option(abi.OPT_GEN_DROP_DOOMED, 1)
    a, b = kept.to_host(reference_order=True), dropped.to_host(reference_order=True)
    ref = oracle.trace(scene, batch.to_host(), max_trace_num=cap)
    assert len(b["ray"]) == len(ref["ray"]) and np.bincount(b["ray"], minlength=n).max() == min(cap, np.bincount(ref["ray"]).max())
    assert torch.equal(kept.capped, dropped.capped)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    np.testing.assert_array_equal(b["ray"], ref["ray"])
    np.testing.assert_array_equal(b["surface"], ref["surface"])
    np.testing.assert_allclose(b["ox"], ref["ox"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(b["intensity"], ref["intensity"], rtol=1e-9, atol=1e-12)


def test_a_long_generation_loop_reports_its_progress(capsys):
    """optical_table.py:99-111: a trace that runs longer than MIN_HINTING_TIME prints its progress once per interval.  The
    generation loop comes back from the library at that interval (Engine.HINT_SECONDS; shortened here), says where it is and
    goes on: the same records as the uninterrupted trace."""
    import numpy as np
    import torch

    import optable_amd as oa
    from optable_amd import abi
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine

    L, D, R = 10 * 4 / 3, 4, 0.9
    comps = [oa.Mirror([0, 0, 0], radius=D, reflectivity=R, transmission=1 - R).RotZ(-np.pi / 4),
             oa.Mirror([L, 0, 0], radius=D, reflectivity=R, transmission=1 - R).RotZ(+np.pi / 4 + 0.02),
             oa.Mirror([L, -L, 0], radius=D, reflectivity=R, transmission=1 - R).RotZ(-np.pi / 4 + 0.02),
             oa.Mirror([0, -L, 0], radius=D, reflectivity=R, transmission=1 - R).RotZ(+np.pi / 4)]
    table = oa.OpticalTable()
    table.add_components(comps)
    n = 2048
    o = np.tile([2.0, 0, 0], (n, 1)) + np.linspace(0, 1e-3, n)[:, None] * np.array([0, 1, 0])
    batch = RayBatch.from_arrays(o, np.tile([1.0, 0, 0], (n, 1)))
    eng = get_engine()
    eng.upload(table.compile())
    whole = eng.trace_tree(batch, 300)
    assert "Tracing..." not in capsys.readouterr().out
    try:
        eng.HINT_SECONDS = 0.0  # (the library comes back after every generation)
        hinted = eng.trace_tree(batch, 300)
    finally:
        del eng.HINT_SECONDS  # back to the class default
    said = capsys.readouterr().out
    assert said.count("Tracing... Time elapsed:") >= 2 and "Alive rays:" in said
    assert hinted.n_valid == whole.n_valid and not hinted.timed_out
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        assert torch.equal(hinted.field(f)[: hinted.n_valid], whole.field(f)[: whole.n_valid]), f

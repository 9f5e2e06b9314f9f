"""GPU (MI355X): k_trace_trees — whole ray trees in one launch, a lane per tree with its FIFO queue in LDS — against the
generation kernels (ot_trace_tree_*): the same segments in the same (reference) order, bit for bit, the same trees capped;
against the oracle; queues that are too small report their trees instead of dropping rays."""
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


def _lattice(k_max=3):
    comps = []
    for k in range(k_max):
        comps.append(oa.BeamSplitter([2.0 * (k + 1), 0, 0], width=6, height=2, eta=0.5).RotZ(np.pi / 4))
        comps.append(oa.Mirror([2.0 * (k + 1), 3.0 + 0.1 * k, 0], radius=2).RotZ(-np.pi / 2))
        comps.append(oa.BeamSplitter([2.0 * (k + 1) + 1.0, 1.5, 0], width=6, height=2, eta=0.3).RotZ(-np.pi / 4))
    t = oa.OpticalTable()
    t.add_components(comps)
    return t.compile()


def _lattice_rays(n, seed, precision="f64"):
    rng = np.random.default_rng(seed)
    o = np.stack([np.zeros(n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.2, 0.2, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.02, 0.02, n), rng.uniform(-0.01, 0.01, n)], 1)
    return RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=Q, precision=precision)


def _same_as_generations(scene, batch, cap):
    eng = get_engine()
    eng.upload(scene)
    plan = eng.trees_plan(batch.precision, cap)
    assert plan["kernel"] and plan["full"], plan
    trees = eng.trace_trees(batch, cap, layout="slots" if plan["slots"] else "append")  # (the [k][tree] slots: the planar preset only)
    gens = eng.trace_tree(batch, cap)
    assert int(trees.count.min()) >= 1
    assert int(trees.count.sum()) == gens.n_valid
    assert torch.equal(trees.capped, gens.capped)
    a, b = trees.to_host(reference_order=True), gens.to_host(reference_order=True)
    np.testing.assert_array_equal(a["ray"], b["ray"])
    np.testing.assert_array_equal(a["surface"], b["surface"])
    for f in abi.SEG_FIELDS:
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    return trees, a


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_trees_equal_generations_on_cfg4_with_reflectivity(precision, oracle):
    table = oa.OpticalTable()
    table.add_components(W.cfg4_components(oa, reflectivity=0.2))
    scene = table.compile()
    o, d, wl = W.cfg4_rays(5_000, 4)  # x 64 wavelengths = 3.2e5 trees
    batch = RayBatch.from_arrays(o, d, wavelength=wl, q=1j * np.pi * W.W0**2 / wl, precision=precision)
    trees, _ = _same_as_generations(scene, batch, 12)
    assert bool(trees.capped.all())
    if precision == "f64":
        small = batch.slice(0, 400)
        got = get_engine().trace_branching(small, 12).to_host(reference_order=True)
        ref = oracle.trace(scene, small.to_host(), max_trace_num=12)
        np.testing.assert_array_equal(got["ray"], ref["ray"])
        np.testing.assert_array_equal(got["surface"], ref["surface"])
        for f in abi.SEG_FIELDS:
            np.testing.assert_allclose(got[f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)


@pytest.mark.parametrize("cap", [1, 2, 3, 7, 12])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_trees_equal_generations_on_bushy_trees(cap, precision):
    """Beam-splitter lattice: trees that double every generation and are cut by the cap in their widest one — the queue bound
    ceil(cap / 2) and the rule that drops children no budget is left for."""
    _same_as_generations(_lattice(), _lattice_rays(3000, 5, precision), cap)


@pytest.mark.parametrize("lds_entries", [1, 2, 3, 8])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_deep_queues_spill_into_the_scratch_ring_and_keep_their_order(lds_entries, precision):
    """Bushy trees under a cap of 48: queues of up to 24 rays per lane, of which 1-8 live in LDS and the rest in the global
    scratch ring — FIFO order across the two rings, whatever the split."""
    eng = get_engine()
    try:
        eng.set_option(abi.OPT_TREES_LDS_ENTRIES, lds_entries)
        _same_as_generations(_lattice(), _lattice_rays(20_000, 11, precision), 48)
        plan = eng.trees_plan(precision, 48)
        assert plan["lds_entries"] == min(lds_entries, 6 if precision == "f64" else 13) and plan["queue"] == plan["lds_entries"] + 24
    finally:
        eng.set_option(abi.OPT_TREES_LDS_ENTRIES, 0)


def test_trees_whose_rays_all_escape_or_die():
    """Trees of one ray (a miss), of dead input rays, and a batch that is not a multiple of the wave."""
    scene = _lattice()
    batch = _lattice_rays(1001, 7)
    batch.dz.fill_(0.9)  # most rays leave the table at once
    batch.flags[::5] |= abi.RAY_DEAD
    _same_as_generations(scene, batch, 6)


def test_a_queue_that_is_too_small_reports_its_trees():
    """A cap whose queue bound does not fit the CU's LDS: the plan says so, small trees still come out right, and a tree
    that overflows its queue is reported (negative count), never silently truncated."""
    scene = _lattice()
    eng = get_engine()
    eng.upload(scene)
    batch = _lattice_rays(2000, 8)
    plan = eng.trees_plan("f64", 2000)
    assert plan["kernel"] and not plan["full"] and plan["queue"] < 1000
    trees = eng.trace_trees(batch, 2000)
    gens = eng.trace_tree(batch, 2000)
    count = trees.count.cpu().numpy()
    per_tree = np.bincount(gens.field("ray")[: gens.n_valid].cpu().numpy(), minlength=batch.n)
    ok = count > 0
    np.testing.assert_array_equal(count[ok], per_tree[ok])
    assert np.all(-count[~ok] <= per_tree[~ok])  # an overflowed tree stopped early, and says how far it got


def _everyday(dove):
    comps = [oa.BeamSplitter([3.0, 0, 0], width=6, height=3, eta=0.4).RotZ(np.pi / 4),
             oa.Mirror([3.0, 3.5, 0], radius=3).RotZ(-np.pi / 2),
             oa.Block([8, 0.4, 0], hole=oa.Circle(0.6), width=3, height=3),
             oa.BiConvexLens([11, 0.4, 0], CT=0.6, R1=12.0, R2=-12.0, diameter=3.0, n=1.5),
             oa.BeamSplitter([13.5, 0.4, 0], width=6, height=3, eta=0.7).RotZ(-np.pi / 4),
             oa.Mirror([16, 0.4, 0], radius=3.0).RotZ(np.pi + 0.05)]
    if dove:
        comps.insert(3, oa.DovePrism([9.5, 0.1, 0], L=1.2, D=0.5, Ng=1.5))
    t = oa.OpticalTable()
    t.add_components(comps)
    return t.compile()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("dove", [False, True], ids=["FE", "FM"])
def test_trees_of_everyday_parts(dove, precision, oracle):
    """Beam splitters around a holed block, a biconvex lens (and a dove prism's tilted polygon faces): the lane-per-tree
    kernels of the presets FE / FM against the generation kernels, bit for bit, and against the oracle."""
    scene = _everyday(dove)
    rng = np.random.default_rng(12)
    n = 4000
    o = np.stack([np.zeros(n), rng.uniform(-0.5, 0.5, n), rng.uniform(-0.4, 0.4, n)], 1)
    d = np.stack([np.ones(n), rng.uniform(-0.03, 0.03, n), rng.uniform(-0.02, 0.02, n)], 1)
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=Q, precision=precision)
    _, got = _same_as_generations(scene, batch, 16)
    assert len(got["ray"]) > 3 * n
    if precision == "f64":
        ref = oracle.trace(scene, batch.to_host(), max_trace_num=16)
        np.testing.assert_array_equal(got["ray"], ref["ray"])
        np.testing.assert_array_equal(got["surface"], ref["surface"])
        for f in abi.SEG_FIELDS:
            np.testing.assert_allclose(got[f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)


def test_trees_under_grids_in_single_precision():
    """cfg 3 with 10 % reflecting slab faces: 32 components under a top-level grid (the planar preset with grid walks, fp32)."""
    table = oa.OpticalTable()
    table.add_components(W.cfg3_components(oa, slab_reflectivity=0.1))
    o, d = W.cfg3_rays(100_000, 2)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q, precision="f32")
    _same_as_generations(table.compile(), batch, 20)


@pytest.mark.parametrize("dove", [False, True], ids=["FE", "FM"])
def test_trees_through_count_limited_faces_match_the_oracle(dove, oracle):
    """A prism whose entrance face splits every ray and whose faces are count-limited (max_interact_count): a tree's rays meet
    the gate one after the other in FIFO order — the reference's order — so the lane-per-tree kernel needs no probe pass;
    through trace_batch (one column of the counts table per ray), against the generation kernels and the oracle."""
    import test_gpu_presets as P

    table = oa.OpticalTable()
    table.add_components(P._parts(dove, split=0.3))
    scene = table.compile()
    assert scene.max_children == 2 and len(scene.limited) == 2
    batch = P._rays(1500, "f64")
    eng = get_engine()
    segs = table.trace_batch(batch, max_segments=14, scene=scene)
    assert segs.count is not None and eng.last_launch()["kernel"] == 4  # the lane-per-tree kernel took it
    got = segs.to_host(reference_order=True)
    ref = oracle.trace(scene, batch.to_host(), max_trace_num=14)
    assert len(ref["ray"]) > 3 * batch.n
    np.testing.assert_array_equal(got["ray"], ref["ray"])
    np.testing.assert_array_equal(got["surface"], ref["surface"])
    for f in abi.SEG_FIELDS:
        np.testing.assert_allclose(got[f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)
    eng.upload(scene)
    gens = eng.trace_tree(batch, 14).to_host(reference_order=True)  # the generation kernels (probe pass + per-slot scans)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(got[f], gens[f], err_msg=f)
    # the counters the trace leaves behind are the generation path's
    table2 = eng.trace_tree(batch, 14).counts_table
    assert torch.equal(segs.counts_table, table2)


def test_rays_sharing_an_id_take_their_counts_in_input_order():
    """Copies of a ray under one id (multiplexed_in_wavelength) share the counters of a limited face: the host API traces them
    in successive rounds, each round one lane-per-tree launch over the same table."""
    import test_gpu_presets as P

    table = oa.OpticalTable()
    table.add_components(P._parts(False, split=0.3))
    scene = table.compile()
    base = P._rays(300, "f64")
    batch = base.multiplexed_in_wavelength(np.array([scenes.WL, 0.9 * scenes.WL, 1.1 * scenes.WL]))
    eng = get_engine()
    segs = table.trace_batch(batch, max_segments=10, scene=scene)
    try:
        eng.set_option(abi.OPT_TREES_LDS_ENTRIES, 1)  # (another split of the queues: same records)
        again = table.trace_batch(batch, max_segments=10, scene=scene)
    finally:
        eng.set_option(abi.OPT_TREES_LDS_ENTRIES, 0)
    a, b = segs.to_host(reference_order=True), again.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    assert torch.equal(segs.counts_table, again.counts_table)


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("away", [0.0, 0.9])
def test_append_layout_of_the_tree_kernel_holds_the_same_records(precision, away):
    """ot_trace_trees_append_*: the records of a step go to consecutive slots of the wave's chunk whatever trees the lanes are on
    (a batch in which nine trees of ten are a single ray: lanes refill from their wave's share all the time) — the same
    records as the [k][tree] slots, in the reference's order after the stable sort by tree."""
    scene = _lattice()
    batch = _lattice_rays(30_000, 13, precision)
    if away:
        gone = torch.rand(batch.n, device=batch.device) < away
        batch.dx[gone] = -1.0
        batch.dy[gone] = 0.0
        batch.dz[gone] = 0.0
    eng = get_engine()
    eng.upload(scene)
    slots = eng.trace_trees(batch, 24)
    dense = eng.trace_trees(batch, 24, layout="append")
    assert dense.layout == "append" and eng.last_launch()["kernel"] == 4
    records = int(slots.count.sum())
    assert torch.equal(dense.count, slots.count) and torch.equal(dense.capped, slots.capped)
    assert records <= dense.n_valid <= records + 512 * 4096  # holes: the tail of every wave's last chunk at most
    a, b = dense.to_host(reference_order=True), slots.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    small = eng.trace_trees(batch, 24, layout="append", capacity=4096)  # a block that is too small loses records, not memory
    with pytest.raises(RuntimeError, match="capacity"):
        small.n_valid


def test_default_call_sizes_the_dense_list_from_a_sample():
    """Engine.trace_branching on a large batch: rays per tree from a 1 % sample; the append block is that x 1.15 + the launch's
    slack, not every tree at its cap — also for few long trees under a large cap."""
    scene = _lattice()
    eng = get_engine()
    eng.upload(scene)
    batch = _lattice_rays(400_000, 17, "f32")
    gone = torch.rand(batch.n, device=batch.device) < 0.9
    batch.dx[gone] = -1.0
    batch.dy[gone] = 0.0
    batch.dz[gone] = 0.0
    segs = eng.trace_branching(batch, 24)
    assert segs.layout == "append" and eng.last_launch()["kernel"] == 4
    records = int(segs.count.sum())
    assert records < 0.2 * batch.n * 24  # (most trees are one ray)
    assert segs.capacity <= 1.3 * records + 512 * 8192 + 64
    ref = eng.trace_tree(batch, 24)
    assert records == ref.n_valid
    a, b = segs.to_host(reference_order=True), ref.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    big = eng.trace_branching(batch, 96)  # few long trees under a large cap: queues of 48 rays per lane, most of them in the scratch ring
    assert big.layout == "append" and int(big.count.sum()) == eng.trace_tree(batch, 96).n_valid


def test_monitors_and_exports_read_tree_outputs_like_the_generation_list(tmp_path):
    """Monitor.record and the CSV export over the outputs of the lane-per-tree kernel ([k][tree] slots, the dense list) equal
    those over the generation kernels' list."""
    table = oa.OpticalTable()
    table.add_components(W.cfg4_components(oa, reflectivity=0.2))
    scene = table.compile()
    mon = oa.Monitor(origin=[-1.0, 0, 0], width=40, height=40)  # in front of the slab: the input rays and every reflection cross it
    o, d, wl = W.cfg4_rays(300, 4)
    batch = RayBatch.from_arrays(o, d, wavelength=wl, q=1j * np.pi * W.W0**2 / wl)
    eng = get_engine()
    eng.upload(scene)
    outs = {"list": eng.trace_tree(batch, 12), "slots": eng.trace_trees(batch, 12, layout="slots"), "append": eng.trace_trees(batch, 12, layout="append")}
    hits = {k: table.record_batch(mon, v) for k, v in outs.items()}
    assert len(hits["list"]) > 0
    for name in ("slots", "append"):
        assert len(hits[name]) == len(hits["list"])
        for acc in ("yList", "zList", "tYList", "IList", "tList"):
            np.testing.assert_array_equal(getattr(hits[name], acc)(None).cpu().numpy(), getattr(hits["list"], acc)(None).cpu().numpy(), err_msg=f"{name} {acc}")
    texts = {}
    for name, segs in outs.items():
        path = tmp_path / f"{name}.csv"
        table.export_batch_csv(segs, str(path), batch)
        texts[name] = path.read_text()
    assert texts["slots"] == texts["list"] and texts["append"] == texts["list"]


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_trees_in_a_scene_of_grids_and_curved_optics(precision, oracle):
    """cfg 5 (asphere, 16 x 16 micro-mirror array under a lattice grid: the all-features preset) behind a partially transmitting end
    mirror and a beam splitter: every scene has a lane-per-tree kernel — against the generation kernels bit for bit, and (double
    precision) a slice against the oracle."""
    comps = W.cfg5_components(oa)
    comps[0] = oa.Mirror([-1, 0, 0], radius=4, reflectivity=0.7, transmission=0.3)
    comps.append(oa.BeamSplitter([10.0, 0, 0], width=5, height=5, eta=0.5).RotZ(np.pi / 4))
    table = oa.OpticalTable()
    table.add_components(comps)
    scene = table.compile()
    assert scene.max_children == 2
    o, d = W.cfg5_rays(3000, 3)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q, precision=precision)
    _, got = _same_as_generations(scene, batch, 24)
    assert len(got["ray"]) > 4 * batch.n
    if precision == "f64":
        small = batch.slice(0, 200)
        mine = get_engine().trace_trees(small, 24, layout="append").to_host(reference_order=True)
        ref = oracle.trace(scene, small.to_host(), max_trace_num=24)
        seq = lambda x: [[int(s) for r, s in zip(x["ray"], x["surface"]) if r == i] for i in range(small.n)]
        same = np.array([a == b for a, b in zip(seq(mine), seq(ref))])
        assert (~same).mean() <= 0.01  # (a hit within an ulp of a micro-mirror's edge may fall to either side)
        keep_m, keep_r = same[mine["ray"]], same[ref["ray"]]
        for f in ("ox", "oy", "oz", "dx", "dy", "dz", "intensity", "pathlength"):
            np.testing.assert_allclose(mine[f][keep_m], ref[f][keep_r], rtol=1e-9, atol=1e-9, err_msg=f)


@pytest.mark.parametrize("refill_at", [1, 40, 64])
def test_records_do_not_depend_on_when_lanes_refill(refill_at):
    """OT_OPT_TREES_REFILL_AT: lanes that take their next tree one by one, in groups, or 64 to a wave at a time — the same records."""
    scene = _lattice()
    batch = _lattice_rays(20_000, 19)
    gone = torch.rand(batch.n, device=batch.device) < 0.6
    batch.dx[gone] = -1.0
    batch.dy[gone] = 0.0
    batch.dz[gone] = 0.0
    eng = get_engine()
    eng.upload(scene)
    ref = eng.trace_trees(batch, 24, layout="append").to_host(reference_order=True)
    try:
        eng.set_option(abi.OPT_TREES_REFILL_AT, refill_at)
        for layout in ("append", "slots"):
            got = eng.trace_trees(batch, 24, layout=layout).to_host(reference_order=True)
            for f in abi.SEG_FIELDS + ("ray", "surface"):
                np.testing.assert_array_equal(got[f], ref[f], err_msg=f"{layout} {f}")
    finally:
        eng.set_option(abi.OPT_TREES_REFILL_AT, 16)


def test_default_call_keeps_moderately_uneven_trees_in_step():
    """Engine.trace_branching reads the spread of the tree sizes off its 1 % sample: trees of 13-30 rays (cfg 4 with R = 0.2 under
    a cap that does not bind) are traced 64 to a wave; a batch in which nine trees of ten are a single ray refills lane by lane."""
    eng = get_engine()
    table = oa.OpticalTable()
    table.add_components(W.cfg4_components(oa, reflectivity=0.2))
    eng.upload(table.compile())
    o, d, wl = W.cfg4_rays(8_000, 4)  # x 64 wavelengths = 5.1e5 trees
    batch = RayBatch.from_arrays(o, d, wavelength=wl, q=1j * np.pi * W.W0**2 / wl)
    segs = eng.trace_branching(batch, 48)
    (rpr, refill_at), = eng._records_per_ray.values()
    assert segs.layout == "append" and 13 < rpr < 30 and refill_at == 64
    assert int(segs.count.min()) >= 8 and int(segs.count.max()) < 48
    eng.upload(_lattice())
    skew = _lattice_rays(400_000, 17, "f32")
    gone = torch.rand(skew.n, device=skew.device) < 0.9
    skew.dx[gone] = -1.0
    skew.dy[gone] = 0.0
    skew.dz[gone] = 0.0
    eng.trace_branching(skew, 24)
    (rpr, refill_at), = eng._records_per_ray.values()
    assert rpr < 6 and refill_at == 16


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_pair_queue_search_of_the_tree_kernel_finds_the_same_hits(precision):
    """Planar scenes under a top-level grid of leaves (cfg 3 with reflecting slabs): the lane-per-tree kernel searches through the
    wave-wide pair queue of the heavy non-branching kernel (OT_OPT_TREES_FLAT, flat_grid_hit) — the same records as with a grid
    walk per lane, and as the generation kernels."""
    table = oa.OpticalTable()
    table.add_components(W.cfg3_components(oa, slab_reflectivity=0.1))
    scene = table.compile()
    o, d = W.cfg3_rays(60_000, 2)
    batch = RayBatch.from_arrays(o, d, wavelength=W.WL, q=Q, precision=precision)
    eng = get_engine()
    trees, with_queue = _same_as_generations(scene, batch, 20)
    assert eng.last_launch is not None
    eng.upload(scene)
    queue_launch = None
    try:
        queue = eng.trace_trees(batch, 20, layout="append")
        queue_launch = eng.last_launch()
        eng.set_option(abi.OPT_TREES_FLAT, 0)
        walk = eng.trace_trees(batch, 20, layout="append")
        walk_launch = eng.last_launch()
    finally:
        eng.set_option(abi.OPT_TREES_FLAT, 1)
    assert queue_launch["pair_queue"] & 1 and not walk_launch["pair_queue"] & 1
    a, b = queue.to_host(reference_order=True), walk.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)

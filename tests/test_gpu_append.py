"""GPU (MI355X): the append layout of ot_trace_append_* (a dense list of segment records, include/optable_hip.h) and
the instanced lattice runs of the device image against the [k][ray] slots, the oracle and the un-accelerated scene.
Everything here is bit-exact: the layouts and the instancing change where records are kept, not what is computed."""
import numpy as np
import pytest
import torch

import scenes
from optable_amd import abi

pytestmark = pytest.mark.gpu


def _batch(o, d, precision="f64"):
    from optable_amd.batch import RayBatch

    q = 1j * np.pi * scenes.W0**2 / scenes.WL
    return RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=q, precision=precision, device="cuda")


def _table(components, accelerate=True):
    import optable_amd as oa

    t = oa.OpticalTable()
    t.accelerate = accelerate
    t.add_components(components)
    return t


CASES = {
    "cfg2": (scenes.cfg2_components, lambda n: scenes.cfg2_rays(n, 0), 20000, 5),
    "cfg3": (scenes.cfg3_components, lambda n: scenes.cfg3_rays(n, 2), 30000, 20),
    "cfg5": (scenes.cfg5_components, lambda n: scenes.cfg5_rays(n, 3), 6000, 50),
}


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_append_layout_equals_slots(case, precision):
    """Same records, bit for bit, in the reference's order; holes only at chunk tails; final states agree."""
    import optable_amd as oa
    from optable_amd import dist
    from optable_amd.engine import get_engine

    comps, gen, n, K = CASES[case]
    table = _table(comps(oa))
    batch = _batch(*gen(n), precision=precision)
    eng = get_engine()
    eng.set_option(abi.OPT_KERNEL, 2)  # the slots through the same (rolling-list) kernel family
    try:
        slots = table.trace_batch(batch, max_segments=K, layout="slots")
    finally:
        eng.set_option(abi.OPT_KERNEL, 0)
    app = table.trace_batch(batch, max_segments=K, layout="append")
    assert app.layout == "append" and eng.last_launch()["pair_queue"] & 4
    a, b = slots.to_host(reference_order=True), app.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface", "count"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    # slots in use = records + holes; holes are marked ray == -1 and number less than one chunk per wave
    used = app.n_valid
    ray = app.ray[:used].cpu().numpy()
    assert (ray >= 0).sum() == len(a["ray"]) == int(np.abs(a["count"]).sum())
    info = eng.last_launch()
    waves = info["workgroups"] * info["threads"] // 64
    assert (ray < 0).sum() < 512 * waves + len(ray) // 128 + 64  # (block pool: up to 63 more where a pass crosses into the workgroup's next chunk)
    # a stable sort by ray is the reference's order: within a ray the records lie at increasing slots in segment order
    keep = ray >= 0
    order = np.argsort(ray[keep], kind="stable")
    np.testing.assert_array_equal(app.length[:used].cpu().numpy()[keep][order], a["length"])
    np.testing.assert_array_equal(dist.final_state(app).cpu().numpy(), dist.final_state(slots).cpu().numpy())


def test_append_capacity_too_small_reports_the_size_that_fits():
    import optable_amd as oa

    table = _table(scenes.cfg3_components(oa))
    batch = _batch(*scenes.cfg3_rays(20000, 2), precision="f32")
    full = table.trace_batch(batch, max_segments=20, layout="append")
    records = int(np.abs(full.count.cpu().numpy()).sum())
    small = table.trace_batch(batch, max_segments=20, layout="append", capacity=records // 2)
    with pytest.raises(RuntimeError, match="capacity >= ") as err:  # (the slots a run claims vary with the holes: records + at most a chunk per wave)
        _ = small.n_valid
    assert int(str(err.value).rsplit(">= ", 1)[1]) >= records
    np.testing.assert_array_equal(small.count.cpu().numpy(), full.count.cpu().numpy())  # the trace itself is complete
    from optable_amd.engine import get_engine

    fits = table.trace_batch(batch, max_segments=20, layout="append", capacity=get_engine().append_capacity(records))
    assert fits.n_valid >= records


def test_append_monitor_and_export_follow_the_list_contract(tmp_path):
    """Monitor.record and the CSV export over an append-layout history equal those over the slots."""
    import optable_amd as oa

    comps = scenes.cfg2_components(oa)
    table = _table(comps)
    mon = oa.Monitor(origin=[7.5, 0, 0], width=6, height=6)
    batch = _batch(*scenes.cfg2_rays(5000, 0))
    slots = table.trace_batch(batch, max_segments=5, layout="slots")
    app = table.trace_batch(batch, max_segments=5, layout="append")
    h0, h1 = table.record_batch(mon, slots), table.record_batch(mon, app)
    assert len(h0) == len(h1) > 0
    for acc in ("yList", "zList", "tYList", "IList", "tList"):
        np.testing.assert_array_equal(getattr(h0, acc)(None).cpu().numpy(), getattr(h1, acc)(None).cpu().numpy(), err_msg=acc)
    a, b = tmp_path / "a.csv", tmp_path / "b.csv"
    table.export_batch_csv(slots, str(a), batch)
    table.export_batch_csv(app, str(b), batch)
    assert a.read_text() == b.read_text()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("family", ["mma", "mla", "dmd"])
def test_instanced_lattices_equal_the_plain_scene(family, precision, oracle):
    """MMA / MLA / DMD children folded into one record + a pose per member, found through the lattice's own raster:
    bit-identical to the same scene without instancing and without any grid, and (fp64) equal to the oracle."""
    import optable_amd as oa
    from optable_amd.engine import get_engine

    if family == "mma":
        comps = lambda: scenes.cfg5_components(oa)
        o, d = scenes.cfg5_rays(4000, 3)
        K = 50
    elif family == "mla":
        comps = lambda: [oa.MLA([6, 0, 0], N=(9, 7), pitch=0.4, focal_length=3.0, radius=0.2).RotZ(0.2), oa.Mirror([9, 0, 0], radius=3).RotZ(np.pi)]
        rng = np.random.default_rng(7)
        n = 4000
        o = np.stack([np.zeros(n), rng.uniform(-1.8, 1.8, n), rng.uniform(-1.4, 1.4, n)], 1)
        d = np.stack([np.ones(n), rng.uniform(-.03, .03, n), rng.uniform(-.03, .03, n)], 1)
        K = 8
    else:
        comps = lambda: [oa.DMD([5, 0, 0], N=(8, 6), pitch=0.5, tilt_angle=np.pi / 5), oa.SquareMirror([0, 4, 0], 12, 12).RotZ(-np.pi / 2)]
        rng = np.random.default_rng(8)
        n = 4000
        o = np.stack([np.zeros(n), rng.uniform(-2.0, 2.0, n), rng.uniform(-1.5, 1.5, n)], 1)
        d = np.stack([np.ones(n), rng.uniform(-.02, .02, n), rng.uniform(-.02, .02, n)], 1)
        K = 8
    batch = _batch(o, d, precision=precision)
    eng = get_engine()
    got = {}
    for label, inst, accel in (("instanced", 1, True), ("plain nodes", 0, True), ("no grid", 0, False)):
        eng.set_option(abi.OPT_INSTANCING, inst)
        try:
            table = _table(comps(), accelerate=accel)
            got[label] = table.trace_batch(batch, max_segments=K).to_host(reference_order=True)
        finally:
            eng.set_option(abi.OPT_INSTANCING, 1)
    for label in ("plain nodes", "no grid"):
        for f in abi.SEG_FIELDS + ("ray", "surface", "count"):
            np.testing.assert_array_equal(got["instanced"][f], got[label][f], err_msg=f"{label} {f}")
    assert len(got["instanced"]["ray"]) > len(o)
    if precision == "f64":
        ref = oracle.trace(_table(comps()).compile(), batch.to_host(), max_trace_num=K)
        np.testing.assert_array_equal(got["instanced"]["surface"], ref["surface"])
        for f in ("ox", "oy", "oz", "dx", "dy", "dz", "length", "pathlength"):
            np.testing.assert_allclose(got["instanced"][f], ref[f], rtol=1e-9, atol=1e-9, err_msg=f)


def test_instancing_shrinks_the_cfg5_image_and_moves_the_records_to_lds():
    import optable_amd as oa
    from optable_amd.engine import get_engine

    table = _table(scenes.cfg5_components(oa))
    batch = _batch(*scenes.cfg5_rays(20000, 3), precision="f32")
    table.trace_batch(batch, max_segments=50)
    info = get_engine().last_launch()
    assert info["kernel"] == 2 and info["pair_queue"] & 2, info  # rolling lists, records of the live rays in LDS
    assert info["threads"] * info["workgroups_per_cu"] >= 12 * 64, info


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n", [20000, 20001, 777])
def test_tiled_layout_equals_slots(precision, n):
    """ot_trace_tiled_*: the [k][ray] slots in 64-slot tiles — the same records bit for bit (even and odd n: lane pairs
    write 16 bytes only when the slot parity allows), monitors and final states read them like slot arrays."""
    import optable_amd as oa
    from optable_amd import dist
    from optable_amd import workloads as W

    for comps, gen, K in ((scenes.cfg2_components, lambda m: scenes.cfg2_rays(m, 0) + (scenes.WL,), 5),
                          (lambda ns: W.cfg4_components(ns), lambda m: W.cfg4_rays(max(m // 8, 1), 4, n_wavelengths=8), 3)):
        table = _table(comps(oa))
        o, d, wl = gen(n)
        from optable_amd.batch import RayBatch

        batch = RayBatch.from_arrays(o, d, wavelength=wl, q=1j * np.pi * scenes.W0**2 / wl, precision=precision)
        slots = table.trace_batch(batch, max_segments=K, layout="slots")
        tiled = table.trace_batch(batch, max_segments=K, layout="tiled")
        assert tiled.layout == "tiled" and tiled.capacity % 64 == 0
        a, b = slots.to_host(reference_order=True), tiled.to_host(reference_order=True)
        for f in abi.SEG_FIELDS + ("ray", "surface", "count"):
            np.testing.assert_array_equal(a[f], b[f], err_msg=f)
        np.testing.assert_array_equal(dist.final_state(tiled).cpu().numpy(), dist.final_state(slots).cpu().numpy())
    mon = oa.Monitor(origin=[1.0, 0, 0], width=8, height=8)
    h0, h1 = table.record_batch(mon, slots), table.record_batch(mon, tiled)
    assert len(h0) == len(h1)
    np.testing.assert_array_equal(h0.yList().cpu().numpy(), h1.yList().cpu().numpy())


def test_tiled_layout_is_for_light_scenes():
    import optable_amd as oa

    table = _table(scenes.cfg3_components(oa))
    batch = _batch(*scenes.cfg3_rays(5000, 2))
    with pytest.raises(RuntimeError, match="tiled layout belongs"):
        table.trace_batch(batch, max_segments=20, layout="tiled")


@pytest.mark.parametrize("case", ["cfg2", "cfg3", "cfg5"])
def test_auto_layout_holds_the_same_records(case):
    """layout="auto" (the default): for light scenes the slot layout this device streams faster, the dense list for heavy ones
    — the records of the slots either way."""
    import optable_amd as oa
    from optable_amd.engine import get_engine

    comps, gen, n, K = CASES[case]
    table = _table(comps(oa))
    batch = _batch(*gen(n), precision="f32")
    slots = table.trace_batch(batch, max_segments=K, layout="slots")
    auto = table.trace_batch(batch, max_segments=K)  # (the default IS "auto")
    assert auto.layout == (get_engine().plan("f32", batch.n, K)["layout"] if case == "cfg2" else "append")
    assert auto.layout in (("tiled", "slots") if case == "cfg2" else ("append",))
    a, b = slots.to_host(reference_order=True), auto.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)


def test_auto_layout_follows_the_library_rule():
    """layout="auto" asks the library (ot_trace_plan) instead of counting nodes: a heavy scene traced with a cap of two
    segments runs on the lane-per-ray kernel (tiles or slot arrays, never the append list its node count suggests); a light
    scene sent to the rolling lists (OT_OPT_KERNEL = 2) is refused by the tiled entry point and must come out as the dense
    list; a worst case beyond the append capacity limit falls back to the slot arrays.  Same records every time."""
    import optable_amd as oa
    from optable_amd.engine import get_engine

    eng = get_engine()
    comps, gen, n, K = CASES["cfg3"]
    table = _table(comps(oa))
    batch = _batch(*gen(n), precision="f32")  # (single precision: in double this scene's grids leave only the lists)
    slots = table.trace_batch(batch, max_segments=2, layout="slots")
    auto = table.trace_batch(batch, max_segments=2)
    plan = eng.plan("f32", n, 2)
    assert plan["kernel"] == 1 and plan["tiled_ok"] and auto.layout == plan["layout"] and auto.layout in ("tiled", "slots"), plan
    a, b = slots.to_host(reference_order=True), auto.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    us_slots, us_tiled = eng.probe_layouts("f32")
    assert plan["layout"] == ("tiled" if us_tiled < 0.985 * us_slots else "slots"), (plan, us_slots, us_tiled)
    assert eng.plan("f64", n, 2)["kernel"] == 2 and eng.plan("f64", n, 2)["layout"] == "append"
    assert eng.plan("f32", 1 << 20, 20)["layout"] == "append"
    assert eng.plan("f32", 1 << 26, 20)["layout"] == "slots"  # 20 x 2^26 slots: beyond the 2^30 an append block may hold
    assert eng.plan("f64", 1 << 25, 20)["layout"] == "slots"
    # a light scene on the rolling lists
    comps, gen, n, K = CASES["cfg2"]
    table = _table(comps(oa))
    batch = _batch(*gen(n), precision="f64")
    slots = table.trace_batch(batch, max_segments=K, layout="slots")
    # ... and, before that, the same scene with its layout MEASURED on the workload itself (Engine.tune_layout): auto follows it
    scene = table.compile()
    eng.upload(scene)
    tuned = eng.tune_layout(batch, K, launches=6)
    assert tuned["chosen"] == ("tiled" if tuned["tiled"] < tuned["slots"] else "slots")
    assert table.trace_batch(batch, max_segments=K, scene=scene).layout == tuned["chosen"]
    try:
        eng.set_option(abi.OPT_KERNEL, 2)
        auto = table.trace_batch(batch, max_segments=K)
        plan = eng.plan("f64", n, K)
    finally:
        eng.set_option(abi.OPT_KERNEL, 0)
    assert plan["kernel"] == 2 and not plan["tiled_ok"] and auto.layout == "append", plan
    a, b = slots.to_host(reference_order=True), auto.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)


def test_append_block_sized_from_a_sample_and_retried_when_too_small():
    """The default append call sizes its block from a 1 % sample (records per ray x 1.15 + slack) instead of the worst case,
    and a block that turns out too small is not an error: the launch reports what it needed and the trace runs again."""
    import optable_amd as oa
    from optable_amd.engine import get_engine

    comps, gen, n, K = CASES["cfg3"]
    n = 600_000  # (beyond the size below which the worst case is simply allocated)
    table = _table(comps(oa))
    batch = _batch(*gen(n), precision="f32")
    eng = get_engine()
    ref = table.trace_batch(batch, max_segments=K, layout="slots").to_host(reference_order=True)
    segs = table.trace_batch(batch, max_segments=K)
    records = int(segs.count.abs().sum().item())
    assert segs.layout == "append" and segs.capacity < n * K and segs.capacity <= 1.3 * records + (1 << 23), (segs.capacity, records)
    got = segs.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(got[f], ref[f], err_msg=f)
    # an estimate that is far too low (as if the sample had seen only rays that leave at once)
    scene = eng.scene
    eng._records_per_ray = {(id(scene), K, "f32"): 0.01}
    old_waves = eng.MAX_WAVES
    eng.MAX_WAVES = 16  # (no slack to hide behind)
    try:
        again = eng.trace(batch, K, layout="append")
    finally:
        eng.MAX_WAVES = old_waves
    assert again.n_valid >= records and again.capacity >= records
    got = again.to_host(reference_order=True)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        np.testing.assert_array_equal(got[f], ref[f], err_msg=f)

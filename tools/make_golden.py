#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (tim4431/optable) on the parity scenes.

Runs only in the build container, where /root/reference exists:
    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [scene ...]
The reference is imported from /root/reference (never copied); what is committed are data
fixtures: the packed input rays, the geometry the reference built (per-leaf pose and boxes, for
checking this package's scene compiler) and every field of every output segment in the order
`OpticalTable.ray_tracing` returned them, plus monitor hits and interact counts.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, os.path.join(ROOT, "tests"))
sys.path.insert(2, ROOT)  # tests/scenes.py takes the BASELINE generators from optable_amd.workloads (plain numpy)

import numpy as np  # noqa: E402
import optable as ref  # noqa: E402  (the reference)
import scenes  # noqa: E402

assert ref.__file__.startswith("/root/reference"), ref.__file__
OUT = os.path.join(ROOT, "tests", "golden")


def flatten(components):
    """Depth-first leaves and groups, the order the scene compiler must reproduce."""
    leaves, groups = [], []

    def walk(c, in_group):
        if hasattr(c, "components"):
            groups.append(c)
            for k in c.components:
                walk(k, True)
        else:
            leaves.append((c, in_group))

    for c in components:
        walk(c, False)
    return leaves, groups


def ray_rows(rays):
    rows = dict(
        origin=np.array([r.origin for r in rays], dtype=float).reshape(-1, 3),
        direction=np.array([r.direction for r in rays], dtype=float).reshape(-1, 3),
        intensity=np.array([r.intensity for r in rays], dtype=float),
        wavelength=np.array([r.wavelength for r in rays], dtype=float),
        length=np.array([np.inf if r.length is None else r.length for r in rays], dtype=float),
        alive=np.array([bool(r.alive) for r in rays]),
        has_q=np.array([r.qo is not None for r in rays]),
        q=np.array([complex(r.qo) if r.qo is not None else complex(np.nan, np.nan) for r in rays]),
        n=np.array([r.n for r in rays], dtype=float),
        pathlength=np.array([r._pathlength for r in rays], dtype=float),
    )
    return rows


def run(name):
    sc = {**scenes.SCENES, **scenes.HOOKED_SCENES}[name](ref)
    table = ref.OpticalTable()
    table.add_components(sc["components"])
    table.add_monitors(sc["monitors"])
    rays = sc["rays"]
    leaves, groups = flatten(table.components)
    ids = [r._id for r in rays]
    class_of = {}
    cls = np.array([class_of.setdefault(i, len(class_of)) for i in ids], dtype=np.int32)

    out = {}
    for key, val in ray_rows(rays).items():
        out["in_" + key] = val
    out["in_class"] = cls
    out["limit"] = np.array([-1 if not sc["limit"] else sc["limit"].get("max_trace_num", -1)])

    # trace exactly as OpticalTable.ray_tracing does (optical_table.py:66-70), keeping the tree index
    segs, tree = [], []
    for i, ray in enumerate(rays):
        traced = table._single_ray_tracing(ray, perfomance_limit=sc["limit"])
        segs.extend(traced)
        tree.extend([i] * len(traced))
    for key, val in ray_rows(segs).items():
        out["seg_" + key] = val
    out["seg_tree"] = np.array(tree, dtype=np.int32)

    # geometry as the reference built it
    out["leaf_origin"] = np.array([c.origin for c, _ in leaves], dtype=float).reshape(-1, 3)
    out["leaf_M"] = np.array([c.transform_matrix for c, _ in leaves], dtype=float).reshape(-1, 3, 3)
    out["leaf_in_group"] = np.array([g for _, g in leaves])
    out["leaf_bbox"] = np.array([c.bbox if g else (np.nan,) * 6 for c, g in leaves], dtype=float).reshape(-1, 6)
    out["group_bbox"] = np.array([g.bbox for g in groups], dtype=float).reshape(-1, 6)
    limited = [c for c, _ in leaves if c.max_interact_count is not None]
    out["counts"] = np.array([[c._interact_count.get(rid, 0) for rid in class_of] for c in limited],
                             dtype=np.int32).reshape(len(limited), len(class_of))

    # monitors: raw hit lists (monitor.py:183-193)
    seg_index = {id(s): k for k, s in enumerate(segs)}
    for m, mon in enumerate(table.monitors):
        raw = mon._data_raw
        out[f"mon{m}_P"] = np.array([d[0] for d in raw], dtype=float).reshape(-1, 3)
        out[f"mon{m}_I"] = np.array([d[1] for d in raw], dtype=float)
        out[f"mon{m}_t"] = np.array([d[2] for d in raw], dtype=float)
        out[f"mon{m}_seg"] = np.array([seg_index[id(d[3])] for d in raw], dtype=np.int64)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {len(rays)} rays, {len(leaves)} leaves, {len(groups)} groups -> {len(segs)} segments")


def exports_fixture():
    """g23: the files the reference's writers produce (optical_table.py:487-500 export_rays_csv, monitor.py:255-269
    export_rays_npz) for g01 (Gaussian q, thin lenses, a slab) and g06 (3-D mirror pair, two monitors): the CSV as the
    list of its text lines, the npz as its arrays.  Output DATA of the reference, not its source."""
    import tempfile

    out = {}
    for name in ("g01_gaussian_beam", "g06_mirror_pair", "g12_dove"):
        sc = scenes.SCENES[name](ref)
        table = ref.OpticalTable()
        table.add_components(sc["components"])
        table.add_monitors(sc["monitors"])
        table.ray_tracing(sc["rays"], perfomance_limit=sc["limit"])
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "rays.csv")
            table.export_rays_csv(path)
            out[name + "_csv"] = np.array(open(path).read().splitlines())
            for m, mon in enumerate(table.monitors):
                mpath = os.path.join(tmp, f"mon{m}.npz")
                mon.export_rays_npz(mpath)
                for key, val in np.load(mpath).items():
                    out[f"{name}_mon{m}_{key}"] = val
    np.savez_compressed(os.path.join(OUT, "g23_exports.npz"), **out)
    print("g23_exports:", {k: v.shape for k, v in out.items()})


def abcd_fixture():
    """g17: OpticalTable.calculate_abcd_matrix on a 4f relay (a caller of the hot path)."""
    sc = scenes.abcd_4f(ref)
    table = ref.OpticalTable()
    table.add_components(sc["components"])
    table.add_monitors(sc["monitors"])
    Ms = table.calculate_abcd_matrix(sc["monitors"][0], sc["monitors"][1], sc["rays"])
    np.savez_compressed(os.path.join(OUT, "g17_abcd.npz"), Ms=Ms)
    print("g17_abcd:", Ms.shape, Ms[3].round(6).tolist())


def slab_vectors():
    """G14: solve_ray_bboxes_intersections unit vectors incl. axis-parallel and flat boxes."""
    rng = np.random.default_rng(14)
    boxes = [(-1, 1, -1, 1, -1, 1), (0, 0, -1, 1, -1, 1), (2, 3, 0, 0, -0.5, 0.5), (-1, 1, -2, -1, 5, 6)]
    o = rng.uniform(-3, 3, (64, 3))
    d = rng.normal(size=(64, 3))
    d[::4, 0] = 0.0
    d[1::8, 1] = 1e-9
    d[2::8, 2] = -1e-8
    o[3::16] = [0.5, 0.5, 0.5]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t1 = np.zeros((64, len(boxes)))
    t2 = np.zeros_like(t1)
    hit = np.zeros_like(t1, dtype=bool)
    for i in range(64):
        a, b, h = ref.solve_ray_bboxes_intersections(o[i], d[i], [tuple(bx) for bx in boxes])
        t1[i], t2[i], hit[i] = a, b, h
    np.savez_compressed(os.path.join(OUT, "g14_slab.npz"), o=o, d=d, boxes=np.array(boxes, dtype=float), t1=t1, t2=t2, hit=hit)
    print("g14_slab: 64 rays x 4 boxes")


def interact_fixture():
    """g20: the single-call API (`component.interact(ray)`, `intersect_point_local`) on single components."""
    out = {}
    names = []
    for name, comp, rays in scenes.interact_cases(ref):
        names.append(name)
        for key, val in ray_rows(rays).items():
            out[f"{name}_in_{key}"] = val
        t_all, rows, owner = [], [], []
        P_loc, t_loc = [], []
        for k, ray in enumerate(rays):
            t, rays_out = comp.interact(ray)
            t_all.append(np.nan if t is None else t)
            for r in (rays_out or []):
                rows.append(r)
                owner.append(k)
            if not hasattr(comp, "components"):
                P, tl = comp.intersect_point_local(comp.ray_to_local_coordinates(ray))
                P_loc.append([np.nan] * 3 if P is None else list(P))
                t_loc.append(np.nan if tl is None else tl)
        out[f"{name}_t"] = np.array(t_all, dtype=float)
        for key, val in ray_rows(rows).items():
            out[f"{name}_out_{key}"] = val
        out[f"{name}_out_owner"] = np.array(owner, dtype=np.int32)
        if P_loc:
            out[f"{name}_local_P"] = np.array(P_loc, dtype=float).reshape(-1, 3)
            out[f"{name}_local_t"] = np.array(t_loc, dtype=float)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g20_interact.npz"), **out)
    print("g20_interact:", names)


def real_example_fixture():
    """g27: the reference's largest example as it stands (examples/ripa_gen2_lensless.py: a multi-pass cavity of micro-mirror
    arrays, 7,689 leaf surfaces in nested groups, count-limited prism faces, ONE Gaussian ray that is reflected about three
    thousand times; the author's own research scene).  The script is executed from /root/reference up to and including its
    `table.ray_tracing(...)` call (the rest renders); what is stored is DATA: the scene as this package's compiler flattens the
    reference's object graph (`ot_node` / material / aux tables: poses, shapes, coefficients), the input ray and every output
    segment in the order the reference returned them.  The GPU test uploads the tables and traces the ray."""
    import contextlib
    import io
    import time

    import optable_amd as oa

    path = "/root/reference/examples/ripa_gen2_lensless.py"
    src = open(path).read()
    cut = src.index("\n", src.index("table.ray_tracing(R1rays0")) + 1
    env = {"__name__": "__main__", "__file__": path}
    cwd = os.getcwd()
    os.chdir(os.path.dirname(path))
    t0 = time.time()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            exec(compile(src[:cut], path, "exec"), env)
    finally:
        os.chdir(cwd)
    seconds = time.time() - t0
    table, rays = env["table"], env["R1rays0"]
    scene = oa.compile_scene(table.components, getattr(table, "unit", 1e-2))
    out = {}
    for key, val in ray_rows(rays).items():
        out["in_" + key] = val
    for key, val in ray_rows(table.rays).items():
        out["seg_" + key] = val
    out["seg_tree"] = np.zeros(len(table.rays), dtype=np.int32)
    out.update(scene.to_tables())
    out["counts"] = np.array([c._interact_count.get(rays[0]._id, 0) for c in scene.limited], dtype=np.int32)
    out["max_trace_num"] = np.array([100000])
    out["reference_seconds"] = np.array([seconds])
    np.savez_compressed(os.path.join(OUT, "g27_real_example.npz"), **out)
    print(f"g27_real_example: {scene.n_leaves} leaves, {scene.n_nodes} nodes, {len(table.rays)} segments, reference {seconds:.1f} s "
          f"(scene construction included)")


def all_examples_fixture():
    """examples/*.py of the reference, every one as it stands: the script is executed from /root/reference with
    `OpticalTable.ray_tracing` wrapped — the FIRST call is let through, recorded (the table's scene as this package's compiler
    flattens the reference's objects, the input rays, every output segment in the order returned, interact counts) and the script
    is stopped there (what follows renders or optimises).  One .npz per example under tests/golden/examples/; scripts that never
    trace (material plots, colour maps) are listed as such.  DATA only: tables, rays, segments."""
    import contextlib
    import glob
    import io
    import time

    import optable_amd as oa

    class Stop(Exception):
        pass

    outdir = os.path.join(OUT, "examples")
    os.makedirs(outdir, exist_ok=True)
    original = ref.OpticalTable.ray_tracing
    summary = []
    for path in sorted(glob.glob("/root/reference/examples/*.py")):
        name = os.path.splitext(os.path.basename(path))[0]
        seen = {}

        def wrapped(self, rays, perfomance_limit=None):
            rays_in = [rays] if isinstance(rays, ref.Ray) else list(rays)
            before = len(self.rays)
            scene = oa.compile_scene(self.components, getattr(self, "unit", 1e-2))  # poses as they are at the call
            ids = [r._id for r in rays_in]
            counts0 = [[c._interact_count.get(i, 0) for i in dict.fromkeys(ids)] for c in scene.limited]
            t0 = time.time()
            segs, tree = [], []
            for i, ray in enumerate(rays_in):  # optical_table.py:66-70, keeping the tree index
                traced = self._single_ray_tracing(ray, perfomance_limit=perfomance_limit)
                self.rays.extend(traced)
                segs.extend(traced)
                tree.extend([i] * len(traced))
            seen.update(scene=scene, rays=rays_in, segs=segs, tree=tree, limit=perfomance_limit, seconds=time.time() - t0, counts0=counts0,
                        accumulated=before)
            raise Stop()

        ref.OpticalTable.ray_tracing = wrapped
        env = {"__name__": "__main__", "__file__": path}
        cwd = os.getcwd()
        os.chdir(os.path.dirname(path))
        np.random.seed(12345)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                exec(compile(open(path).read(), path, "exec"), env)
            status = "never traces"
        except Stop:
            status = "traced"
        except Exception as exc:  # noqa: BLE001 - a script that fails before its first trace (a missing file, a display)
            status = f"stopped before its first trace: {type(exc).__name__}: {exc}"
        finally:
            os.chdir(cwd)
            ref.OpticalTable.ray_tracing = original
            import matplotlib.pyplot as plt

            plt.close("all")
        if status != "traced":
            summary.append((name, status))
            continue
        if name == "ripa_gen2_lensless":  # fixture g27 (real_example_fixture) is this example: 328 KB once
            summary.append((name, "see ../g27_real_example.npz (7,689 leaves, 1 ray -> 3,202 segments, cap 100000)"))
            continue
        scene, rays = seen["scene"], seen["rays"]
        out = {}
        for key, val in ray_rows(rays).items():
            out["in_" + key] = val
        ids = [r._id for r in rays]
        class_of = {}
        out["in_class"] = np.array([class_of.setdefault(i, len(class_of)) for i in ids], dtype=np.int32)
        out["in_unit_scale"] = np.array([getattr(r, "unit", scene.unit) / scene.unit for r in rays], dtype=float)
        for key, val in ray_rows(seen["segs"]).items():
            out["seg_" + key] = val
        out["seg_tree"] = np.array(seen["tree"], dtype=np.int32)
        out.update(scene.to_tables())
        cap = 2000 if not seen["limit"] or "max_trace_num" not in seen["limit"] else int(seen["limit"]["max_trace_num"])
        out["max_trace_num"] = np.array([cap])
        out["counts_before"] = np.array(seen["counts0"], dtype=np.int32).reshape(len(scene.limited), len(class_of))
        out["counts"] = np.array([[c._interact_count.get(i, 0) for i in class_of] for c in scene.limited], dtype=np.int32).reshape(len(scene.limited), len(class_of))
        out["reference_seconds"] = np.array([seen["seconds"]])
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
        summary.append((name, f"{len(rays)} rays, {scene.n_leaves} leaves -> {len(seen['segs'])} segments, cap {cap}, reference {seen['seconds']:.2f} s"))
    with open(os.path.join(outdir, "INDEX.txt"), "w") as fh:
        fh.write("# examples/*.py of the reference at their first OpticalTable.ray_tracing call (tools/make_golden.py all_examples_fixture)\n")
        for name, line in summary:
            fh.write(f"{name}: {line}\n")
            print(f"{name}: {line}")


def calibrate_fixture():
    """g22: OpticalTable.calibrate_symmetric_4f (optical_table.py:299-422), a caller of the hot path: the cost
    terms at a fixed (F1, F2) and the Nelder-Mead result for two criteria."""
    import contextlib
    import io

    sc = scenes.calibrate_case(ref)
    out = {}  # (optimize=False renders upstream, :346-359 — plotting, out of scope; the fixed-point matrices are g17)
    for crit in ("M=-I", "min_stdtY"):
        with contextlib.redirect_stdout(io.StringIO()):
            F1, F2 = ref.OpticalTable.calibrate_symmetric_4f(sc["lens"], sc["rays"], sc["F10"], sc["F20"], criterion=crit)
        out["opt_" + crit.replace("=", "").replace("-", "m")] = np.array([F1, F2])
    np.savez_compressed(os.path.join(OUT, "g22_calibrate.npz"), **out)
    print("g22_calibrate:", {k: np.round(v, 5).tolist() if v.size <= 4 else v.shape for k, v in out.items()})


if __name__ == "__main__":
    names = sys.argv[1:] or list(scenes.SCENES) + list(scenes.HOOKED_SCENES)
    for nm in names:
        if nm in ("g14_slab", "g17_abcd", "g20_interact", "g22_calibrate", "g23_exports", "g27_real_example", "examples"):
            continue
        np.random.seed(12345)
        run(nm)
    if not sys.argv[1:] or "g14_slab" in sys.argv[1:]:
        slab_vectors()
    if not sys.argv[1:] or "g17_abcd" in sys.argv[1:]:
        abcd_fixture()
    if not sys.argv[1:] or "g22_calibrate" in sys.argv[1:]:
        np.random.seed(12345)
        calibrate_fixture()
    if not sys.argv[1:] or "g23_exports" in sys.argv[1:]:
        np.random.seed(12345)
        exports_fixture()
    if "examples" in sys.argv[1:]:  # (every example of the reference at its first trace: ~40 s, on request only)
        all_examples_fixture()
    if "g27_real_example" in sys.argv[1:]:  # (18 s of the reference: on request only)
        real_example_fixture()
    if not sys.argv[1:] or "g20_interact" in sys.argv[1:]:
        np.random.seed(12345)
        interact_fixture()

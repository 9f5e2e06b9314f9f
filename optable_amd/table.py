"""OpticalTable: scene container whose `ray_tracing` runs on the MI355X engine.

Drop-in for optable/optical_table.py:9-147 (`add_components`, `add_monitors`, `ray_tracing`
with the misspelt `perfomance_limit` kwarg, `.rays` accumulating across calls, a deep copy
returned).  What changes is where the work happens: the component list is compiled to flat
tables (scene.py), the rays are packed structure-of-arrays into HBM (batch.py) and
liboptable_hip.so traces every ray tree; `Ray` objects are rebuilt from the segment stream
only at the end.  `trace_batch` is the same path without Python objects, for 1e6+ rays.

The wall-clock cap (`perfomance_limit["max_trace_time"]`, 600 s by default, optical_table.py:84-97) is honoured
where a trace takes more than one launch: ray trees that go generation by generation read the clock after each
(engine.trace_tree), and say where they are once per second as the reference's loop does (optical_table.py:99-111).  A
non-branching scene, and a batch of ray trees whose queues fit the chip (engine.trace_trees), is ONE launch bounded by
`max_trace_num`: milliseconds, nothing can be cut inside it.  Render / GUI are out of scope.
"""
import copy
from typing import List, Union

import numpy as np

from . import abi
from .assemblies import *  # noqa: F401,F403  (reference re-exports, optical_table.py:1-3)
from .components import *  # noqa: F401,F403
from .components import OpticalComponent
from .geometry import base_merge_bboxs, _NO_BOX, out_of_scope
from .monitors import Monitor
from .rays import Ray
from .scene import compile_scene

MAX_TRACE_NUM = 2000  # optical_table.py:87
MAX_TRACE_TIME = 600.0  # seconds, optical_table.py:86
_FUSED_MAX_SEGMENTS = 64


def _engine():
    from .engine import get_engine  # imports torch + the HIP library lazily

    return get_engine()


class OpticalTable:
    render = out_of_scope("render")
    gather_components = out_of_scope("gather_components")
    export_components_csv = out_of_scope("export_components_csv")
    add_wavelength_legend = out_of_scope("add_wavelength_legend")

    def __init__(self, **kwargs):
        self.components = []
        self.rays = []
        self.monitors = []
        self.norender_set = set()
        self._bbox = _NO_BOX
        self.unit = kwargs.get("unit", 1e-2)

    # -- scene building (optical_table.py:25-43) ------------------------------------------------
    def add_components(self, component: Union[OpticalComponent, List]):
        self._collect(component, self.components, OpticalComponent)

    def add_monitors(self, monitor: Union[Monitor, List]):
        self._collect(monitor, self.monitors, Monitor)

    @staticmethod
    def _collect(item, sink, cls):
        if isinstance(item, cls):
            sink.append(item)
        elif isinstance(item, list):
            for entry in item:
                if isinstance(entry, cls):
                    sink.append(entry)
                elif isinstance(entry, list):
                    sink.extend(entry)

    @property
    def bbox(self):
        if self._bbox[0] is None:
            self.get_bbox()
        return self._bbox

    def get_bbox(self):
        self._bbox = base_merge_bboxs([c.bbox for c in self.components])
        return self._bbox

    accelerate = True  # attach the acceleration grids (scene.py); results do not depend on it

    def compile(self):
        """Flatten the current components into device tables (poses are read now)."""
        return compile_scene(self.components, self.unit, accelerate=self.accelerate)

    # -- the hot path ---------------------------------------------------------------------------
    def ray_tracing(self, rays: Union[Ray, List[Ray]], perfomance_limit=None):
        """Trace `rays`; append the finished segments to `self.rays`; return a deep copy of
        everything accumulated so far (optical_table.py:57-72)."""
        if isinstance(rays, Ray):
            rays = [rays]
        cap, max_time = MAX_TRACE_NUM, MAX_TRACE_TIME
        if perfomance_limit is not None and "max_trace_num" in perfomance_limit:
            cap = int(perfomance_limit["max_trace_num"])
        if perfomance_limit is not None and "max_trace_time" in perfomance_limit:
            max_time = float(perfomance_limit["max_trace_time"])
        if len(rays) and cap > 0 and max_time > 0:  # the reference's loop does not start on an exhausted limit
            traced, capped = self._trace_objects(list(rays), cap, max_time)
            if capped:
                # the reference prints the number of traces its loop got through (optical_table.py:138-143): `cap` when
                # the count limit cut a tree, what was done by then when the clock did (the clock here runs over the
                # whole call and is read after every generation of the batch, not per input ray)
                done = min(cap, self._last_trace_num) if getattr(self, "_last_trace_num", None) else cap
                print(f"Ray tracing time exceeds the maximum tracing time after {done} traces. "
                      f"({capped} ray tree(s) truncated)")
            self.rays.extend(traced)
        elif len(rays):  # an exhausted limit: the reference's loop does not start, nothing is archived, the message is printed
            print(f"Ray tracing time exceeds the maximum tracing time after 0 traces. ({len(rays)} ray tree(s) truncated)")
        return _clone_rays(self.rays)

    def trace_batch(self, batch, max_segments=None, counts=None, scene=None, layout="auto", capacity=None):
        """Scalable entry: `RayBatch` in, `SegmentBatch` out, no Python objects.  Non-branching
        scenes run as one launch with [segment][ray] output slots; branching scenes run generation by
        generation; both in the batch's precision.
        `scene`: a `table.compile()` result to reuse when the components have not changed since (flattening
        a few hundred components in Python costs milliseconds — 10 ms for cfg 5 — and the engine skips the
        upload when it already holds that very scene); default: compile now, poses are read at call time.
        `layout`: "auto" (the default: what the scene's kernels write fastest on this device, Engine.plan — the dense
        "append" list for heavy scenes, "tiled" or "slots" for light ones, whichever the device streams faster), "slots"
        ([segment][ray] slots), "tiled" (the same slots in 64-slot tiles) or "append" (a dense list in append order,
        `capacity` slots: see Engine.trace) for the non-branching launch; ray trees come back as [k][tree] slots when one
        launch takes them (Engine.trace_branching: light scenes, caps whose queues fit the LDS) and as a list in generation
        order otherwise.  Every reader of a SegmentBatch takes all of them."""
        eng = _engine()
        if scene is None:
            scene = self.compile()
        with eng.lock:
            eng.upload(scene)
            cap = MAX_TRACE_NUM if max_segments is None else int(max_segments)
            if scene.limited:
                return self._trace_batch_limited(eng, scene, batch, cap, max_segments is not None, counts)
            speculate = scene.max_children == 2 and not scene.always_branches
            if max_segments is not None and (scene.max_children <= 1 or speculate):
                # max_children == 2: speculate that no tree actually branches (e.g. mirror-coated
                # interfaces only split on total internal reflection); fall back when one does.
                segs = eng.trace(batch, cap, layout=layout, capacity=capacity)
                if scene.max_children <= 1 or not bool((segs.count < 0).any()):
                    return segs
            return eng.trace_branching(batch, cap)

    def _trace_batch_limited(self, eng, scene, batch, cap, fused_ok, counts):
        """`trace_batch` for scenes with `max_interact_count` surfaces.  Their counters are keyed by ray id
        (optical_component.py:140-149): the device table has one column per DISTINCT id of the batch
        (`segs.count_ids` names the columns, ascending), and rays that share an id — copies made by
        `multiplexed_in_wavelength`, explicit duplicates — must see each other's updates in input order,
        as the reference finishes one input ray before it starts the next (optical_table.py:66-70).  Rays
        of one id are therefore traced in successive rounds (round r = the r-th ray of every id) over the
        same table; rays with different ids never interact, so a round is one ordinary launch."""
        import torch
        from .batch import SegmentBatch

        n, dev = batch.n, batch.device
        uniq, inverse = torch.unique(batch.id, return_inverse=True)
        n_classes, n_slots = int(uniq.numel()), len(scene.limited)
        if counts is None:
            counts = torch.zeros((n_slots, max(n_classes, 1)), dtype=torch.int32, device=dev)
        elif tuple(counts.shape) != (n_slots, n_classes):
            raise ValueError(f"counts must be [{n_slots}, {n_classes}] (limited surfaces x distinct ray ids)")
        work = batch.with_ids(inverse)
        fused = fused_ok and scene.max_children <= 1

        def run(sub):
            # (a run holds every id once: the lane-per-tree kernel's FIFO order per tree is the reference's)
            return eng.trace(sub, cap, counts=counts) if fused else eng.trace_branching(sub, cap, counts=counts, distinct_ids=True)

        if n_classes == n:
            segs = run(work)
            segs.count_ids = uniq
            return segs
        order = torch.argsort(inverse, stable=True)
        grouped = inverse[order]
        rounds = torch.empty(n, dtype=torch.int64, device=dev)
        rounds[order] = torch.arange(n, device=dev) - torch.searchsorted(grouped, grouped)
        parts = []
        for r in range(int(rounds.max()) + 1):
            idx = torch.nonzero(rounds == r).flatten()  # ascending: input order inside a round
            parts.append((idx, run(work.take(idx))))
        if fused:  # [k][ray] slots of the whole batch
            out = SegmentBatch(n * cap, batch.precision, dev)
            out.count, out.n_rays = torch.empty(n, dtype=torch.int32, device=dev), n
            for idx, part in parts:
                m = int(idx.numel())
                out.count[idx] = part.count
                for f in abi.SEG_FIELDS + ("surface",):
                    out.field(f).view(cap, n)[:, idx] = part.field(f).view(cap, m)
                out.ray.view(cap, n)[:, idx] = idx.to(torch.int32).unsqueeze(0).expand(cap, m)
        else:  # flat lists: concatenate, tree indices back to positions in `batch`
            def valid(part):  # slots a round's trace filled: the first n_valid of a list; of [k][tree] slots (lane-per-tree launch) row by
                if part.count is None:  # row, which keeps every tree's rays in their FIFO order
                    return slice(0, part.n_valid), part.n_valid
                keep = part.valid_mask()
                return keep, int(keep.sum().item())

            picks = [valid(part) for _, part in parts]
            total = sum(v for _, v in picks)
            out = SegmentBatch(total, batch.precision, dev)
            out.capped = torch.zeros(n, dtype=torch.bool, device=dev)
            at = 0
            for (idx, part), (keep, v) in zip(parts, picks):
                for f in abi.SEG_FIELDS + ("surface",):
                    out.field(f)[at:at + v] = part.field(f)[keep]
                out.ray[at:at + v] = idx[part.ray[keep].long()].to(torch.int32)
                out.capped[idx] = part.capped
                at += v
            out.n_valid = total
        out.counts_table, out.count_ids = counts, uniq
        return out

    def trace_host(self, origin, direction, **kwargs):
        """Host numpy rays in, host numpy results out, PCIe copies overlapped with the trace (stream.py)."""
        from .stream import trace_host

        return trace_host(self, origin, direction, **kwargs)

    def record_batch(self, monitor, segs):
        """Monitor.record over a SegmentBatch without Python objects: a `MonitorHits` (device tensors
        + the Monitor accessors: yList, tYList, IList, ... with the reference's sort orders)."""
        from .monitors import MonitorHits

        segs = segs.to_slots()  # a tiled history: as slot arrays (the accessors index segments by slot)
        slot, P, t = _engine().monitor_record(monitor_struct(monitor), segs)
        return MonitorHits(monitor, segs, slot, P, t)

    # -- ABCD extraction (optical_table.py:211-297), a caller of the hot path ----------------------
    def calculate_abcd_matrix(self, mon0, mon1, rays, disp=1e-5, rot=1e-5, debugaxs=None):
        """Per-ray 2x2 ABCD matrix between two monitors by finite differences: three traces
        (nominal, displaced along mon0's Y tangent, rotated about mon0's Z tangent at the nominal
        hit point).  Each trace is one device batch.  The reference's in-place biasing is kept:
        the angular probe is applied to the already displaced rays (optical_table.py:272-284)."""
        y_axis, z_axis = mon0.tangent_Y, mon0.tangent_Z

        def simulate(batch):
            self.rays = []
            mon0.clear()
            mon1.clear()
            self.ray_tracing(batch)

        assert len(rays) > 0, "No rays to trace in ABCD calculation."
        ids = [r._id for r in rays]
        assert len(set(ids)) == len(rays), "Redundant ray ids in ABCD calculation."
        probe = [copy.deepcopy(rays[i]) for i in np.argsort(ids)]
        simulate(probe)
        for mon, label in ((mon0, "mon0"), (mon1, "mon1")):
            assert set(ids) == {r._id for r in mon.get_rays(sort="ID")}, f"Rays at {label} do not match the input rays."
        pivots = mon0.get_PList(sort="ID")
        y0, ty0 = mon1.get_yList(sort="ID"), mon1.get_tYList(sort="ID")
        simulate([r._Translate(y_axis * disp) for r in probe])
        y1, ty1 = mon1.get_yList(sort="ID"), mon1.get_tYList(sort="ID")
        simulate([r._RotAround(z_axis, pivots[k], rot) for k, r in enumerate(probe)])
        y2, ty2 = mon1.get_yList(sort="ID"), mon1.get_tYList(sort="ID")
        Ms = np.zeros((len(rays), 2, 2))
        Ms[:, 0, 0], Ms[:, 1, 0] = (y1 - y0) / disp, (ty1 - ty0) / disp
        Ms[:, 0, 1], Ms[:, 1, 1] = (y2 - y0) / rot, (ty2 - ty0) / rot
        return Ms

    def abcd_batch(self, mon0, mon1, batch, disp=1e-5, rot=1e-5, max_segments=64):
        """`calculate_abcd_matrix` for a RayBatch, entirely on the device: three traces, monitor
        passes and the finite differences are tensor operations; returns a [N, 2, 2] tensor (ray i of
        the batch = row i).  Same biasing sequence as the object version (the angular probe acts on
        the already displaced rays and pivots about mon0's LOCAL hit point used as a lab point,
        optical_table.py:259-284).  Every ray must cross each monitor exactly once."""
        import torch
        from .geometry import rotation_matrix

        scene = self.compile()  # the components do not move between the three traces

        def probe(b):
            segs = self.trace_batch(b, max_segments=max_segments, scene=scene)
            h0, h1 = self.record_batch(mon0, segs), self.record_batch(mon1, segs)
            for h, label in ((h0, "mon0"), (h1, "mon1")):
                if len(h) != b.n or not torch.equal(h.ray_index("ID"), torch.arange(b.n, device=b.device)):
                    raise AssertionError(f"Rays at {label} do not match the input rays.")
            return h0, h1

        work = batch.clone()
        h0, h1 = probe(work)
        pivots = h0.PList("ID")
        y0, ty0 = h1.yList("ID"), h1.tYList("ID")
        work.translate_(np.asarray(mon0.tangent_Y, dtype=float) * disp)
        _, h1 = probe(work)
        y1, ty1 = h1.yList("ID"), h1.tYList("ID")
        work.rotate_around_(rotation_matrix(mon0.tangent_Z, rot), pivots)
        _, h1 = probe(work)
        y2, ty2 = h1.yList("ID"), h1.tYList("ID")
        Ms = torch.empty((batch.n, 2, 2), dtype=y0.dtype, device=batch.device)
        Ms[:, 0, 0], Ms[:, 1, 0] = (y1 - y0) / disp, (ty1 - ty0) / disp
        Ms[:, 0, 1], Ms[:, 1, 1] = (y2 - y0) / rot, (ty2 - ty0) / rot
        return Ms

    @staticmethod
    def calibrate_symmetric_4f(lens, rays, F10, F20, criterion="M=-I", debugaxs=None, optimize=True, display_M=False):
        """Tune the distances of mon0 - F1 - lens - 2 F2 - lens(turned) - F1 - mon1 with Nelder-Mead
        (optical_table.py:299-422).  A caller of the hot path: every cost evaluation is one
        `ray_tracing` + one `calculate_abcd_matrix` (4 device traces)."""
        def simulate(F1, F2):
            first = lens.copy()._Translate(np.array([F1, 0, 0]) - lens.origin)
            second = lens.copy()._Translate(np.array([F1 + 2 * F2, 0, 0]) - lens.origin).RotZ(np.pi)
            mon0 = Monitor(origin=[0, 0, 0], width=5, height=5)
            mon1 = Monitor(origin=[2 * F1 + 2 * F2, 0, 0], width=5, height=5)
            table = OpticalTable()
            table.add_components([first, second])
            table.add_monitors([mon0, mon1])
            table.ray_tracing(rays)
            y, ty = mon1.get_yList(), mon1.get_tYList()
            Ms = table.calculate_abcd_matrix(mon0, mon1, rays)
            if display_M:
                for M in Ms:
                    print(M)
            return Ms, y, ty

        costs = {
            "M=-I": lambda Ms, ty: np.mean([np.linalg.norm(M + np.eye(2)) for M in Ms]),
            "flat_field": lambda Ms, ty: np.mean([abs((1.5 * M[0, 0] - M[0, 1]) / (M[1, 1] - M[1, 0] * 1.5) - 1.5) for M in Ms]),
            "min_stdtY": lambda Ms, ty: float(np.std(ty)),
        }
        if criterion not in costs:
            raise ValueError(f"Unknown criterion: {criterion}")
        if not optimize:
            return simulate(F10, F20)
        from scipy.optimize import minimize

        def cost(x):
            Ms, _, ty = simulate(x[0], x[1])
            return costs[criterion](Ms, ty)

        res = minimize(cost, x0=[F10, F20], method="Nelder-Mead", options={"disp": True, "xatol": 1e-5, "maxiter": 50})
        return res.x[0], res.x[1]

    # -- exports (optical_table.py:447-500), written from columns (export.py) ---------------------------
    def gather_rays_csv(self):
        from . import export

        cols, has_q = export.columns_of_rays(self.rays)
        return [dict(zip(export.HEADER, row)) for row in export.rays_csv_rows(cols, has_q)]

    def export_rays_csv(self, filename: str):
        from . import export

        cols, has_q = export.columns_of_rays(self.rays)
        export.write_rays_csv(filename, cols, has_q)

    def export_batch_csv(self, segs, filename: str, rays=None):
        """`export_rays_csv` for a SegmentBatch, straight from its columns: no Ray objects.  `rays`: the traced
        RayBatch (tells which rays carry a Gaussian q, ray.py:98-104); without it every segment prints its q."""
        segs.export_rays_csv(filename, rays)

    def materialize(self, segs, sources, select=None):
        """SegmentBatch -> List[Ray] for the input rays in `select` only (all when None), in the
        reference's order; `sources` are the input Ray objects (ids, wavelengths, custom attributes
        are inherited from them as upstream copies do).  Lets plotting code consume a device trace
        without building millions of objects."""
        host = segs.to_host(reference_order=True)
        keep = np.ones(len(host["ray"]), dtype=bool) if select is None else np.isin(host["ray"], np.asarray(list(select)))
        host = {k: (v[keep] if k != "count" else v) for k, v in host.items()}
        per_ray = [None] * len(sources)
        _scatter_segments(host, sources, np.arange(len(sources)), per_ray)
        return [seg for chunk in per_ray if chunk for seg in chunk]

    # -- List[Ray] plumbing ------------------------------------------------------------------------
    def _trace_objects(self, rays, cap, max_time=MAX_TRACE_TIME):
        with _engine().lock:  # upload + every trace of this call as one unit (see Engine.lock)
            return OpticalTable._trace_objects_locked(self, rays, cap, max_time)  # (`self` may be the reference package's table: adapter.install)

    def _trace_objects_locked(self, rays, cap, max_time=MAX_TRACE_TIME):
        import torch

        eng = _engine()
        scene = self.compile()
        eng.upload(scene)
        n = len(rays)
        ids = [r._id for r in rays]
        class_of = {}
        cls = np.array([class_of.setdefault(i, len(class_of)) for i in ids], dtype=np.int32)
        n_classes = len(class_of)
        # rays sharing an _id share interact counters and must see each other's updates in
        # input order (optical_component.py:140-149): trace them in successive rounds.
        rounds = np.zeros(n, dtype=np.int64)
        if scene.limited and n_classes < n:
            seen = {}
            for k, c in enumerate(cls):
                rounds[k] = seen.get(c, 0)
                seen[c] = rounds[k] + 1
        counts = None
        if scene.limited:
            host = np.zeros((len(scene.limited), n_classes), dtype=np.int32)
            for s, comp in enumerate(scene.limited):
                for rid, c in class_of.items():
                    host[s, c] = comp._interact_count.get(rid, 0)
            counts = torch.from_numpy(host).to(eng.device)
        per_ray = [None] * n
        total_capped = 0
        for rnd in range(int(rounds.max()) + 1):
            pick = np.nonzero(rounds == rnd)[0]
            sub = [rays[k] for k in pick]
            if scene.hooks:  # leaves whose interact_local is the user's Python: generation by generation, device search + host physics
                chunks, capped = OpticalTable._trace_hooked(self, eng, scene, sub, cls[pick], cap, counts, max_time)
                total_capped += capped
                for k, chunk in zip(pick, chunks):
                    per_ray[k] = chunk
                continue
            batch = _pack(sub, cls[pick], eng.device, scene.unit)
            if scene.max_children <= 1 and cap <= _FUSED_MAX_SEGMENTS:
                segs = eng.trace(batch, cap, counts=counts)
                host_segs = segs.to_host(reference_order=True)
                capped = _fused_capped(host_segs, cap)
            else:
                segs = eng.trace_branching(batch, cap, counts=counts, max_trace_time=max_time, distinct_ids=True)  # (a round holds every id once)
                host_segs = segs.to_host(reference_order=True)
                capped = int(segs.capped.sum().item())
                if capped:  # traces done by the largest truncated tree (what the cap message reports)
                    per_tree = np.bincount(host_segs["ray"], minlength=len(sub))
                    self._last_trace_num = int(per_tree[segs.capped.cpu().numpy()].max())
            total_capped += capped
            _scatter_segments(host_segs, sub, pick, per_ray)
        if scene.limited:
            host = counts.cpu().numpy()
            for s, comp in enumerate(scene.limited):
                for rid, c in class_of.items():
                    if host[s, c] or rid in comp._interact_count:
                        comp._interact_count[rid] = int(host[s, c])
        traced = [seg for chunk in per_ray for seg in chunk]
        for mon in self.monitors:
            record_monitor_hits(mon, traced)
        return traced, total_capped


    def _trace_hooked(self, eng, scene, sources, cls, cap, counts, max_time):
        """Ray trees through a scene with USER-DEFINED components (subclasses that override `interact_local`,
        optical_component.py:235-240).  The trees advance generation by generation: the device finds every queued ray's
        nearest hit among ALL leaves (boxes, grids, count gates, the strict first minimum: `Engine.generation_step`, the
        kernels of every other trace) and emits the children of the built-in leaves; for a ray whose winner is a hooked leaf
        the user's method gets the ray in the leaf's frame and its return is taken to the lab frame — the statements of
        optical_component.py:354-372.  Within a tree, generation order is the reference's FIFO order (optical_table.py:
        115-134): a hit contributes [truncated ray, its dead children ...] to the output and its live children to the queue;
        the cap counts pops per tree and drops what is still queued (:93-98, :138-144).  Deviation, by design: the reference
        calls `interact_local` on every component a ray geometrically hits and discards all results but the nearest's (:119-
        123); here only the nearest hit's is called, which is the same for a method without side effects.
        Returns (per-tree lists of finished rays, number of trees the cap or the clock cut)."""
        import time

        t0 = time.time()
        n = len(sources)
        done = [[] for _ in range(n)]
        pops, cut = [0] * n, [False] * n
        queue = [(r, t) for t, r in enumerate(sources)]
        while queue:
            if time.time() - t0 >= max_time:
                for _, t in queue:
                    cut[t] = True
                break
            live = []
            for r, t in queue:
                if pops[t] < cap:
                    pops[t] += 1
                    live.append((r, t))
            if not live:
                break
            m = len(live)
            eng.upload(scene)  # (a hook may have used the engine for a scene of its own: intersect_point_local)
            batch = _pack([r for r, _ in live], np.asarray(cls)[[t for _, t in live]], eng.device, scene.unit)
            segs, kids, parent = eng.generation_step(batch, counts, tree=[t for _, t in live])  # (a tree's rays meet a count gate in FIFO order)
            surface = segs.surface[:m].cpu().numpy()
            length = segs.length[:m].cpu().numpy()
            kid = {f: kids.field(f).cpu().numpy() for f in abi.RAY_FIELDS} if kids.n else None
            first = np.searchsorted(parent.cpu().numpy(), np.arange(m + 1))  # children come in parent order
            queue = []
            for i, (r, t) in enumerate(live):
                if surface[i] < 0:  # escaped, or dead on input: archived as it is (optical_table.py:133-134)
                    done[t].append(_clone_rays([r])[0])
                    continue
                stub = _clone_rays([r])[0]
                stub.length, stub.alive = float(length[i]), False
                done[t].append(stub)
                comp = scene.hooks.get(int(surface[i]))
                if comp is None:
                    children = [_child_ray(r, kid, j) for j in range(first[i], first[i + 1])]
                else:  # (the children the device emitted for a hooked leaf are its class's built-in physics: the hook may ask for them)
                    children = _call_hook(comp, r, length[i], (lambda r=r, a=first[i], b=first[i + 1]: [_child_ray(r, kid, j) for j in range(a, b)]))
                for c in children:
                    if c.alive:
                        queue.append((c, t))
                    else:
                        done[t].append(c)
        capped = [cut[t] or pops[t] >= cap for t in range(n)]
        if any(capped):
            self._last_trace_num = max(pops[t] for t in range(n) if capped[t])
        return done, int(sum(capped))


# What the device already knows about the ray a hook is being called for: id(local ray) -> (component, local origin, local
# direction, t, built-in children in the lab frame or None).  A hook that starts with `P, t = self.intersect_point_local(ray)`
# (every interact_local upstream does: optical_component.py:543, 624, 936) or calls `super().interact_local(ray)` asks for
# exactly that hit again; answering from here saves a one-ray launch per question (0.6 ms each: 126 -> 50 ms per call on the
# scene of fixture g25).  Only for the very ray object handed to the hook, and only while it still has the origin and direction
# it was handed over with; any other question goes to the device.
_HOOK_MEMO = {}


def _memo_for(comp, ray_local):
    memo = _HOOK_MEMO.get(id(ray_local))
    if memo is None or memo[0] is not comp:
        return None
    if not (np.array_equal(memo[1], ray_local.origin) and np.array_equal(memo[2], ray_local.direction)):
        return None
    return memo


def _call_hook(comp, ray, t=None, builtin_children=None):
    """The user's `interact_local` on a lab-frame ray the device found to hit `comp` first: local frame in, lab frame out
    (optical_component.py:354, 366-372).  `t`: the hit distance the device found; `builtin_children`: a callable giving the
    lab-frame rays the class's own physics emits for this hit (or None)."""
    local = comp.ray_to_local_coordinates(ray)
    if t is not None:
        _HOOK_MEMO[id(local)] = (comp, local.origin.copy(), local.direction.copy(), float(t), builtin_children)
    try:
        out = comp.interact_local(local)
    finally:
        _HOOK_MEMO.pop(id(local), None)
    if out is None:
        return []
    return [comp.ray_to_lab_coordinates(c) for c in out]


def _child_ray(parent, kid, j):
    """Ray object of child `j` of a generation step's output: the parent's clone (it keeps _id, wavelength, unit and user
    attributes, like the copy chain upstream) with the traced fields replaced."""
    from .materials import Material

    child = _clone_rays([parent])[0]
    d = child.__dict__
    d["origin"] = np.array([kid["ox"][j], kid["oy"][j], kid["oz"][j]])
    d["_direction"] = np.array([kid["dx"][j], kid["dy"][j], kid["dz"][j]])
    d["intensity"] = float(kid["intensity"][j])
    d["length"], d["alive"] = None, True
    if parent.qo is not None:
        d["qo"] = complex(kid["q_re"][j], kid["q_im"][j])
    d["_n"] = Material("Constant", n=float(kid["n"][j]))
    d["_pathlength"] = float(kid["pathlength"][j])
    return child


def _clone_rays(rays):
    """What `copy.deepcopy(self.rays)` (optical_table.py:72) yields for ordinary rays, without
    walking every object graph: the arrays are copied, scalars and the (immutable) material are
    shared.  ~20x cheaper than deepcopy, which is the reference's own bottleneck here."""
    out = []
    new = object.__new__
    for r in rays:
        c = new(type(r))
        d = c.__dict__
        d.update(r.__dict__)
        d["origin"] = r.origin.copy()
        d["_direction"] = r._direction.copy()
        out.append(c)
    return out


def _pack(rays, cls, device, scene_unit=1e-2):
    """List[Ray] -> RayBatch (fp64).  The reference evaluates materials at `ray.wavelength * ray.unit`
    — the RAY's unit (optical_component.py:627-628, base.py:31) — while the kernel multiplies by the
    scene's; the wavelength column is rescaled so that both give the same metres for every ray."""
    from .batch import RayBatch

    n = len(rays)
    origin = np.array([r.origin for r in rays], dtype=float).reshape(n, 3)
    direction = np.array([r.direction for r in rays], dtype=float).reshape(n, 3)
    has_q = np.array([r.qo is not None for r in rays])
    q = np.array([complex(r.qo) if r.qo is not None else 0j for r in rays], dtype=np.complex128)
    b = RayBatch.from_arrays(origin, direction,
                             wavelength=[r.wavelength * (getattr(r, "unit", scene_unit) / scene_unit) for r in rays],
                             intensity=[r.intensity for r in rays],
                             q=q, n_index=[r.n for r in rays], pathlength=[r._pathlength for r in rays],
                             ids=cls, device=device, normalize=False)
    import torch

    flags = np.where(has_q, abi.RAY_HAS_Q, 0) | np.where([bool(r.alive) for r in rays], 0, abi.RAY_DEAD)
    b.flags.copy_(torch.from_numpy(flags.astype(np.int32)))
    if any(r.length is not None for r in rays):
        lengths = np.array([np.inf if r.length is None else r.length for r in rays], dtype=float)
        b.length = torch.from_numpy(lengths).to(device)
    return b


def _fused_capped(host_segs, cap):
    """Trees for which the reference prints its cap message: the loop ran `cap` iterations that
    each processed a ray, so `exit_flag` was never set (optical_table.py:93-98, 138-143) — i.e.
    the tree filled all `cap` slots, whether or not a queued ray was actually dropped."""
    return int(np.sum(host_segs["count"] >= cap))


def _scatter_segments(host_segs, sources, pick, per_ray):
    """Rebuild Ray objects, grouped by input ray, in the reference's order.  Every segment starts as a
    shallow clone of its input ray (it inherits _id, wavelength, unit and any user attribute, like the copy
    chain upstream) with the traced fields replaced.  Columns are converted to Python lists once and the
    clone is `__new__` + a dict update: ~2 us per segment instead of ~10 with copy.copy and per-field numpy
    scalars."""
    from .materials import Material

    tree = host_segs["ray"].tolist()
    surface = host_segs["surface"].tolist()
    length = host_segs["length"].tolist()
    intensity = host_segs["intensity"].tolist()
    q_re, q_im = host_segs["q_re"].tolist(), host_segs["q_im"].tolist()
    index, path = host_segs["n"].tolist(), host_segs["pathlength"].tolist()
    origin = np.stack([host_segs["ox"], host_segs["oy"], host_segs["oz"]], axis=1)
    direction = np.stack([host_segs["dx"], host_segs["dy"], host_segs["dz"]], axis=1)
    inf = float("inf")
    constants = {}  # one shared constant Material per distinct index value (Materials are immutable)
    for k in pick:
        per_ray[k] = []
    new = object.__new__
    for s, t in enumerate(tree):
        src = sources[t]
        seg = new(type(src))
        d = seg.__dict__
        d.update(src.__dict__)
        d["origin"] = origin[s].copy()
        d["_direction"] = direction[s].copy()
        d["intensity"] = intensity[s]
        d["length"] = None if length[s] == inf else length[s]
        d["alive"] = surface[s] == -1
        if d.get("qo") is not None:
            d["qo"] = complex(q_re[s], q_im[s])
        n = index[s]
        mat = constants.get(n)
        if mat is None:
            mat = constants[n] = Material("Constant", n=n)
        d["_n"] = mat  # what the RefractiveIndex descriptor would store for a float
        d["_pathlength"] = path[s]
        per_ray[pick[t]].append(seg)


def interact_component(comp, ray, call_hooks=True):
    """`component.interact(ray)` (optical_component.py:337-378; component_group.py:93-122 for groups) through
    the engine: one generation step over a scene made of this component alone.  Returns `(t, [truncated,
    *children])` or `(None, None)`; interact counters of the component (and of a group's children: every
    geometric hit consumes one) are read from and written back to the objects."""
    from .scene import compile_scene as _compile

    if not ray.alive:
        return None, None
    import torch

    eng = _engine()
    scene = _compile([comp])
    batch = _pack([ray], np.zeros(1, dtype=np.int32), eng.device, scene.unit)
    counts = None
    if scene.limited:
        host = np.array([[c._interact_count.get(ray._id, 0)] for c in scene.limited], dtype=np.int32)
        counts = torch.from_numpy(host).to(eng.device)
    with eng.lock:
        eng.upload(scene)
        segs, kids, _ = eng.generation_step(batch, counts)
    if scene.limited:
        after = counts.cpu().numpy()
        for s, c in enumerate(scene.limited):
            if after[s, 0] or ray._id in c._interact_count:
                c._interact_count[ray._id] = int(after[s, 0])
    if int(segs.surface[0].item()) < 0:
        return None, None
    t = float(segs.length[0].item())
    truncated = _clone_rays([ray])[0]
    truncated.length, truncated.alive = t, False
    out = [truncated]
    hook = scene.hooks.get(int(segs.surface[0].item())) if call_hooks else None
    if hook is not None:  # the leaf's physics is the user's Python (a group is asked for the leaf that was hit)
        return t, out + _call_hook(hook, ray, t)
    if kids.n:
        k = {f: kids.field(f).cpu().numpy() for f in abi.RAY_FIELDS}
        for j in range(kids.n):
            child = _clone_rays([ray])[0]
            child.origin = np.array([k["ox"][j], k["oy"][j], k["oz"][j]])
            child._direction = np.array([k["dx"][j], k["dy"][j], k["dz"][j]])
            child.intensity = float(k["intensity"][j])
            child.length, child.alive = None, True
            if ray.qo is not None:
                child.qo = complex(k["q_re"][j], k["q_im"][j])
            child._n = float(k["n"][j])
            child._pathlength = float(k["pathlength"][j])
            out.append(child)
    return t, out


def _pose_free_copy(comp):
    if hasattr(comp, "components"):
        raise NotImplementedError("local-frame calls are defined for leaf components")
    probe = copy.copy(comp)
    probe.origin, probe.transform_matrix = np.zeros(3), np.identity(3)
    probe._bbox, probe.max_interact_count, probe._interact_count = _NO_BOX, None, {}
    return probe


def interact_leaf_local(comp, ray_local):
    """`leaf.interact_local(ray_local)` (optical_component.py:536-570, 617-717, 930-948): the rays a hit emits,
    in the leaf's frame; an empty list when the local ray misses (upstream would fail on `P is None` there)."""
    memo = _memo_for(comp, ray_local)
    if memo is not None and memo[4] is not None:  # asked from inside a hook about the hit the device has just found
        return [comp.ray_to_local_coordinates(c) for c in memo[4]()]
    probe = _pose_free_copy(comp)
    probe._builtin_physics = True  # a user's override that calls super().interact_local() gets the class's own physics here
    _, rays = interact_component(probe, ray_local)
    return [] if rays is None else rays[1:]


def intersect_leaf_local(comp, ray_local):
    """`leaf.intersect_point_local(ray_local)` (optical_component.py:151-233): the ray is already in the
    leaf's frame, so the leaf is traced with an identity pose; count gates do not apply here."""
    memo = _memo_for(comp, ray_local)
    if memo is not None:  # asked from inside a hook about the hit the device has just found
        t = memo[3]
        return np.asarray(ray_local.origin, dtype=float) + t * np.asarray(ray_local.direction, dtype=float), t
    probe = _pose_free_copy(comp)
    t, rays = interact_component(probe, ray_local, call_hooks=False)  # (a user's interact_local asks for the hit point: no recursion)
    if t is None:
        return None, None
    return np.asarray(ray_local.origin, dtype=float) + t * np.asarray(ray_local.direction, dtype=float), t


def monitor_struct(monitor):
    """ot_monitor for a Monitor's current pose."""
    mon = abi.OtMonitor()
    mon.M[:] = np.asarray(monitor.transform_matrix, dtype=float).ravel().tolist()
    mon.origin[:] = np.asarray(monitor.origin, dtype=float).tolist()
    mon.half_width, mon.half_height = monitor.width / 2, monitor.height / 2
    return mon


def record_monitor_hits(monitor, rays):
    """Monitor.record (monitor.py:183-193) through `ot_monitor_record_f64`."""
    if not rays:
        monitor._updated = True
        return
    import torch
    from .batch import SegmentBatch

    eng = _engine()
    n = len(rays)
    segs = SegmentBatch(n, "f64", eng.device)
    cols = {
        "ox": [r.origin[0] for r in rays], "oy": [r.origin[1] for r in rays], "oz": [r.origin[2] for r in rays],
        "dx": [r.direction[0] for r in rays], "dy": [r.direction[1] for r in rays], "dz": [r.direction[2] for r in rays],
        "length": [np.inf if r.length is None else r.length for r in rays],
        "intensity": [r.intensity for r in rays],
    }
    for name, values in cols.items():
        segs.field(name).copy_(torch.tensor(values, dtype=torch.float64))
    segs.n_valid = n
    idx, P, t = eng.monitor_record(monitor_struct(monitor), segs)
    idx, P, t = idx.cpu().numpy(), P.cpu().numpy(), t.cpu().numpy()
    monitor._data_raw.extend([(P[k].copy(), rays[int(i)].intensity, float(t[k]), rays[int(i)]) for k, i in enumerate(idx)])
    monitor._updated = True

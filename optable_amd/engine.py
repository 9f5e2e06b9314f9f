"""Python handle on the HIP trace engine (one `ot_ctx` per device).

Everything that touches rays goes through liboptable_hip.so; this module only moves
pointers.  If the library or a GPU is missing the constructor raises — there is no CPU path.
"""
import ctypes as C
import threading

import torch

from . import abi
from .batch import RayBatch, SegmentBatch

_engines = {}


def append_slots(n_records, launch, chunk=512):
    """Slots an append-layout block needs for `n_records` segment records written by a launch of shape `launch`
    (`Engine.last_launch()`), holes included.  Per-wave lists: every wave may leave the tail of its last chunk unused.
    Block pool (bit 4 of `pair_queue`): a workgroup fills one chunk of 16 x chunk slots at a time, loses at most 63 slots
    where a pass crosses into the next chunk and leaves the tail of its last chunk unused.  A multiple of 64."""
    n_records, chunk = int(n_records), int(chunk)
    if launch["kernel"] == 2 and launch["pair_queue"] & 16:
        wg_chunk = min(16 * chunk, 1 << 19)
        slots = n_records + 64 * (n_records // (wg_chunk - 64) + 1) + (wg_chunk + 64) * max(launch["workgroups"], 1)
    else:
        waves = max(launch["workgroups"] * launch["threads"] // 64, 1) if launch["kernel"] == 2 else 256 * 16
        slots = n_records + chunk * waves
    return (slots + 63) // 64 * 64


class Engine:
    def __init__(self, device=0):
        self.lib = abi.load()
        if not torch.cuda.is_available():
            raise abi.EngineUnavailable("no GPU visible: optable_amd traces only on an MI355X (no CPU fallback)")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self._ctx = C.c_void_p()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        abi.check(self.lib.ot_ctx_create(device, C.c_void_p(stream), C.byref(self._ctx)), self.lib)
        self.scene = None
        self.append_chunk = 512  # OT_OPT_APPEND_CHUNK as last set through set_option (the library's default)
        self.trees_refill_at = 16  # OT_OPT_TREES_REFILL_AT likewise
        self._policy_sets_refill = False
        self._records_per_ray = {}  # (scene, cap, precision) -> records per ray seen in a sample trace (append capacity estimates)
        self._layout_choice = {}    # (scene, precision) -> "slots" | "tiled" measured on the workload itself (tune_layout)
        # An ot_ctx holds one scene and one set of scratch buffers: calls on it are serialised (include/
        # optable_hip.h).  The table-level entry points hold this lock across their upload + trace sequence so that
        # Python threads sharing the engine cannot interleave them (ctypes releases the GIL during a call).
        self.lock = threading.RLock()

    def use_stream(self, stream=None):
        """Launch on `stream` (a torch.cuda.Stream; default: torch's current stream) from now on."""
        handle = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        abi.check(self.lib.ot_ctx_set_stream(self._ctx, C.c_void_p(handle)), self.lib)

    def close(self):
        if self._ctx:
            self.lib.ot_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    # -- scene -----------------------------------------------------------------------------
    def upload(self, scene):
        if scene is self.scene:  # the very object already on the device (a CompiledScene is immutable)
            return
        desc = scene.desc()
        abi.check(self.lib.ot_scene_upload(self._ctx, C.byref(desc)), self.lib)
        self.scene = scene
        self._records_per_ray = {}
        self._layout_choice = {}

    def _refuse_hooks(self):
        """Whole-trace launches cannot call back into Python: a scene with user-defined `interact_local` leaves
        (CompiledScene.hooks) is traced generation by generation by `table.ray_tracing` (table.py: _trace_hooked), where
        the device finds the hits and the user's method says what they emit."""
        if self.scene is not None and getattr(self.scene, "hooks", None):
            from .scene import SceneError

            names = sorted({type(c).__name__ for c in self.scene.hooks.values()})
            raise SceneError(f"{', '.join(names)}: a user-defined interact_local runs on the host; trace such a scene with "
                             "table.ray_tracing (Ray objects) — the batch entry points are whole-trace device launches")

    def _check_wavelengths(self, rays):
        """Scenes with a dispersion SERIES (Material(n = callable), fitted over a wavelength interval: materials.py) say
        nothing outside that interval; the reference would call the function there.  Refuse rather than extrapolate."""
        rng = getattr(self.scene, "wavelength_range", None)
        if rng is None or rays.n == 0:
            return
        wl = rays.wavelength.double() * self.scene.unit
        lo, hi = float(wl.min().item()), float(wl.max().item())
        tol = 1e-12 if rays.precision == "f64" else 1e-6  # (a single-precision wavelength exactly at an edge of the interval has moved by its rounding)
        if lo < rng[0] * (1 - tol) or hi > rng[1] * (1 + tol):
            raise ValueError(f"ray wavelengths {lo:.4g} .. {hi:.4g} m leave the interval {rng[0]:.4g} .. {rng[1]:.4g} m on which the "
                             "scene's dispersion functions have their device form (Material(..., wavelength_range=(lo, hi)) widens it)")

    def plan(self, precision, n_rays, max_segments):
        """What the library would launch for the uploaded scene and such a batch, and the output layout its kernels write
        fastest (include/optable_hip.h: ot_trace_plan).  The first call for a light scene measures the device's stream rate in
        both slot layouts (ot_probe_layouts: ~15 ms, once per engine and precision)."""
        info = (C.c_int32 * 8)()
        abi.check(self.lib.ot_trace_plan(self._ctx, 8 if precision == "f64" else 4, int(n_rays), int(max_segments), C.byref(info)), self.lib)
        return {"kernel": int(info[0]), "tiled_ok": bool(info[1]), "append_limit": 1 << int(info[2]),
                "layout": ("slots", "tiled", "append")[int(info[3])], "probe_us": (info[5] / 100.0, info[6] / 100.0)}

    def tune_layout(self, rays, max_segments, launches=20):
        """Measure THIS workload in both slot layouts and keep the faster one for the uploaded scene: `layout="auto"` then
        takes it instead of the generic stream probe (which times a copy-like kernel on buffers of its own; the real trace on
        the caller's buffers can come out the other way round — the two layouts differ by how their streams fall on the
        memory channels, and that depends on the box and on where the buffers lie).  Light scenes only (heavy ones have one
        dense layout); returns {"slots": us per launch, "tiled": us per launch, "rounds": ..., "chosen": ...}.  Needs room for
        both layouts' outputs at once (2 x n x K records).  A caller who times particular output buffers measures on THOSE
        (bench.py does): where the buffers lie is part of what is being measured."""
        K = int(max_segments)
        plan = self.plan(rays.precision, rays.n, K)
        if plan["kernel"] != 1 or not plan["tiled_ok"] or rays.n == 0:
            return {"chosen": plan["layout"]}
        # Both layouts' buffers live side by side and the rounds alternate, after a pre-load that brings the chip to its steady
        # clocks: measured one after the other from cold, the first layout pays for the clocks (a bench run chose tiles at 119
        # against 130 us that way and then ran 125 against 119 in its timed region).  The best round of a layout counts.
        outs = {lay: SegmentBatch(rays.n * K, rays.precision, rays.device, tiled=(lay == "tiled")) for lay in ("slots", "tiled")}
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for it in range(2000):  # ~60 ms of load, at most 2000 launches
            self.trace(rays, K, out=outs["slots"], layout="slots")
            if it % 4 == 3:
                ev1.record()
                ev1.synchronize()
                if ev0.elapsed_time(ev1) > 60.0:
                    break
        rounds = {"slots": [], "tiled": []}
        for rnd in range(3):
            for layout in (("slots", "tiled") if rnd % 2 == 0 else ("tiled", "slots")):
                for _ in range(3):
                    self.trace(rays, K, out=outs[layout], layout=layout)
                ev0.record()
                for _ in range(launches):
                    self.trace(rays, K, out=outs[layout], layout=layout)
                ev1.record()
                torch.cuda.synchronize()
                rounds[layout].append(ev0.elapsed_time(ev1) / launches * 1e3)
        del outs
        res = {"slots": min(rounds["slots"]), "tiled": min(rounds["tiled"]), "rounds": rounds}
        res["chosen"] = "tiled" if res["tiled"] < res["slots"] else "slots"
        self._layout_choice = {(id(self.scene), rays.precision): res["chosen"]}
        return res

    def probe_layouts(self, precision="f64"):
        """(microseconds per launch into the 14 slot arrays, into 64-slot tiles) of cfg 2's streams on this device."""
        a, b = C.c_double(), C.c_double()
        abi.check(self.lib.ot_probe_layouts(self._ctx, 8 if precision == "f64" else 4, C.byref(a), C.byref(b)), self.lib)
        return a.value, b.value

    # -- non-branching trace ---------------------------------------------------------------
    def trace(self, rays: RayBatch, max_segments, out: SegmentBatch = None, counts=None, layout="slots", capacity=None):
        """All segments of every ray in one launch; returns the SegmentBatch.
        layout="slots": [k][ray] slots (ot_trace_*).  layout="auto": what the library recommends for this scene, batch and
        device (`plan`): "append" for scenes that take the rolling-list / block-pool kernels, for light ones "tiled" or
        "slots", whichever this device streams faster — every reader of a SegmentBatch (to_host, monitors, exports,
        final_state) takes all layouts.  layout="append": a dense list in append order (ot_trace_append_*,
        include/optable_hip.h) — `capacity` slots; default: estimated from a 1 % sample of the batch (records per ray x
        1.15 + the launch's chunk slack; the rare batch that needs more is traced again into a block of the size the
        first launch reported); pass what the job needs to skip the estimate: if that turns out too small a RuntimeError
        names the size that fits."""
        self._refuse_hooks()
        if self.scene is None:
            raise RuntimeError("upload a scene first")
        n, K = rays.n, int(max_segments)
        self._check_wavelengths(rays)
        if layout == "auto":  # what this scene's kernels write fastest ON THIS DEVICE: a measurement of this very workload
            # (tune_layout) where there is one, else the library's own rule (ot_trace_plan: a generic stream probe for light scenes)
            tuned = self._layout_choice.get((id(self.scene), rays.precision))
            plan = self.plan(rays.precision, n, K) if n else None
            layout = "slots" if plan is None else (tuned if (tuned and plan["kernel"] == 1 and plan["tiled_ok"]) else plan["layout"])
        if layout == "append":
            return self._trace_append(rays, K, out, counts, capacity)
        if layout == "tiled":
            return self._trace_tiled(rays, K, out, counts)
        if layout != "slots":
            raise ValueError(f"unknown layout {layout!r}")
        if out is None:
            out = SegmentBatch(n * K, rays.precision, rays.device)
        elif out.capacity < n * K or out.precision != rays.precision:
            raise ValueError("output SegmentBatch too small or of the wrong precision")
        if out.count is None or out.count.numel() != n:
            out.count = torch.empty(n, dtype=torch.int32, device=rays.device)
        out.n_rays = n
        out.counts_table = counts
        if n == 0:  # nothing to launch (zero-size tensors have no address to hand over)
            return out
        n_slots = len(self.scene.limited)
        if n_slots and counts is None:
            counts = torch.zeros((n_slots, n), dtype=torch.int32, device=rays.device)
        n_classes = 0 if counts is None else counts.shape[1]
        fn = self.lib.ot_trace_f64 if rays.precision == "f64" else self.lib.ot_trace_f32
        rs, ss = rays.c_struct(), out.c_struct()
        abi.check(fn(self._ctx, C.byref(rs), n, K, C.byref(ss), out.count.data_ptr(),
                     None if counts is None else counts.data_ptr(), n_classes), self.lib)
        out.counts_table = counts
        return out

    def _trace_tiled(self, rays, K, out, counts):
        """[k][ray] slots in 64-slot tiles (ot_trace_tiled_*): light scenes, the layout the HBM streams like best."""
        n = rays.n
        if out is None:
            out = SegmentBatch(n * K, rays.precision, rays.device, tiled=True)
        elif not out.tiled or out.capacity < n * K or out.precision != rays.precision:
            raise ValueError("tiled layout needs a SegmentBatch(tiled=True) of the rays' precision with max_segments * n slots")
        if out.count is None or out.count.numel() != n:
            out.count = torch.empty(n, dtype=torch.int32, device=rays.device)
        out.n_rays, out.counts_table = n, counts
        if n == 0:
            return out
        n_slots_table = len(self.scene.limited)
        if n_slots_table and counts is None:
            counts = torch.zeros((n_slots_table, n), dtype=torch.int32, device=rays.device)
        n_classes = 0 if counts is None else counts.shape[1]
        fn = self.lib.ot_trace_tiled_f64 if rays.precision == "f64" else self.lib.ot_trace_tiled_f32
        rs = rays.c_struct()
        abi.check(fn(self._ctx, C.byref(rs), n, K, out.block.data_ptr(), out.capacity, out.count.data_ptr(),
                     None if counts is None else counts.data_ptr(), n_classes), self.lib)
        out.counts_table = counts
        return out

    def append_capacity(self, n_records):
        """Slots an append-layout block needs for `n_records` segment records on this device: every wave of the launch
        may leave the tail of its last chunk (512 slots) unused.  Call after a trace of the same scene (the launch shape
        is taken from it); `sum(|count|)` of that trace is the record count."""
        return append_slots(n_records, self.last_launch(), self.append_chunk)

    MAX_WAVES = 256 * 32  # most waves a launch can have resident (256 CUs x 8 per SIMD): each may leave one chunk's tail unused

    def _append_worst_case(self, n, K):
        """Slots that hold the records of ANY trace of n rays capped at K segments, holes included: the tail of every wave's last
        chunk (one wave per ticket of 64 rays at most); with the block pool (curved scenes, fp32) 63 slots per workgroup chunk
        and the last chunk of every workgroup."""
        wg_chunk = min(16 * self.append_chunk, 1 << 19)
        waves = min(self.MAX_WAVES, (n + 63) // 64 + 1)
        return (n * K + 64 * (n * K // (wg_chunk - 64) + 1)
                + max(self.append_chunk * waves, (wg_chunk + 64) * min((n + 1023) // 1024, 512)))

    def _append_estimate(self, rays, K, counts):
        """Capacity for an append trace without tracing the batch twice: records per ray from a strided 1 % sample (its own
        small trace), x 1.15, plus the slack of the launch; remembered per scene and cap, so the next batch skips the sample."""
        n = rays.n
        worst = self._append_worst_case(n, K)
        if n * K <= 1 << 22 or counts is not None:  # small: the worst case costs nothing; count tables: one ray per id per launch, never sampled
            return worst, None
        key = (id(self.scene), K, rays.precision)
        rpr = self._records_per_ray.get(key)
        if rpr is None:
            m = max(n // 100, 4096)
            idx = torch.arange(0, n, max(n // m, 1), device=rays.device)
            sample = self.trace(rays.take(idx), K, layout="append", capacity=self._append_worst_case(int(idx.numel()), K))
            rpr = float(sample.count.abs().sum().item()) / float(idx.numel())
            self._records_per_ray = {key: rpr}  # (one scene at a time)
        wg_chunk = min(16 * self.append_chunk, 1 << 19)
        slack = max(self.append_chunk * self.MAX_WAVES, (wg_chunk + 64) * 256)
        est = int(n * rpr * 1.15) + int(n * rpr * 1.15) // 128 + slack
        return (min(est, worst) + 63) // 64 * 64, rpr

    def _trace_append(self, rays, K, out, counts, capacity):
        n = rays.n
        estimated = False
        if capacity is None and out is None:
            capacity, rpr = self._append_estimate(rays, K, counts)
            estimated = rpr is not None
        elif capacity is None:
            capacity = out.capacity
        if out is None:
            out = SegmentBatch(capacity, rays.precision, rays.device, block=True)
        elif out.block is None or out.tiled or out.precision != rays.precision:
            raise ValueError("append layout needs a SegmentBatch(block=True) of the rays' precision")
        out.tiled = False
        if out.count is None or out.count.numel() != n:
            out.count = torch.empty(n, dtype=torch.int32, device=rays.device)
        out.n_rays, out.append, out.n_valid, out.counts_table = n, True, 0, counts
        if n == 0:
            return out
        n_slots_table = len(self.scene.limited)
        if n_slots_table and counts is None:
            counts = torch.zeros((n_slots_table, n), dtype=torch.int32, device=rays.device)
        n_classes = 0 if counts is None else counts.shape[1]
        fn = self.lib.ot_trace_append_f64 if rays.precision == "f64" else self.lib.ot_trace_append_f32
        cursor = torch.zeros(1, dtype=torch.int64, device=rays.device)
        rs, blk = rays.c_struct(), out.block_struct()
        abi.check(fn(self._ctx, C.byref(rs), n, K, C.byref(blk), cursor.data_ptr(), out.count.data_ptr(),
                     None if counts is None else counts.data_ptr(), n_classes), self.lib)
        out.cursor = cursor  # device scalar: read lazily (n_valid) so that back-to-back launches do not synchronise
        out.n_valid = None
        out.counts_table = counts
        if estimated:  # an estimated block can be too small: look (one 8-byte read-back), and trace again into what the launch asked for
            for _ in range(3):
                need = int(cursor.item())
                if need <= out.capacity or need >= 1 << 62:
                    break
                # (plus room for the holes to fall differently: which wave claims which chunk is not the same from run to run)
                self._records_per_ray = {}
                del out
                out = self._trace_append(rays, K, None, counts, (int(need * 1.02) + (1 << 20) + 63) // 64 * 64)
                cursor = out.cursor
        return out

    # -- branching trace: breadth-first, one generation per launch ------------------------------
    def trees_plan(self, precision, max_trace_num, n_rays=0):
        """ot_trace_trees_plan: does the uploaded scene have a lane-per-tree kernel (k_trace_trees), how many queue entries
        per lane would a cap of `max_trace_num` get for a batch of `n_rays` trees (0: one that fills the device), and is that
        enough for every possible tree."""
        info = (C.c_int32 * 8)()
        abi.check(self.lib.ot_trace_trees_plan(self._ctx, 8 if precision == "f64" else 4, int(max_trace_num), int(n_rays), info), self.lib)
        return {"kernel": bool(info[0] & 1), "slots": bool(info[0] & 2), "queue": int(info[1]), "full": bool(info[2]), "lds_entries": int(info[3]),
                "chunk": int(info[4]), "waves": int(info[5])}

    def trace_trees(self, rays: RayBatch, max_trace_num, out: SegmentBatch = None, counts=None, layout="slots", capacity=None):
        """Whole ray trees in one launch, a lane per tree with its FIFO on the chip (ot_trace_trees_*).
        layout "slots": a SegmentBatch in the `slots` layout — slot k * n + i = the k-th ray of tree i in the reference's order;
        layout "append": the dense list of ot_trace_append_* (`capacity` slots, default: what any trace of the batch can need) —
        a stable sort by `ray` is the reference's order; the layout for batches whose trees differ widely in size.
        count[i] = rays of tree i (negative: the tree's queue overflowed, possible only when `trees_plan()["full"]` is False;
        take trace_tree then); `capped` as trace_tree reports it.  `counts`: the interact-count table of scenes with limited
        surfaces ([surfaces, classes], rays.id = column; default: a zeroed column per ray) — exact when no two trees of the
        call share a column."""
        self._refuse_hooks()
        if self.scene is None:
            raise RuntimeError("upload a scene first")
        self._check_wavelengths(rays)
        prec, n, K = rays.precision, rays.n, int(max_trace_num)
        n_slots = len(self.scene.limited)
        if n_slots and counts is None:
            counts = torch.zeros((n_slots, max(n, 1)), dtype=torch.int32, device=rays.device)
        n_classes = 0 if counts is None else counts.shape[1]
        cp = None if counts is None else counts.data_ptr()
        rs = rays.c_struct()
        if layout == "append":
            if out is None:
                if capacity is None:  # every tree at its cap + the tail of every wave's last chunk
                    plan = self.trees_plan(prec, K, n)
                    capacity = (n * K + plan["chunk"] * plan["waves"] + 63) // 64 * 64
                out = SegmentBatch(capacity, prec, rays.device, block=True)
            elif out.block is None or out.tiled or out.precision != prec:
                raise ValueError("append layout needs a SegmentBatch(block=True) of the rays' precision")
            out.count = torch.empty(n, dtype=torch.int32, device=rays.device)
            out.n_rays, out.append, out.tiled = n, True, False
            cursor = torch.zeros(1, dtype=torch.int64, device=rays.device)
            if n:
                fn = self.lib.ot_trace_trees_append_f64 if prec == "f64" else self.lib.ot_trace_trees_append_f32
                blk = out.block_struct()
                abi.check(fn(self._ctx, C.byref(rs), n, K, C.byref(blk), cursor.data_ptr(), out.count.data_ptr(), cp, n_classes), self.lib)
            out.cursor, out.n_valid = cursor, None  # device scalar: read lazily (n_valid)
        elif layout == "slots":
            if out is None:
                out = SegmentBatch(n * K, prec, rays.device)
            if out.capacity < n * K or out.precision != prec or out.tiled or out.block is not None:
                raise ValueError("out: plain slot arrays of max_trace_num * n_rays slots in the rays' precision")
            out.count = torch.empty(n, dtype=torch.int32, device=rays.device)
            out.n_rays = n
            fn = self.lib.ot_trace_trees_f64 if prec == "f64" else self.lib.ot_trace_trees_f32
            ss = out.c_struct()
            abi.check(fn(self._ctx, C.byref(rs), n, K, C.byref(ss), out.count.data_ptr(), cp, n_classes), self.lib)
        else:
            raise ValueError("layout: 'slots' or 'append'")
        out.capped = out.count >= K
        out.timed_out = False
        out.counts_table = counts
        out.trees = True
        return out

    LANE_PER_TREE = True  # False: trace_branching always takes the generation loop (A/B measurements)
    HINT_SECONDS = 1.0  # MIN_HINTING_TIME of the reference's loop (optical_table.py:85): a longer trace reports its progress at this interval

    # A lane-per-tree launch whose queues may overflow (caps beyond ~170 in double precision) is a speculation on small trees:
    # small batches only, and it may allocate this many slots at most.
    TREES_SMALL_BATCH = 64 * 256
    TREES_SPECULATIVE_SLOTS = 1 << 22  # (440 MB of records in double precision; 40 rays under the reference's largest example's cap of 1e5)

    def _trees_estimate(self, rays, K):
        """Rays per tree of a large batch, and how its waves should refill, from a strided 1 % sample (a lane-per-tree launch of
        its own), remembered per scene, cap and precision.  Rays per tree sizes the append block (x 1.15 + the launch's slack
        instead of every tree at its cap).  The spread of the tree sizes picks OT_OPT_TREES_REFILL_AT:
        trees that differ moderately (standard deviation below 0.35 of the mean: cfg 4 with R = 0.2 without a binding cap, 13-30
        rays) are traced 64 to a wave, in step — 1.9 instead of 2.6 ms on 3.2e6 of them; batches of mostly tiny trees, or of trees of
        every size up to the cap, keep their lanes busy one by one (1.33 vs 2.13 and 3.2 vs 3.8 ms) (kernels.h)."""
        n = rays.n
        key = ("trees", id(self.scene), K, rays.precision)
        known = self._records_per_ray.get(key)
        if known is None:
            m = max(n // 100, 4096)
            sample = rays.take(torch.arange(0, n, max(n // m, 1), device=rays.device))
            sizes = self.trace_trees(sample, K, layout="append").count.abs().double()
            rpr = float(sizes.mean().item())
            even = float(sizes.std(unbiased=False).item()) < 0.35 * rpr
            known = (rpr, 64 if even else 16)
            self._records_per_ray = {key: known}  # (one scene at a time)
        self._policy_sets_refill = True
        try:
            self.set_option(abi.OPT_TREES_REFILL_AT, known[1])
        finally:
            self._policy_sets_refill = False
        return known[0]

    def trace_branching(self, rays: RayBatch, max_trace_num, counts=None, max_trace_time=None, distinct_ids=None):
        """Ray trees by whichever path the scene and the cap allow: ONE launch with a lane per tree (`trace_trees`) when the
        scene has such a kernel and its queues hold every possible tree — or, for larger caps, speculatively when the batch
        is small (a tree that overflows its queue sends the call to the generations) — else the generation loop
        (`trace_tree`: a list in generation order).  The lane-per-tree launch writes [k][tree] slots (the reference's order
        as they lie) for small batches of planar scenes and the dense append list otherwise (whole lines per field however
        much the trees differ in size; its block sized from a 1 % sample of a large batch, traced again into what the launch
        asked for should that be too small); readers take all layouts; `capped` / `timed_out` are set either way.
        Scenes with count-limited surfaces: the lane-per-tree kernel meets them in each tree's FIFO order, which is the
        reference's as long as no two trees of the call share a column of `counts` (`distinct_ids=True`: the caller vouches
        for it, as the host API's rounds do; default: such scenes take the generations)."""
        self._refuse_hooks()
        n, K = rays.n, int(max_trace_num)
        gates_ok = not self.scene.limited or bool(distinct_ids)
        if n and gates_ok and self.LANE_PER_TREE and (max_trace_time is None or max_trace_time > 1.0):
            plan = self.trees_plan(rays.precision, K, n)
            small = n <= self.TREES_SMALL_BATCH
            if plan["kernel"] and (plan["full"] or (small and n * K <= self.TREES_SPECULATIVE_SLOTS)):
                before = counts.clone() if (counts is not None and not plan["full"]) else None  # a speculation must not leave counts behind
                if plan["slots"] and small:
                    segs = self.trace_trees(rays, K, counts=counts, layout="slots")
                elif n * K <= 1 << 22 or self.scene.limited:  # small: the worst case costs nothing; count tables are never sampled
                    segs = self.trace_trees(rays, K, counts=counts, layout="append")
                else:
                    rpr = self._trees_estimate(rays, K)  # (also sets how the waves of this batch refill)
                    if plan["slots"] and rays.precision == "f32" and rpr >= 0.9 * K:
                        # nearly every tree runs into the cap: lanes stay in step, [k][tree] rows are whole lines and cost no claims — in
                        # single precision, where a step is short: 0.29 vs 0.42 ms on 1e6 bushy trees under a cap of 12 (double: 0.57
                        # either way, cfg 4 R = 0.2 4.29 vs 4.07 for the dense list)
                        try:
                            return self.trace_trees(rays, K, layout="slots")
                        finally:
                            self.set_option(abi.OPT_TREES_REFILL_AT, self.trees_refill_at)
                    slack = plan["chunk"] * plan["waves"]
                    capacity = min(int(n * rpr * 1.15) + slack, n * K + slack)
                    try:
                        for _ in range(3):
                            segs = self.trace_trees(rays, K, layout="append", capacity=(capacity + 63) // 64 * 64)
                            need = int(segs.cursor.item())
                            if need <= segs.capacity:
                                break
                            del segs
                            capacity = int(need * 1.02) + (1 << 20)  # (holes fall differently from run to run)
                    finally:
                        self.set_option(abi.OPT_TREES_REFILL_AT, self.trees_refill_at)  # (the default for calls that do not sample)
                if plan["full"] or not bool((segs.count < 0).any()):
                    return segs
                del segs
                if before is not None:
                    counts.copy_(before)
        return self.trace_tree(rays, K, counts=counts, max_trace_time=max_trace_time)

    def trace_tree(self, rays: RayBatch, max_trace_num, counts=None, out_capacity=None, max_trace_time=None):
        """Full ray trees (beam splitters, partial reflections, any cap).  Returns a flat
        SegmentBatch in generation order plus, per tree, whether a cap cut it short.
        `max_trace_time` (seconds): the reference's wall-clock cap (optical_table.py:84-97, `perfomance_limit
        ["max_trace_time"]`), honoured at generation granularity — the clock is read after every generation of
        the whole batch, and once it has run out the rays still queued are dropped, as the reference drops a
        tree's queue (`:138-144`); the trees they belonged to are reported in `capped` (`timed_out` says why)."""
        self._refuse_hooks()
        import time

        t_start = time.time()
        if self.scene is None:
            raise RuntimeError("upload a scene first")
        prec = rays.precision
        self._check_wavelengths(rays)
        gen_fn = self.lib.ot_trace_generation_f64 if prec == "f64" else self.lib.ot_trace_generation_f32
        dev, n = rays.device, rays.n
        if n == 0:
            out = SegmentBatch(0, prec, dev)
            out.n_valid, out.counts_table = 0, counts
            out.capped = torch.zeros(0, dtype=torch.bool, device=dev)
            out.timed_out = False
            return out
        fan = max(self.scene.max_children, 1)
        if out_capacity is None:
            out_capacity = max(4 * n, 1024)
        out = SegmentBatch(out_capacity, prec, dev)
        budget = torch.full((n,), int(max_trace_num), dtype=torch.int32, device=dev)
        state = torch.zeros(2, dtype=torch.int64, device=dev)  # [segment cursor, rays in the next generation]
        tree = torch.arange(n, dtype=torch.int32, device=dev)
        n_slots = len(self.scene.limited)
        if n_slots and counts is None:
            counts = torch.zeros((n_slots, n), dtype=torch.int32, device=dev)
        n_classes = 0 if counts is None else counts.shape[1]
        # The generation loop runs inside the library (ot_trace_tree_*: one 16-byte read-back per generation); it comes back
        # when the queue is empty, when the time is up, or when the pending generation needs more room — then the
        # buffers are grown here and the call repeated with that generation as input.
        tree_fn = self.lib.ot_trace_tree_f64 if prec == "f64" else self.lib.ot_trace_tree_f32
        cap = max(n * fan, 1024)
        bufs = [RayBatch(cap, prec, dev, initialise=False), RayBatch(cap, prec, dev, initialise=False)]
        trees = [torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev)]
        result = (C.c_int64 * 5)()
        cur, cur_n, written = rays, n, 0
        timed_out = None
        # The library comes back at least once per HINT seconds: a trace that runs longer says so once per second, as the
        # reference does per input ray (optical_table.py:99-111; here the clock is the batch's and the counts are the batch's).
        HINT = self.HINT_SECONDS
        while cur_n > 0:
            left_total = None if max_trace_time is None else max(max_trace_time - (time.time() - t_start), 0.0)
            left = HINT * 1.01 if left_total is None else min(left_total, HINT * 1.01)  # (just over the interval: chains of small generations need max_seconds > 1)
            rs, ss, sa, sb = cur.c_struct(), out.c_struct(), bufs[0].c_struct(), bufs[1].c_struct()
            abi.check(tree_fn(self._ctx, C.byref(rs), tree.data_ptr(), cur_n, budget.data_ptr(), C.byref(ss), out.capacity,
                              state.data_ptr(), C.byref(sa), trees[0].data_ptr(), C.byref(sb), trees[1].data_ptr(), bufs[0].n,
                              None if counts is None else counts.data_ptr(), n_classes, left, result), self.lib)
            written, cur_n, where, _, reason = (int(x) for x in result)
            if where:  # the pending generation sits in one of the buffers
                cur, tree = bufs[where - 1].slice(0, cur_n), trees[where - 1][:cur_n]
                if where == 1:  # a call writes its first generation into buf_a: the pending one must be in the other buffer
                    bufs.reverse()
                    trees.reverse()
            if cur_n == 0:
                break
            if reason == 3:
                elapsed = time.time() - t_start
                if max_trace_time is None or elapsed < max_trace_time:  # a hint interval ended, not the caller's time
                    print("Tracing... Time elapsed: {:.2f} s, Trace num: {}, Alive rays: {}, Dead rays: {}".format(elapsed, written, cur_n, written))
                    continue
                timed_out = torch.zeros(n, dtype=torch.bool, device=dev)
                timed_out[tree.long()] = True  # trees with rays still queued
                break
            if reason == 1:
                out = _grow(out, max(written + cur_n, 2 * out.capacity), written)
            elif reason == 2:  # new, larger buffers; the pending generation stays where it is until the next call has read it
                cap = max(cur_n * fan, 2 * bufs[0].n)
                keep = (cur, tree)  # noqa: F841 - holds the old buffer alive across the call
                bufs = [RayBatch(cap, prec, dev, initialise=False), RayBatch(cap, prec, dev, initialise=False)]
                trees = [torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev)]
        out.n_valid = int(written)
        out.counts_table = counts
        out.capped = budget <= 0  # cap reached: queued rays were dropped (optical_table.py:138-144)
        out.timed_out = timed_out is not None
        if timed_out is not None:
            out.capped = out.capped | timed_out
        return out

    def generation_step(self, rays: RayBatch, counts=None, tree=None):
        """ONE generation of the breadth-first trace: every input ray is processed once (nearest hit,
        interaction).  Returns (segments, children, parent): a SegmentBatch with one record per input ray
        (slot i = ray i), the emitted rays as a RayBatch in parent order then child order, and for each of them
        the index of its parent.  This is `component.interact(ray)` for a batch (optical_component.py:337-378).
        `tree` (int32 per ray, ascending: a generation lists its rays tree by tree): which rays belong to one ray tree — the
        rays of a tree meet a count-limited surface one after the other in input order, as the reference's FIFO makes them
        (optical_component.py:140-149, 359-362); default: every ray a tree of its own."""
        if self.scene is None:
            raise RuntimeError("upload a scene first")
        prec = rays.precision
        self._check_wavelengths(rays)
        gen_fn = self.lib.ot_trace_generation_f64 if prec == "f64" else self.lib.ot_trace_generation_f32
        dev, n = rays.device, rays.n
        out = SegmentBatch(n, prec, dev)
        out.n_valid = 0
        if n == 0:
            return out, RayBatch(0, prec, dev), torch.zeros(0, dtype=torch.int32, device=dev)
        fan = max(self.scene.max_children, 1)
        state = torch.zeros(2, dtype=torch.int64, device=dev)
        if tree is None:
            tree = torch.arange(n, dtype=torch.int32, device=dev)
            budget = torch.full((n,), 2, dtype=torch.int32, device=dev)  # (one ray per tree, room for one more: the children are wanted)
        else:
            tree = torch.as_tensor(tree, dtype=torch.int32, device=dev).contiguous()
            if tree.numel() != n or (n > 1 and bool((tree[1:] < tree[:-1]).any())) or int(tree[0]) < 0:
                raise ValueError("tree: one non-negative id per ray, ascending")
            budget = torch.full((int(tree[-1]) + 1,), n + 1, dtype=torch.int32, device=dev)  # (every ray processed, children wanted)
        nxt = RayBatch(n * fan, prec, dev, initialise=False)
        nxt_tree = torch.empty(n * fan, dtype=torch.int32, device=dev)
        n_slots = len(self.scene.limited)
        if n_slots and counts is None:
            counts = torch.zeros((n_slots, n), dtype=torch.int32, device=dev)
        n_classes = 0 if counts is None else counts.shape[1]
        rs, ss, ns = rays.c_struct(), out.c_struct(), nxt.c_struct()
        self.set_option(abi.OPT_GEN_PARENT_INDEX, 1)  # nxt_tree = the parent's index, whatever `tree` groups
        try:
            abi.check(gen_fn(
                self._ctx, C.byref(rs), tree.data_ptr(), n, budget.data_ptr(), C.byref(ss), out.capacity,
                state.data_ptr(), C.byref(ns), nxt_tree.data_ptr(), nxt.n, state.data_ptr() + 8,
                None if counts is None else counts.data_ptr(), n_classes), self.lib)
        finally:
            self.set_option(abi.OPT_GEN_PARENT_INDEX, 0)
        written, n_next = state.tolist()
        out.n_valid, out.counts_table = int(written), counts
        kids = nxt.slice(0, int(n_next))
        # bits 8.. of the flags name the node a child starts on IN THIS SCENE (include/optable_hip.h); the batch is handed to
        # the caller, who may trace it through any scene: drop them (the start-plane rule then falls back to the |t| guard)
        kids.flags.bitwise_and_(0xff)
        return out, kids, nxt_tree[: int(n_next)]

    # -- monitors -------------------------------------------------------------------------------
    def monitor_record(self, monitor_struct, segs: SegmentBatch, n_segments=None):
        """Device pass of Monitor.record over a SegmentBatch.  Returns (slot index, P_local [h,3], t)
        in ascending slot order.  For the [k][ray] layout every slot is scanned and unused ones
        are skipped on the device."""
        with self.lock:
            return self._monitor_record(monitor_struct, segs, n_segments)

    def _monitor_record(self, monitor_struct, segs, n_segments):
        dev = segs.device
        segs = segs.to_slots()  # (a tiled history: one copy into slot arrays)
        if segs.precision != "f64":  # the monitor pass is fp64: widen an fp32 history once
            segs = segs.astype("f64")
        if segs.layout == "slots":
            n_segments, count_ptr, n_rays = segs.capacity // segs.n_rays * segs.n_rays, segs.count.data_ptr(), segs.n_rays
        else:
            n_segments = segs.n_valid if n_segments is None else n_segments
            count_ptr, n_rays = None, (-1 if segs.append else 0)
        idx = torch.empty(n_segments, dtype=torch.int64, device=dev)
        P = [torch.empty(n_segments, dtype=torch.float64, device=dev) for _ in range(3)]
        t = torch.empty(n_segments, dtype=torch.float64, device=dev)
        nh = torch.zeros(1, dtype=torch.int64, device=dev)
        ss = segs.c_struct()
        abi.check(self.lib.ot_monitor_record_f64(self._ctx, C.byref(monitor_struct), C.byref(ss), n_segments, count_ptr, n_rays,
                                                 idx.data_ptr(), P[0].data_ptr(), P[1].data_ptr(), P[2].data_ptr(),
                                                 t.data_ptr(), nh.data_ptr()), self.lib)
        k = int(nh.item())
        return idx[:k], torch.stack([p[:k] for p in P], dim=1), t[:k]

    # -- measurement ---------------------------------------------------------------------------
    def timing(self, enabled=True):
        abi.check(self.lib.ot_timing_enable(self._ctx, int(enabled)), self.lib)
        abi.check(self.lib.ot_timing_reset(self._ctx), self.lib)

    def timing_reset(self):
        abi.check(self.lib.ot_timing_reset(self._ctx), self.lib)

    def timing_read(self):
        ms, cnt = C.c_double(), C.c_int64()
        abi.check(self.lib.ot_timing_read(self._ctx, C.byref(ms), C.byref(cnt)), self.lib)
        return ms.value, cnt.value

    def set_option(self, option, value):
        abi.check(self.lib.ot_set_option(self._ctx, option, value), self.lib)
        if option == abi.OPT_APPEND_CHUNK:
            self.append_chunk = int(value)  # (the capacity estimates of the append layout count holes in chunks)
        if option == abi.OPT_TREES_REFILL_AT and not self._policy_sets_refill:
            self.trees_refill_at = int(value)  # (what the caller asked for: trace_branching's own choice for a batch is undone after it)

    def stream_ceiling(self, rays: RayBatch, max_segments, out: SegmentBatch):
        """Same bytes as `trace` with no tracing (roofline companion), in the layout of `out` (slots or tiled)."""
        if out.count is None or out.count.numel() != rays.n:
            out.count = torch.empty(rays.n, dtype=torch.int32, device=rays.device)
        if out.tiled:
            fn = self.lib.ot_bench_stream_tiled_f64 if rays.precision == "f64" else self.lib.ot_bench_stream_tiled_f32
            rs = rays.c_struct()
            abi.check(fn(self._ctx, C.byref(rs), rays.n, int(max_segments), out.block.data_ptr(), out.capacity, out.count.data_ptr()), self.lib)
            return
        rs, ss = rays.c_struct(), out.c_struct()
        fn = self.lib.ot_bench_stream_f64 if rays.precision == "f64" else self.lib.ot_bench_stream_f32
        abi.check(fn(self._ctx, C.byref(rs), rays.n, int(max_segments), C.byref(ss),
                                               out.count.data_ptr()), self.lib)

    def last_launch(self):
        """Shape of the last trace launch (include/optable_hip.h: ot_debug_last_launch) as a dict."""
        info = (C.c_int32 * 8)()
        abi.check(self.lib.ot_debug_last_launch(self._ctx, C.byref(info)), self.lib)
        keys = ("kernel", "threads", "workgroups_per_cu", "workgroups", "lds_bytes", "list_cap", "mixed", "pair_queue")  # pair_queue: bit flags, see the header
        return dict(zip(keys, (int(v) for v in info)))

    def generation_mismatches(self):
        """Diagnostic counter of the two-pass generation kernels (include/optable_hip.h); expected 0."""
        v = C.c_int64()
        abi.check(self.lib.ot_debug_generation_mismatches(self._ctx, C.byref(v)), self.lib)
        return v.value

    def synchronize(self):
        abi.check(self.lib.ot_ctx_synchronize(self._ctx), self.lib)


def _grow(segs, capacity, n_keep):
    big = SegmentBatch(capacity, segs.precision, segs.device)
    for f in abi.SEG_FIELDS + ("ray", "surface"):
        big.field(f)[:n_keep].copy_(segs.field(f)[:n_keep])
    return big


def get_engine(device=None) -> Engine:
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device not in _engines:
        _engines[device] = Engine(device)
    return _engines[device]

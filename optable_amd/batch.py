"""Structure-of-arrays containers for ray and segment streams.

torch tensors are used only as the device-array container (allocation, dtype, device);
every field is a separate contiguous 1-D tensor so a wave of 64 lanes reads/writes 64
consecutive elements per field (coalesced 512 B at fp64, 256 B at fp32).  Layouts mirror
`ot_rays` / `ot_segments` in include/optable_hip.h.
"""
import numpy as np
import torch

from . import abi

_REAL = {"f64": torch.float64, "f32": torch.float32}


def real_dtype(precision):
    return _REAL[precision]


class RayBatch:
    """N rays (reference: N `Ray` objects, ray.py:63-104)."""

    def __init__(self, n, precision="f64", device="cuda", initialise=True):
        self.n, self.precision, self.device = int(n), precision, torch.device(device)
        dt = _REAL[precision]
        if not initialise:  # scratch the engine overwrites (generation ping-pong buffers): no fills
            for f in abi.RAY_FIELDS:
                setattr(self, "n_index" if f == "n" else f, torch.empty(self.n, dtype=dt, device=self.device))
            self.id = torch.empty(self.n, dtype=torch.int32, device=self.device)
            self.flags = torch.empty(self.n, dtype=torch.int32, device=self.device)
            self.length = None
            return
        for f in abi.RAY_FIELDS:
            if f != "n":
                setattr(self, f, torch.zeros(self.n, dtype=dt, device=self.device))
        self.intensity.fill_(1.0)
        self.n_index = torch.ones(self.n, dtype=dt, device=self.device)
        self.id = torch.arange(self.n, dtype=torch.int32, device=self.device)
        self.flags = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        self.length = None  # optional finite input lengths (+inf = None)

    # `n` names both the ray count and the refractive-index field in the C struct; the tensor
    # lives in `n_index` on the Python side.
    def field(self, name):
        return self.n_index if name == "n" else getattr(self, name)

    @classmethod
    def from_arrays(cls, origin, direction, wavelength=0.0, intensity=1.0, q=None, n_index=1.0,
                    pathlength=0.0, ids=None, precision="f64", device="cuda", normalize=True):
        """Build from host arrays: origin/direction (N,3); scalars broadcast."""
        origin = np.asarray(origin, dtype=np.float64).reshape(-1, 3)
        direction = np.asarray(direction, dtype=np.float64).reshape(-1, 3)
        if normalize:  # Ray.direction setter normalises (ray.py:115-119)
            direction = direction / np.linalg.norm(direction, axis=1, keepdims=True)
        nrays = origin.shape[0]
        dt = _REAL[precision]
        # one staging block [12, stride] -> ONE host-to-device copy; the fields are its rows (contiguous 1-D views,
        # each starting on a 256-byte boundary)
        stride = (nrays + 31) // 32 * 32
        host = np.zeros((len(abi.RAY_FIELDS), stride), dtype=np.float64)[:, :nrays]
        row = {f: k for k, f in enumerate(abi.RAY_FIELDS)}
        host[row["ox"]], host[row["oy"]], host[row["oz"]] = origin[:, 0], origin[:, 1], origin[:, 2]
        host[row["dx"]], host[row["dy"]], host[row["dz"]] = direction[:, 0], direction[:, 1], direction[:, 2]
        host[row["wavelength"]] = wavelength
        host[row["intensity"]] = intensity
        host[row["n"]] = n_index
        host[row["pathlength"]] = pathlength
        if q is not None:
            qc = np.broadcast_to(np.asarray(q, dtype=np.complex128), (nrays,))
            host[row["q_re"]], host[row["q_im"]] = qc.real, qc.imag
        else:
            host[row["q_re"]] = host[row["q_im"]] = 0.0
        dev = torch.device(device)
        block = torch.from_numpy(host.base if host.base is not None else host).to(dt).to(dev)[:, :nrays]
        b = object.__new__(cls)
        b.n, b.precision, b.device = int(nrays), precision, dev
        for f, k in row.items():
            setattr(b, "n_index" if f == "n" else f, block[k])
        b.flags = torch.full((nrays,), abi.RAY_HAS_Q if q is not None else 0, dtype=torch.int32, device=dev)
        if ids is not None:
            b.id = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int32)).to(dev)
        else:
            b.id = torch.arange(nrays, dtype=torch.int32, device=dev)
        b.length = None
        return b

    def slice(self, lo, hi):
        """Contiguous shard [lo, hi) sharing storage (multi-GPU sharding, tests)."""
        out = object.__new__(RayBatch)
        out.n, out.precision, out.device = hi - lo, self.precision, self.device
        for f in abi.RAY_FIELDS:
            name = "n_index" if f == "n" else f
            setattr(out, name, self.field(f)[lo:hi])
        out.id, out.flags = self.id[lo:hi], self.flags[lo:hi]
        out.length = None if self.length is None else self.length[lo:hi]
        return out

    def take(self, index):
        """New batch made of rays `index` (device int64 tensor), in that order."""
        out = object.__new__(RayBatch)
        out.n, out.precision, out.device = int(index.numel()), self.precision, self.device
        for f in abi.RAY_FIELDS:
            setattr(out, "n_index" if f == "n" else f, self.field(f)[index].contiguous())
        out.id, out.flags = self.id[index].contiguous(), self.flags[index].contiguous()
        out.length = None if self.length is None else self.length[index].contiguous()
        return out

    def astype(self, precision):
        """Copy of the batch with its real fields in another precision ("f64" / "f32")."""
        if precision == self.precision:
            return self
        out = object.__new__(RayBatch)
        out.n, out.precision, out.device = self.n, precision, self.device
        dt = _REAL[precision]
        for f in abi.RAY_FIELDS:
            setattr(out, "n_index" if f == "n" else f, self.field(f).to(dt))
        out.id, out.flags = self.id, self.flags
        out.length = None if self.length is None else self.length.to(dt)
        return out

    def with_ids(self, ids):
        """Same rays (storage shared) under other ids (int32 device tensor)."""
        out = self.slice(0, self.n)
        out.id = ids.to(torch.int32).contiguous()
        return out

    def sorted_spatially(self, cells=32):
        """(batch in a spatially coherent order, order) with `sorted.field(f) == self.field(f)[order]`.

        Heavy scenes are latency/VALU-bound and their cost depends on how alike the 64 rays of a wave
        are (which groups they prune, which grid cells they walk): BASELINE cfg 3 traces 2.3x faster
        when its rays arrive sorted by origin than in random order.  The trace kernels never reorder
        rays themselves (their [k][ray] slots are addressed by ray index, and a permuted visit would
        turn every coalesced stream access into a gather/scatter — measured: 2x SLOWER), so callers
        with unordered rays sort once here; `order` maps results back (ray j of the sorted batch is
        ray order[j] of this one).  Round 4, re-measured at full size in the append layout: the pair queue
        (cfg 3) and the workgroup-wide block pool (cfg 5) pool the candidates / rays of a whole wave or
        workgroup, so the order the rays arrive in hardly matters any more — cfg 3 2.25 ms sorted against
        2.18 unsorted, cfg 5 11.4 against 12.0: a sort (milliseconds) no longer pays for itself.  Key: origin quantised to `cells`^3 over the batch's own bounding
        box, then direction octant."""
        o = torch.stack([self.ox, self.oy, self.oz], dim=1).double()
        lo, hi = o.min(dim=0).values, o.max(dim=0).values
        q = ((o - lo) / (hi - lo).clamp_min(1e-300) * (cells - 1e-9)).long().clamp_(0, cells - 1)
        octant = (self.dx < 0).long() + 2 * (self.dy < 0).long() + 4 * (self.dz < 0).long()
        key = ((q[:, 0] * cells + q[:, 1]) * cells + q[:, 2]) * 8 + octant
        order = torch.argsort(key, stable=True)
        out = object.__new__(RayBatch)
        out.n, out.precision, out.device = self.n, self.precision, self.device
        for f in abi.RAY_FIELDS:
            setattr(out, "n_index" if f == "n" else f, self.field(f)[order].contiguous())
        out.id, out.flags = self.id[order].contiguous(), self.flags[order].contiguous()
        out.length = None if self.length is None else self.length[order].contiguous()
        return out, order

    def multiplexed_in_wavelength(self, wavelengths):
        """Wavelength-major copies of the batch, `multiplex_rays_in_wavelength` (ray.py:428-445) for
        a device batch: ray j of copy w sits at w*n + j, carries wavelengths[w] and — like the
        reference's `ray.copy(wavelength=wl)` — shares id, q and every other field with its source."""
        wl = torch.as_tensor(np.asarray(wavelengths, dtype=np.float64), device=self.device)
        W = int(wl.numel())
        out = object.__new__(RayBatch)
        out.n, out.precision, out.device = self.n * W, self.precision, self.device
        for f in abi.RAY_FIELDS:
            name = "n_index" if f == "n" else f
            if f == "wavelength":
                setattr(out, name, wl.to(self.wavelength.dtype).repeat_interleave(self.n))
            else:
                setattr(out, name, self.field(f).repeat(W))
        out.id, out.flags = self.id.repeat(W), self.flags.repeat(W)
        out.length = None if self.length is None else self.length.repeat(W)
        return out

    def clone(self):
        """Deep copy on the device."""
        out = object.__new__(RayBatch)
        out.n, out.precision, out.device = self.n, self.precision, self.device
        for f in abi.RAY_FIELDS:
            setattr(out, "n_index" if f == "n" else f, self.field(f).clone())
        out.id, out.flags = self.id.clone(), self.flags.clone()
        out.length = None if self.length is None else self.length.clone()
        return out

    def translate_(self, vec):
        """origin += vec (Vector._Translate, base.py:141-144), in place."""
        for k, ax in enumerate("xyz"):
            self.field("o" + ax).add_(float(vec[k]))
        return self

    def rotate_around_(self, R, points):
        """Every ray turned by the rotation matrix R about its own lab point `points[i]`
        (Ray._RotAround -> _RotAroundLocal, ray.py:153-168): direction = R d (renormalised),
        origin = origin + R(-lp) + lp with lp = point - origin.  In place."""
        dt = self.ox.dtype
        R = torch.as_tensor(np.asarray(R, dtype=float), dtype=dt, device=self.device)
        o = torch.stack([self.ox, self.oy, self.oz], dim=1)
        d = torch.stack([self.dx, self.dy, self.dz], dim=1)
        lp = points.to(dt) - o
        o = o - lp @ R.T + lp
        d = d @ R.T
        d = d / torch.linalg.norm(d, dim=1, keepdim=True)
        for k, ax in enumerate("xyz"):
            self.field("o" + ax).copy_(o[:, k])
            self.field("d" + ax).copy_(d[:, k])
        return self

    def c_struct(self):
        s = abi.OtRays()
        for f in abi.RAY_FIELDS:
            setattr(s, f, self.field(f).data_ptr())
        s.id, s.flags = self.id.data_ptr(), self.flags.data_ptr()
        s.length = None if self.length is None else self.length.data_ptr()
        return s

    def to_host(self):
        out = {f: self.field(f).cpu().numpy() for f in abi.RAY_FIELDS}
        out["id"], out["flags"] = self.id.cpu().numpy(), self.flags.cpu().numpy()
        return out


class SegmentBatch:
    """`capacity` segment slots (reference: the List[Ray] `ray_tracing` returns).  Three layouts:

    slots   non-branching trace (ot_trace_*): slot k*n_rays + i = k-th segment of ray i, valid for k < count[i];
    list    breadth-first trace (ot_trace_generation_*): the first `n_valid` slots, in generation order;
    append  non-branching trace, dense (ot_trace_append_*): the first `n_valid` slots are records in append order or
            holes (`ray == -1`, the unused tail of a wave's last chunk); `count[i]` as for slots.
    For `list` and `append` a STABLE sort by `ray` gives the reference's order (input ray major, then segment order).
    """

    tiled = append = False          # defaults for batches assembled field by field (astype, to_slots, table.py)
    trees = False                   # True: `count[i]` rays of a ray TREE per input ray (Engine.trace_trees), not the segments of one path
    block = cursor = _n_valid = count = n_rays = None

    def __init__(self, capacity, precision="f64", device="cuda", block=False, tiled=False):
        self.capacity, self.precision, self.device = int(capacity), precision, torch.device(device)
        dt = _REAL[precision]
        self.block = None
        self.tiled = bool(tiled)
        if tiled:  # 64-slot tiles (include/optable_hip.h: ot_trace_tiled_*); the fields are strided [tiles, 64] views of one block
            self.capacity = cap = (self.capacity + 63) // 64 * 64
            width = 8 if precision == "f64" else 4
            tile_bytes = 64 * (12 * width + 8)
            self.block = torch.empty(cap // 64 * tile_bytes, dtype=torch.uint8, device=self.device)
            reals, ints = self.block.view(dt), self.block.view(torch.int32)
            rs, is_ = tile_bytes // width, tile_bytes // 4
            for k, f in enumerate(abi.SEG_FIELDS):
                setattr(self, "n_index" if f == "n" else f, reals.as_strided((cap // 64, 64), (rs, 1), 64 * k))
            self.ray = ints.as_strided((cap // 64, 64), (is_, 1), 12 * 64 * width // 4)
            self.surface = ints.as_strided((cap // 64, 64), (is_, 1), 12 * 64 * width // 4 + 64)
        elif block:  # ONE allocation of 14 planes (include/optable_hip.h: ot_segment_block); the fields are its rows
            self.capacity = cap = (self.capacity + 63) // 64 * 64
            width = 8 if precision == "f64" else 4
            self.block = torch.empty(12 * cap * width + 2 * cap * 4, dtype=torch.uint8, device=self.device)
            reals = self.block[: 12 * cap * width].view(dt).view(12, cap)
            ints = self.block[12 * cap * width:].view(torch.int32).view(2, cap)
            for k, f in enumerate(abi.SEG_FIELDS):
                setattr(self, "n_index" if f == "n" else f, reals[k])
            self.ray, self.surface = ints[0], ints[1]
        else:
            for f in abi.SEG_FIELDS:
                name = "n_index" if f == "n" else f
                setattr(self, name, torch.empty(self.capacity, dtype=dt, device=self.device))
            self.ray = torch.empty(self.capacity, dtype=torch.int32, device=self.device)
            self.surface = torch.empty(self.capacity, dtype=torch.int32, device=self.device)
        self.count = None      # int32 [n_rays] (slots and append layouts)
        self.n_rays = None
        self._n_valid = None   # number of slots in use when the layout is a list (list, append)
        self.cursor = None     # append layout: device scalar the kernel leaves the slot count in (read on first use)
        self.append = False

    @property
    def n_valid(self):
        if self._n_valid is None and self.cursor is not None:
            claimed = int(self.cursor.item())  # synchronises with the trace
            if claimed >= 1 << 62:
                raise RuntimeError("append layout: the trace kernel stopped at an internal bound (rays may be untraced): a defect in "
                                   "liboptable_hip.so, not a capacity problem — please report the scene")
            if claimed > self.capacity:
                raise RuntimeError(f"append layout: the trace needed {claimed} slots, the block holds {self.capacity}; "
                                   f"records beyond it were dropped — trace again with capacity >= {claimed}")
            self._n_valid = claimed
        return self._n_valid

    @n_valid.setter
    def n_valid(self, value):
        self._n_valid = value

    @property
    def layout(self):
        return "append" if self.append else ("tiled" if self.tiled else ("slots" if self.count is not None else "list"))

    def block_struct(self):
        s = abi.OtSegmentBlock()
        s.base, s.capacity = self.block.data_ptr(), self.capacity
        return s

    def field(self, name):
        return self.n_index if name == "n" else getattr(self, name)

    def to_slots(self):
        """A tiled batch as plain [k][ray] arrays (one device copy per field); any other batch as it is.  The tiled
        layout is what the lane-per-ray kernel writes fastest; everything that reads segments (monitors, exports,
        to_host) reads slot arrays."""
        if not self.tiled:
            return self
        out = object.__new__(SegmentBatch)
        out.capacity, out.precision, out.device = self.capacity, self.precision, self.device
        for f in abi.SEG_FIELDS:
            setattr(out, "n_index" if f == "n" else f, self.field(f).reshape(-1))
        out.ray, out.surface = self.ray.reshape(-1), self.surface.reshape(-1)
        out.count, out.n_rays, out._n_valid, out.cursor = self.count, self.n_rays, self._n_valid, None
        out.append, out.block, out.tiled = False, None, False
        for extra in ("capped", "counts_table", "count_ids", "trees", "timed_out"):
            if hasattr(self, extra):
                setattr(out, extra, getattr(self, extra))
        return out

    def as_kray_slots(self, max_segments=None):
        """The history of a non-branching trace as plain [k][ray] slot arrays whatever layout it was written in: slots and
        tiled batches as `to_slots()` gives them, an append-order list scattered into fresh arrays (slot = k * n_rays + ray, k =
        the record's rank among its ray's records: a stable sort by ray is segment order).  For checks and tools that index
        segments by (k, ray); the trace itself never needs it."""
        if not self.append:
            return self.to_slots()
        n = self.n_rays
        K = int(max_segments) if max_segments is not None else int(self.count.abs().max().item())
        m = self.n_valid
        idx = torch.nonzero(self.ray[:m] >= 0).flatten()
        ray = self.ray[idx].long()
        order = torch.argsort(ray, stable=True)
        idx, ray = idx[order], ray[order]
        cnt = self.count.abs().long()
        start = torch.cumsum(cnt, 0) - cnt
        k = torch.arange(ray.numel(), device=self.device) - start[ray]
        slot = k * n + ray
        out = SegmentBatch(n * K, self.precision, self.device)
        for f in abi.SEG_FIELDS + ("ray", "surface"):
            out.field(f)[slot] = self.field(f)[idx]
        out.count, out.n_rays = self.count, n
        for extra in ("counts_table", "count_ids"):
            if hasattr(self, extra):
                setattr(out, extra, getattr(self, extra))
        return out

    def c_struct(self):
        if self.tiled:
            raise ValueError("a tiled SegmentBatch has no 14-array form: use to_slots()")
        s = abi.OtSegments()
        for f in abi.SEG_FIELDS:
            setattr(s, f, self.field(f).data_ptr())
        s.ray, s.surface = self.ray.data_ptr(), self.surface.data_ptr()
        return s

    def astype(self, precision):
        """The same segments with their real fields in another precision (a copy unless already there)."""
        if self.tiled:
            return self.to_slots().astype(precision)
        if precision == self.precision:
            return self
        out = object.__new__(SegmentBatch)
        out.tiled = False
        out.capacity, out.precision, out.device = self.capacity, precision, self.device
        dt = _REAL[precision]
        for f in abi.SEG_FIELDS:
            setattr(out, "n_index" if f == "n" else f, self.field(f).to(dt))
        out.ray, out.surface = self.ray, self.surface
        out.count, out.n_rays, out._n_valid, out.cursor = self.count, self.n_rays, self.n_valid, None
        out.append, out.block = self.append, None
        for extra in ("capped", "counts_table", "count_ids", "trees", "timed_out"):
            if hasattr(self, extra):
                setattr(out, extra, getattr(self, extra))
        return out

    def valid_mask(self):
        """Boolean mask over slots (device)."""
        if self.tiled:
            return self.to_slots().valid_mask()
        if self.layout == "slots":
            k = torch.arange(self.capacity // self.n_rays, device=self.device, dtype=torch.int32).unsqueeze(1)
            return (k < self.count.abs().unsqueeze(0)).reshape(-1)
        m = torch.zeros(self.capacity, dtype=torch.bool, device=self.device)
        m[: self.n_valid] = True
        if self.append:
            m[: self.n_valid] &= self.ray[: self.n_valid] >= 0
        return m

    def _columns_to_host(self, m):
        """The first m slots of every field as numpy arrays.  Small histories (the object API) go through one
        stacked device tensor = one device-to-host copy instead of fourteen."""
        names = abi.SEG_FIELDS + ("ray", "surface")
        if m > 1_000_000:
            return {f: self.field(f)[:m].cpu().numpy() for f in names}
        reals = torch.stack([self.field(f)[:m] for f in abi.SEG_FIELDS]).cpu().numpy()
        ints = torch.stack([self.ray[:m], self.surface[:m]]).cpu().numpy()
        out = {f: reals[k] for k, f in enumerate(abi.SEG_FIELDS)}
        out["ray"], out["surface"] = ints[0], ints[1]
        return out

    def to_host(self, reference_order=True):
        """Valid segments as numpy arrays.  reference_order: input-ray-major, then segment
        order within the ray (the order OpticalTable.ray_tracing returns, optical_table.py:66-70)."""
        if self.tiled:
            return self.to_slots().to_host(reference_order)
        if self.count is not None and self.n_rays == 0:
            out = {f: np.zeros(0) for f in abi.SEG_FIELDS}
            out.update(ray=np.zeros(0, np.int32), surface=np.zeros(0, np.int32), count=np.zeros(0, np.int32))
            return out
        if self.append:  # records in append order with holes: drop the holes, then as for a list
            out = self._columns_to_host(self.n_valid)
            keep = out["ray"] >= 0
            out = {k: v[keep] for k, v in out.items()}
            if reference_order:
                order = np.argsort(out["ray"], kind="stable")
                out = {k: v[order] for k, v in out.items()}
            out["count"] = np.abs(self.count.cpu().numpy())
            return out
        if self.count is not None:
            cnt = np.abs(self.count.cpu().numpy())  # a negative count marks a tree that branched (its slots are still valid)
            K = min(self.capacity // self.n_rays, int(cnt.max()))  # (only the planes in use: a generous cap costs no copy)
            full = bool(cnt.min() == K)                            # every slot valid: no masking needed
            keep = None if full else np.arange(K)[:, None] < cnt[None, :]            # [K, N]
            out = {}
            cols = self._columns_to_host(K * self.n_rays)
            for f in abi.SEG_FIELDS + ("ray", "surface"):
                a = cols[f].reshape(K, self.n_rays)
                if full:
                    out[f] = np.ascontiguousarray(a.T).reshape(-1) if reference_order else a.reshape(-1)
                else:
                    out[f] = a.T[keep.T] if reference_order else a[keep]
            out["count"] = cnt
            return out
        out = self._columns_to_host(self.n_valid)
        if reference_order:  # generation order -> tree-major; within a tree generation order IS FIFO order
            order = np.argsort(out["ray"], kind="stable")
            out = {k: v[order] for k, v in out.items()}
        return out

    def export_rays_csv(self, filename, rays=None):
        """The reference's ray CSV (optical_table.py:447-500) for this history, in the reference's row order
        (input-ray-major, then segment order), written from the columns (export.py)."""
        from . import export

        host = self.to_host(reference_order=True)
        if rays is None:
            has_q = True
        else:
            flags = rays.flags.cpu().numpy()
            has_q = (flags[host["ray"]] & abi.RAY_HAS_Q) != 0
        export.write_rays_csv(filename, host, has_q)


"""Host arrays in, host arrays out, with the PCIe copies overlapped with the trace.

`OpticalTable.trace_batch` wants the rays resident in HBM.  When they live in host memory (numpy) and the
results are wanted back there, the kernel is the smallest part of the job: 104 B per ray go up and 96 B (final
state) or 104 B per segment (full history) come back.  `trace_host` cuts the batch into chunks and runs them
through two engines on two HIP streams, so the upload of chunk c+1 overlaps the trace and the download of chunk c
(PCIe is full duplex); inputs go through two pinned staging blocks, outputs land directly in pinned result arrays
(no second host copy).  Non-branching scenes only (the [segment][ray] layout); everything else: `trace_batch`.
"""
import numpy as np
import torch

from . import abi
from .batch import RayBatch, SegmentBatch, _REAL
from .dist import FINAL_FIELDS
from .engine import Engine


def trace_host(table, origin, direction, wavelength=0.0, q=None, max_segments=8, chunk=1 << 20, history=False,
               precision="f64", device=None):
    """Trace host rays chunk by chunk.  Returns a dict of numpy arrays (views of pinned memory):
    `count` [n]; `final` = {field: [n]} (dist.FINAL_FIELDS of each ray's last segment); with `history=True` also one
    [max_segments, n] array per segment field plus `surface` (slot k, ray i valid for k < count[i])."""
    origin = np.asarray(origin, dtype=np.float64).reshape(-1, 3)
    direction = np.asarray(direction, dtype=np.float64).reshape(-1, 3)
    n, K = origin.shape[0], int(max_segments)
    scene = table.compile()
    if scene.max_children > 1 or scene.limited:
        raise NotImplementedError("trace_host streams non-branching scenes without interact-count limits; use trace_batch")
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    dt, np_dt = _REAL[precision], (np.float64 if precision == "f64" else np.float32)
    wl_arr = None if np.ndim(wavelength) == 0 else np.ascontiguousarray(wavelength, dtype=np.float64).reshape(n)
    q_arr = None if (q is None or np.ndim(q) == 0) else np.ascontiguousarray(q, dtype=np.complex128).reshape(n)
    origin, direction = np.ascontiguousarray(origin), np.ascontiguousarray(direction)

    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    engines = []
    for s in streams:
        e = Engine(dev.index)
        e.use_stream(s)
        e.upload(scene)
        engines.append(e)
    rows = {f: k for k, f in enumerate(abi.RAY_FIELDS)}
    # Host work per chunk is two or three straight memcpys into pinned staging ([m, 3] origins and directions as
    # they are, per-ray wavelengths / q if given); the transposition into structure-of-arrays and the
    # normalisation of the directions (Ray.direction setter, ray.py:115-119) happen on the device.
    stage_o = [torch.empty((chunk, 3), dtype=torch.float64).pin_memory() for _ in streams]
    stage_d = [torch.empty((chunk, 3), dtype=torch.float64).pin_memory() for _ in streams]
    stage_w = [torch.empty(chunk, dtype=torch.float64).pin_memory() for _ in streams] if wl_arr is not None else None
    stage_q = [torch.empty((chunk, 2), dtype=torch.float64).pin_memory() for _ in streams] if q_arr is not None else None
    ready = [None, None]                      # event: the staging blocks' uploads have finished
    out_count = torch.empty(n, dtype=torch.int32).pin_memory()
    out_final = {f: torch.empty(n, dtype=dt).pin_memory() for f in FINAL_FIELDS}   # contiguous per field: direct DMA
    out_hist = out_surf = None
    if history:
        out_hist = {f: torch.empty((K, n), dtype=dt).pin_memory() for f in abi.SEG_FIELDS}
        out_surf = torch.empty((K, n), dtype=torch.int32).pin_memory()
    for c, lo in enumerate(range(0, n, chunk)):
        hi, slot = min(lo + chunk, n), c % 2
        m = hi - lo
        if ready[slot] is not None:
            ready[slot].synchronize()         # the staging blocks are free again
        stage_o[slot].numpy()[:m] = origin[lo:hi]
        stage_d[slot].numpy()[:m] = direction[lo:hi]
        if wl_arr is not None:
            stage_w[slot].numpy()[:m] = wl_arr[lo:hi]
        if q_arr is not None:
            stage_q[slot].numpy()[:m] = q_arr[lo:hi].view(np.float64).reshape(m, 2)
        with torch.cuda.stream(streams[slot]):
            o_dev = stage_o[slot][:m].to(dev, non_blocking=True)
            d_dev = stage_d[slot][:m].to(dev, non_blocking=True)
            w_dev = stage_w[slot][:m].to(dev, non_blocking=True) if wl_arr is not None else None
            q_dev = stage_q[slot][:m].to(dev, non_blocking=True) if q_arr is not None else None
            ready[slot] = torch.cuda.Event()
            ready[slot].record(streams[slot])
            stride = (m + 31) // 32 * 32
            block = torch.empty((len(rows), stride), dtype=dt, device=dev)[:, :m]
            block[rows["ox"]:rows["oz"] + 1] = o_dev.T
            block[rows["dx"]:rows["dz"] + 1] = (d_dev / torch.linalg.norm(d_dev, dim=1, keepdim=True)).T
            block[rows["wavelength"]] = w_dev if w_dev is not None else float(wavelength)
            block[rows["intensity"]], block[rows["n"]], block[rows["pathlength"]] = 1.0, 1.0, 0.0
            if q_dev is not None:
                block[rows["q_re"]], block[rows["q_im"]] = q_dev[:, 0], q_dev[:, 1]
            elif q is not None:
                block[rows["q_re"]], block[rows["q_im"]] = float(np.real(q)), float(np.imag(q))
            else:
                block[rows["q_re"]], block[rows["q_im"]] = 0.0, 0.0
            batch = object.__new__(RayBatch)
            batch.n, batch.precision, batch.device = m, precision, dev
            for f, k in rows.items():
                setattr(batch, "n_index" if f == "n" else f, block[k])
            batch.id = torch.arange(m, dtype=torch.int32, device=dev)
            batch.flags = torch.full((m,), abi.RAY_HAS_Q if q is not None else 0, dtype=torch.int32, device=dev)
            batch.length = None
            segs = engines[slot].trace(batch, K, out=SegmentBatch(m * K, precision, dev))
            out_count[lo:hi].copy_(segs.count, non_blocking=True)
            last = (segs.count.long() - 1).clamp_(min=0) * m + torch.arange(m, device=dev)
            for f in FINAL_FIELDS:
                out_final[f][lo:hi].copy_(segs.field(f)[last], non_blocking=True)
            if history:
                for k in range(K):  # every destination is a contiguous run of pinned memory
                    for f in abi.SEG_FIELDS:
                        out_hist[f][k, lo:hi].copy_(segs.field(f)[k * m:(k + 1) * m], non_blocking=True)
                    out_surf[k, lo:hi].copy_(segs.surface[k * m:(k + 1) * m], non_blocking=True)
            # (the device tensors of this chunk may go out of scope now: the caching allocator hands their memory
            #  only to later work on the same stream, i.e. after the copies above)
    for s in streams:
        s.synchronize()
    for e in engines:
        e.close()
    out = {"count": out_count.numpy(), "final": {f: t.numpy() for f, t in out_final.items()}}
    if history:
        out.update({f: t.numpy() for f, t in out_hist.items()})
        out["surface"] = out_surf.numpy()
    return out

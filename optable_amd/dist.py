"""Multi-GPU: rays are independent, so the path shards by contiguous ray ranges with the scene
replicated and NO collective inside the trace.  The only communication is one gather of the
per-ray final state at the end (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for
tests).  Reference: the per-ray loop at optical_table.py:66-70 has no cross-ray coupling except
shared `_id` counters — shard by id class when a scene has limited surfaces.
"""
import torch
import torch.distributed as dist

FINAL_FIELDS = ("ox", "oy", "oz", "dx", "dy", "dz", "length", "intensity", "q_re", "q_im", "n", "pathlength")


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of rank `rank` out of `world` (sizes differ by at most 1)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_indices_by_id(ids, rank, world):
    """Ray indices (ascending) owned by `rank` when rays that share an id must stay on one GPU — scenes
    with `max_interact_count` surfaces, whose counters are keyed by id (e.g. the copies made
    by `multiplexed_in_wavelength` sit n apart and a contiguous split would separate them).  Distinct ids
    are dealt out in contiguous, equally sized ranges of their sorted order, so shards stay balanced to
    within one id class; `RayBatch.take(indices)` builds the shard."""
    uniq, inverse = torch.unique(ids, return_inverse=True)
    owner = (inverse * world) // max(int(uniq.numel()), 1)
    return torch.nonzero(owner == rank).flatten()


def final_state(segs):
    """[12, n_rays] tensor: each ray's last segment (non-branching [k][ray] layout)."""
    if segs.count is None or getattr(segs, "trees", False):
        raise ValueError("final_state needs a non-branching trace (slots or append layout); a ray tree has no single last segment")
    n = segs.n_rays
    if getattr(segs, "tiled", False):  # slot s at [s // 64, s % 64] of the strided field views
        last = (segs.count.long().abs() - 1).clamp_(min=0) * n + torch.arange(n, device=segs.device)
        return torch.stack([segs.field(f)[last >> 6, last & 63] for f in FINAL_FIELDS])
    if segs.append:  # the records of a ray lie at increasing slots: its last segment is its highest slot
        m = segs.n_valid
        ray = segs.ray[:m].long()
        slot = torch.arange(m, device=segs.device)
        keep = ray >= 0
        last = torch.zeros(n, dtype=torch.int64, device=segs.device).scatter_reduce_(0, ray[keep], slot[keep], "amax", include_self=False)
        return torch.stack([segs.field(f)[last] for f in FINAL_FIELDS])
    last = (segs.count.long().abs() - 1).clamp_(min=0) * n + torch.arange(n, device=segs.device)
    return torch.stack([segs.field(f)[last] for f in FINAL_FIELDS])


def gather_final_state(local, dst=0, group=None, timings=None):
    """The one collective of a job: every rank's [12, n_local] block to `dst`; returns [12, n_total] there, None
    elsewhere.  The root allocates the result ONCE and every field row of every shard is received straight into its
    place (row f of rank r's shard is the contiguous slice out[f, off_r : off_r + n_r]): no padding to the widest shard
    and no `world` staging blocks on the root.  Two steps, timed apart into `timings` (a dict) when given:
      sizes    all-gather of the shard sizes (one int64 per rank)
      payload  12 sends per rank, 12 (world - 1) receives on the root, issued as one batch (RCCL: one group)
    Shard sizes may differ (contiguous shards differ by at most one ray, id-class shards by one class)."""
    import time

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    on_gpu = dist.get_backend(group) != "gloo"
    if not on_gpu:  # CPU rehearsal of the collective (tests, 1-GPU boxes)
        local = local.cpu()
    local = local.contiguous()

    def now():
        if on_gpu:
            torch.cuda.synchronize(local.device)
        return time.perf_counter()

    t0 = now()
    n_local = torch.tensor([local.shape[1]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    t1 = now()
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + sz)
    rows = local.shape[0]
    out, ops = None, []
    if rank == dst:
        out = torch.empty((rows, offs[-1]), dtype=local.dtype, device=local.device)
        out[:, offs[dst]:offs[dst + 1]] = local
        for r in range(world):
            if r != dst and sizes[r] > 0:
                ops.extend(dist.P2POp(dist.irecv, out[f, offs[r]:offs[r + 1]], r, group) for f in range(rows))
    elif sizes[rank] > 0:
        ops.extend(dist.P2POp(dist.isend, local[f], dst, group) for f in range(rows))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    t2 = now()
    if timings is not None:
        timings.update(sizes_ms=(t1 - t0) * 1e3, payload_ms=(t2 - t1) * 1e3, shard_sizes=sizes)
    return out

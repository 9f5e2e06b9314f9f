"""CPU, world_size 2, gloo: the sharding arithmetic and the single end-of-job gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from optable_amd.dist import shard_range, gather_final_state, shard_indices_by_id


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_id_class_shards_keep_shared_ids_together():
    """Count-limited scenes: every id lands on exactly one rank, shards cover all rays, stay balanced."""
    rng = np.random.default_rng(5)
    base = torch.from_numpy(1000 + 3 * rng.permutation(500)).to(torch.int32)
    ids = base.repeat(3)                                   # three wavelength copies, n apart
    for world in (1, 2, 4, 8):
        shards = [shard_indices_by_id(ids, r, world) for r in range(world)]
        allidx = torch.cat(shards)
        assert torch.equal(torch.sort(allidx).values, torch.arange(ids.numel()))
        owners = {}
        for r, idx in enumerate(shards):
            assert bool((idx[1:] > idx[:-1]).all())        # ascending: input order kept inside a shard
            for i in torch.unique(ids[idx]).tolist():
                assert owners.setdefault(i, r) == r        # an id never appears on two ranks
        sizes = [int(s.numel()) for s in shards]
        assert max(sizes) - min(sizes) <= 3                # one id class (3 copies) at most


def _worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n, rank, world)
    full = torch.arange(12 * n, dtype=torch.float64).reshape(12, n)
    got = gather_final_state(full[:, lo:hi].clone(), dst=0)
    if rank == 0:
        out.put(bool(torch.equal(got, full)))
    else:
        assert got is None
    dist.destroy_process_group()


def test_gather_reassembles_shards_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    world, n = 2, 101  # uneven shards
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    ok = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok

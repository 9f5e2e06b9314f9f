"""BASELINE configs 3-5 at the size ONE GPU sees in the quoted configuration (cfg 3: all 1e7 rays,
fp32; cfg 4: the 4-GPU shard of 1e7 rays x 64 wavelengths, fp64; cfg 5: the 8-GPU shard of 1e8 rays,
fp32), checked through properties that need no oracle: segment chains are continuous, directions are
unit vectors, a shard traced on its own reproduces its part of the whole bit for bit (what
`bench.py --gpus N` relies on), and cfg 4 has a closed-form answer (Snell through a flat slab).
Small-size parity against the oracle for the same scenes is in test_gpu_parity.py."""
import numpy as np
import pytest

import scenes
from optable_amd import abi

pytestmark = pytest.mark.gpu


def _table(components):
    import optable_amd as oa

    t = oa.OpticalTable()
    t.add_components(components)
    return t


def _chain_properties(segs, n, K, tol, unit_tol):
    """Device-side, one segment index at a time (no [K, n] temporaries)."""
    import torch

    cnt = segs.count
    f = {name: segs.field(name).view(K, n) for name in ("ox", "oy", "oz", "dx", "dy", "dz", "length")}
    surf = segs.surface.view(K, n)
    worst_gap = worst_norm = 0.0
    for k in range(K):
        valid = cnt > k
        if not bool(valid.any()):
            break
        norm = torch.sqrt(f["dx"][k].double() ** 2 + f["dy"][k].double() ** 2 + f["dz"][k].double() ** 2)
        worst_norm = max(worst_norm, float((norm - 1).abs()[valid].max()))
        assert bool((f["length"][k][valid] > 0).all())
        last = cnt == k + 1          # a ray's final segment either escaped (open length) or hit the cap / a block
        open_end = last & (surf[k] < 0)
        assert bool(torch.isinf(f["length"][k][open_end]).all())
        if k + 1 < K:
            link = cnt > k + 1
            if bool(link.any()):
                assert bool((surf[k][link] >= 0).all())
                for a in "xyz":
                    end = f["o" + a][k].double() + f["length"][k].double() * f["d" + a][k].double()
                    worst_gap = max(worst_gap, float((end - f["o" + a][k + 1].double()).abs()[link].max()))
    assert worst_gap < tol, worst_gap
    assert worst_norm < unit_tol, worst_norm


def _shard_equals_whole(table, batch, segs, lo, hi, K):
    import torch

    n = batch.n
    part = table.trace_batch(batch.slice(lo, hi), max_segments=K).as_kray_slots(K)  # (the default layout, whatever it is for this scene)
    assert torch.equal(part.count, segs.count[lo:hi])
    m = hi - lo
    keep = torch.arange(K, device=part.count.device).unsqueeze(1) < part.count.unsqueeze(0)
    for name in abi.SEG_FIELDS + ("surface",):
        whole = segs.field(name).reshape(K, n)[:, lo:hi]
        mine = part.field(name).reshape(K, m)
        assert torch.equal(whole[keep], mine[keep]), name


def test_cfg3_full_size_fp32():
    """1e7 rays, 32 mixed components (56 leaves), cap 20, fp32 — BASELINE configs[2] as quoted."""
    import torch
    import optable_amd as oa
    from optable_amd.batch import RayBatch

    n, K = 10_000_000, 20
    table = _table(scenes.cfg3_components(oa))
    o, d = scenes.cfg3_rays(n, 2)
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j * np.pi * scenes.W0**2 / scenes.WL, precision="f32")
    del o, d
    from optable_amd.engine import get_engine

    segs = table.trace_batch(batch, max_segments=K)  # the DEFAULT call: the dense append-order list through the pair queue
    info = get_engine().last_launch()
    assert segs.layout == "append" and info["pair_queue"] & 1 and info["pair_queue"] & 4, info
    records = int(segs.count.abs().sum().item())
    assert segs.capacity <= 1.3 * records + (1 << 23), (segs.capacity, records)  # sized from a 1 % sample, not for the worst case
    cnt = segs.count
    assert int(cnt.min()) >= 1 and int(cnt.max()) <= K
    mean = float(cnt.double().mean())
    assert 4.6 < mean < 5.2, mean        # mean 4.8 segments per ray (oracle-validated)
    segs = segs.as_kray_slots(K)         # (indexable by segment and ray for the checks below)
    _chain_properties(segs, n, K, tol=6e-3, unit_tol=1e-6)
    _shard_equals_whole(table, batch, segs, 3_000_000, 4_250_000, K)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("nb", [2_500_000, 5_000_000], ids=["4gpu-shard-1.6e8", "2gpu-shard-3.2e8"])
def test_cfg4_shard_closed_form_fp64(nb):
    """The per-GPU shard of configs[3] (1e7 rays x 64 wavelengths, "sharded 2 and 4 GPUs"): 2.5e6 base rays on 4 GPUs
    = 1.6e8 ray-wavelength pairs, 5e6 on 2 GPUs = 3.2e8 pairs (33 GB of rays, 100 GB of segment history: what
    `bench.py --workload cfg4 --gpus 2` traces per rank), through an N-BK7 slab, fp64.
    Every pair yields exactly 3 segments, the exit ray is parallel
    to the entry ray, displaced sideways by thickness*sin(ti - tt)/cos(tt) with sin(tt) = sin(ti)/n(wl),
    and its optical path is the geometric lengths weighted by n — all evaluated here independently
    from the Sellmeier table on the host (material.py:115-120)."""
    import torch
    import optable_amd as oa
    from optable_amd.batch import RayBatch

    nwl, K = 64, 3
    thickness = 0.5
    rng = np.random.default_rng(4)
    jit = rng.uniform(-0.3, 0.3, (nb, 2))
    o = np.stack([np.full(nb, -3.0), 2 + jit[:, 0], jit[:, 1]], 1)
    d = np.tile([np.cos(np.pi / 6), -np.sin(np.pi / 6), 0.0], (nb, 1))
    wls = np.linspace(400e-7, 1100e-7, nwl)
    base = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j * np.pi * scenes.W0**2 / scenes.WL)
    batch = base.multiplexed_in_wavelength(wls)
    del base, o, d
    n = nb * nwl
    glass = oa.Glass_NBK7()
    table = _table([oa.GlassSlab([0, 0, 0], width=2, height=2, thickness=thickness, n1=oa.Vacuum(), n2=glass, reflectivity=0)])
    segs = table.trace_batch(batch, max_segments=K)  # the DEFAULT call: tiles or slot arrays, whichever this device streams faster
    assert segs.layout in ("tiled", "slots")
    assert int(segs.count.min()) == K and int(segs.count.max()) == K

    class Fields:  # field -> [K, n]: a view of a slot array, a copy out of the tiles — made when asked for, one at a time
        def __getitem__(self, name):
            return segs.field(name).reshape(K, n)

    f = Fields()
    surf = segs.surface.reshape(K, n)
    assert bool((surf[:2] >= 0).all()) and bool((surf[2] == -1).all())
    # host-side Sellmeier (wavelength in model units x unit 1e-2 m -> micrometres)
    um = wls * 1e-2 * 1e6
    B = np.array([1.03961212, 0.231792344, 1.01046945])      # Schott N-BK7 (material.py:123-130)
    C = np.array([0.00600069867, 0.0200179144, 103.560653])
    n_glass = np.sqrt(1 + sum(B[i] * um**2 / (um**2 - C[i]) for i in range(3)))
    n_dev = torch.as_tensor(n_glass, device="cuda").repeat_interleave(nb)
    assert float((f["n"][1] - n_dev).abs().max()) < 1e-12          # index carried by the inner segment
    assert float((f["n"][0] - 1).abs().max()) == 0.0 and float((f["n"][2] - 1).abs().max()) == 0.0
    for a in "xyz":                                                  # exit parallel to entry
        assert float((f["d" + a][2] - f["d" + a][0]).abs().max()) < 1e-12
    ti = np.pi / 6
    tt = torch.arcsin(np.sin(ti) / n_dev)
    shift = thickness * torch.sin(ti - tt) / torch.cos(tt)
    # sideways displacement of the exit ray from the entry line (direction d0 through o0)
    rx, ry, rz = (f["o" + a][2] - f["o" + a][0] for a in "xyz")
    along = rx * f["dx"][0] + ry * f["dy"][0] + rz * f["dz"][0]
    perp = torch.sqrt((rx - along * f["dx"][0]) ** 2 + (ry - along * f["dy"][0]) ** 2 + (rz - along * f["dz"][0]) ** 2)
    assert float((perp - shift).abs().max()) < 1e-10
    inner = thickness / torch.cos(tt)
    assert float((f["length"][1] - inner).abs().max()) < 1e-10
    opl = f["length"][0] + inner * n_dev
    assert float((f["pathlength"][2] - opl).abs().max()) < 1e-9
    # transmission 1, reflectivity 0: lossless
    assert float((f["intensity"] - 1).abs().max()) == 0.0
    del f, surf, perp, along, rx, ry, rz
    _shard_equals_whole(table, batch, segs, 40_000_000, 41_000_000, K)
    del segs, batch, n_dev, tt, shift, inner, opl
    torch.cuda.empty_cache()


def test_cfg5_shard_fp32():
    """The per-GPU shard of configs[4] on 8 GPUs: 1.25e7 rays, asphere + MMA 16x16 (260 leaves),
    cap 50, fp32."""
    import torch
    import optable_amd as oa
    from optable_amd.batch import RayBatch

    n, K = 12_500_000, 50
    table = _table(scenes.cfg5_components(oa))
    o, d = scenes.cfg5_rays(n, 3)
    batch = RayBatch.from_arrays(o, d, wavelength=scenes.WL, q=1j * np.pi * scenes.W0**2 / scenes.WL, precision="f32")
    del o, d
    from optable_amd.engine import get_engine

    segs = table.trace_batch(batch, max_segments=K)  # the DEFAULT call: the workgroup-wide block pool into the dense append-order list
    info = get_engine().last_launch()
    assert segs.layout == "append" and info["pair_queue"] & 16 and info["pair_queue"] & 4, info
    records = int(segs.count.abs().sum().item())
    assert segs.capacity <= 1.3 * records + (1 << 23), (segs.capacity, records)  # 17 GB of records, not the worst case's 35
    cnt = segs.count
    assert int(cnt.min()) >= 1 and int(cnt.max()) == K      # some rays ring between mirror and MMA to the cap
    segs = segs.as_kray_slots(K)                             # (indexable by segment and ray for the checks below)
    mean = float(cnt.double().mean())
    assert 20 < mean < 28, mean                              # 24.1 at 4e5 rays (oracle-validated at small n)
    _chain_properties(segs, n, K, tol=6e-3, unit_tol=1e-6)
    _shard_equals_whole(table, batch, segs, 5_000_000, 5_400_000, K)
    torch.cuda.empty_cache()


def test_cfg4_ray_trees_full_size_fp64():
    """cfg 4 with reflectivity 0.2 at the size bench.py quotes (1.28e7 ray trees x 12 rays, fp64) through the default call
    (one lane-per-tree launch; the dense list): every tree is cut by the cap with exactly 12
    records, the first record of a tree is its input ray, every direction is a unit vector, energy never grows along a tree —
    and a strided sample of 20 000 trees traced on its own through the GENERATION kernels gives the records the big launch
    holds for those trees, bit for bit, in the reference's order."""
    import torch

    import optable_amd as oa
    from optable_amd import workloads as W
    from optable_amd.batch import RayBatch
    from optable_amd.engine import get_engine

    table = _table(W.cfg4_components(oa, reflectivity=0.2))
    scene = table.compile()
    eng = get_engine()
    eng.upload(scene)
    o, d, _ = W.cfg4_rays(200_000, 4, n_wavelengths=1)
    base = RayBatch.from_arrays(o, d, wavelength=W.WL, q=1j * np.pi * W.W0**2 / W.WL, precision="f64")
    batch = base.multiplexed_in_wavelength(np.linspace(400e-7, 1100e-7, W.CFG4_WAVELENGTHS))
    n, K = batch.n, 12
    segs = eng.trace_branching(batch, K)
    assert eng.last_launch()["kernel"] == 4 and segs.layout == "append"
    assert bool((segs.count == K).all()) and bool(segs.capped.all())
    m = segs.n_valid
    ray = segs.ray[:m]
    real = ray >= 0
    plan = eng.trees_plan("f64", K, n)
    assert int(real.sum()) == n * K and m - n * K <= plan["chunk"] * plan["waves"] <= n * K // 16 + 512 * plan["waves"]  # holes: chunk tails only
    norm = torch.sqrt(segs.dx[:m] ** 2 + segs.dy[:m] ** 2 + segs.dz[:m] ** 2)
    assert float((norm - 1).abs()[real].max()) < 1e-12
    assert float(segs.intensity[:m][real].max()) <= 1.0 + 1e-12 and float(segs.intensity[:m][real].min()) >= 0.0
    # the first record of every tree (its lowest slot) is the input ray
    slot = torch.arange(m, device=ray.device)
    first = torch.full((n,), m, dtype=torch.int64, device=ray.device).scatter_reduce_(0, ray[real].long(), slot[real], "amin", include_self=True)
    assert bool((first < m).all())
    for f, src in (("ox", batch.ox), ("oy", batch.oy), ("oz", batch.oz), ("dx", batch.dx), ("dy", batch.dy), ("dz", batch.dz)):
        assert torch.equal(segs.field(f)[first], src), f
    # a strided sample of trees through the generation kernels
    pick = torch.arange(0, n, n // 20_000, device=ray.device)
    sample = eng.trace_tree(batch.take(pick), K).to_host(reference_order=True)
    index_of = torch.full((n,), -1, dtype=torch.int64, device=ray.device)
    index_of[pick] = torch.arange(pick.numel(), device=ray.device)
    mine = real & (index_of[ray.clamp(min=0).long()] >= 0)
    where = torch.nonzero(mine).flatten()
    order = torch.argsort(index_of[ray[where].long()], stable=True)  # (records of a tree lie at increasing slots: FIFO order)
    where = where[order].cpu()
    assert where.numel() == len(sample["ray"]) == pick.numel() * K
    np.testing.assert_array_equal(index_of[ray[where.to(ray.device)].long()].cpu().numpy(), sample["ray"])
    np.testing.assert_array_equal(segs.surface[where.to(ray.device)].cpu().numpy(), sample["surface"])
    for f in abi.SEG_FIELDS:
        np.testing.assert_array_equal(segs.field(f)[where.to(ray.device)].cpu().numpy(), sample[f], err_msg=f)

"""bench.py as the driver runs it (a subprocess printing ONE JSON line), at reduced ray counts so that the test
takes seconds: the line parses, carries every contract field, and its numbers are self-consistent."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "1",
                          "--sustained-seconds", "1", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,rays", [("cfg2", 200_000), ("cfg5", 60_000)])
def test_bench_line_contract(workload, rays):
    line = _run("--workload", workload, "--rays", str(rays))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["vs_baseline"] is None
    # --rays fixes the per-GPU size: weak scaling whatever the workload (cfg 5 without it shards its 1e8-ray total)
    assert line["scaling"] == "weak"
    assert line["higher_is_better"] is True and line["data"] == "synthetic"
    assert line["config"]["rays_per_gpu"] == rays and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"])
    # value = segments x leaves x steps / wall time: consistent with ms_per_step
    segs = line["config"]["segments_per_ray"] * rays
    assert line["value"] == pytest.approx(segs * line["config"]["leaf_surfaces"] / (line["ms_per_step"] / 1e3), rel=1e-6)
    # the event-timed launch cannot be longer than the wall-clock step
    assert roof["kernel_us"] <= line["ms_per_step"] * 1e3 * 1.02
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and "sample" in cpu
    assert line["value"] > 50 * cpu["value"]          # a GPU against one host core, even at this size
    assert cpu["reference_python"]["kind"] == "reference" and "provenance" in cpu["reference_python"]
    # measurement hygiene: what ran before the warmup is reported, rotated inputs exceed the Infinity
    # Cache, a cold and a sustained figure stand beside the steady one, the PMC traffic names its source
    assert line["preload_launches"] > 0 and line["cold"]["us_per_step"] > 0
    assert line["sustained"]["seconds"] >= 1.0 and line["sustained"]["steps"] > 0
    nb, rec = line["config"]["input_batches_rotated"], 104 if line["dtype"] == "f64" else 56
    assert nb * rays * rec >= 256 * 2**20 or nb == 4
    if roof["traffic"] is not None:
        assert "profiles/" in roof["traffic_source"]


def test_two_ranks_on_one_card_gloo_rehearsal():
    """`bench.py --gpus 2` must start its two ranks itself (the driver runs exactly this command for N > 1, with RCCL);
    here both ranks share the one card and the collective runs over gloo."""
    line = _run("--gpus", "2", "--backend", "gloo", "--rays", "20000", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["config"]["rays_per_gpu"] == 20000 and line["config"]["rays_total"] == 40000
    assert line["gathered_shape"] == [12, 40000] and line["gather_ms"] > 0 and "gather_error" not in line
    assert line["config"]["segments_per_ray"] == 5.0


def test_sharded_configs_of_the_multi_gpu_line_gloo_rehearsal():
    """With N > 1 the line also carries the configs BASELINE quotes on several GPUs (cfg 4 on 2 / 4, cfg 5 on 8) at their
    quoted totals, each with a timed region of at least 50 ms between its barriers; rehearsed here with two ranks on the one
    card, over gloo, at a few thousand rays per rank."""
    line = _run("--gpus", "2", "--backend", "gloo", "--rays", "20000", "--no-cpu-baseline", "--sharded-configs", "cfg4,cfg5",
                "--sharded-rays", "64000")
    recs = line["sharded_configs"]
    assert [r.get("error") for r in recs] == [None, None], recs
    assert recs[0]["workload"].startswith("cfg4") and recs[1]["workload"].startswith("cfg5")
    for r in recs:
        assert r["n_gpus"] == 2 and r["rays_total"] == 128000 and r["timed_region_ms"] >= 50.0 and r["steps"] >= 3
        assert len(r["ranks"]["ms_per_step_by_rank"]) == 2 and min(r["ranks"]["segments_by_rank"]) > 64000
        assert r["value"] / r["segments_per_s"] == pytest.approx(2 if r["workload"].startswith("cfg4") else 260)  # x leaf surfaces
    assert recs[0]["layout"] in ("tiled", "slots") and recs[1]["layout"] == "append"


def test_four_ranks_on_one_card_gloo_rehearsal():
    """The multi-rank line with four ranks sharing the one card of this box (the box allows six processes on its card and
    the test runner is one of them; the eight-rank line is rehearsed without a GPU in tests/test_bench_launcher_cpu.py):
    rank -> device modulo the card count, per-rank diagnostics, the gather received straight into one [12, n_total]
    block on the root."""
    line = _run("--gpus", "4", "--backend", "gloo", "--workload", "cfg5", "--rays", "20000", "--no-cpu-baseline")
    assert line["n_gpus"] == 4 and line["config"]["rays_total"] == 80000 and line["scaling"] == "weak"
    assert line["gathered_shape"] == [12, 80000] and "gather_error" not in line
    assert len(line["ranks"]["ms_per_step"]["by_rank"]) == 4 and min(line["ranks"]["segments_per_step_by_rank"]) > 20000
    assert line["gather"]["payload_ms"] > 0 and line["gather"]["shard_sizes"] == [20000] * 4
    assert line["comm"]["backend"] == "gloo"


def test_one_rank_rccl_rehearsal():
    """The RCCL calls of the multi-GPU path (init with device_id, barrier, all_reduce MAX / SUM, the end-of-job gather) with
    the one rank a 1-GPU box allows: `torchrun --nproc-per-node 1 bench.py --gpus 1 --force-dist` uses backend nccl."""
    import socket

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sock:  # a free port, as bench.py's own launcher picks one
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1",
                          "--rays", "20000", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["gathered_shape"] == [12, 20000] and line["gather_ms"] > 0 and "gather_error" not in line
    assert line["comm"]["backend"] == "rccl" and line["comm"]["rccl_version"]

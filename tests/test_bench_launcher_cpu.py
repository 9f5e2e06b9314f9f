"""bench.py's multi-rank entry without a GPU: `--gpus N` must start N ranks by itself, size the shards from the
world size (cfg 2 weak, cfg 4 / cfg 5 strong: total / N) and run the one end-of-job gather (gloo here, RCCL on the
GPU node).  `--dry-run` replaces the trace by fabricated per-ray state; everything around it is the real code.
Reference semantics being sharded: optical_table.py:66-70 (rays are traced one after the other, independently)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env,
                          timeout=300)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    return proc, [json.loads(ln) for ln in lines]


def test_gpus_2_starts_two_ranks_and_gathers():
    proc, lines = run_bench("--gpus", "2", "--backend", "gloo", "--rays", "20000", "--dry-run", "--steps", "2", "--warmup", "0")
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert len(lines) == 1, proc.stdout  # rank 0 alone prints
    line = lines[0]
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert line["config"]["rays_per_gpu"] == 20000 and line["config"]["rays_total"] == 40000
    assert line["gathered_shape"] == [12, 40000]
    assert line["gather_ms"] > 0 and "gather_error" not in line
    assert line["scaling"] == "weak"


def test_strong_scaling_sizes_follow_the_world_size():
    # cfg 4: 1e7 rays x 64 wavelengths = 6.4e8 pairs over 2 GPUs = 3.2e8 per rank; cfg 5: 1e8 rays over 4 ranks
    proc, lines = run_bench("--gpus", "2", "--workload", "cfg4", "--dry-run", "--steps", "1", "--warmup", "0")
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = lines[0]
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["rays_per_gpu"] == 320_000_000 and line["config"]["rays_total"] == 640_000_000
    proc, lines = run_bench("--gpus", "4", "--workload", "cfg5", "--dry-run", "--steps", "1", "--warmup", "0")
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert lines[0]["config"]["rays_per_gpu"] == 25_000_000 and lines[0]["n_gpus"] == 4


def test_mismatched_world_size_fails_loudly():
    proc, lines = run_bench("--gpus", "2", "--dry-run", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert proc.returncode != 0 and not lines
    assert "--gpus 2 but WORLD_SIZE=1" in proc.stderr


def test_eight_ranks_report_per_rank_diagnostics():
    """The line of a multi-rank run names every rank's own step time and work, the two halves of the gather and the
    communication backend: when the scaling curve is finally measured a straggler can be attributed."""
    proc, lines = run_bench("--gpus", "8", "--backend", "gloo", "--rays", "20001", "--dry-run", "--steps", "2", "--warmup", "0")
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = lines[0]
    assert line["n_gpus"] == 8 and line["config"]["rays_total"] == 8 * 20001
    ranks = line["ranks"]
    assert len(ranks["ms_per_step"]["by_rank"]) == 8 and len(ranks["segments_per_step_by_rank"]) == 8
    assert ranks["ms_per_step"]["min"] <= ranks["ms_per_step"]["median"] <= ranks["ms_per_step"]["max"]
    assert line["ms_per_step"] == ranks["ms_per_step"]["max"] or abs(line["ms_per_step"] - ranks["ms_per_step"]["max"]) < 1e-9
    g = line["gather"]
    assert g["sizes_ms"] > 0 and g["payload_ms"] > 0 and len(g["shard_sizes"]) == 8
    assert line["gathered_shape"] == [12, sum(g["shard_sizes"])] and "gather_error" not in line
    assert line["comm"]["backend"] == "gloo"


def test_a_clamped_total_is_labelled_weak():
    # cfg 5 names 1e8 rays; two ranks take 2.5e7 each (the per-rank HBM cap): the total grows with N
    proc, lines = run_bench("--gpus", "2", "--workload", "cfg5", "--dry-run", "--steps", "1", "--warmup", "0")
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = lines[0]
    assert line["clamped"] is True and line["total_requested"] == 100_000_000 and line["scaling"] == "weak"
    assert line["config"]["rays_per_gpu"] == 25_000_000

"""bench.py's own multi-rank entry point, rehearsed on the CPU (BASELINE north_star: "reported at 1/2/4/8 GPUs"): `python bench.py --gpus N`
without a launcher starts the N ranks itself; `--dry-run` replaces the GPU work by the configuration-only controller of every shard and a
gloo gather of slabs of the real shape.  What is asserted: n_gpus is the number of ranks the process group counts, the shard ranges, the
gather shape, config 4's 8192 instances per GPU, and that a run which cannot be the one asked for prints NO line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=300, env=e)


def test_two_ranks_started_by_bench_itself():
    r = _bench("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout      # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["valid"] is True and d["value"] is None
    assert d["config"]["shards"] == [[0, 4096], [4096, 8192]] and d["config"]["global_batch"] == 8192
    assert d["config"]["gather_shape"] == [2, 7, 4096] and d["scaling"] == "weak"


def test_config4_is_8192_instances_per_gpu():
    r = _bench("--gpus", "2", "--dry-run", "--config", "4")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["batch_per_gpu"] == 8192 and d["config"]["gather_shape"] == [2, 7, 8192]
    assert "65536" in d["config"]["workload"] or "config 4" in d["config"]["workload"]


def test_world_size_that_contradicts_gpus_prints_nothing():
    r = _bench("--gpus", "2", "--dry-run", env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE=1" in r.stderr


def test_more_gpus_than_visible_prints_nothing():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than two GPUs")
    r = _bench("--gpus", "2")
    # (the launcher parent counts GPUs from the KFD topology without touching the runtime; where that cannot tell, the ranks themselves refuse)
    assert r.returncode != 0 and r.stdout.strip() == "" and ("GPU(s) are visible" in r.stderr or "needs a GPU" in r.stderr)


def test_gather_behind_every_step():
    """--gather every-step: one all-gather per cycle (the reference's consumer takes the torques every control period,
    examples/05-using_robot_controller/05-using_robot_controller.cpp:193-196); the line says how many gathers ran and what one cost"""
    r = _bench("--gpus", "2", "--dry-run", "--gather", "every-step", "--steps", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["valid"] is True and c["gather_mode"] == "every-step" and c["gathers"] == 5 and c["gather_us"] > 0
    assert "behind every cycle" in c["parallelism"]
    d0 = json.loads(_bench("--gpus", "2", "--dry-run").stdout.strip().splitlines()[-1])
    assert d0["config"]["gather_mode"] == "final" and d0["config"]["gathers"] == 1


def test_single_process_launcher():
    """--launcher multi: ONE torch-free process for all GPUs (saip_multi_*); the dry run builds every shard's configuration-only controller"""
    r = _bench("--gpus", "4", "--dry-run", "--launcher", "multi", "--config", "4")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["dry_run"] is True and d["valid"] is True
    assert d["config"]["launcher"].startswith("multi") and d["config"]["gather_shape"] == [4, 7, 8192] and d["config"]["batch_per_gpu"] == 8192
    r = _bench("--gpus", "2", "--dry-run", "--launcher", "multi", env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"))
    assert r.returncode != 0 and r.stdout.strip() == "" and "ONE process" in r.stderr


def test_a_rank_that_leaves_early_ends_the_run():
    """the self-started ranks are polled: when one exits non-zero before the rendezvous the others are ended, no line is printed and the
    launcher returns non-zero promptly (it used to wait for rank 0 for ever)"""
    import time
    t0 = time.time()
    r = _bench("--gpus", "2", "--dry-run", env=dict(SAIP_BENCH_FAIL_RANK="1"))
    assert r.returncode != 0 and r.stdout.strip() == "" and "rank 1 exited" in r.stderr
    assert time.time() - t0 < 120
    t0 = time.time()
    r = _bench("--gpus", "2", "--dry-run", "--spawn-timeout", "3", env=dict(SAIP_BENCH_HANG_RANK="1"))
    assert r.returncode != 0 and r.stdout.strip() == "" and "no result within" in r.stderr
    assert time.time() - t0 < 120

"""The round-4 measurement entry points on the one GPU of the test box: `bench.py --launcher multi` (one torch-free process through
saip_multi_*, n = 1: ncclCommInitAll over one device, no collective issued), `--gather every-step` on one rank (nothing to gather: the line
says so), saip_batch_time_steps_gather without a communicator, saip_batch_time_steps_begin / _end against saip_batch_time_steps.  What N > 1
adds (the collective itself) cannot run here: tests/test_sharding_gloo.py and tests/test_bench_cli.py rehearse it on the CPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import workloads as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_single_process_launcher_on_one_gpu():
    d = _bench("--launcher", "multi", "--gpus", "1", "--steps", "50", "--warmup", "5", "--no-cpu-baseline")
    assert d["valid"] is True and d["n_gpus"] == 1 and d["config"]["launcher"].startswith("multi")
    assert d["config"]["kernel"] == "saip_cycle_oct" and d["config"]["gathers"] == 0 and d["config"]["gather_us"] == 0.0
    assert 2.0 < d["roofline"]["avg_launch_us"] < 30.0 and d["roofline"]["bound"] == "fp64-valu"
    assert d["value"] > 1e8 and d["value_event_timed"] >= d["value"]


def test_every_step_gather_mode_on_one_rank_and_the_oracle_error_in_the_line():
    d = _bench("--gather", "every-step", "--steps", "20", "--warmup", "5")
    assert d["valid"] is True and d["config"]["gather_mode"] == "every-step" and d["config"]["gathers"] == 0 and d["config"]["gather"].startswith("none")
    assert d["max_rel_err_vs_oracle"] is not None and d["max_rel_err_vs_oracle"] < 1e-5      # the second half of BASELINE.json's metric
    cb = d["cpu_baseline"]
    assert cb["cores"] >= 1 and str(cb["cores"]) in cb["thread_scan"] and cb["cpu_model"] and cb["value"] > 0
    d5 = _bench("--config", "5", "--steps", "5", "--warmup", "2", "--no-cpu-baseline")
    assert d5["roofline"]["bound"] == "mfma" and d5["config"]["kernel"] == "saip_cycle_wave"


def test_time_steps_entry_points_agree():
    import ctypes as C
    from sai_primitives_amd import capi
    from sai_primitives_amd.controller import controller_from_specs
    d = W.make_inputs(2, 1024)
    robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], 1024, device=0)
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    ctrl.setGoals(d["goals"])
    ms0 = ctrl.timeSteps(200, 20)
    ctrl.timeStepsBegin(200)
    ctrl.synchronize()
    ms1 = ctrl.timeStepsEnd()
    ms2, g2 = ctrl.timeStepsGather(200, None, None, False)     # no communicator: no collective, gather time 0
    ms3, g3 = ctrl.timeStepsGather(200, None, None, True)
    assert g2 == 0.0 and g3 == 0.0
    for ms in (ms1, ms2, ms3):
        assert 0.5 * ms0 < ms < 2.0 * ms0, (ms0, ms1, ms2, ms3)
    tau = ctrl.getTorques()
    assert np.isfinite(tau).all() and (ctrl.status & 1).sum() == 0
    with pytest.raises(Exception):
        capi.check(capi.lib().saip_batch_time_steps_gather(ctrl._h, None, None, 0, 0, C.byref(C.c_double()), C.byref(C.c_double())))


def test_flagged_instances_in_the_kernel_tail_or_on_the_list_give_the_same_torques():
    """saip_batch_set_flagged_recompute: config 3's stack with an eighth of the batch near the elbow singularity, packed into whole groups of eight
    (the slow tail's worst case) -- the in-kernel tail and the list launch must agree bit for bit, and with the oracle"""
    from oracle import Oracle
    from sai_primitives_amd.controller import controller_from_specs
    B = 512
    d = W.make_inputs(3, B)
    q = d["q"].copy()
    q[:64, 3] = -0.07 - 0.2 * np.random.default_rng(3).uniform(size=64)
    out = []
    for on_list in (False, True):
        robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
        ctrl.setFlaggedRecompute(on_list)
        robot.setQ(q)
        robot.setDq(d["dq"])
        robot.updateModel()
        ctrl.updateControllerTaskModels()
        ctrl.setGoals(d["goals"])
        out.append((ctrl.computeControlTorques(), ctrl.status.copy(), ctrl.kernelName()))
    assert out[0][2] == out[1][2] == "saip_cycle_oct"
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
    ref, st = Oracle(d["model"], d["tasks"]).step(q, d["dq"], d["goals"], nthreads=8)
    assert np.array_equal(out[0][1] & 9, st & 9) and (st & 8).sum() > 0
    ok = (st & 1) == 0
    assert W.torque_error(out[0][0][ok], ref[ok]) < 1e-5

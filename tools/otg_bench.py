#!/usr/bin/env python3
"""Internal OTG kernels alone, B = 4096 x 7-dof joint task: acceleration-limited against jerk-limited mode, a cycle in which every robot
computes a new trajectory (fresh goal) against the sampling cycles that follow.  Run under rocprofv3 --kernel-trace --stats for the kernel
durations; the wall-clock numbers printed here include the control cycle behind the OTG step.
   python tools/otg_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import workloads as W  # noqa: E402
from sai_primitives_amd.controller import controller_from_specs  # noqa: E402

B = int(os.environ.get("BATCH", "4096"))
d = W.make_inputs(1, B) if False else W.make_inputs(2, B)
tasks = [W.joint_task("joint_task")]
rng = np.random.default_rng(3)
for mode in ("acceleration", "jerk"):
    robot, ctrl, (jt,) = controller_from_specs(d["model"].name, tasks, B, device=0, disable_otg=False)
    if mode == "jerk":
        jt.enableInternalOtgJerkLimited(np.pi / 3, 2 * np.pi, 10 * np.pi)
    else:
        jt.enableInternalOtgAccelerationLimited(np.pi / 3, 2 * np.pi)
    robot.setQ(d["q"])
    robot.setDq(np.zeros((B, 7)))
    robot.updateModel()
    ctrl.reinitializeTasks()
    ctrl.updateControllerTaskModels()
    ctrl.computeControlTorques()
    t_new, t_samp = [], []
    for rep in range(6):
        jt.setGoalPosition(d["q"] + rng.uniform(-0.4, 0.4, d["q"].shape))   # every robot gets a new goal: the next cycle computes 4096 trajectories
        ctrl.synchronize()
        t0 = time.perf_counter()
        ctrl.stepAsync()
        ctrl.synchronize()
        t_new.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(100):
            ctrl.stepAsync()
        ctrl.synchronize()
        t_samp.append((time.perf_counter() - t0) / 100)
    print(f"{mode}-limited OTG, B = {B}: cycle with {B} new trajectories {np.median(t_new) * 1e6:.1f} us wall, sampling cycle {np.median(t_samp) * 1e6:.1f} us wall (OTG + control cycle)")

#!/usr/bin/env python3
"""Closed-loop period of a stack WITHOUT internal OTGs (cycle kernel integrates the state itself: one launch per period), cfg2 tasks.
   SAIP_ROLLOUT_NO_FUSE=1 selects the two-launch form for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import workloads as W
from sai_primitives_amd.controller import controller_from_specs

for B in [int(x) for x in os.environ.get("BATCHES", "4096,256").split(",")]:
    d = W.make_inputs(2, B)
    robot, ctrl, tasks = controller_from_specs(d["model"].name, d["tasks"], B, device=0, disable_otg=True)
    robot.setQ(d["q"]); robot.setDq(np.zeros((B, 7))); robot.updateModel()
    ctrl.reinitializeTasks()
    ctrl.updateControllerTaskModels()
    ctrl.rolloutAsync(50, 5e-4, 2, gravity=(0, 0, 0)); ctrl.synchronize()
    K = 400
    t0 = time.perf_counter()
    ctrl.rolloutAsync(K, 5e-4, 2, gravity=(0, 0, 0)); ctrl.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B}: closed-loop period without internal OTGs {dt / K * 1e6:.1f} us = {B * K / dt / 1e6:.1f} M robot-periods/s ({ctrl.kernelName()})")

#!/usr/bin/env python3
"""Closed-loop period on the device (internal OTGs -> control cycle -> 2 dynamics substeps) and the integrate kernel alone, cfg2 tasks.
   SAIP_LIB=<path> selects the build (same-session A/B)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import workloads as W  # noqa: E402
from sai_primitives_amd.controller import controller_from_specs  # noqa: E402

for B in [int(x) for x in os.environ.get("BATCHES", "4096,65536").split(",")]:
    d = W.make_inputs(2, B)
    robot, ctrl, tasks = controller_from_specs(d["model"].name, d["tasks"], B, device=0, disable_otg=False)
    robot.setQ(d["q"])
    robot.setDq(np.zeros((B, 7)))
    robot.updateModel()
    ctrl.reinitializeTasks()
    mf, jt = tasks
    mf.setGoalPosition(mf.getGoalPosition() + np.array([0.05, -0.04, 0.03]))
    ctrl.updateControllerTaskModels()
    ctrl.rolloutAsync(50, 5e-4, 2, gravity=(0, 0, 0))
    ctrl.synchronize()
    K = 400
    t0 = time.perf_counter()
    ctrl.rolloutAsync(K, 5e-4, 2, gravity=(0, 0, 0))
    ctrl.synchronize()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(200):
        ctrl.integrate(5e-4, 2, gravity=(0, 0, 0))
    ctrl.synchronize()
    di = time.perf_counter() - t0
    print(f"{os.path.basename(os.environ.get('SAIP_LIB', 'libsaip.so'))} B={B}: closed-loop period {dt / K * 1e6:.1f} us = {B * K / dt / 1e6:.1f} M robot-periods/s; "
          f"integrate (2 substeps) {di / 200 * 1e6:.1f} us")

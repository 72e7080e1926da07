"""diagnostic: the lane-per-instance kernel's general-law instantiations (selector 2) on the configurations with velocity saturation /
force and moment spaces, against the oracle.   python tools/lane_general_probe.py [configs...]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
import workloads as W
from oracle import Oracle
from sai_primitives_amd.controller import controller_from_specs
for cfg in [int(a) for a in sys.argv[1:]] or [9, 10]:
    B = 256
    d = W.make_inputs(cfg, B)
    robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
    ctrl.setKernel(2)
    robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.updateControllerTaskModels(); ctrl.setGoals(d["goals"])
    tau = ctrl.computeControlTorques()
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=4)
    print("cfg", cfg, ctrl.kernelName(), "err", W.torque_error(tau, ref), "status sums", int(ctrl.status.sum()), int(st.sum()), flush=True)

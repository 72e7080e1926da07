import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, workloads as W
from sai_primitives_amd.controller import controller_from_specs
B = 4096
d = W.make_inputs(2, B)
for always in (False, True, False, True):
    robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
    ctrl.setIntegratorTracking(always)
    robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.setGoals(d["goals"])
    ctrl.timeSteps(100, 0)
    ms = ctrl.timeSteps(1000, 0)
    print("integrators tracked every cycle" if always else "integrators tracked only while ki != 0", "%.3f us per launch" % (ms), ctrl.kernelName())

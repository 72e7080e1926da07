import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import numpy as np, workloads as W
from oracle import Oracle
from sai_primitives_amd.controller import controller_from_specs
for cfg, sizes in ((5, (1, 3, 65, 1000)), (6, (1, 7, 9, 4097)), (15, (1, 5, 8193))):
    for B in sizes:
        d = W.make_inputs(cfg, B)
        robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
        ctrl.setFlaggedTorquePolicy(True)
        robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.updateControllerTaskModels(); ctrl.setGoals(d["goals"])
        tau = ctrl.computeControlTorques()
        ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=8)
        ok = st != 1
        err = W.torque_error(tau[ok], ref[ok]) if ok.any() else 0.0
        print("config", cfg, "B", B, ctrl.kernelName(), "err %.2e" % err, "status equal", np.array_equal(ctrl.status & 9, st & 9))
        assert err < 1e-6 and np.array_equal(ctrl.status & 9, st & 9)
print("all sizes ok")

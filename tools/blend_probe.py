"""diagnostic: config 14 (blended strategies) on the eight-lane kernel against the oracle, per-instance error summary"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
import workloads as W
from oracle import Oracle
from sai_primitives_amd.controller import controller_from_specs
B = 256
d = W.make_inputs(14, B)
robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
ctrl.setKernel(3)
robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.updateControllerTaskModels(); ctrl.setGoals(d["goals"])
tau = ctrl.computeControlTorques()
ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=4)
err = np.abs(tau - ref).max(axis=1) / np.abs(ref).max(axis=1)
print("status equal", np.array_equal(ctrl.status, st), "blended", int((st == 8).sum()), "nan rows", int(np.isnan(tau).any(axis=1).sum()))
bad = np.flatnonzero(~(err < 1e-5))
print("bad", len(bad), bad[:10], "st", st[bad[:10]], "err", err[bad[:10]])
if len(bad):
    i = bad[0]
    print(tau[i], ref[i])

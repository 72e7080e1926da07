"""diagnostic: generic (rolled task loop) general-law instantiations of the lane kernel, selector 2, against the oracle"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
import workloads as W
from oracle import Oracle
from sai_primitives_amd.controller import controller_from_specs
B = 128
d = W.make_inputs(2, B)
stacks = {
    "7dof: pos(vel_sat) + ori + joint": ([W.motion_force_task("pos", "end-effector", (0, 0, 0.07), dirs_trans=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], vel_sat=True, lin_sat=0.05, ang_sat=0.2),
                                          W.motion_force_task("ori", "end-effector", (0, 0, 0.07), dirs_rot=[[1, 0, 0], [0, 1, 0], [0, 0, 1]]),
                                          W.joint_task("joint")], [d["goals"][0], d["goals"][0], d["goals"][1]]),
    "7dof: ori(force space) + pos + joint": ([W.motion_force_task("ori", "end-effector", (0, 0, 0.07), dirs_rot=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], moment_dim=1, moment_axis=[0, 0, 1.0]),
                                              W.motion_force_task("pos", "end-effector", (0, 0, 0.07), dirs_trans=[[1, 0, 0], [0, 1, 0], [0, 0, 1]]),
                                              W.joint_task("joint", vel_sat=True, sat=0.2)], None),
}
for name, (spec, goals) in stacks.items():
    if goals is None:
        g0 = np.concatenate([d["goals"][0], np.zeros((B, 6))], axis=1)
        g0[:, 27:30] = 0.3
        goals = [g0, d["goals"][0], d["goals"][1]]
    robot, ctrl, _ = controller_from_specs(d["model"].name, spec, B, device=0)
    ctrl.setKernel(2)
    robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.updateControllerTaskModels(); ctrl.setGoals(goals)
    tau = ctrl.computeControlTorques()
    ref, st = Oracle(d["model"], spec).step(d["q"], d["dq"], goals, nthreads=4)
    ok = (st & 1) == 0
    print(name, ctrl.kernelName(), "err", W.torque_error(tau[ok], ref[ok]), "flagged", int((st & 1).sum()), int((ctrl.status & 1).sum()), flush=True)

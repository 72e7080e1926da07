"""Closed-loop behaviour of the reference's example controllers, reproduced entirely on the device (control cycle + resident forward
dynamics, 1 kHz control / 2 kHz simulation, gravity-free world as in the examples' world.urdf).  The reference has no tests; the
comments in its example loops state what each scenario must show ("cannot do it because the partial task does not control X
direction", "this should be doable", ...).  These tests assert exactly those statements for a batch of perturbed start poses: they
decide signs, frames, selection projectors and the nullspace ordering through behaviour rather than through a second
implementation of the formulas."""
import numpy as np
import pytest

import workloads as W

pytestmark = pytest.mark.gpu

Q_NOM = np.array([0.0, -0.45, 0.0, -2.0, 0.0, 1.6, 0.8])   # elbow bent, wrist away from alignment: no singular direction nearby
ZERO_G = (0.0, 0.0, 0.0)


def _start(B, seed):
    return Q_NOM[None] + 0.12 * np.random.default_rng(seed).uniform(-1.0, 1.0, (B, 7))


def _ee(model, q, pos_in_link):
    fr = W.fk(model, q)
    li = model.link_index("end-effector")
    return fr[li][1] + fr[li][0] @ np.asarray(pos_in_link, float), fr[li][0]


def _rot(axis, angle):
    a = np.asarray(axis, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def _rotvec(R):
    """rotation vectors of a batch of rotation matrices"""
    ang = np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1))
    v = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
    return v * (ang / np.maximum(2 * np.sin(ang), 1e-12))[:, None]


def test_example05_circle_tracking_and_nullspace_joint_motion():
    """examples/05-using_robot_controller/05-using_robot_controller.cpp:140-203: the end effector tracks a circle in the y-z plane and
    an orientation oscillation about Y with full feed-forward (goal velocity and acceleration set every cycle), OTG disabled; later the
    joint task is asked to move joint 0 by 1.5 rad: it may only do so in the nullspace, the end effector keeps tracking."""
    import sai_primitives_amd as sp
    B = 24
    model = W.load_robot("panda_arm")
    robot = sp.SaiModel("panda_arm", B, device=0)
    mf = sp.MotionForceTask(robot, "end-effector", (0.0, 0.0, 0.07))
    jt = sp.JointTask(robot)
    mf.disableInternalOtg()                                            # :113
    jt.disableInternalOtg()                                            # (the example leaves it on; off keeps the step response sharp)
    ctrl = sp.RobotController(robot, [mf, jt])
    q0 = _start(B, 5)
    robot.setQ(q0)
    robot.setDq(np.zeros((B, 7)))
    robot.updateModel()
    ctrl.reinitializeTasks()
    ctrl.updateControllerTaskModels()
    x0, R0 = _ee(model, q0, (0, 0, 0.07))
    w_ori, amp = 2 * np.pi * 0.2, np.pi / 8
    rad, w = 0.05, 2 * np.pi * 0.33
    worst = {False: [0.0, 0.0], True: [0.0, 0.0]}                      # [position, orientation] error before / after the joint goal moves
    cycles, switch, settled = 2000, 1300, 1000                         # the start is at rest but the goal is moving: 1 s = 10 time constants
    for k in range(cycles):
        t = k * 1e-3
        ang = amp * np.sin(w_ori * t)
        Rg = _rot((0, 1, 0), ang).T @ R0                               # :159-162  R.transpose() * initial_orientation
        mf.setGoalOrientation(Rg)
        # the example feeds +ang_vel about Y forward (:163-166) although R^T R0 turns by -angle: its own goal is inconsistent by a sign
        # and it tracks with a 50 mrad lag (reproduced here on the first run of this test).  The derivative of the goal that IS set:
        mf.setGoalAngularVelocity(np.array([0.0, -amp * w_ori * np.cos(w_ori * t), 0.0]))
        mf.setGoalAngularAcceleration(np.array([0.0, amp * w_ori * w_ori * np.sin(w_ori * t), 0.0]))
        xg = x0 + rad * np.array([0.0, np.sin(w * t), 1 - np.cos(w * t)])
        mf.setGoalPosition(xg)
        mf.setGoalLinearVelocity(rad * w * np.array([0.0, np.cos(w * t), np.sin(w * t)]))
        mf.setGoalLinearAcceleration(rad * w * w * np.array([0.0, -np.sin(w * t), np.cos(w * t)]))
        if k == switch:                                                # :181-186
            qg = q0.copy()
            qg[:, 0] += 1.5
            jt.setGoalPosition(qg)
        ctrl.stepAsync()
        ctrl.integrate(5e-4, 2, gravity=ZERO_G)
        if k % 50 == 49 or k == cycles - 1:
            q, dq = ctrl.pullState()
            assert np.isfinite(ctrl.getTorques()).all() and (ctrl.status & 1).sum() == 0
            x, R = _ee(model, q, (0, 0, 0.07))
            t1 = (k + 1) * 1e-3
            xg1 = x0 + rad * np.array([0.0, np.sin(w * t1), 1 - np.cos(w * t1)])
            Rg1 = _rot((0, 1, 0), amp * np.sin(w_ori * t1)).T @ R0
            if k >= settled:
                w_ = worst[k >= switch]
                w_[0] = max(w_[0], np.abs(x - xg1).max())
                w_[1] = max(w_[1], np.linalg.norm(_rotvec(R @ Rg1.transpose(0, 2, 1)), axis=1).max())
            if k == switch - 1:
                assert np.abs(q[:, 0] - q0[:, 0]).max() < 0.3          # before the joint goal moves, joint 0 only follows the task
    print(f"example 05: worst tracking error {worst[False][0] * 1e3:.3f} mm / {worst[False][1] * 1e3:.3f} mrad before, "
          f"{worst[True][0] * 1e3:.3f} mm / {worst[True][1] * 1e3:.3f} mrad after the joint goal moved; joint 0 moved {np.abs(q[:, 0] - q0[:, 0]).min():.2f} rad")
    # feed-forward PD tracking of a 5 cm circle: sub-millimetre with exact dynamics; the 1.5 rad step of the joint goal acts through the
    # nullspace and disturbs it only through the 1 kHz discretisation
    assert worst[False][0] < 1e-3 and worst[False][1] < 5e-3
    assert worst[True][0] < 5e-3 and worst[True][1] < 2e-2
    # the joint task moved joint 0 towards its goal through the nullspace (one redundant degree of freedom: it cannot reach 1.5 rad,
    # but it must move the arm a lot while the end effector keeps tracking)
    assert np.abs(q[:, 0] - q0[:, 0]).min() > 0.15 and np.all((q[:, 0] - q0[:, 0]) > 0)


def test_example08_partial_task_moves_only_in_controlled_directions():
    """examples/08-partial_motion_force_task/08-partial_motion_force_task.cpp:110-190: translation controlled along Y and Z only,
    rotation about X only, compliant frame at (0.07, 0, 0); goals change in steps, so each phase is one on-device rollout
    (saip_batch_rollout_async: no host in the loop).  "try to move in X, cannot do it"; "move in Y and Z, this should be doable";
    "try to rotate around Z, should not be able to do it"; "rotate around X, this should be doable"; joint 0 moves in the nullspace."""
    import sai_primitives_amd as sp
    B = 24
    model = W.load_robot("panda_arm")
    robot = sp.SaiModel("panda_arm", B, device=0)
    pil = (0.07, 0.0, 0.0)
    mf = sp.MotionForceTask(robot, "end-effector", pil, None, [[0, 1, 0], [0, 0, 1]], [[1, 0, 0]])   # :113-122
    mf.disableInternalOtg()
    mf.disableVelocitySaturation()
    mf.setPosControlGains(100.0, 20.0)
    mf.setOriControlGains(100.0, 20.0)
    jt = sp.JointTask(robot)
    jt.disableInternalOtg()
    ctrl = sp.RobotController(robot, [mf, jt])
    jt.setGains(100.0, 20.0)                                           # :145
    q0 = _start(B, 8)
    robot.setQ(q0)
    robot.setDq(np.zeros((B, 7)))
    robot.updateModel()
    ctrl.reinitializeTasks()
    x0, R0 = _ee(model, q0, pil)

    def phase(cycles=700):
        ctrl.updateControllerTaskModels()
        ctrl.rolloutAsync(cycles, 5e-4, 2, gravity=ZERO_G)
        ctrl.synchronize()
        q, dq = ctrl.pullState()
        assert (ctrl.getTorques() is not None) and (ctrl.status & 1).sum() == 0
        x, R = _ee(model, q, pil)
        return q, dq, x, R

    goal = x0 + np.array([0.1, 0.0, 0.0])                              # :163-165  not controlled: nothing may happen
    mf.setGoalPosition(goal)
    q, dq, x, R = phase()
    assert np.abs(x - x0).max() < 2e-3 and np.abs(q - q0).max() < 2e-2
    goal = goal + np.array([0.0, 0.1, 0.1])                            # :167-169
    mf.setGoalPosition(goal)
    q, dq, x, R = phase(1500)
    assert np.abs(x[:, 1:] - goal[:, 1:]).max() < 2e-3                 # y and z arrive
    assert np.abs(x[:, 0] - goal[:, 0]).min() > 0.03                   # x does not follow its goal (it is free, not regulated)
    mf.setGoalOrientation(_rot((0, 0, 1), np.pi / 6) @ R0)             # :172-175  rotation about Z: not controlled
    q_before, R_before = q, R
    q, dq, x, R = phase()
    # (the free rotations have drifted a little from R0 during the translation, so the new goal has a small component about X:
    # a few hundredths of a radian of motion are legitimate; the 0.52 rad about Z are not followed)
    assert np.abs(q - q_before).max() < 0.1
    assert np.linalg.norm(_rotvec(R @ R_before.transpose(0, 2, 1)), axis=1).max() < 0.1
    Rg = _rot((1, 0, 0), np.pi / 6) @ R0                               # :177-180  rotation about X: controlled
    mf.setGoalOrientation(Rg)
    q, dq, x, R = phase(1500)
    err = _rotvec(Rg @ R.transpose(0, 2, 1))                           # remaining rotation to the goal, world frame
    assert np.abs(err[:, 0]).max() < 5e-3                              # no error left about the controlled axis
    assert np.abs(x[:, 1:] - goal[:, 1:]).max() < 2e-3
    qg = q.copy()
    qg[:, 0] += 0.5                                                    # :183-187  "move the first joint in the nullspace of the partial task"
    jt.setGoalPosition(qg)
    q1, dq1, x1, R1 = phase(2500)
    # the joint goal (0.5, 0, ..., 0) is not a nullspace direction: the four redundant degrees of freedom take joint 0 most of the way
    # (0.3 rad measured) and the controlled coordinates do not move
    assert (q1[:, 0] - q[:, 0]).min() > 0.2 and (q1[:, 0] - qg[:, 0]).max() < 0.0 and np.abs(dq1).max() < 1e-2
    err = _rotvec(Rg @ R1.transpose(0, 2, 1))
    assert np.abs(err[:, 0]).max() < 5e-3 and np.abs(x1[:, 1:] - goal[:, 1:]).max() < 2e-3
    print(f"example 08: joint 0 moved {(q1[:, 0] - q[:, 0]).min():.3f} .. {(q1[:, 0] - q[:, 0]).max():.3f} of 0.5 rad, y/z residual {np.abs(x1[:, 1:] - goal[:, 1:]).max() * 1e3:.3f} mm")


def test_example09_free_space_position_task_with_default_otg():
    """examples/09-3d_position_force_controller/09-3d_position_force_controller.cpp:113-170 before contact: a translation-only task
    (three controlled directions, no rotation) with the DEFAULT internal OTG moves 7 cm back and forth in x and y; the orientation is
    free.  (The contact phase of the example needs a collision world and is outside the engine.)"""
    import sai_primitives_amd as sp
    B = 24
    model = W.load_robot("panda_arm")
    robot = sp.SaiModel("panda_arm", B, device=0)
    mf = sp.MotionForceTask(robot, "end-effector", (0.0, 0.0, 0.0), None, [[1, 0, 0], [0, 1, 0], [0, 0, 1]], None)
    jt = sp.JointTask(robot)
    ctrl = sp.RobotController(robot, [mf, jt])
    q0 = _start(B, 13)
    robot.setQ(q0)
    robot.setDq(np.zeros((B, 7)))
    robot.updateModel()
    ctrl.reinitializeTasks()
    x0, _ = _ee(model, q0, (0, 0, 0))
    goal = x0.copy()
    for leg, step in enumerate((-0.07, +0.07)):                        # :160-166
        goal[:, 0] += step
        goal[:, 1] += step
        mf.setGoalPosition(goal)
        ctrl.updateControllerTaskModels()
        ctrl.rolloutAsync(1000, 5e-4, 2, gravity=ZERO_G)
        ctrl.synchronize()
        q, dq = ctrl.pullState()
        assert (ctrl.getTorques() is not None) and (ctrl.status & 1).sum() == 0
        x, _ = _ee(model, q, (0, 0, 0))
        gr, _, _ = mf.getInternalOtgStatus()
        print(f"example 09 leg {leg}: OTG reached {int(gr.sum())}/{B}, residual {np.abs(x - goal).max() * 1e3:.3f} mm")
        assert gr.all() and np.abs(x - goal).max() < 2e-3
    assert np.abs(x - x0).max() < 2e-3                                 # back where it started

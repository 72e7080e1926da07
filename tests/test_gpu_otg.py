"""GPU parity of the internal joint OTG (acceleration-limited, the reference default) through the C-ABI / facade:
 (i)  against the committed fixtures tests/golden/otg_*.npz = outputs of the reference's own vendored Ruckig (oracle/_ref),
 (ii) against that reference build directly on fresh random scenarios when the prebuilt library travelled to this box,
 (iii) the control law tracks the OTG output (JointTask.cpp:313-319): torques equal the oracle's for goal := desired state.
Tolerance: 1e-9 absolute on desired position / velocity / acceleration (rad, rad/s, rad/s^2); the expressions are evaluated in the
reference's order without FMA contraction, observed differences are ~1e-13."""
import glob
import os
import sys

import numpy as np
import pytest

import workloads as W

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import make_otg_golden as G  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-9
ALL = sorted(glob.glob(os.path.join(HERE, "golden", "otg_*.npz")))
FIX = [p for p in ALL if "otg_cart_" not in p]
CART = [p for p in ALL if "otg_cart_" in p]


def _controller(robot_name, S, q0):
    from sai_primitives_amd.controller import JointTask, RobotController, SaiModel
    B = q0.shape[0]
    robot = SaiModel(robot_name, B, device=0)
    robot.setQ(q0)
    robot.setDq(np.zeros_like(q0))
    robot.updateModel()
    jt = JointTask(robot, S)
    ctrl = RobotController(robot, [jt])   # OTG enabled by default, like the reference
    return robot, ctrl, jt


def _apply(ev, k, robot, ctrl, jt, state):
    for st, kind, payload in ev:
        if st != k:
            continue
        if kind == "goal":
            state["gp"], state["gv"] = payload
        elif kind == "limits":
            if len(payload) == 3:
                jt.enableInternalOtgJerkLimited(payload[0], payload[1], payload[2])
            else:
                jt.enableInternalOtgAccelerationLimited(payload[0], payload[1])
        elif kind == "mode":   # the other OTG mode: the wrapper re-initialises the OTG at the task's current position (JointTask.cpp:374-376, 400-402)
            which, qcur, lim = payload
            if which == "jerk":
                jt.enableInternalOtgJerkLimited(*lim)
            else:
                jt.enableInternalOtgAccelerationLimited(*lim)
            xc = qcur if jt.getTaskDof() == qcur.shape[1] else state["gp"] * 0 + (qcur @ state["S"].T)
            state["gp"], state["gv"] = xc, np.zeros_like(state["gv"])
        elif kind == "reinit":
            robot.setQ(payload)
            robot.updateModel()
            ctrl.reinitializeTasks()
            state["gp"], state["gv"] = jt.getGoalPosition(), np.zeros_like(state["gv"])


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_otg_matches_reference_fixture(path):
    sc, steps, desired, reached, meta = G.load(path)
    robot, ctrl, jt = _controller(sc["robot"], sc["S"], sc["q0"])
    if sc.get("jmax") is not None:
        jt.enableInternalOtgJerkLimited(sc["vmax"], sc["amax"], sc["jmax"])
    else:
        jt.enableInternalOtgAccelerationLimited(sc["vmax"], sc["amax"])
    x0 = sc["q0"] if sc["S"] is None else sc["q0"] @ sc["S"].T
    state = dict(gp=x0.copy(), gv=np.zeros_like(x0), S=sc["S"])
    want = {int(s): r for r, s in enumerate(steps)}
    worst = 0.0
    for k in range(sc["K"]):
        _apply(sc["events"], k, robot, ctrl, jt, state)
        jt.setGoalPosition(state["gp"])
        jt.setGoalVelocity(state["gv"])
        ctrl.updateControllerTaskModels()
        if k in want:
            ctrl.computeControlTorques()
            r = want[k]
            got = np.stack([jt.getDesiredPosition(), jt.getDesiredVelocity(), jt.getDesiredAcceleration()])
            err = np.abs(got - desired[r]).max()
            gr, fl, res = jt.getInternalOtgStatus()
            assert err < TOL, (k, err)
            assert np.array_equal(gr, reached[r]), k
            assert (res >= 0).all()
            worst = max(worst, err)
        else:
            ctrl.stepAsync()
    ctrl.synchronize()
    gr, fl, res = jt.getInternalOtgStatus()
    assert (((fl & 16) != 0).sum() > 0) == (meta["finished_with_velocity"] > 0)  # sticky "finished while moving" flag
    print(os.path.basename(path), "max |desired - reference|", worst)


def test_otg_matches_reference_build_random():
    import otg_ref as R
    if not R.available():
        pytest.skip("oracle/_ref/libsaip_otg_ref.so did not travel to this box")
    rng = np.random.default_rng(11)
    B = 96
    q_nom = np.array([0.0, -0.436, 0.0, -2.356, 0.0, 1.833, 0.0])
    q0 = q_nom + rng.uniform(-0.3, 0.3, (B, 7))
    robot, ctrl, jt = _controller("panda_arm", None, q0)
    refs = [R.JointsOtgRef(q0[i], 1e-3, np.pi / 3, 2 * np.pi) for i in range(B)]
    goal = q0.copy()
    gvel = np.zeros((B, 7))
    worst = 0.0
    for k in range(900):
        if k % 150 == 0:  # new goals every 150 cycles, most of them in flight; a third with a goal velocity
            goal = q_nom + rng.uniform(-0.9, 0.9, (B, 7))
            gvel = np.where(rng.uniform(size=(B, 1)) < 0.33, rng.uniform(-0.4, 0.4, (B, 7)), 0.0)
            if k == 450:
                gvel[:4] = 2.0  # beyond the velocity limit: ErrorInvalidInput, the wrapper keeps the previous output
        jt.setGoalPosition(goal)
        jt.setGoalVelocity(gvel)
        ctrl.updateControllerTaskModels()
        ctrl.computeControlTorques()
        got = np.stack([jt.getDesiredPosition(), jt.getDesiredVelocity(), jt.getDesiredAcceleration()], axis=1)
        gr, fl, res = jt.getInternalOtgStatus()
        for i in range(B):
            p, v, a, g, r = refs[i].step(goal[i], gvel[i])
            err = max(np.abs(p - got[i, 0]).max(), np.abs(v - got[i, 1]).max(), np.abs(a - got[i, 2]).max())
            assert err < TOL, (k, i, err, r, res[i])
            assert g == gr[i] and (r == res[i] or g), (k, i, r, res[i])
            worst = max(worst, err)
    print("random scenarios: max |desired - reference|", worst)


def test_jerk_limited_otg_matches_reference_build_random():
    """jerk-limited mode (third-order profiles) against the reference build on fresh random scenarios: new goals mostly in flight (non-zero
    velocity and acceleration at the switch), a third with goal velocities, per-joint limits"""
    import otg_ref as R
    if not R.available():
        pytest.skip("oracle/_ref/libsaip_otg_ref.so did not travel to this box")
    rng = np.random.default_rng(23)
    B = 64
    q_nom = np.array([0.0, -0.436, 0.0, -2.356, 0.0, 1.833, 0.0])
    q0 = q_nom + rng.uniform(-0.3, 0.3, (B, 7))
    vm = np.array([1.0, 0.8, 1.2, 0.9, 1.5, 1.1, 2.0])
    am = np.array([4.0, 3.0, 6.0, 5.0, 8.0, 7.0, 9.0])
    jm = np.array([30.0, 20.0, 50.0, 40.0, 60.0, 45.0, 80.0])
    robot, ctrl, jt = _controller("panda_arm", None, q0)
    jt.enableInternalOtgJerkLimited(vm, am, jm)
    refs = [R.JointsOtgRef(q0[i], 1e-3, np.pi / 3, 2 * np.pi) for i in range(B)]
    for i, o in enumerate(refs):
        o.reinitialize(q0[i])
        o.set_jerk_limits(vm, am, jm)
    goal = q0.copy()
    gvel = np.zeros((B, 7))
    worst = 0.0
    for k in range(1200):
        if k % 170 == 0:
            goal = q_nom + rng.uniform(-0.9, 0.9, (B, 7))
            gvel = np.where(rng.uniform(size=(B, 1)) < 0.33, rng.uniform(-0.3, 0.3, (B, 7)), 0.0)
        jt.setGoalPosition(goal)
        jt.setGoalVelocity(gvel)
        ctrl.updateControllerTaskModels()
        ctrl.computeControlTorques()
        got = np.stack([jt.getDesiredPosition(), jt.getDesiredVelocity(), jt.getDesiredAcceleration()], axis=1)
        gr, fl, res = jt.getInternalOtgStatus()
        for i in range(B):
            p, v, a, g, r = refs[i].step(goal[i], gvel[i])
            err = max(np.abs(p - got[i, 0]).max(), np.abs(v - got[i, 1]).max(), np.abs(a - got[i, 2]).max())
            assert err < TOL, (k, i, err, r, res[i])
            assert g == gr[i] and (r == res[i] or g), (k, i, r, res[i])
            worst = max(worst, err)
    print("jerk-limited random scenarios: max |desired - reference|", worst)


def test_control_law_tracks_otg_output():
    from oracle import Oracle
    d = W.make_inputs(1, 1)
    B = 32
    rng = np.random.default_rng(3)
    model = d["model"]
    q = np.repeat(d["q"], B, 0) + rng.uniform(-0.2, 0.2, (B, 7))
    dq = rng.uniform(-0.3, 0.3, (B, 7))
    robot, ctrl, jt = _controller("panda_arm", None, q)
    jt.setGains(100.0, 20.0, 0.0)
    goal = q + rng.uniform(-0.4, 0.4, (B, 7))
    tasks = [W.joint_task("joint_task", kp=100.0, kv=20.0)]
    orc = Oracle(model, tasks)
    robot.setDq(dq)
    robot.updateModel()
    for k in range(40):
        jt.setGoalPosition(goal)
        ctrl.updateControllerTaskModels()
        tau = ctrl.computeControlTorques()
    des = np.concatenate([jt.getDesiredPosition(), jt.getDesiredVelocity(), jt.getDesiredAcceleration()], axis=1)
    assert np.abs(des[:, :7] - q).max() > 1e-4          # the desired state has left the start ...
    assert np.abs(des[:, :7] - goal).max() > 1e-2        # ... and is not the goal: the law really tracks the OTG
    tau_ref, st = orc.step(q, dq, [des])
    assert W.torque_error(tau, tau_ref) < 1e-5


def test_otg_default_enabled_and_disable():
    """reference default: OTG on (JointTask.h:38).  disableInternalOtg makes the desired state the goal itself (JointTask.cpp:309-311)."""
    d = W.make_inputs(1, 1)
    q = np.repeat(d["q"], 4, 0)
    robot, ctrl, jt = _controller("panda_arm", None, q)
    goal = q + 0.3
    jt.setGoalPosition(goal)
    ctrl.updateControllerTaskModels()
    ctrl.computeControlTorques()
    assert jt.getInternalOtgEnabled()
    first = jt.getDesiredPosition()
    assert 0 < np.abs(first - q).max() < 1e-4            # one period at <= 2 pi rad/s^2
    jt.disableInternalOtg()
    ctrl.computeControlTorques()
    assert np.array_equal(jt.getDesiredPosition(), goal)
    with pytest.raises(ValueError):
        jt.enableInternalOtgAccelerationLimited(0.0, 1.0)   # OTG_joints.cpp:51-55
    with pytest.raises(ValueError):
        jt.enableInternalOtgAccelerationLimited([1.0, 2.0], [1.0, 2.0])  # JointTask.cpp:367-373


# ------------------------------------------------------------------ Cartesian OTG of a motion-force task (OTG_6dof_cartesian)
def _cart_controller(q0):
    from sai_primitives_amd.controller import MotionForceTask, RobotController, SaiModel
    B = q0.shape[0]
    robot = SaiModel("panda_arm", B, device=0)
    robot.setQ(q0)
    robot.setDq(np.zeros_like(q0))
    robot.updateModel()
    mf = MotionForceTask(robot, G.CART_LINK, G.CART_POINT)
    ctrl = RobotController(robot, [mf])   # OTG enabled by default, like the reference
    return robot, ctrl, mf


def _set_cart_goal(mf, goal):
    mf.setGoalPosition(goal[0])
    mf.setGoalOrientation(goal[1])
    mf.setGoalLinearVelocity(goal[2])
    mf.setGoalAngularVelocity(goal[3])


# Tolerance of the Cartesian wrapper: its frame algebra (rotation vector <-> matrix, sin/cos/atan2 of the device math library vs
# glibc) is evaluated once per goal change and differs by a few ulp, which the trajectory then carries: 1e-9 absolute.
@pytest.mark.parametrize("path", CART, ids=[os.path.basename(p)[:-4] for p in CART])
def test_cartesian_otg_matches_reference_fixture(path):
    sc, steps, desired, reached, meta = G.load_cart(path)
    robot, ctrl, mf = _cart_controller(sc["q0"])
    if len(sc["limits"]) == 6:
        mf.enableInternalOtgJerkLimited(*sc["limits"])
    else:
        mf.enableInternalOtgAccelerationLimited(*sc["limits"])
    B = sc["q0"].shape[0]
    p0, R0 = G.cart_pose(sc["q0"])
    goal = [p0, R0, np.zeros((B, 3)), np.zeros((B, 3))]
    want = {int(s): r for r, s in enumerate(steps)}
    worst = 0.0
    for k in range(sc["K"]):
        for st, kind, payload in sc["events"]:
            if st != k:
                continue
            if kind == "goal":
                goal = [np.array(x, float) for x in payload]
            elif kind == "limits":
                if len(payload) == 6:
                    mf.enableInternalOtgJerkLimited(*payload)
                else:
                    mf.enableInternalOtgAccelerationLimited(*payload)
            elif kind == "reinit":
                q, mask = payload
                robot.setQ(q)
                robot.updateModel()
                pr, Rr = G.cart_pose(q)
                if mask == 3:
                    ctrl.reinitializeTasks()
                    goal = [pr, Rr, np.zeros((B, 3)), np.zeros((B, 3))]
                else:  # reInitializeLinear: what parametrizeForceMotionSpaces triggers when the force space changes (MotionForceTask.cpp:828-856)
                    assert mask == 1
                    dim = 1 if not getattr(mf, "_toggle", False) else 0
                    mf._toggle = not getattr(mf, "_toggle", False)
                    assert mf.parametrizeForceMotionSpaces(dim, [0, 0, 1]) is True
                    mf.parametrizeForceMotionSpaces(0)      # back to pure motion control (second linear re-initialisation, same pose)
                    goal[0], goal[2] = pr, np.zeros((B, 3))
        _set_cart_goal(mf, goal)
        ctrl.updateControllerTaskModels()
        if k in want:
            ctrl.computeControlTorques()
            r = want[k]
            got = mf._desired_block()[:, :24]
            err = np.abs(got - desired[r]).max()
            gr, fl, res = mf.getInternalOtgStatus()
            assert err < TOL, (k, err, np.unravel_index(np.abs(got - desired[r]).argmax(), got.shape))
            assert np.array_equal(gr, reached[r]), k
            assert (res >= 0).all() and ((fl & 64) == 0).all()
            worst = max(worst, err)
        else:
            ctrl.stepAsync()
    ctrl.synchronize()
    print(os.path.basename(path), "max |desired - reference|", worst)


def test_cartesian_otg_default_on_and_invalid_goal():
    d = W.make_inputs(2, 8)
    robot, ctrl, mf = _cart_controller(d["q"])
    p0, R0 = G.cart_pose(d["q"])
    mf.setGoalPosition(p0 + 0.05)
    ctrl.updateControllerTaskModels()
    ctrl.computeControlTorques()
    first = mf.getDesiredPosition()
    assert 0 < np.abs(first - p0).max() < 1e-5 and np.abs(first - (p0 + 0.05)).max() > 0.04   # moving towards, not at, the goal
    bad = R0.copy()
    bad[:, 0, 0] += 0.1                                   # not a rotation: the reference throws, the engine flags and ignores
    mf.setGoalOrientation(bad)
    ctrl.computeControlTorques()
    gr, fl, res = mf.getInternalOtgStatus()
    assert ((fl & 64) != 0).all()
    mf.setGoalOrientation(R0)
    ctrl.computeControlTorques()
    gr, fl, res = mf.getInternalOtgStatus()
    assert ((fl & 64) == 0).all()
    with pytest.raises(ValueError):
        mf.enableInternalOtgAccelerationLimited(0.3, 2.0, -1.0, 1.0)   # OTG_6dof_cartesian.cpp:107-113

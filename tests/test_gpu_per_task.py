"""The reference's per-task plug-in interface on the GPU (TemplateTask.h:43-60): updateTaskModel(N_prec), computeTorques(),
computeTorques(tau_prec), getTaskNullspace / getPreviousTasksNullspace / getTaskAndPreviousNullspace -- driven by hand the way
/root/reference/examples/04-task_and_redundancy/04-task_and_redundancy.cpp:141-206 drives it, against the CPU oracle's per-task
functions (orc_task_step = mf/jt_update_task_model + mf/jt_compute_torques) and against RobotController."""
import numpy as np
import pytest

import workloads as W

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _standalone(cfg, B):
    """robot + tasks that belong to NO RobotController (example 04 builds none)"""
    import sai_primitives_amd as sp
    from sai_primitives_amd.controller import tasks_from_specs
    d = W.make_inputs(cfg, B)
    robot = sp.SaiModel(d["model"].name, B, device=0)
    return d, robot, tasks_from_specs(robot, d["tasks"])


def _set_goal(task, spec, g):
    if spec["type"] == "motion_force":
        task.setGoalPosition(g[:, 0:3])
        task.setGoalOrientation(g[:, 3:12].reshape(-1, 3, 3))
        task.setGoalLinearVelocity(g[:, 12:15])
        task.setGoalAngularVelocity(g[:, 15:18])
        task.setGoalLinearAcceleration(g[:, 18:21])
        task.setGoalAngularAcceleration(g[:, 21:24])
    else:
        m = task.getTaskDof()
        task.setGoalPosition(g[:, :m])
        task.setGoalVelocity(g[:, m:2 * m])
        task.setGoalAcceleration(g[:, 2 * m:3 * m])


def _manual_cycle(d, robot, tasks, *, device, compensate):
    """example 04's loop body for any hierarchy: returns per-task torques, the N_prec each task was updated with, its N and N N_prec"""
    n, B = robot.dof(), robot.batch_size
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    N_prec = np.eye(n)
    out = dict(tau=[], N_prec=[], N=[], N_total=[])
    for t in tasks:
        t.updateTaskModel(N_prec)                                     # 04-...cpp:144, 150
        N_prec = t.getTaskAndPreviousNullspace(device=device)         # :145
        out["N_prec"].append(t.getPreviousTasksNullspace())
        out["N"].append(t.getTaskNullspace())
        out["N_total"].append(N_prec.numpy() if device else N_prec)
    total = np.zeros((B, n))
    for t, spec, g in zip(tasks, d["tasks"], d["goals"]):
        _set_goal(t, spec, g)
        tau_t = t.computeTorques(total) if compensate else t.computeTorques()   # :192-193 / RobotController.cpp:82-84
        assert (t.status & 1).sum() == 0
        out["tau"].append(tau_t)
        total = total + tau_t
    out["total"] = total
    return out


@pytest.mark.parametrize("cfg,B", [(2, 512), (3, 512), (6, 256), (5, 64), (7, 128), (8, 128)])
def test_example04_manual_hierarchy_matches_oracle(cfg, B):
    from oracle import Oracle
    d, robot, tasks = _standalone(cfg, B)
    orc = Oracle(d["model"], d["tasks"])
    n = robot.dof()
    for compensate in (False, True):
        got = _manual_cycle(d, robot, tasks, device=False, compensate=compensate)
        prec = np.zeros((B, n))
        Np = np.broadcast_to(np.eye(n), (B, n, n)).copy()
        for i in range(len(tasks)):
            tau_ref, N_ref, Nt_ref, st = orc.task_step(i, d["q"], d["dq"], d["goals"][i], N_prec=Np, tau_prec=prec if compensate else None)
            assert (st & 1).sum() == 0
            assert np.abs(got["N_prec"][i] - Np).max() < 1e-8   # the N N_prec of the task above, as the device computed it
            assert np.abs(got["N"][i] - N_ref).max() < 1e-8, (cfg, i)
            assert np.abs(got["N_total"][i] - Nt_ref).max() < 1e-8, (cfg, i)
            err = W.torque_error(got["tau"][i], tau_ref)
            print(f"cfg{cfg} task {i} compensate={compensate}: per-task torque err {err:.2e}")
            assert err < TOL, (cfg, i, compensate)
            Np = Nt_ref
            prec = prec + tau_ref
    # the manual sum with computeTorques(tau_prec) IS RobotController::computeControlTorques (RobotController.cpp:79-85)
    ref, st = orc.step(d["q"], d["dq"], d["goals"], nthreads=4)
    assert st.sum() == 0
    assert W.torque_error(got["total"], ref) < TOL
    # nullspaces chained on the device (nothing crosses PCIe) give the same numbers bit for bit
    dev = _manual_cycle(d, robot, tasks, device=True, compensate=True)
    for a, b in zip(got["tau"], dev["tau"]):
        assert np.array_equal(a, b)
    # the same tasks moved into a RobotController afterwards: configuration and goals travel with them
    import sai_primitives_amd as sp
    ctrl = sp.RobotController(robot, tasks)
    robot.updateModel()
    ctrl.updateControllerTaskModels()
    tau_ctrl = ctrl.computeControlTorques()
    assert ctrl.status.sum() == 0
    assert W.torque_error(tau_ctrl, ref) < TOL


def test_arbitrary_nprec_and_order_contract():
    """N_prec need not come from this hierarchy: any (dof, dof) matrix, one for all instances or one per instance"""
    import sai_primitives_amd as sp
    from oracle import Oracle
    B = 96
    d, robot, (mf, jt) = _standalone(3, B)
    orc = Oracle(d["model"], d["tasks"])
    n = robot.dof()
    rng = np.random.default_rng(5)
    # a projector that removes two random joint-space directions (per instance), and one shared full-rank matrix
    V = np.linalg.qr(rng.standard_normal((B, n, 2)))[0]
    Np_inst = np.eye(n)[None] - V @ V.transpose(0, 2, 1)
    Np_shared = np.eye(n) + 0.1 * rng.standard_normal((n, n))
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    for task, i in ((jt, 1), (mf, 0)):
        _set_goal(task, d["tasks"][i], d["goals"][i])
        for Np in (Np_inst, Np_shared):
            task.updateTaskModel(Np)
            tau = task.computeTorques()
            Npb = np.broadcast_to(Np, (B, n, n))
            tau_ref, N_ref, Nt_ref, st = orc.task_step(i, d["q"], d["dq"], d["goals"][i], N_prec=Npb)
            ok = (st & 1) == 0   # (a motion-force task can land inside the singularity bounds under an arbitrary N_prec: blended, status 8)
            assert ok.sum() > B // 2 and np.array_equal(st, task.status)
            assert W.torque_error(tau[ok], tau_ref[ok]) < TOL
            assert np.abs(task.getTaskAndPreviousNullspace()[ok] - Nt_ref[ok]).max() < 1e-8
            assert np.isnan(tau[~ok]).all()
    # shape errors of the reference (JointTask.cpp:219-229)
    with pytest.raises(ValueError, match="not square"):
        jt.updateTaskModel(np.zeros((n, n + 1)))
    with pytest.raises(ValueError, match="not consistent with robot dof"):
        jt.updateTaskModel(np.eye(n + 1))
    # a new robot state invalidates the task model: computeTorques must be preceded by updateTaskModel again
    robot.setQ(d["q"] + 0.01)
    robot.updateModel()
    with pytest.raises(sp.SaipError, match="updateTaskModel"):
        jt.computeTorques()
    jt.updateTaskModel(np.eye(n))
    assert np.isfinite(jt.computeTorques()).all()


def test_per_task_calls_inside_a_controller_leave_its_results_alone():
    """tasks of a RobotController can be driven by hand as well (the controller does nothing else, RobotController.cpp:68-85); the per-task
    launches write to buffers of their own, so the controller's torques / status of the last cycle stay what they were"""
    from oracle import Oracle
    from sai_primitives_amd.controller import controller_from_specs
    B = 128
    d = W.make_inputs(2, B)
    robot, ctrl, (mf, jt) = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    ctrl.updateControllerTaskModels()
    ctrl.setGoals(d["goals"])
    tau = ctrl.computeControlTorques()
    n = robot.dof()
    N_hier = jt.getPreviousTasksNullspace()   # hierarchy diagnostics (re-evaluates the models)
    assert np.array_equal(ctrl.getTorques(), tau)
    mf.updateTaskModel(np.eye(n))
    jt.updateTaskModel(mf.getTaskAndPreviousNullspace(device=True))
    assert np.abs(jt.getPreviousTasksNullspace() - N_hier).max() < 1e-9
    t0 = mf.computeTorques()
    t1 = jt.computeTorques(t0)
    assert W.torque_error(t0 + t1, tau) < 1e-9
    assert np.array_equal(ctrl.getTorques(), tau) and ctrl.status.sum() == 0
    ref, _ = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"])
    assert W.torque_error(t0 + t1, ref) < TOL


def test_per_task_integrators_and_otg_advance_in_compute_torques_only():
    """updateTaskModel is pure (model only); computeTorques advances the integrators (JointTask.cpp:323-324) and the internal OTG
    (:313-319) once per call, like the reference"""
    import sai_primitives_amd as sp
    from oracle import Oracle
    B = 64
    d = W.make_inputs(2, B)
    spec = [dict(d["tasks"][1], ki=3.0)]   # the joint task of config 2 alone, with an integral gain
    goals = [d["goals"][1]]
    robot = sp.SaiModel(d["model"].name, B, device=0)
    from sai_primitives_amd.controller import tasks_from_specs
    (jt,) = tasks_from_specs(robot, spec)
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    _set_goal(jt, spec[0], goals[0])
    n = robot.dof()
    orc = Oracle(d["model"], spec)
    refs = [orc.step(d["q"], d["dq"], goals)[0] for _ in range(3)]   # the oracle keeps its integrators between steps
    for c in range(3):
        jt.updateTaskModel(np.eye(n))
        jt.updateTaskModel(np.eye(n))   # twice: no state is advanced by the model update
        tau = jt.computeTorques()
        assert W.torque_error(tau, refs[c]) < TOL, c
    assert W.torque_error(refs[0], refs[2]) > 1e-6


def test_joint_task_rank_decision_at_the_range_basis_tolerance():
    """JointTask.cpp:156 -> SaiModel::matrixRangeBasis: directions of S N_prec below 1e-3 of the largest singular value are dropped.
    N_prec = Q diag(1, ..., 1, s) Q^T puts the smallest singular value of the projected Jacobian of a full joint task at exactly s:
    the device's rank decision (certificates first, exact eigen-solve when they cannot tell) must agree with the oracle's SVD on both
    sides of the tolerance, down to one part in 1e9 from it (tests/test_oracle.py pins the oracle's side against LAPACK)."""
    import sai_primitives_amd as sp
    from sai_primitives_amd.controller import tasks_from_specs
    from oracle import Oracle
    ratios = [1e-3 * (1 + 1e-4), 1e-3 * (1 - 1e-4), 1e-3 * (1 + 1e-9), 1e-3 * (1 - 1e-9), 0.5, 1e-8, 2e-3, 5e-4]
    B, n = 8 * len(ratios), 7
    d = W.make_inputs(2, B)
    specs = [W.joint_task("joint_task", decoupling=W.FULL_DYNAMIC_DECOUPLING)]
    robot = sp.SaiModel(d["model"].name, B, device=0)
    (jt,) = tasks_from_specs(robot, specs)
    orc = Oracle(d["model"], specs)
    rng = np.random.default_rng(23)
    Np = np.empty((B, n, n))
    keep = np.empty(B, bool)
    for b in range(B):
        s = ratios[b % len(ratios)]
        Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
        D = np.ones(n)
        D[rng.integers(n)] = s
        Np[b] = Q @ np.diag(D) @ Q.T
        keep[b] = s >= 1e-3
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    goal = d["goals"][1]
    _set_goal(jt, specs[0], goal)
    jt.updateTaskModel(Np)
    tau = jt.computeTorques()
    N = jt.getTaskNullspace()
    tau_ref, N_ref, _, st = orc.task_step(0, d["q"], d["dq"], goal, N_prec=Np)
    assert (st & 1).sum() == 0 and np.array_equal(jt.status, st)
    rank = np.rint(np.trace(np.eye(n)[None] - N, axis1=1, axis2=2)).astype(int)
    rank_ref = np.rint(np.trace(np.eye(n)[None] - N_ref, axis1=1, axis2=2)).astype(int)
    assert np.array_equal(rank_ref, np.where(keep, n, n - 1))
    assert np.array_equal(rank, rank_ref), (rank, rank_ref)
    err = W.torque_error(tau, tau_ref)
    print(f"range-basis straddle: torque err {err:.2e}")
    assert err < TOL

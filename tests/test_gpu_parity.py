"""GPU parity tests proper: the HIP engine (through the C-ABI / the reference-shaped facade) against
the CPU oracle and the committed golden vectors.  FP64; tolerance from BASELINE.json's north_star:
1e-5 relative per joint torque (measured as SURVEY.md 8(d): |tau-tau_ref| / max(|tau_ref|, 1e-3*||tau_ref||_inf))."""
import glob
import os

import numpy as np
import pytest

import workloads as W
import restatement as RS  # oracle/restatement.py: the NumPy restatement of the reference path (test infrastructure)

pytestmark = pytest.mark.gpu
TOL = 1e-5

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cfg*.npz")))


def _engine(model_name, tasks, B, **kw):
    from sai_primitives_amd.controller import controller_from_specs
    robot, ctrl, objs = controller_from_specs(model_name, tasks, B, device=0, **kw)
    ctrl.setFlaggedTorquePolicy(True)   # tests: an instance the engine refuses (status 1) shows as NaN, like in the oracle (the default holds the last torque)
    return robot, ctrl, objs


def _flagging(spec):
    """engine extra: the blended singularity strategies switched off -> instances outside the non-singular branch are flagged (status 1)"""
    return [dict(t, singularity_strategies=False) if t["type"] == "motion_force" else t for t in spec]


def _run(robot, ctrl, q, dq, goals):
    robot.setQ(q)
    robot.setDq(dq)
    robot.updateModel()
    ctrl.updateControllerTaskModels()
    ctrl.setGoals(goals)
    return ctrl.computeControlTorques()


def _load(path):
    z = np.load(path)
    cfg = int(z["cfg"])
    tasks = W.config_tasks(cfg)
    goals = [z[f"goal{i}"] for i in range(len(tasks))]
    jla = bool(z["joint_limit_avoidance"]) if "joint_limit_avoidance" in z else False
    return cfg, W.CONFIGS[cfg]["robot"], tasks, z["q"], z["dq"], goals, z["tau"], bool(z["gravity_comp"]), bool(z["torque_saturation"]), jla, z["status"]


@pytest.mark.parametrize("kernel", [0, 1], ids=["auto", "general"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_engine_matches_golden(path, kernel):
    cfg, robot_name, tasks, q, dq, goals, tau_ref, grav, sat, jla, st_ref = _load(path)
    robot, ctrl, _ = _engine(robot_name, tasks, q.shape[0])
    ctrl.setKernel(kernel)   # 0: lane kernel where it applies; 1: force the general workgroup-per-instance kernel
    ctrl.enableJointLimitAvoidance(jla)
    ctrl.enableGravityCompensation(grav)
    ctrl.enableTorqueSaturation(sat)
    tau = _run(robot, ctrl, q, dq, goals)
    assert np.array_equal(ctrl.status, st_ref)   # all 0, except config 12: 2 = task reduced to its non-singular subspace
    err = W.torque_error(tau, tau_ref)
    print(os.path.basename(path), ctrl.kernelName(), "err", err)
    assert err < TOL


@pytest.mark.parametrize("cfg,B", [(2, 4096), (3, 4096), (5, 4096), (6, 256), (9, 1024), (10, 1024)])
def test_engine_matches_oracle_full_size(cfg, B):
    from oracle import Oracle
    d = W.make_inputs(cfg, B)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=8)
    assert st.sum() == 0 and ctrl.status.sum() == 0
    err = W.torque_error(tau, ref)
    print("cfg", cfg, "B", B, ctrl.kernelName(), "err", err)
    assert err < TOL


@pytest.mark.parametrize("B", [1, 63, 65, 4097])
@pytest.mark.parametrize("kernel", [0, 1], ids=["auto", "general"])
def test_ragged_batches(B, kernel):
    """batch sizes that do not fill a wavefront / the 32-instance padding of the device arrays: every instance still matches the
    oracle and nothing is read or written past B (B = 1 is the reference's own single-robot case)"""
    from oracle import Oracle
    d = W.make_inputs(2, B)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B)
    ctrl.setKernel(kernel)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert tau.shape == (B, 7) and ctrl.status.shape == (B,) and ctrl.status.sum() == 0
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=4)
    assert W.torque_error(tau, ref) < TOL


def test_config4_shard_size_and_lean_variant():
    """BASELINE config 4 = 8 x 8192 instances of the config-2 stack: one per-GPU shard (rank 3 of the seeded stream) against the
    oracle; then the same rows tiled to 20480 (eight-lane kernel in rounds) and 28672 instances (large-batch instantiation of the lane kernel)"""
    from oracle import Oracle
    d = W.make_inputs(4, 8192, shard=3)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], 8192)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=8)
    assert st.sum() == 0 and ctrl.status.sum() == 0
    assert W.torque_error(tau, ref) < TOL
    # past 8192 instances the eight-lane kernel runs in rounds (automatic choice up to 24576 for this stack), beyond that the lane kernel's
    # large-batch instantiation takes over: both against the same oracle rows
    for B, kernel in ((20480, "saip_cycle_oct"), (28672, "saip_cycle_lane")):
        reps = B // 4096
        q, dq = np.tile(d["q"][:4096], (reps, 1)), np.tile(d["dq"][:4096], (reps, 1))
        goals = [np.tile(g[:4096], (reps, 1)) for g in d["goals"]]
        robot2, ctrl2, _ = _engine(d["model"].name, d["tasks"], B)
        tau2 = _run(robot2, ctrl2, q, dq, goals)
        assert ctrl2.kernelName() == kernel
        assert ctrl2.status.sum() == 0
        assert W.torque_error(tau2[:4096], ref[:4096]) < TOL and W.torque_error(tau2[-4096:], ref[:4096]) < TOL


def test_config4_whole_batch_equals_its_eight_shards():
    """BASELINE config 4 at its FULL size on one GPU: 65536 instances in one batch, and the same stream as the eight 8192-instance shards a
    node would run (make_inputs(shard=r) = rank r's block).  Size-independent property of the sharded path: an instance's torque does not
    depend on which shard (or batch size, or kernel: the whole batch runs the lane kernel, a shard the eight-lane kernel) computed it;
    the concatenation of the shards is what the final all-gather delivers.  Plus the oracle on a random sample of the whole batch."""
    from oracle import Oracle
    world, per = 8, 8192
    d = W.make_inputs(4, world * per)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], world * per)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.status.sum() == 0 and np.isfinite(tau).all()
    whole_kernel = ctrl.kernelName()
    del robot, ctrl
    worst = 0.0
    for r in range(world):
        ds = W.make_inputs(4, per, shard=r)
        assert np.array_equal(ds["q"], d["q"][r * per:(r + 1) * per])          # rank r's block of the seeded stream
        rb, cs, _ = _engine(ds["model"].name, ds["tasks"], per)
        ts = _run(rb, cs, ds["q"], ds["dq"], ds["goals"])
        assert cs.status.sum() == 0 and cs.kernelName() == "saip_cycle_oct"
        worst = max(worst, W.torque_error(ts, tau[r * per:(r + 1) * per]))
        del rb, cs
    print("config 4: whole batch on", whole_kernel, "vs eight shards on saip_cycle_oct: max rel difference", worst)
    assert worst < 1e-9
    idx = np.random.default_rng(4).choice(world * per, 2048, replace=False)
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"][idx], d["dq"][idx], [g[idx] for g in d["goals"]], nthreads=8)
    assert st.sum() == 0 and W.torque_error(tau[idx], ref) < TOL


def test_general_kernel_explicitly_selected():
    from oracle import Oracle
    d = W.make_inputs(2, 512)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], 512)
    ctrl.setKernel(1)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.kernelName().startswith("saip_cycle_wg")
    ref, _ = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=4)
    assert W.torque_error(tau, ref) < TOL


def test_hierarchy_property_full_size():
    """size-independent property at the full config-2 batch: the joint task must not
    accelerate the motion-force task, J M^-1 (tau_total - tau_task1_only) = 0."""
    d = W.make_inputs(2, 4096)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], 4096)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    robot1, ctrl1, _ = _engine(d["model"].name, d["tasks"][:1], 4096)
    tau1 = _run(robot1, ctrl1, d["q"], d["dq"], d["goals"][:1])
    model = d["model"]
    fr = W.fk(model, d["q"])
    li = model.link_index("end-effector")
    p = fr[li][1] + fr[li][0] @ np.array([0, 0, 0.07])
    J = W.jacobian(model, fr, li, p)
    Minv = np.linalg.inv(W.mass_matrix(model, fr))
    acc = np.einsum("bij,bjk,bk->bi", J, Minv, tau - tau1)
    assert np.abs(acc).max() < 1e-7 * np.abs(tau).max()


def test_singular_instances_default_blended_and_flagged_when_switched_off():
    """reference default: an instance inside the singularity bounds runs the blended type-1 / type-2 strategies (status bit 8, valid
    torques).  With the strategies switched off (engine extra) it is refused: status 1, its torques held (default) or NaN."""
    from oracle import Oracle
    from sai_primitives_amd.controller import controller_from_specs
    d = W.make_inputs(2, 64)
    q = d["q"].copy()
    q[5] = np.array([0.0, 0.0, 0.0, -0.0698, 0.0, 0.0, 0.0])  # outstretched arm
    q[17] = np.array([0.3, 0.5, 0.0, -0.08, 0.0, 0.2, 0.0])
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], 64)
    tau = _run(robot, ctrl, q, d["dq"], d["goals"])
    ref, st = Oracle(d["model"], d["tasks"]).step(q, d["dq"], d["goals"])
    assert np.array_equal(ctrl.status, st) and st[5] == 8 and (st & 1).sum() == 0
    assert np.isfinite(tau).all() and W.torque_error(tau, ref) < TOL
    spec = _flagging(d["tasks"])
    robot, ctrl, _ = _engine(d["model"].name, spec, 64)
    tau = _run(robot, ctrl, q, d["dq"], d["goals"])
    ref, st = Oracle(d["model"], spec).step(q, d["dq"], d["goals"])
    assert np.array_equal(ctrl.status, st) and st[5] == 1
    assert np.isnan(tau[st == 1]).all()
    assert W.torque_error(tau[st == 0], ref[st == 0]) < TOL
    # default policy: a refused instance keeps the torques of its last valid cycle, nothing is ever NaN
    robot, ctrl, _ = controller_from_specs(d["model"].name, spec, 64, device=0)
    tau0 = _run(robot, ctrl, d["q"], d["dq"], d["goals"])          # every instance valid
    assert ctrl.status.sum() == 0
    tau1 = _run(robot, ctrl, q, d["dq"], d["goals"])               # instances 5 and 17 refused
    assert ctrl.status[5] == 1 and ctrl.status[17] == 1
    assert np.array_equal(tau1[st == 1], tau0[st == 1]) and W.torque_error(tau1[st == 0], ref[st == 0]) < TOL


@pytest.mark.parametrize("cfg", [2, 3, 6])
def test_branch_predicates_on_unfiltered_configurations(cfg):
    """the workloads reject near-singular samples; here 4096 joint configurations are drawn over the WHOLE joint range without any
    rejection, so the certificate / exact-eigenvalue fast paths of the kernels see the full spread of conditioning, including
    the blending region of SingularityHandler.  Every instance must get the oracle's status (1 = outside the non-singular branch)
    and, where valid, the oracle's torques, on both kernel paths."""
    from oracle import Oracle
    B = 4096
    d = W.make_inputs(cfg, B)
    model = d["model"]
    rng = np.random.default_rng(100 + cfg)
    q = model.q_lower + (model.q_upper - model.q_lower) * rng.uniform(0.02, 0.98, (B, model.dof))
    # (a) the blended strategies switched off: the branch predicate alone decides between status 0 and status 1
    spec = _flagging(d["tasks"])
    ref, st = Oracle(model, spec).step(q, d["dq"], d["goals"], nthreads=8)
    assert 0 < (st == 1).sum() < B
    for kernel in (0, 1):
        robot, ctrl, _ = _engine(model.name, spec, B)
        ctrl.setKernel(kernel)
        tau = _run(robot, ctrl, q, d["dq"], d["goals"])
        differ = np.flatnonzero(ctrl.status != st)
        assert differ.size == 0, (ctrl.kernelName(), differ[:10], W._task_sv_ratio(model, d["tasks"], q[differ[:10]]))
        assert np.isnan(tau[st == 1]).all()
        err = W.torque_error(tau[st == 0], ref[st == 0])
        print("cfg", cfg, ctrl.kernelName(), "flagged", int((st == 1).sum()), "of", B, "err on the rest", err)
        assert err < TOL
    # (b) the default configuration (the reference's: handling enforced, blended strategies): EVERY instance gets finite torques equal to
    # the oracle's, the near-singular ones through the type-1 / type-2 strategies (status bit 8)
    ref, st = Oracle(model, d["tasks"]).step(q, d["dq"], d["goals"], nthreads=8)
    assert (st & 8).sum() > 0
    for kernel in (0, 1):
        robot, ctrl, _ = _engine(model.name, d["tasks"], B)
        ctrl.setKernel(kernel)
        tau = _run(robot, ctrl, q, d["dq"], d["goals"])
        differ = np.flatnonzero(ctrl.status != st)
        assert differ.size == 0, (ctrl.kernelName(), differ[:10], ctrl.status[differ[:10]], st[differ[:10]])
        ok = (st & 1) == 0
        assert ok.sum() >= B - 2 and np.isfinite(tau[ok]).all()   # (sigma_min / sigma_0 < 1e-6 would be refused: not met by random postures)
        err = W.torque_error(tau[ok], ref[ok])
        print("cfg", cfg, ctrl.kernelName(), "default configuration: blended", int((st & 8 != 0).sum()), "of", B, "err", err)
        assert err < TOL


def test_singularity_handling_disabled_full_size():
    """disableSingularityHandling() (MotionForceTask.h:723): near-singular instances are neither blended nor flagged, the motion-force task
    acts on its non-singular subspace (SingularityHandler.cpp:107-115, 146-148, 319-330).  Lane kernel + device-side slow path and
    the general kernel alone against the oracle, 4096 instances of which a third sit inside the blending region; with the handling
    strategy enabled (default) the same instances are flagged."""
    from oracle import Oracle
    B = 4096
    d = W.make_inputs(12, B)
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=8)
    assert (st == 2).sum() > B // 4 and (st == 1).sum() == 0
    for kernel in (0, 1):
        robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B)
        ctrl.setKernel(kernel)
        tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
        assert np.array_equal(ctrl.status, st)
        err = W.torque_error(tau, ref)
        print("handling disabled, kernel", ctrl.kernelName(), "truncated", int((st == 2).sum()), "err", err)
        assert err < TOL
    tasks_on = _flagging(W.config_tasks(2))   # handling enforced, blended strategies switched off
    robot, ctrl, _ = _engine(d["model"].name, tasks_on, B)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert np.array_equal(ctrl.status == 1, st == 2) and np.isnan(tau[st == 2]).all() and np.isfinite(tau[st == 0]).all()


def test_closed_loop_force_control_over_cycles():
    """setClosedLoopForceControl / MomentControl (MotionForceTask.cpp:327-349, 357-378, 484-487): sensed force / moment from a rotated,
    offset sensor, PI feedback with saturation, feed-forward gains; the force / moment integrators advance over cycles with changing
    sensor readings.  Both kernels against the oracle; toggling the flag resets the integrators (:973-986); passivity is refused."""
    import sai_primitives_amd as sp
    from oracle import Oracle
    B = 512
    d = W.make_inputs(13, B)
    rng = np.random.default_rng(9)
    for kernel in (0, 1):
        orc = Oracle(d["model"], d["tasks"])
        robot, ctrl, tasks = _engine(d["model"].name, d["tasks"], B)
        ctrl.setKernel(kernel)
        goals = [g.copy() for g in d["goals"]]
        robot.setQ(d["q"])
        robot.setDq(d["dq"])
        robot.updateModel()
        ctrl.updateControllerTaskModels()
        ctrl.setGoals(goals)
        for k in range(4):
            fs, ms = rng.uniform(-8, 8, (B, 3)), rng.uniform(-1.5, 1.5, (B, 3))
            goals[0][:, 30:33], goals[0][:, 33:36] = fs, ms
            tasks[0].updateSensedForceAndMoment(fs, ms)
            tau = ctrl.computeControlTorques()
            ref, st = orc.step(d["q"], d["dq"], goals, nthreads=8)
            assert st.sum() == 0 and ctrl.status.sum() == 0
            err = W.torque_error(tau, ref)
            print("closed-loop force control, kernel", ctrl.kernelName(), "cycle", k, "err", err)
            assert err < TOL
            if kernel == 0:   # closed-loop stacks (without the passivity controller) run in the lane kernel's general-law instantiations
                assert ctrl.kernelName() == "saip_cycle_lane"
        # toggling closed-loop force control resets the linear integrators: the next cycle equals a fresh oracle's first cycle
        tasks[0].setClosedLoopForceControl(False)
        tasks[0].setClosedLoopForceControl(True)
        tasks[0].setClosedLoopMomentControl(False)
        tasks[0].setClosedLoopMomentControl(True)
        tau = ctrl.computeControlTorques()
        ref, _ = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], goals, nthreads=8)
        assert W.torque_error(tau, ref) < TOL


def test_closed_loop_force_control_with_passivity_controller():
    """enablePassivity() (MotionForceTask.h:630; examples 07 / 09): windowed passivity observer + passivity controller of
    POPCExplicitForceControl.cpp:35-95 around the force loop.  600 cycles with changing sensor readings and joint velocities:
    the observer detects activity, the controller scales the feedback down (Rc < 1) and recovers; torques match the oracle in
    every compared cycle.  disablePassivity() re-initialises the observer (:24-29)."""
    from oracle import Oracle
    B = 64
    d = W.make_inputs(13, B)
    tasks_spec = W.config_tasks(13)
    tasks_spec[0]["passivity"] = True
    rng = np.random.default_rng(21)
    orc = Oracle(d["model"], tasks_spec)
    robot, ctrl, tasks = _engine(d["model"].name, tasks_spec, B)
    goals = [g.copy() for g in d["goals"]]
    robot.setQ(d["q"])
    ctrl.setGoals(goals)
    worst, taus = 0.0, []
    for k in range(600):
        if k % 9 == 0:
            scale = 0.05 if (k // 150) % 2 else 1.0          # alternate rough and quiet sensor phases
            fs = goals[0][:, 24:27] + scale * rng.uniform(-8, 8, (B, 3))
            goals[0][:, 30:33] = fs
            tasks[0].updateSensedForceAndMoment(fs, goals[0][:, 33:36])
        dq = d["dq"] * (1.0 + 0.5 * np.sin(0.05 * k))
        robot.setDq(dq)
        robot.updateModel()
        ctrl.updateControllerTaskModels()
        tau = ctrl.computeControlTorques()
        ref, st = orc.step(d["q"], dq, goals, nthreads=8)
        assert st.sum() == 0 and ctrl.status.sum() == 0
        worst = max(worst, W.torque_error(tau, ref))
        taus.append(tau)
    print("passivity controller, 600 cycles, kernel", ctrl.kernelName(), "worst err", worst)
    assert worst < TOL
    # the controller did something: with passivity off the same final cycle gives different torques
    orc2 = Oracle(d["model"], W.config_tasks(13))
    for k in range(3):
        ref_off, _ = orc2.step(d["q"], dq, goals, nthreads=8)
    assert np.abs(ref_off - taus[-1]).max() > 1e-3
    tasks[0].disablePassivity()
    tasks[0].enablePassivity()


def test_everything_on_at_once():
    """interaction test: internal OTGs on both tasks (Cartesian + joint), closed-loop force control with the passivity controller,
    singularity handling disabled, torque saturation and gravity compensation, 200 cycles.  The engine's desired states are fed to
    the oracle as goals, so every cycle checks the whole chain OTG -> law -> hierarchy -> post-processing."""
    from oracle import Oracle
    B = 96
    d = W.make_inputs(13, B)
    spec = W.config_tasks(13)
    spec[0]["passivity"] = True
    spec[0]["singularity_handling"] = False
    orc = Oracle(d["model"], spec, gravity_comp=True, torque_saturation=True)
    robot, ctrl, tasks = _engine(d["model"].name, spec, B, disable_otg=False)
    ctrl.enableGravityCompensation(True)
    ctrl.enableTorqueSaturation(True)
    mf, jt = tasks
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    ctrl.reinitializeTasks()
    rng = np.random.default_rng(4)
    g0, g1 = d["goals"][0].copy(), d["goals"][1].copy()
    mf.setGoalPosition(g0[:, 0:3])
    mf.setGoalOrientation(g0[:, 3:12].reshape(B, 3, 3))
    mf.setGoalForce(g0[:, 24:27])
    mf.setGoalMoment(g0[:, 27:30])
    jt.setGoalPosition(g1[:, 0:7])
    worst = 0.0
    for k in range(200):
        if k % 10 == 0:
            g0[:, 30:33], g0[:, 33:36] = g0[:, 24:27] + rng.uniform(-3, 3, (B, 3)), rng.uniform(-1, 1, (B, 3))
            mf.updateSensedForceAndMoment(g0[:, 30:33], g0[:, 33:36])
        ctrl.updateControllerTaskModels()
        tau = ctrl.computeControlTorques()
        des0 = mf._desired_block()                     # OTG output + pass-through force rows, goal layout
        des1 = jt._desired_block()
        assert np.array_equal(des0[:, 24:36], g0[:, 24:36])
        ref, st = orc.step(d["q"], d["dq"], [des0, des1], nthreads=8)
        assert np.array_equal(ctrl.status & 3, st & 3) and (st != 1).all()
        worst = max(worst, W.torque_error(tau, ref))
    print("everything on, 200 cycles, kernel", ctrl.kernelName(), "worst err", worst)
    assert worst < TOL
    assert np.abs(des0[:, 0:3] - d["goals"][0][:, 0:3]).max() < 0.06 and np.abs(des0[:, 0:3] - g0[:, 0:3]).max() < 0.051


@pytest.mark.parametrize("cfg,B,opts", [(12, 192, {}), (11, 128, dict(joint_limit_avoidance=True, gravity_comp=True, torque_saturation=True)),
                                        (5, 24, {}), (3, 128, {}), (9, 128, {})])
def test_internal_otg_with_every_stack(cfg, B, opts):
    """the control law must track the internal OTGs' output on every kind of stack and kernel path: near-singular instances through
    the slow path (12), joint-limit avoidance + gravity + saturation (11), the 30-dof chain with two Cartesian OTGs and a 32-lane joint
    OTG (5), partial tasks (3), hybrid force/velocity-saturation laws (9).  Torques = oracle evaluated at the desired states, 6 cycles,
    a re-initialisation in the middle."""
    from oracle import Oracle
    d = W.make_inputs(cfg, B)
    orc = Oracle(d["model"], d["tasks"], **opts)
    robot, ctrl, tasks = _engine(d["model"].name, d["tasks"], B, disable_otg=False)
    ctrl.enableJointLimitAvoidance(opts.get("joint_limit_avoidance", False))
    ctrl.enableGravityCompensation(opts.get("gravity_comp", False))
    ctrl.enableTorqueSaturation(opts.get("torque_saturation", False))
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    ctrl.reinitializeTasks()
    ctrl.setGoals(d["goals"])
    worst = 0.0
    for k in range(6):
        if k == 3:
            ctrl.reinitializeTasks()
            orc.reset_state()
            ctrl.setGoals(d["goals"])
        ctrl.updateControllerTaskModels()
        tau = ctrl.computeControlTorques()
        des = []
        for t, g in zip(tasks, d["goals"]):
            x = t._desired_block()
            des.append(x[:, :g.shape[1]])
        ref, st = orc.step(d["q"], d["dq"], des, nthreads=8)
        assert np.array_equal(ctrl.status, st) and (st != 1).all()
        worst = max(worst, W.torque_error(tau, ref))
        for t, x, g in zip(tasks, des, d["goals"]):     # the desired state has not jumped to the goal
            if g.shape[1] >= 24:
                assert np.abs(x[:, 0:3] - g[:, 0:3]).max() > 1e-3
    print("cfg", cfg, ctrl.kernelName(), "OTG on every task, worst err", worst)
    assert worst < TOL


@pytest.mark.parametrize("cfg", [3, 6, 8])
def test_singularity_handling_disabled_on_partial_and_multi_task_stacks(cfg):
    """rank-truncated motion-force tasks beyond the full 6-dof case: a 3-dof position task (3), two motion-force tasks on the 8-dof
    sliding-base Panda (6), a planar partial task with IMPEDANCE decoupling (8); every second instance has the elbow nearly straight.
    Status (0 / 2) and torques against the oracle on both kernel paths."""
    from oracle import Oracle
    B = 96
    d = W.make_inputs(cfg, B)
    spec = W.config_tasks(cfg)
    for t in spec:
        if t["type"] == "motion_force":
            t["singularity_handling"] = False
    q = d["q"].copy()
    col = 3 if d["model"].dof == 7 else 4
    q[::2, col] = -0.07 - 0.25 * np.random.default_rng(cfg).uniform(size=B // 2)
    ref, st = Oracle(d["model"], spec).step(q, d["dq"], d["goals"], nthreads=8)
    assert (st == 2).sum() >= 8 and (st == 1).sum() == 0
    for kernel in (0, 1):
        robot, ctrl, _ = _engine(d["model"].name, spec, B)
        ctrl.setKernel(kernel)
        tau = _run(robot, ctrl, q, d["dq"], d["goals"])
        assert np.array_equal(ctrl.status, st)
        err = W.torque_error(tau, ref)
        print("cfg", cfg, ctrl.kernelName(), "truncated", int((st == 2).sum()), "err", err)
        assert err < TOL


def test_task_nullspaces_match_oracle():
    from oracle import Oracle
    d = W.make_inputs(3, 32)
    robot, ctrl, tasks = _engine(d["model"].name, d["tasks"], 32)
    _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    N0 = tasks[0].getTaskNullspace()
    N1 = tasks[1].getTaskNullspace()
    o = Oracle(d["model"], d["tasks"])
    for b in range(32):
        det = o.details(d["q"][b], d["dq"][b], [g[b] for g in d["goals"]])
        assert np.abs(N0[b] - det["N"][0]).max() < 1e-8
        assert np.abs(N1[b] - det["N"][1]).max() < 1e-7
    # TemplateTask.h:79-89: the nullspace a task was updated with and the one it hands on
    assert np.array_equal(tasks[0].getPreviousTasksNullspace(), np.broadcast_to(np.eye(N0.shape[1]), N0.shape))
    assert np.allclose(tasks[1].getPreviousTasksNullspace(), N0, atol=0, rtol=0)
    assert np.allclose(tasks[1].getTaskAndPreviousNullspace(), N1 @ N0, atol=1e-15)


def test_integrators_and_repeated_cycles():
    from oracle import Oracle
    d = W.make_inputs(2, 128)
    tasks = W.config_tasks(2)
    tasks[0].update(ki_pos=5.0, ki_ori=7.0)
    tasks[1].update(ki=3.0)
    robot, ctrl, _ = _engine(d["model"].name, tasks, 128)
    o = Oracle(d["model"], tasks)
    for _ in range(3):
        tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
        ref, _ = o.step(d["q"], d["dq"], d["goals"])
        assert W.torque_error(tau, ref) < TOL


def test_reinitialize_tasks_snaps_goals_to_current_pose():
    d = W.make_inputs(2, 64)
    robot, ctrl, tasks = _engine(d["model"].name, d["tasks"], 64)
    robot.setQ(d["q"])
    robot.setDq(np.zeros_like(d["dq"]))
    robot.updateModel()
    ctrl.reinitializeTasks()
    model = d["model"]
    fr = W.fk(model, d["q"])
    li = model.link_index("end-effector")
    p = fr[li][1] + fr[li][0] @ np.array([0, 0, 0.07])
    assert np.abs(tasks[0].getGoalPosition() - p).max() < 1e-12
    assert np.abs(tasks[0].getGoalOrientation() - fr[li][0]).max() < 1e-12
    assert np.abs(tasks[1].getGoalPosition() - d["q"]).max() == 0
    ctrl.updateControllerTaskModels()
    tau = ctrl.computeControlTorques()
    assert np.abs(tau).max() < 1e-9  # zero error, zero velocity -> zero command


def test_call_order_and_otg_are_loud():
    import sai_primitives_amd as sp
    d = W.make_inputs(2, 8)
    robot, ctrl, tasks = _engine(d["model"].name, d["tasks"], 8)
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    with pytest.raises(sp.SaipError):  # compute before update
        ctrl.computeControlTorques()
    ctrl.updateControllerTaskModels()
    with pytest.raises(ValueError):                    # OTG_joints::setMaxJerk: "max jerk cannot be 0 or negative" (OTG_joints.cpp:79-83)
        tasks[1].enableInternalOtgJerkLimited(1.0, 2.0, 0.0)
    with pytest.raises(ValueError):                    # JointTask.cpp:391-397: vector sizes
        tasks[1].enableInternalOtgJerkLimited([1.0, 2.0], [1.0, 2.0], [1.0, 2.0])
    with pytest.raises(ValueError):                    # OTG_6dof_cartesian::setMaxJerk (OTG_6dof_cartesian.cpp:128-132)
        tasks[0].enableInternalOtgJerkLimited(0.3, 2.0, 10.0, 1.0, 6.0, -30.0)
    tasks[1].disableInternalOtg()
    tasks[0].disableInternalOtg()
    ctrl.setGoals(d["goals"])
    assert np.isfinite(ctrl.computeControlTorques()).all()


def test_force_space_reparametrisation_resets_linear_goal():
    """parametrizeForceMotionSpaces returns whether it changed and, if so, snaps the linear goal to the current position
    and clears the integrators (MotionForceTask.cpp:828-856); the angular goal is untouched."""
    d = W.make_inputs(2, 64)
    robot, ctrl, tasks = _engine(d["model"].name, d["tasks"], 64)
    _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    ori_before = tasks[0].getGoalOrientation()
    assert tasks[0].parametrizeForceMotionSpaces(1, [0, 0, 1]) is True
    assert tasks[0].parametrizeForceMotionSpaces(1, [0, 0, 2]) is False     # same normalised axis
    model = d["model"]
    fr = W.fk(model, d["q"])
    li = model.link_index("end-effector")
    p = fr[li][1] + fr[li][0] @ np.array([0, 0, 0.07])
    assert np.abs(tasks[0].getGoalPosition() - p).max() < 1e-12
    assert np.array_equal(tasks[0].getGoalOrientation(), ori_before)
    # and the cycle now matches the oracle configured the same way, with the reset goals
    from oracle import Oracle
    spec = W.config_tasks(2)
    spec[0].update(force_dim=1, force_axis=[0, 0, 1])
    goals = [np.concatenate([tasks[0]._get_goal()], axis=1), d["goals"][1]]
    tasks[0].setGoalForce(np.tile([1.0, -2.0, 3.0], (64, 1)))
    goals[0][:, 24:27] = [1.0, -2.0, 3.0]
    ctrl.updateControllerTaskModels()
    tau = ctrl.computeControlTorques()
    ref, _ = Oracle(model, spec).step(d["q"], d["dq"], goals)
    assert W.torque_error(tau, ref) < TOL


def test_joint_limit_avoidance_full_size():
    """RobotController's JLA wrap (RobotController.cpp:96-112) on 2048 instances pushed into the limit zones, vs the oracle"""
    from oracle import Oracle
    d = W.make_inputs(11, 2048)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], 2048)
    ctrl.enableJointLimitAvoidance(True)
    ctrl.enableTorqueSaturation(True)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    o = Oracle(d["model"], d["tasks"], joint_limit_avoidance=True, torque_saturation=True)
    ref, st = o.step(d["q"], d["dq"], d["goals"], nthreads=8)
    assert np.array_equal(ctrl.status, st)
    ok = st == 0
    assert ok.sum() > 1000
    assert W.torque_error(tau[ok], ref[ok]) < TOL
    # the wrap really acted: it differs from the un-wrapped torques on most instances
    ctrl.enableJointLimitAvoidance(False)
    tau0 = ctrl.computeControlTorques()
    assert (np.abs(tau0[ok] - tau[ok]).max(axis=1) > 1e-6).mean() > 0.5


def test_closed_loop_tracking_toy_integrator():
    """behavioural check of SURVEY 8(c)(iii): driving 64 Pandas with the engine's torques in a toy semi-implicit Euler loop
    (zero gravity, dynamics M(q) ddq = tau from the NumPy model) brings the end-effector to its goal and holds the posture
    task's nullspace motion bounded -- the scenario of examples/05-using_robot_controller (:150-207) without graphics."""
    d = W.make_inputs(2, 64)
    model = d["model"]
    robot, ctrl, tasks = _engine(model.name, d["tasks"], 64)
    q, dq = d["q"].copy(), np.zeros_like(d["dq"])
    robot.setQ(q)
    robot.setDq(dq)
    robot.updateModel()
    ctrl.reinitializeTasks()
    li = model.link_index("end-effector")
    fr = W.fk(model, q)
    x0 = fr[li][1] + fr[li][0] @ np.array([0, 0, 0.07])
    x_goal = x0 + np.array([0.03, -0.02, 0.02])
    tasks[0].setGoalPosition(x_goal)
    dt = 1e-3
    err0 = np.linalg.norm(x_goal - x0, axis=1).max()
    for step in range(600):
        robot.setQ(q)
        robot.setDq(dq)
        robot.updateModel()
        ctrl.updateControllerTaskModels()
        tau = ctrl.computeControlTorques()
        ok = ctrl.status == 0
        assert ok.mean() > 0.9  # a few instances may wander into the singular band; they are flagged, not wrong
        tau = np.where(ok[:, None], tau, 0.0)
        M = W.mass_matrix(model, W.fk(model, q))
        ddq = np.linalg.solve(M, tau[..., None])[..., 0]
        dq = dq + dt * ddq
        q = q + dt * dq
    fr = W.fk(model, q)
    x = fr[li][1] + fr[li][0] @ np.array([0, 0, 0.07])
    err = np.linalg.norm(x_goal - x, axis=1)[ok]
    print("closed loop: initial error", err0, "final max error", err.max())
    assert err.max() < 0.1 * err0


def test_per_task_reinitialise_reset_integrators_and_getters():
    """TemplateTask::reInitializeTask of one task, MotionForceTask::resetIntegrators{,Linear,Angular}, JointTask::resetIntegrators and the
    host-side getters of the facade (MotionForceTask.h / JointTask.h accessors)"""
    from oracle import Oracle
    B = 64
    d = W.make_inputs(2, B)
    spec = W.config_tasks(2)
    spec[0].update(ki_pos=5.0, ki_ori=7.0)
    spec[1].update(ki=3.0)
    robot, ctrl, (mf, jt) = _engine(d["model"].name, spec, B)
    orc = Oracle(d["model"], spec)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    ref, _ = orc.step(d["q"], d["dq"], d["goals"], nthreads=4)
    assert W.torque_error(tau, ref) < TOL
    # second cycle without a reset differs from the first (integrators advanced); after resetting every integrator it repeats the first
    tau2 = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert W.torque_error(tau2, ref) > 1e-6
    mf.resetIntegrators()
    jt.resetIntegrators()
    tau3 = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert W.torque_error(tau3, ref) < TOL
    # linear part only: the orientation integrator keeps its two cycles -> equals an oracle whose linear / joint integrators were cleared
    mf.resetIntegratorsLinear()
    mf.resetIntegratorsAngular()
    jt.resetIntegrators()
    assert W.torque_error(_run(robot, ctrl, d["q"], d["dq"], d["goals"]), ref) < TOL
    # per-task reinitialisation: the joint task's goal becomes the current posture, the motion-force goal stays
    g_before = mf.getGoalPosition().copy()
    jt.reInitializeTask()
    assert np.allclose(jt.getGoalPosition(), d["q"]) and np.array_equal(mf.getGoalPosition(), g_before)
    assert np.allclose(jt.getCurrentPosition(), d["q"]) and np.allclose(jt.getCurrentVelocity(), d["dq"])
    # getters
    assert mf.getPosControlGains()[0].kp == spec[0]["kp_pos"] and mf.getOriControlGains()[0].ki == 7.0 and jt.getGains()[0].ki == 3.0
    assert mf.getForceSpaceDimension() == 0 and not mf.getVelocitySaturationEnabled() and jt.getJointSelectionMatrix().shape == (7, 7)
    mf.setFeedforwardForceGain(0.8)
    assert mf.getFeedforwardForceGain() == 0.8 and mf.getMaxForceControlFeedbackOutput() == 20.0
    import sai_primitives_amd as sp
    with pytest.raises(sp.SaipError, match="updateTaskModel"):   # the per-task interface needs the task's own model first (TemplateTask.h:43)
        mf.computeTorques()


def test_current_pose_and_goal_reached():
    """MotionForceTask::getCurrentPosition / getCurrentOrientation and goalPositionReached / goalOrientationReached (MotionForceTask.cpp:548-579,
    sigma matrices :892-971), JointTask::goalPositionReached (JointTask.cpp:437-446)"""
    B = 48
    d = W.make_inputs(2, B)
    model = d["model"]
    robot, ctrl, (mf, jt) = _engine(model.name, d["tasks"], B)
    _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    fr = W.fk(model, d["q"])
    li = model.link_index("end-effector")
    x = fr[li][1] + fr[li][0] @ np.array([0.0, 0.0, 0.07])
    assert np.abs(mf.getCurrentPosition() - x).max() < 1e-14 and np.abs(mf.getCurrentOrientation() - fr[li][0]).max() < 1e-14
    e = np.linalg.norm(d["goals"][0][:, 0:3] - x, axis=1)
    assert np.array_equal(mf.goalPositionReached(0.04), e < 0.04) and 0 < (e < 0.04).sum() < B
    Rd = d["goals"][0][:, 3:12].reshape(B, 3, 3)
    dphi = np.array([RS.orientation_error(Rd[b], fr[li][0][b]) for b in range(B)])
    assert np.array_equal(mf.goalOrientationReached(0.15), np.linalg.norm(dphi, axis=1) < 0.15)
    ctrl.reinitializeTasks()
    assert mf.goalPositionReached(1e-9).all() and mf.goalOrientationReached(1e-9).all() and jt.goalPositionReached(1e-9).all()
    # force space along world z: the position error along z no longer counts (sigmaPosition = I - z z^T)
    mf.parametrizeForceMotionSpaces(1, [0.0, 0.0, 1.0])
    g = mf.getGoalPosition() + np.array([0.0, 0.0, 0.2])
    mf.setGoalPosition(g)
    assert mf.goalPositionReached(1e-9).all()
    mf.setGoalPosition(g + np.array([0.01, 0.0, 0.0]))
    assert not mf.goalPositionReached(5e-3).any() and mf.goalPositionReached(2e-2).all()
    # the joint task behind a 6-dof task only sees the one direction left in its range: a goal offset inside the range of the
    # motion-force task does not count
    jt.setGoalPosition(d["q"] + 0.1)
    r = jt.goalPositionReached(0.05)
    assert r.dtype == bool and r.shape == (B,)


@pytest.mark.parametrize("kernel", [0, 1], ids=["auto", "general"])
def test_integrators_advance_once_when_a_lower_task_sends_the_instance_to_the_slow_path(kernel):
    """Two motion-force tasks with integral gains; the LOWER one is near-singular in a third of the batch and has its singularity
    handling disabled, so the lane kernel flags those instances after the upper task's control law has run and the general kernel
    launched behind recomputes them.  The upper task's integrators must advance exactly once per cycle (they are committed at the end of
    the cycle, and only for instances that do not end flagged), whichever kernel ran: three cycles against the oracle."""
    from oracle import Oracle
    B = 192
    d = W.make_inputs(12, B)   # every third instance has the elbow nearly straight
    tasks = [W.motion_force_task("ori_task", "end-effector", (0, 0, 0.07), dirs_rot=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], ki_ori=4.0, singularity_handling=False),
             W.motion_force_task("pos_task", "end-effector", (0, 0, 0.07), dirs_trans=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], ki_pos=3.0, singularity_handling=False),
             W.joint_task("joint_task", ki=2.0)]
    goals = [d["goals"][0], d["goals"][0], d["goals"][1]]
    robot, ctrl, _ = _engine(d["model"].name, tasks, B)
    ctrl.setKernel(kernel)
    orc = Oracle(d["model"], tasks)
    for cycle in range(3):
        tau = _run(robot, ctrl, d["q"] + 1e-3 * cycle, d["dq"], goals)   # a new state each cycle (the models must be re-evaluated)
        ref, st = orc.step(d["q"] + 1e-3 * cycle, d["dq"], goals, nthreads=4)
        assert np.array_equal(ctrl.status & 1, st & 1) and (st & 1).sum() == 0
        if cycle == 0:
            assert (ctrl.status & 2).sum() > B // 8, "the scenario must send a sizeable part of the batch through the reduced-task path"
        err = W.torque_error(tau, ref)
        print(f"cycle {cycle} kernel {ctrl.kernelName()}: err {err:.2e}, reduced instances {(ctrl.status & 2 != 0).sum()}")
        assert err < TOL, cycle

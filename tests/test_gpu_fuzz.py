"""Randomised parity sweep: the headline stack (MotionForceTask + JointTask on the Panda) with every scalar option drawn at random -- per-axis
gains, integrators, decoupling type per task, bounded-inertia thresholds, velocity saturation, loop time step, singularity bounds, compliant
frame, gravity compensation, torque saturation -- on a batch that mixes ordinary postures with config 14's type-1 / type-2 singular ones,
three cycles of a drifting state, through the automatic kernel choice (eight-lane kernel, one- or two-wavefront form by batch size) and
the general kernel, against the CPU oracle.  Complements the fixed scenarios of test_gpu_parity.py / test_gpu_oct.py: the options are not
exercised one at a time but in combinations nobody wrote down."""
import numpy as np
import pytest

import workloads as W

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _random_spec(rng):
    mf = W.motion_force_task("motion_force_task", "end-effector", tuple(rng.uniform(-0.05, 0.1, 3)))
    jt = W.joint_task("joint_task")
    aniso = rng.random() < 0.5
    mf["kp_pos"] = list(rng.uniform(50, 200, 3)) if aniso else float(rng.uniform(50, 200))
    mf["kv_pos"] = list(rng.uniform(10, 30, 3)) if aniso else float(rng.uniform(10, 30))
    mf["kp_ori"] = list(rng.uniform(100, 300, 3)) if aniso else float(rng.uniform(100, 300))
    mf["kv_ori"] = list(rng.uniform(15, 35, 3)) if aniso else float(rng.uniform(15, 35))
    if rng.random() < 0.5:
        mf["ki_pos"] = list(rng.uniform(0, 8, 3)) if aniso else float(rng.uniform(0, 8))
        mf["ki_ori"] = list(rng.uniform(0, 8, 3)) if aniso else float(rng.uniform(0, 8))
    jt["kp"] = list(rng.uniform(20, 120, 7)) if aniso else float(rng.uniform(20, 120))
    jt["kv"] = list(rng.uniform(8, 25, 7)) if aniso else float(rng.uniform(8, 25))
    if rng.random() < 0.5:
        jt["ki"] = list(rng.uniform(0, 5, 7)) if aniso else float(rng.uniform(0, 5))
    dec = [W.FULL_DYNAMIC_DECOUPLING, W.BOUNDED_INERTIA_ESTIMATES, W.BOUNDED_INERTIA_ESTIMATES, W.IMPEDANCE]
    mf["decoupling"] = dec[rng.integers(4)]
    jt["decoupling"] = dec[rng.integers(4)]
    thr = float(rng.choice([0.05, 0.1, 0.3, 0.6]))   # one shared threshold or two: several clamped entries from 0.3 up
    mf["bie_threshold"] = thr
    jt["bie_threshold"] = thr if rng.random() < 0.7 else float(rng.choice([0.1, 0.4]))
    if rng.random() < 0.4:
        jt.update(vel_sat=True, sat=float(rng.uniform(0.2, 1.0)))
    dt = float(rng.choice([5e-4, 1e-3, 2e-3]))
    mf["dt"] = jt["dt"] = dt
    smax = float(rng.uniform(3e-2, 9e-2))
    mf["s_max"], mf["s_min"] = smax, smax * float(rng.uniform(0.05, 0.5))
    opts = dict(gravity_comp=bool(rng.random() < 0.5), torque_saturation=bool(rng.random() < 0.5))
    return [mf, jt], opts


@pytest.mark.parametrize("seed", range(8))
def test_random_option_combinations_match_oracle(seed):
    from oracle import Oracle
    from sai_primitives_amd.controller import controller_from_specs
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.choice([96, 520, 4100]))
    spec, opts = _random_spec(rng)
    da, db = W.make_inputs(2, B), W.make_inputs(14, B)
    pick = rng.random(B) < 0.5                     # half ordinary postures, half config 14's (a third of those type-1, a third type-2 singular)
    q0 = np.where(pick[:, None], da["q"], db["q"])
    dq = np.where(pick[:, None], da["dq"], db["dq"])
    goals = [np.where(pick[:, None], ga, gb) for ga, gb in zip(da["goals"], db["goals"])]
    orc = Oracle(da["model"], spec, **opts)
    seq = [q0]
    for _ in range(2):
        seq.append(seq[-1] + 1e-3 * dq + 1e-3 * rng.standard_normal(q0.shape))
    refs = [orc.step(q, dq, goals, nthreads=8) for q in seq]
    for kernel in (0, 1):
        robot, ctrl, _ = controller_from_specs(da["model"].name, spec, B, device=0)
        ctrl.setFlaggedTorquePolicy(True)
        ctrl.setKernel(kernel)
        ctrl.enableGravityCompensation(opts["gravity_comp"])
        ctrl.enableTorqueSaturation(opts["torque_saturation"])
        for c, q in enumerate(seq):
            robot.setQ(q)
            robot.setDq(dq)
            robot.updateModel()
            ctrl.updateControllerTaskModels()
            ctrl.setGoals(goals)
            tau = ctrl.computeControlTorques()
            ref, st = refs[c]
            assert np.array_equal(ctrl.status, st), (seed, kernel, c, np.flatnonzero(ctrl.status != st)[:8])
            ok = (st & 1) == 0
            assert ok.sum() > B // 2 and np.isfinite(tau[ok]).all()
            err = W.torque_error(tau[ok], ref[ok])
            print(f"fuzz seed {seed} B {B} kernel {ctrl.kernelName()} cycle {c}: err {err:.2e}, blended {int((st == 8).sum())}, refused {int((~ok).sum())}")
            assert err < TOL, (seed, kernel, c)


def _random_partial_spec(rng):
    """partial motion-force task (1 - 3 translation directions, 0 - 3 rotation directions, not necessarily orthogonal), optionally with
    velocity saturation or an open-loop force / moment space, and a joint task on a random selection of 1 - 4 joints or on all of them"""
    nt, nr = int(rng.integers(1, 4)), int(rng.integers(0, 4))
    dirs_t = [list(rng.standard_normal(3)) for _ in range(nt)]
    dirs_r = [list(rng.standard_normal(3)) for _ in range(nr)] if nr else None
    mf = W.motion_force_task("motion_force_task", "end-effector", tuple(rng.uniform(-0.03, 0.08, 3)), dirs_trans=dirs_t, dirs_rot=dirs_r)
    mode = rng.integers(3)
    if mode == 1:
        mf.update(vel_sat=True, lin_sat=float(rng.uniform(0.05, 0.3)), ang_sat=float(rng.uniform(0.2, 1.0)))
    elif mode == 2 and nt == 3 and nr == 3:
        mf.update(force_dim=int(rng.integers(1, 3)), force_axis=list(rng.standard_normal(3)), moment_dim=int(rng.integers(0, 2)),
                  moment_axis=list(rng.standard_normal(3)), kv_force=float(rng.uniform(5, 15)), kv_moment=float(rng.uniform(5, 15)))
    if rng.random() < 0.5:
        mf.update(ki_pos=float(rng.uniform(0, 5)), ki_ori=float(rng.uniform(0, 5)))
    if rng.random() < 0.6:
        rows = sorted(rng.choice(7, size=int(rng.integers(1, 5)), replace=False).tolist())
        jt = W.joint_task("partial_joint_task", S=rows)
    else:
        jt = W.joint_task("joint_task")
    if rng.random() < 0.5:
        jt.update(ki=float(rng.uniform(0, 4)))
    dec = [W.FULL_DYNAMIC_DECOUPLING, W.BOUNDED_INERTIA_ESTIMATES, W.IMPEDANCE]
    mf["decoupling"] = dec[rng.integers(3)]
    jt["decoupling"] = dec[rng.integers(3)]
    opts = dict(gravity_comp=bool(rng.random() < 0.5), torque_saturation=bool(rng.random() < 0.5), joint_limit_avoidance=bool(rng.random() < 0.4))
    return [mf, jt], opts


@pytest.mark.parametrize("seed", range(10))
def test_random_partial_stacks_match_oracle(seed):
    """the FULL / general-joint-task instantiations of the eight-lane kernel (two-wavefront form at these batch sizes), the lane kernel
    and the general kernel, whichever the engine picks for the stack, against the oracle"""
    from oracle import Oracle
    from sai_primitives_amd.controller import controller_from_specs
    rng = np.random.default_rng(2000 + seed)
    B = int(rng.choice([64, 333, 1024]))
    spec, opts = _random_partial_spec(rng)
    d = W.make_inputs(11 if opts["joint_limit_avoidance"] else 2, B)   # config 11: states pushed into the joint-limit zones
    goals = [d["goals"][0], d["goals"][1]]
    if spec[1]["S"] is not None:
        m = len(spec[1]["S"])
        g = d["goals"][1]
        goals[1] = np.concatenate([g[:, :7][:, spec[1]["S"]], g[:, 7:14][:, spec[1]["S"]], g[:, 14:21][:, spec[1]["S"]]], axis=1)
        assert goals[1].shape[1] == 3 * m
    if spec[0].get("force_dim", 0) or spec[0].get("moment_dim", 0):
        goals[0] = np.concatenate([goals[0], rng.uniform(-3, 3, (B, 6))], axis=1)   # goal force / moment of the open-loop force space
    orc = Oracle(d["model"], spec, **opts)
    seq = [d["q"]]
    for _ in range(2):
        seq.append(seq[-1] + 1e-3 * d["dq"] + 5e-4 * rng.standard_normal(d["q"].shape))
    refs = [orc.step(q, d["dq"], goals, nthreads=8) for q in seq]
    robot, ctrl, _ = controller_from_specs(d["model"].name, spec, B, device=0)
    ctrl.setFlaggedTorquePolicy(True)
    ctrl.enableGravityCompensation(opts["gravity_comp"])
    ctrl.enableTorqueSaturation(opts["torque_saturation"])
    ctrl.enableJointLimitAvoidance(opts["joint_limit_avoidance"])
    for c, q in enumerate(seq):
        robot.setQ(q)
        robot.setDq(d["dq"])
        robot.updateModel()
        ctrl.updateControllerTaskModels()
        ctrl.setGoals(goals)
        tau = ctrl.computeControlTorques()
        ref, st = refs[c]
        assert np.array_equal(ctrl.status & 1, st & 1), (seed, c, np.flatnonzero((ctrl.status & 1) != (st & 1))[:8])
        ok = (st & 1) == 0
        assert ok.sum() > B // 2 and np.isfinite(tau[ok]).all()
        err = W.torque_error(tau[ok], ref[ok])
        print(f"fuzz partial seed {seed} B {B} kernel {ctrl.kernelName()} dirs {len(spec[0]['dirs_trans'])}+{len(spec[0]['dirs_rot'] or [])} "
              f"S {spec[1]['S']} cycle {c}: err {err:.2e}, refused {int((~ok).sum())}")
        assert err < TOL, (seed, c)

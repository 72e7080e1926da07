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

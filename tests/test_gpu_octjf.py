"""The eight-lane kernel for hierarchies that start with a joint task (csrc/saip_kernel_octjf.hip: { partial JointTask, full
MotionForceTask in its nullspace } on 7- and 8-dof chains, the stack of examples/06 = config 6) against the golden vectors, the CPU
oracle, and the lane-per-instance kernel that ran this stack before, on the same inputs: decoupling types, gains, integrators over
cycles, velocity saturation, the general (open-loop force) control laws, gravity compensation, torque saturation, one to two selected
joints on the 8-dof arm and one on the 7-dof arm, and postures outside the non-singular branch (flagged, recomputed behind)."""
import glob
import os

import numpy as np
import pytest

import workloads as W

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _engine(desc, tasks, B, kernel):
    from sai_primitives_amd.controller import controller_from_specs
    robot, ctrl, objs = controller_from_specs(desc, tasks, B, device=0)
    ctrl.setFlaggedTorquePolicy(True)
    ctrl.setKernel(kernel)
    return robot, ctrl, objs


def _run(robot, ctrl, q, dq, goals):
    robot.setQ(q)
    robot.setDq(dq)
    robot.updateModel()
    ctrl.updateControllerTaskModels()
    ctrl.setGoals(goals)
    return ctrl.computeControlTorques()


def test_config6_golden():
    path = glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cfg6_*.npz"))[0]
    z = np.load(path)
    model = W.load_robot(W.CONFIGS[6]["robot"])
    tasks = W.config_tasks(6)
    goals = [z[f"goal{i}"] for i in range(len(tasks))]
    B = z["q"].shape[0]
    robot, ctrl, _ = _engine(model.name, tasks, B, 3)
    ctrl.enableGravityCompensation(bool(z["gravity_comp"]))
    ctrl.enableTorqueSaturation(bool(z["torque_saturation"]))
    tau = _run(robot, ctrl, z["q"], z["dq"], goals)
    assert ctrl.kernelName() == "saip_cycle_octjf"
    assert np.array_equal(ctrl.status, z["status"])
    err = W.torque_error(tau, z["tau"])
    print("config 6 golden, saip_cycle_octjf err", err)
    assert err < TOL


def test_config6_auto_choice_and_oracle():
    """the automatic choice picks this kernel at every batch size; 4096 instances of the seeded stream against the oracle"""
    from oracle import Oracle
    B = 4096
    d = W.make_inputs(6, B)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B, 0)
    tau = _run(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.kernelName() == "saip_cycle_octjf"
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=8)
    ok = st != 1
    assert np.array_equal(ctrl.status == 1, st == 1)
    err = W.torque_error(tau[ok], ref[ok])
    print("config 6, B = 4096 err", err, "refused", int((~ok).sum()), "status set", sorted(set(ctrl.status)))
    assert err < TOL


def _sliding(**mf_kw):
    return W.motion_force_task("hand", "end-effector", (0.0, 0.0, 0.07), **mf_kw)


VARIANTS = {
    # (robot, tasks, controller options)
    "full_decoupling_gains": ("panda_sliding_base", lambda: [W.joint_task("base", S=[0, 7], kp=[80.0, 30.0], kv=[18.0, 9.0], decoupling=W.FULL_DYNAMIC_DECOUPLING),
                                                             _sliding(decoupling=W.FULL_DYNAMIC_DECOUPLING, kp_pos=150.0, kv_pos=25.0)], {}),
    "impedance_gravity_saturation": ("panda_sliding_base", lambda: [W.joint_task("base", S=[7, 0], decoupling=W.IMPEDANCE),
                                                                    _sliding(decoupling=W.IMPEDANCE)], dict(gravity_comp=True, torque_saturation=True)),
    "integrators_velocity_saturation": ("panda_sliding_base", lambda: [W.joint_task("base", S=[0], ki=3.0, vel_sat=True, sat=0.2, bie_threshold=0.6),
                                                                       _sliding(ki_pos=4.0, ki_ori=2.0, bie_threshold=0.6)], dict(gravity_comp=True)),
    "high_bie_threshold_one_row": ("panda_sliding_base", lambda: [W.joint_task("last", S=[7], bie_threshold=0.5), _sliding(bie_threshold=0.5)], {}),
    "general_laws": ("panda_sliding_base", lambda: [W.joint_task("base", S=[0, 7]),
                                                    _sliding(force_dim=1, force_axis=[0.1, -0.2, 1.0], moment_dim=2, moment_axis=[0.0, 1.0, 0.3],
                                                             param_in_compliant_frame=True, vel_sat=True, lin_sat=0.05, ang_sat=0.2)], {}),
    "panda7_one_row": ("panda_arm", lambda: [W.joint_task("elbow", S=[2], kp=70.0), W.motion_force_task("hand", "link7", (0.0, 0.0, 0.1))],
                       dict(torque_saturation=True)),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variants_match_lane_kernel_and_oracle(name):
    from oracle import Oracle
    robot_name, mk, opts = VARIANTS[name]
    model = W.load_robot(robot_name)
    tasks = mk()
    n = model.dof
    B = 200
    rng = np.random.default_rng(len(name))
    base = W.make_inputs(6 if n == 8 else 2, B, shard=2)
    q, dq = base["q"], base["dq"]
    frames = W.fk(model, q)
    goals = []
    for t in tasks:
        if t["type"] == "motion_force":
            li = model.link_index(t["link"])
            R, o = frames[li]
            x = o + np.einsum("bij,j->bi", R, np.asarray(t["pos_in_link"], float))
            Rg = W._expm_so3(rng.uniform(-0.2, 0.2, (B, 3))) @ (R @ np.asarray(t["rot_in_link"], float))
            g = np.concatenate([x + rng.uniform(-0.05, 0.05, (B, 3)), Rg.reshape(B, 9), rng.uniform(-0.1, 0.1, (B, 6)), rng.uniform(-0.5, 0.5, (B, 6))], axis=1)
            gs = W.goal_size(t, n)
            if gs > 24:
                g = np.concatenate([g, rng.uniform(-2, 2, (B, gs - 24))], axis=1)
            goals.append(g)
        else:
            S = W.joint_selection(t, n)
            m = S.shape[0]
            goals.append(np.concatenate([q @ S.T + rng.uniform(-0.3, 0.3, (B, m)), rng.uniform(-0.1, 0.1, (B, m)), rng.uniform(-0.5, 0.5, (B, m))], axis=1))
    out = {}
    for kernel in (3, 2):
        robot, ctrl, _ = _engine(model.name, tasks, B, kernel)
        ctrl.enableGravityCompensation(opts.get("gravity_comp", False))
        ctrl.enableTorqueSaturation(opts.get("torque_saturation", False))
        taus = [_run(robot, ctrl, q + 0.01 * cyc, dq, goals).copy() for cyc in range(3)]  # integrators advance over the cycles
        out[kernel] = (np.array(taus), ctrl.status.copy(), ctrl.kernelName())
    assert out[3][2] == "saip_cycle_octjf" and out[2][2] == "saip_cycle_lane"
    orc = Oracle(model, tasks, **opts)
    for cyc in range(3):
        ref, st = orc.step(q + 0.01 * cyc, dq, goals, nthreads=4)
        ok = st != 1
        assert ok.sum() > B // 2
        plain = st == 0  # (instances the blended strategies handle come from the general kernel behind either kernel: looser bound)
        e3, e2 = W.torque_error(out[3][0][cyc][plain], ref[plain]), W.torque_error(out[2][0][cyc][plain], ref[plain])
        e3b = W.torque_error(out[3][0][cyc][ok], ref[ok])
        print(f"{name} cycle {cyc}: octjf err {e3:.2e} (with blended instances {e3b:.2e}), lane err {e2:.2e}, refused {int((~ok).sum())}, blended {int(((st & 8) > 0).sum())}")
        assert e3 < TOL and e2 < TOL and e3b < 1e-6
    assert np.array_equal(out[3][1], out[2][1]) and np.array_equal(out[3][1] == 1, st == 1)


def test_postures_outside_the_nonsingular_branch_are_recomputed_behind():
    """every third instance with the arm nearly stretched out (inside SingularityHandler's blending region): this kernel flags them, the
    general kernel launched behind over the flag list runs the reference's blended strategies: results and status bits = oracle"""
    from oracle import Oracle
    B = 192
    d = W.make_inputs(6, B)
    q = d["q"].copy()
    rng = np.random.default_rng(11)
    q[0::3, 4] = rng.uniform(-0.03, 0.03, q[0::3, 4].shape)   # joint 4 of the sliding-base arm = the Panda's elbow (joint 4 of 7 -> index 3 + 1)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B, 3)
    orc = Oracle(d["model"], d["tasks"])
    for cyc in range(2):
        tau = _run(robot, ctrl, q, d["dq"], d["goals"]).copy()
        ref, st = orc.step(q, d["dq"], d["goals"], nthreads=8)
        ok = st != 1
        err = W.torque_error(tau[ok], ref[ok])
        print("cycle", cyc, "octjf + slow path err", err, "blended", int(((st & 8) > 0).sum()), "refused", int((~ok).sum()))
        assert err < 1e-7 and np.array_equal(ctrl.status & 9, st & 9)
    assert ((st & 8) > 0).sum() >= B // 6


def test_non_unit_selection_rows_stay_on_the_lane_kernel():
    model = W.load_robot("panda_sliding_base")
    S = np.zeros((2, 8))
    S[0, 0] = 1.0
    S[1, 6] = S[1, 7] = np.sqrt(0.5)
    tasks = [W.joint_task("mixed", S=S.tolist()), _sliding()]
    robot, ctrl, _ = _engine(model.name, tasks, 64, 0)
    d = W.make_inputs(6, 64)
    goals = [np.concatenate([d["q"] @ S.T, np.zeros((64, 4))], axis=1), d["goals"][1]]
    _run(robot, ctrl, d["q"], d["dq"], goals)
    assert ctrl.kernelName() == "saip_cycle_lane"


def _random_joint_first_spec(rng):
    """every scalar option of the { partial JointTask, full MotionForceTask } stack drawn at random (the style of tests/test_gpu_fuzz.py)"""
    rows = [[0, 7], [7, 0], [0], [7], [7, 1], [0, 3]][rng.integers(6)]
    m = len(rows)
    aniso = rng.random() < 0.5
    jt = W.joint_task("partial_joint_task", S=rows, kp=list(rng.uniform(20, 120, m)) if aniso else float(rng.uniform(20, 120)),
                      kv=list(rng.uniform(8, 25, m)) if aniso else float(rng.uniform(8, 25)))
    mf = W.motion_force_task("motion_force_task", "end-effector", tuple(rng.uniform(-0.05, 0.1, 3)))
    mf["kp_pos"] = list(rng.uniform(50, 200, 3)) if aniso else float(rng.uniform(50, 200))
    mf["kv_pos"] = list(rng.uniform(10, 30, 3)) if aniso else float(rng.uniform(10, 30))
    mf["kp_ori"] = list(rng.uniform(100, 300, 3)) if aniso else float(rng.uniform(100, 300))
    mf["kv_ori"] = list(rng.uniform(15, 35, 3)) if aniso else float(rng.uniform(15, 35))
    if rng.random() < 0.5:
        mf["ki_pos"], mf["ki_ori"] = float(rng.uniform(0, 8)), float(rng.uniform(0, 8))
        jt["ki"] = float(rng.uniform(0, 5))
    dec = [W.FULL_DYNAMIC_DECOUPLING, W.BOUNDED_INERTIA_ESTIMATES, W.BOUNDED_INERTIA_ESTIMATES, W.IMPEDANCE]
    mf["decoupling"], jt["decoupling"] = dec[rng.integers(4)], dec[rng.integers(4)]
    mf["bie_threshold"] = jt["bie_threshold"] = float(rng.choice([0.05, 0.1, 0.3, 0.6]))
    if rng.random() < 0.4:
        jt.update(vel_sat=True, sat=float(rng.uniform(0.2, 1.0)))
    if rng.random() < 0.3:
        mf.update(vel_sat=True, lin_sat=float(rng.uniform(0.02, 0.3)), ang_sat=float(rng.uniform(0.1, 1.0)))
    mf["dt"] = jt["dt"] = float(rng.choice([5e-4, 1e-3, 2e-3]))
    smax = float(rng.uniform(3e-2, 9e-2))
    mf["s_max"], mf["s_min"] = smax, smax * float(rng.uniform(0.05, 0.5))
    return [jt, mf], dict(gravity_comp=bool(rng.random() < 0.5), torque_saturation=bool(rng.random() < 0.5))


@pytest.mark.parametrize("seed", range(6))
def test_random_option_combinations_match_oracle(seed):
    """random option combinations on the 8-dof arm, a quarter of the batch with the elbow nearly straight (flagged, recomputed behind by the
    blended strategies), three cycles of a drifting state, batches on both sides of the two- / one-wavefront boundary"""
    from oracle import Oracle
    rng = np.random.default_rng(4200 + seed)
    B = int(rng.choice([96, 1000, 4100]))
    spec, opts = _random_joint_first_spec(rng)
    d = W.make_inputs(6, B, shard=1)
    model = d["model"]
    q0 = d["q"].copy()
    q0[0::4, 4] = rng.uniform(-0.04, 0.04, q0[0::4, 4].shape)
    S = W.joint_selection(spec[0], 8)
    goals = [np.concatenate([q0 @ S.T + rng.uniform(-0.3, 0.3, (B, S.shape[0])), rng.uniform(-0.1, 0.1, (B, S.shape[0])), rng.uniform(-0.5, 0.5, (B, S.shape[0]))], axis=1),
             d["goals"][1]]
    robot, ctrl, _ = _engine(model.name, spec, B, 0)
    ctrl.enableGravityCompensation(opts["gravity_comp"])
    ctrl.enableTorqueSaturation(opts["torque_saturation"])
    orc = Oracle(model, spec, **opts)
    for cyc in range(3):
        q = q0 + 0.004 * cyc
        tau = _run(robot, ctrl, q, d["dq"], goals).copy()
        assert ctrl.kernelName() == "saip_cycle_octjf"
        ref, st = orc.step(q, d["dq"], goals, nthreads=8)
        ok, plain = st != 1, st == 0
        assert np.array_equal(ctrl.status & 9, st & 9)
        e0, e1 = W.torque_error(tau[plain], ref[plain]), W.torque_error(tau[ok], ref[ok])
        print(f"seed {seed} B {B} rows {spec[0]['S']} cycle {cyc}: err {e0:.2e} (with blended instances {e1:.2e}), blended {int(((st & 8) > 0).sum())}, refused {int((~ok).sum())}")
        assert plain.sum() > B // 2 and e0 < TOL and e1 < 1e-6


def test_exact_fallback_when_forced():
    """the one-lane Gram eigen-solve behind the certificates is rare on real data; the test-only build libsaip_octexact.so (-DSAIP_OCT_FORCE_EXACT)
    sends EVERY instance of this kernel through it and must give the oracle's answers and statuses too (child process: the library is chosen
    at load time through SAIP_LIB)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "sai-primitives_amd", "libsaip_octexact.so")
    assert os.path.exists(lib), "build() links the test-only variant next to libsaip.so"
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import workloads as W
from oracle import Oracle
from sai_primitives_amd.controller import controller_from_specs
B = 256
d = W.make_inputs(6, B)
q = d["q"].copy()
q[0::3, 4] = np.random.default_rng(2).uniform(-0.03, 0.03, q[0::3, 4].shape)   # a third of the batch inside the singular region
ref, st = Oracle(d["model"], d["tasks"]).step(q, d["dq"], d["goals"], nthreads=8)
robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
ctrl.setFlaggedTorquePolicy(True)
ctrl.setKernel(3)
robot.setQ(q); robot.setDq(d["dq"]); robot.updateModel(); ctrl.updateControllerTaskModels(); ctrl.setGoals(d["goals"])
tau = ctrl.computeControlTorques()
assert ctrl.kernelName() == "saip_cycle_octjf" and np.array_equal(ctrl.status & 9, st & 9)
ok = st == 0
err = W.torque_error(tau[ok], ref[ok])
print("forced exact path, config 6:", int(ok.sum()), "plain instances, err", err, "blended", int(((st & 8) > 0).sum()))
assert err < 1e-9 and ((st & 8) > 0).sum() > B // 6
''' % (root, os.path.join(root, "oracle"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SAIP_LIB=lib), capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr[-2000:])
    assert out.returncode == 0

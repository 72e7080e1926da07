"""Eight-lanes-per-instance kernel (saip_cycle_oct, the small-batch path of the headline stack) against the CPU oracle."""
import numpy as np
import pytest

import workloads as W

pytestmark = pytest.mark.gpu
TOL = 1e-5
OCT = 3


def _engine(model_name, tasks, B):
    from sai_primitives_amd.controller import controller_from_specs
    robot, ctrl, objs = controller_from_specs(model_name, tasks, B, device=0)
    ctrl.setFlaggedTorquePolicy(True)   # tests: an instance the engine refuses (status 1) shows as NaN, like in the oracle (the default holds the last torque)
    return robot, ctrl, objs


def _flagging(spec):
    """engine extra: the blended singularity strategies switched off -> instances outside the non-singular branch are flagged (status 1)"""
    return [dict(t, singularity_strategies=False) if t["type"] == "motion_force" else t for t in spec]


def _cycle(robot, ctrl, q, dq, goals):
    robot.setQ(q)
    robot.setDq(dq)
    robot.updateModel()
    ctrl.updateControllerTaskModels()
    ctrl.setGoals(goals)
    return ctrl.computeControlTorques()


@pytest.mark.parametrize("B", [4096, 1001, 5])
def test_oct_matches_oracle_config2(B):
    from oracle import Oracle
    d = W.make_inputs(2, B)
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=8)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B)
    ctrl.setKernel(OCT)
    tau = _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.kernelName() == "saip_cycle_oct"
    assert np.array_equal(ctrl.status, st)
    err = W.torque_error(tau, ref)
    print("oct cfg2 B", B, "err", err)
    assert err < TOL


def test_oct_is_the_default_for_the_headline_batch():
    d = W.make_inputs(2, 4096)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], 4096)
    _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.kernelName() == "saip_cycle_oct"
    d = W.make_inputs(3, 64)   # partial motion-force task + 4-row joint task: the general eight-lane path
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], 64)
    _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.kernelName() == "saip_cycle_oct"
    d = W.make_inputs(6, 64)   # 8-dof chain, joint task first: the joint-first eight-lane kernel (tests/test_gpu_octjf.py)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], 64)
    _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.kernelName() == "saip_cycle_octjf"
    d = W.make_inputs(13, 64)   # closed-loop force control: not covered by the eight-lane kernels
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], 64)
    _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.kernelName() == "saip_cycle_lane"
    ctrl.setKernel(OCT)
    with pytest.raises(Exception):
        _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])


@pytest.mark.parametrize("variant", ["gravity_saturation", "full_decoupling", "impedance", "integrators_velsat", "two_thresholds_off"])
def test_oct_options(variant):
    from oracle import Oracle
    B = 512
    d = W.make_inputs(2, B)
    spec = W.config_tasks(2)
    opts = {}
    cycles = 1
    if variant == "gravity_saturation":
        opts = dict(gravity_comp=True, torque_saturation=True)
    elif variant == "full_decoupling":
        spec[0]["decoupling"] = W.FULL_DYNAMIC_DECOUPLING
        spec[1]["decoupling"] = W.FULL_DYNAMIC_DECOUPLING
    elif variant == "impedance":
        spec[0]["decoupling"] = W.IMPEDANCE
        spec[1]["decoupling"] = W.IMPEDANCE
    elif variant == "integrators_velsat":
        spec[0].update(ki_pos=5.0, ki_ori=7.0)
        spec[1].update(ki=3.0, vel_sat=True, sat=0.4)
        cycles = 4
    elif variant == "two_thresholds_off":
        spec[0]["bie_threshold"] = 0.5   # several diagonal entries of M clamped: one Sherman-Morrison update each
        spec[1]["bie_threshold"] = 0.5
    orc = Oracle(d["model"], spec, **opts)
    robot, ctrl, _ = _engine(d["model"].name, spec, B)
    ctrl.setKernel(OCT)
    ctrl.enableGravityCompensation(opts.get("gravity_comp", False))
    ctrl.enableTorqueSaturation(opts.get("torque_saturation", False))
    rng = np.random.default_rng(8)
    q = d["q"].copy()
    for c in range(cycles):
        tau = _cycle(robot, ctrl, q, d["dq"], d["goals"])
        assert ctrl.kernelName() == "saip_cycle_oct"
        ref, st = orc.step(q, d["dq"], d["goals"], nthreads=8)
        assert np.array_equal(ctrl.status, st)
        err = W.torque_error(tau, ref)
        print("oct", variant, "cycle", c, "err", err)
        assert err < TOL
        q = q + 1e-3 * rng.standard_normal(q.shape)


@pytest.mark.parametrize("opts", [dict(joint_limit_avoidance=True), dict(joint_limit_avoidance=True, gravity_comp=True, torque_saturation=True)],
                         ids=["jla", "jla_gravity_saturation"])
def test_oct_joint_limit_avoidance(opts):
    """RobotController::enableJointLimitAvoidance (RobotController.cpp:96-112) on the oct kernel: config 11's states (joints inside the
    position and velocity zones of their limits), two cycles"""
    from oracle import Oracle
    B = 1024
    d = W.make_inputs(11, B)
    orc = Oracle(d["model"], d["tasks"], **opts)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B)
    ctrl.setKernel(OCT)
    ctrl.enableJointLimitAvoidance(True)
    ctrl.enableGravityCompensation(opts.get("gravity_comp", False))
    ctrl.enableTorqueSaturation(opts.get("torque_saturation", False))
    q = d["q"].copy()
    for c in range(2):
        tau = _cycle(robot, ctrl, q, d["dq"], d["goals"])
        assert ctrl.kernelName() == "saip_cycle_oct"
        ref, st = orc.step(q, d["dq"], d["goals"], nthreads=8)
        assert np.array_equal(ctrl.status, st)
        ok = st == 0
        err = W.torque_error(tau[ok], ref[ok])
        print("oct jla", opts, "cycle", c, "err", err)
        assert err < TOL
        q = q + 1e-3 * d["dq"]


@pytest.mark.parametrize("cfg", [9, 10])
def test_oct_general_control_laws(cfg):
    """velocity saturation + hybrid force / motion control (9) and a full force space (10), MotionForceTask.cpp:306-487, on the oct kernel;
    integrators on, three cycles"""
    from oracle import Oracle
    B = 512
    d = W.make_inputs(cfg, B)
    spec = W.config_tasks(cfg)
    for t in spec:
        if t["type"] == "motion_force":
            t.update(ki_pos=4.0, ki_ori=6.0)
    orc = Oracle(d["model"], spec)
    robot, ctrl, _ = _engine(d["model"].name, spec, B)
    ctrl.setKernel(OCT)
    rng = np.random.default_rng(cfg)
    q = d["q"].copy()
    for c in range(3):
        tau = _cycle(robot, ctrl, q, d["dq"], d["goals"])
        assert ctrl.kernelName() == "saip_cycle_oct"
        ref, st = orc.step(q, d["dq"], d["goals"], nthreads=8)
        assert np.array_equal(ctrl.status, st)
        err = W.torque_error(tau, ref)
        print("oct general laws cfg", cfg, "cycle", c, "err", err)
        assert err < TOL
        q = q + 1e-3 * rng.standard_normal(q.shape)


@pytest.mark.parametrize("cfg", [12, 14])
def test_oct_flags_and_slow_path(cfg):
    """near-singular instances: flagged by the oct kernel, recomputed by the general kernel behind it (truncated task / blended strategies)"""
    from oracle import Oracle
    B = 768
    d = W.make_inputs(cfg, B)
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=8)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B)
    ctrl.setKernel(OCT)
    tau = _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.kernelName() == "saip_cycle_oct"
    assert np.array_equal(ctrl.status, st) and (st != 0).sum() > B // 4
    assert W.torque_error(tau, ref) < TOL
    # handling enforced with the blended strategies switched off: the same instances are flagged, the others are right
    robot, ctrl, _ = _engine(d["model"].name, _flagging(W.config_tasks(2)), B)
    ctrl.setKernel(OCT)
    tau = _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert np.array_equal(ctrl.status == 1, st != 0) and np.isnan(tau[st != 0]).all()
    assert W.torque_error(tau[st == 0], ref[st == 0]) < TOL


def test_oct_uniform_postures_predicate_matches_oracle():
    """uniformly random postures (more than half outside the non-singular branch): the certificate / exact fallback of the oct kernel
    takes the same decisions as the oracle's SVD"""
    from oracle import Oracle
    B = 2048
    d = W.make_inputs(2, B)
    rng = np.random.default_rng(31)
    q = rng.uniform(d["model"].q_lower, d["model"].q_upper, size=d["q"].shape)
    ref, st = Oracle(d["model"], d["tasks"]).step(q, d["dq"], d["goals"], nthreads=8)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B)
    ctrl.setKernel(OCT)
    tau = _cycle(robot, ctrl, q, d["dq"], d["goals"])
    assert np.array_equal(ctrl.status, st)
    ok = st == 0
    print("uniform postures: valid", int(ok.sum()), "of", B, "err", W.torque_error(tau[ok], ref[ok]))
    assert W.torque_error(tau[ok], ref[ok]) < TOL


@pytest.mark.parametrize("cfg,B,kernel", [(2, 4096, 2), (2, 28672, 0), (12, 768, 2), (14, 768, 2)])
def test_lane_kernel_on_the_same_stack(cfg, B, kernel):
    """the lane-per-instance kernel stays the path of the lean stack for batches above 24576 (auto) and of every stack on request (selector 2)"""
    from oracle import Oracle
    d = W.make_inputs(cfg, B)
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=8)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B)
    ctrl.setKernel(kernel)
    tau = _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])
    assert ctrl.kernelName() == "saip_cycle_lane"
    assert np.array_equal(ctrl.status, st)
    assert W.torque_error(tau, ref) < TOL


def test_oct_instances_are_isolated():
    """eight instances share a wavefront and two share a DPP row: a non-finite state in one instance must not reach its neighbours
    (their torques stay bit-identical to a clean run), and the bad instance itself is reported, not silently wrong"""
    B = 64
    d = W.make_inputs(2, B)
    robot, ctrl, _ = _engine(d["model"].name, d["tasks"], B)
    ctrl.setKernel(OCT)
    clean = _cycle(robot, ctrl, d["q"], d["dq"], d["goals"]).copy()
    for bad, how in ((9, np.nan), (14, np.inf), (16, np.nan)):
        q = d["q"].copy()
        q[bad, 3] = how
        tau = _cycle(robot, ctrl, q, d["dq"], d["goals"])
        others = np.arange(B) != bad
        assert np.array_equal(tau[others], clean[others])
        assert not np.isfinite(tau[bad]).all() and ctrl.status[bad] == 1
        assert (ctrl.status[others] == 0).all()
    dq = d["dq"].copy()
    dq[23, :] = np.nan
    tau = _cycle(robot, ctrl, d["q"], dq, d["goals"])
    others = np.arange(B) != 23
    assert np.array_equal(tau[others], clean[others]) and not np.isfinite(tau[23]).any()


def test_oct_exact_fallbacks_when_forced():
    """the Gram-eigenvalue fallbacks of the oct kernel (singular-value predicate, range basis) are rare on real data; a test-only build
    of the same source sends EVERY instance through them (libsaip_octexact.so, -DSAIP_OCT_FORCE_EXACT) and must give the oracle's
    answers too.  Runs in a child process because the library is chosen at load time (SAIP_LIB)."""
    import os
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
rng = np.random.default_rng(31)
for uniform in (False, True):
    B = 768
    d = W.make_inputs(2, B)
    q = rng.uniform(d["model"].q_lower, d["model"].q_upper, size=d["q"].shape) if uniform else d["q"]
    ref, st = Oracle(d["model"], d["tasks"]).step(q, d["dq"], d["goals"], nthreads=8)
    robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
    ctrl.setKernel(3)
    robot.setQ(q); robot.setDq(d["dq"]); robot.updateModel(); ctrl.updateControllerTaskModels(); ctrl.setGoals(d["goals"])
    tau = ctrl.computeControlTorques()
    assert ctrl.kernelName() == "saip_cycle_oct" and np.array_equal(ctrl.status, st)
    ok = st == 0
    err = W.torque_error(tau[ok], ref[ok])
    print("forced exact paths, uniform postures" if uniform else "forced exact paths, config 2", int(ok.sum()), "valid, err", err)
    assert err < 1e-5
''' % (root, os.path.join(root, "oracle"))
    env = dict(os.environ, SAIP_LIB=lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr[-2000:])
    assert out.returncode == 0


@pytest.mark.parametrize("cfg,B", [(3, 4096), (3, 1001), (7, 256), (8, 256)])
def test_oct_partial_stacks(cfg, B):
    """partial motion-force tasks (zero-padded rows, projection-aware law) and the general joint-task range basis (<= 4 rows, Gram
    eigen-pairs by the register Jacobi solve): BASELINE config 3 (3-dof position task + joint task on joints {0,2,4,6}) and, where the
    stack is covered, the orientation-only (7) and planar (8) scenarios; integrators on, three cycles"""
    from oracle import Oracle
    d = W.make_inputs(cfg, B)
    spec = W.config_tasks(cfg)
    for t in spec:
        if t["type"] == "motion_force":
            t.update(ki_pos=4.0, ki_ori=6.0)
        else:
            t.update(ki=3.0)
    robot, ctrl, _ = _engine(d["model"].name, spec, B)
    ctrl.setKernel(OCT)
    try:
        _cycle(robot, ctrl, d["q"], d["dq"], d["goals"])
    except Exception as e:  # stack not covered by the eight-lane kernel (e.g. a full joint task behind a partial motion-force task)
        assert cfg != 3, e
        pytest.skip("stack not covered: " + str(e)[:80])
    robot, ctrl, _ = _engine(d["model"].name, spec, B)
    ctrl.setKernel(OCT)
    orc = Oracle(d["model"], spec)
    rng = np.random.default_rng(cfg)
    q = d["q"].copy()
    for c in range(3):
        tau = _cycle(robot, ctrl, q, d["dq"], d["goals"])
        assert ctrl.kernelName() == "saip_cycle_oct"
        ref, st = orc.step(q, d["dq"], d["goals"], nthreads=8)
        assert np.array_equal(ctrl.status, st)
        ok = st == 0
        err = W.torque_error(tau[ok], ref[ok])
        print("oct partial cfg", cfg, "B", B, "cycle", c, "err", err, "flagged", int((~ok).sum()))
        assert err < TOL
        q = q + 1e-3 * rng.standard_normal(q.shape)


@pytest.mark.parametrize("cfg", [3, 7, 12])
def test_oct_general_paths_with_every_controller_option(cfg):
    """the general joint-task paths (Gram path: 3; multi-pivot path: 7; reduced task: 12) together with joint limit avoidance, gravity
    compensation, torque saturation and integrators, states pushed towards the joint limits, four cycles"""
    from oracle import Oracle
    B = 768
    d = W.make_inputs(cfg, B)
    d11 = W.make_inputs(11, B)          # states inside the limit zones
    spec = W.config_tasks(cfg)
    for t in spec:
        if t["type"] == "motion_force":
            t.update(ki_pos=4.0, ki_ori=6.0)
        else:
            t.update(ki=3.0)
    opts = dict(joint_limit_avoidance=True, gravity_comp=True, torque_saturation=True)
    orc = Oracle(d["model"], spec, **opts)
    robot, ctrl, _ = _engine(d["model"].name, spec, B)
    ctrl.setKernel(OCT)
    ctrl.enableJointLimitAvoidance(True)
    ctrl.enableGravityCompensation(True)
    ctrl.enableTorqueSaturation(True)
    q = np.where((np.arange(B) % 2 == 0)[:, None], d["q"], d11["q"])
    dq = np.where((np.arange(B) % 2 == 0)[:, None], d["dq"], d11["dq"])
    rng = np.random.default_rng(cfg)
    worst = 0.0
    for c in range(4):
        tau = _cycle(robot, ctrl, q, dq, d["goals"])
        assert ctrl.kernelName() == "saip_cycle_oct"
        ref, st = orc.step(q, dq, d["goals"], nthreads=8)
        assert np.array_equal(ctrl.status, st)
        ok = (st & 1) == 0
        worst = max(worst, W.torque_error(tau[ok], ref[ok]))
        q = q + 1e-3 * rng.standard_normal(q.shape)
    print("oct general paths + options, cfg", cfg, "worst err", worst, "last statuses", np.unique(st, return_counts=True))
    assert worst < TOL


@pytest.mark.parametrize("variant", ["default", "gravity_saturation", "full_decoupling", "impedance", "integrators_velsat", "several_clamped_entries"])
@pytest.mark.parametrize("B", [768, 4099])
def test_oct_options_on_blended_postures(variant, B):
    """config 14's postures (a third type-1, a third type-2 singular: SingularityHandler's blended strategies) with every option of the
    headline stack.  B = 768 runs the two-wavefront form, in which wavefront B evaluates the blended block from what wavefront A left
    in LDS (beta, the clamped entry, columns of M^-1 and of M_BIE^-1) and hands torque and flags back; B = 4099 the one-wavefront form."""
    from oracle import Oracle
    d = W.make_inputs(14, B)
    spec = W.config_tasks(14)
    opts = {}
    cycles = 2
    if variant == "gravity_saturation":
        opts = dict(gravity_comp=True, torque_saturation=True)
    elif variant == "full_decoupling":
        spec[0]["decoupling"] = W.FULL_DYNAMIC_DECOUPLING
        spec[1]["decoupling"] = W.FULL_DYNAMIC_DECOUPLING
    elif variant == "impedance":
        spec[0]["decoupling"] = W.IMPEDANCE
        spec[1]["decoupling"] = W.IMPEDANCE
    elif variant == "integrators_velsat":
        spec[0].update(ki_pos=5.0, ki_ori=7.0)
        spec[1].update(ki=3.0, vel_sat=True, sat=0.4)
        cycles = 4
    elif variant == "several_clamped_entries":
        spec[0]["bie_threshold"] = 0.5
        spec[1]["bie_threshold"] = 0.5
    orc = Oracle(d["model"], spec, **opts)
    robot, ctrl, _ = _engine(d["model"].name, spec, B)
    ctrl.setKernel(OCT)
    ctrl.enableGravityCompensation(opts.get("gravity_comp", False))
    ctrl.enableTorqueSaturation(opts.get("torque_saturation", False))
    rng = np.random.default_rng(18)
    q = d["q"].copy()
    for c in range(cycles):
        tau = _cycle(robot, ctrl, q, d["dq"], d["goals"])
        assert ctrl.kernelName() == "saip_cycle_oct"
        ref, st = orc.step(q, d["dq"], d["goals"], nthreads=8)
        assert np.array_equal(ctrl.status, st), (variant, c, np.flatnonzero(ctrl.status != st)[:8])
        assert (st == 8).sum() > B // 3 and (st == 1).sum() == 0
        err = W.torque_error(tau, ref)
        print("oct blended", variant, "B", B, "cycle", c, "err", err)
        assert err < TOL
        q = q + 1e-3 * rng.standard_normal(q.shape)


def _six_dof_chain(rng):
    """a 6-dof arm with general axes (5 revolute joints + one prismatic), the generator of tests/test_gpu_wave.py at n = 6"""
    links = []
    for i in range(6):
        prismatic = i == 2
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        if i % 2 == 0:
            ax = np.eye(3)[i // 2]
        ine = rng.uniform(0.004, 0.03, 3)
        off = rng.uniform(-0.002, 0.002, 3)
        links.append(dict(name=f"link{i + 1}", joint_type="prismatic" if prismatic else "revolute",
                          origin_xyz=list(rng.uniform(-0.05, 0.25, 3)) if i else [0.0, 0.0, 0.2], origin_rpy=list(rng.uniform(-1.0, 1.0, 3)), axis=list(ax),
                          mass=float(rng.uniform(0.5, 3.0)), com=list(rng.uniform(-0.05, 0.05, 3)), inertia=[ine[0], ine[1], ine[2], off[0], off[1], off[2]],
                          q_lower=-0.3 if prismatic else -2.5, q_upper=0.3 if prismatic else 2.5, velocity_limit=3.0, effort_limit=float(rng.uniform(20, 60)),
                          q_nominal_deg=0.0))
    return dict(name="arm6", links=links)


@pytest.mark.parametrize("case", ["arm8_full_task", "arm8_position_task", "arm8_planar_task_impedance", "arm6_full_task", "arm6_position_task_gains"])
def test_six_and_eight_dof_chains(case):
    """{ MotionForceTask, full JointTask } on chains of 6 and 8 dof: the general instantiation of this kernel (Jp = N_1 of rank n - k, 0 .. 5)
    with N = 6 / 8 against the lane kernel (their path before) and the oracle; three cycles, integrators on in one case"""
    from oracle import Oracle
    rng = np.random.default_rng(len(case))
    B = 200
    if case.startswith("arm8"):
        model = W.load_robot("panda_sliding_base")
        desc = model.name
        base = W.make_inputs(6, B, shard=4)
        q, dq = base["q"], base["dq"]
        link, pos = "end-effector", (0.0, 0.0, 0.07)
    else:
        desc = _six_dof_chain(rng)
        model = W.RobotModel(desc)
        q = rng.uniform(0.7 * model.q_lower, 0.7 * model.q_upper, (B, 6))
        dq = rng.uniform(-0.5, 0.5, (B, 6))
        link, pos = "link6", (0.02, 0.0, 0.1)
    n = model.dof
    if case == "arm8_full_task":
        tasks = [W.motion_force_task("hand", link, pos), W.joint_task("posture")]
    elif case == "arm8_position_task":
        tasks = [W.motion_force_task("hand", link, pos, dirs_trans=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], dirs_rot=None, decoupling=W.FULL_DYNAMIC_DECOUPLING),
                 W.joint_task("posture", decoupling=W.FULL_DYNAMIC_DECOUPLING)]
    elif case == "arm8_planar_task_impedance":
        tasks = [W.motion_force_task("hand", link, pos, dirs_trans=[[1, 0, 0], [0, 1, 1]], dirs_rot=[[0, 0, 1], [1, 0, 0]], decoupling=W.IMPEDANCE),
                 W.joint_task("posture", decoupling=W.IMPEDANCE, vel_sat=True, sat=0.5)]
    elif case == "arm6_full_task":
        tasks = [W.motion_force_task("hand", link, pos), W.joint_task("posture")]
    else:
        tasks = [W.motion_force_task("hand", link, pos, dirs_trans=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], dirs_rot=None, kp_pos=[120.0, 90.0, 150.0], ki_pos=3.0, bie_threshold=0.4),
                 W.joint_task("posture", kp=list(rng.uniform(20, 90, 6)), ki=1.5, bie_threshold=0.4)]
    opts = dict(gravity_comp=case != "arm8_full_task", torque_saturation=case.endswith("gains"))
    frames = W.fk(model, q)
    goals = []
    for t in tasks:
        if t["type"] == "motion_force":
            li = model.link_index(t["link"])
            R, o = frames[li]
            x = o + np.einsum("bij,j->bi", R, np.asarray(t["pos_in_link"], float))
            Rg = W._expm_so3(rng.uniform(-0.2, 0.2, (B, 3))) @ (R @ np.asarray(t["rot_in_link"], float))
            goals.append(np.concatenate([x + rng.uniform(-0.05, 0.05, (B, 3)), Rg.reshape(B, 9), rng.uniform(-0.1, 0.1, (B, 6)), rng.uniform(-0.5, 0.5, (B, 6))], axis=1))
        else:
            goals.append(np.concatenate([q + rng.uniform(-0.3, 0.3, (B, n)), rng.uniform(-0.1, 0.1, (B, n)), rng.uniform(-0.5, 0.5, (B, n))], axis=1))
    out = {}
    for kernel in (OCT, 2):
        robot, ctrl, _ = _engine(desc, tasks, B)
        ctrl.setKernel(kernel)
        ctrl.enableGravityCompensation(opts["gravity_comp"])
        ctrl.enableTorqueSaturation(opts["torque_saturation"])
        taus = [_cycle(robot, ctrl, q + 0.01 * cyc, dq, goals).copy() for cyc in range(3)]
        out[kernel] = (np.array(taus), ctrl.status.copy(), ctrl.kernelName())
    assert out[OCT][2] == "saip_cycle_oct" and out[2][2] == "saip_cycle_lane"
    orc = Oracle(model, tasks, **opts)
    for cyc in range(3):
        ref, st = orc.step(q + 0.01 * cyc, dq, goals, nthreads=4)
        plain = (st == 0) & (out[2][1] == 0)
        assert plain.sum() > (B // 2 if n == 8 else 20)   # (a random 6-dof arm with a 6-dof task: most postures sit inside the handler's singular region)
        e8, e1 = W.torque_error(out[OCT][0][cyc][plain], ref[plain]), W.torque_error(out[2][0][cyc][plain], ref[plain])
        print(f"{case} cycle {cyc}: eight-lane err {e8:.2e}, lane err {e1:.2e}, plain {int(plain.sum())} of {B}, status set {sorted(set(out[OCT][1]))}")
        assert e8 < TOL and e1 < TOL
    assert np.array_equal(out[OCT][1], out[2][1])

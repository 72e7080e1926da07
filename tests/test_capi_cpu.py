"""CPU-side tests of the C-ABI library (no compute calls: there is no GPU here):
  * libsaip.so loads and exports every function include/saip.h declares,
  * host logic of the boundary: robot model build (fixed-link merge), task constructors and
    RobotController constructor validation with the reference's error conditions,
  * every compute entry point fails loudly without a device (the engine has no CPU path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sp():
    import sai_primitives_amd as sp
    sp.build_library()
    return sp


def _cfg_only(sp, robot="panda_arm", B=4):
    return sp.SaiModel(robot, B, device=-1)  # configuration-only batch: host logic without a device


def test_library_exports_every_declared_symbol(sp):
    hdr = open(os.path.join(ROOT, "include", "saip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(saip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 45
    L = C.CDLL(os.path.join(ROOT, "sai-primitives_amd", "libsaip.so"))
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    # the ctypes binding covers the same set
    assert set(sp.lib()._declared) == declared


def test_model_merges_fixed_links_like_rbdl(sp):
    robot = _cfg_only(sp)
    assert robot.dof() == 7
    assert robot.linkIndex("end-effector") == 7 and robot.linkIndex("link4") == 3 and robot.linkIndex("nope") == -1
    lim = robot.jointLimits()
    m = W.load_robot("panda_arm")
    assert np.allclose(lim["position_lower"], m.q_lower) and np.allclose(lim["effort"], m.effort)
    assert _cfg_only(sp, "panda_sliding_base").dof() == 8
    assert _cfg_only(sp, "chain30").dof() == 30


def test_robot_controller_constructor_validation(sp):
    robot = _cfg_only(sp)
    other = _cfg_only(sp)
    with pytest.raises(ValueError, match="at least one task"):
        sp.RobotController(robot, [])
    jt = sp.JointTask(robot, task_name="a")
    with pytest.raises(ValueError, match="same robot model"):
        sp.RobotController(other, [jt])
    with pytest.raises(ValueError, match="unique names"):
        sp.RobotController(robot, [sp.MotionForceTask(robot, "end-effector", task_name="x"), sp.JointTask(robot, task_name="x")])
    with pytest.raises(ValueError, match="same loop timestep"):
        sp.RobotController(robot, [sp.MotionForceTask(robot, "end-effector", task_name="m", loop_timestep=0.002), sp.JointTask(robot, task_name="j")])
    with pytest.raises(ValueError, match="nullspace of a full joint task"):
        sp.RobotController(robot, [sp.JointTask(robot, task_name="full"), sp.MotionForceTask(robot, "end-effector", task_name="late")])
    ctrl = sp.RobotController(robot, [sp.MotionForceTask(robot, "end-effector", (0, 0, 0.07), task_name="mf"), sp.JointTask(robot, task_name="jt")])
    assert ctrl.getTaskNames() == ["mf", "jt"]
    assert ctrl.getJointTaskByName("jt").getTaskType() == sp.TaskType.JOINT_TASK
    with pytest.raises(ValueError, match="is not a JointTask"):
        ctrl.getJointTaskByName("mf")
    with pytest.raises(ValueError, match="not found"):
        ctrl.getMotionForceTaskByName("zzz")


def test_task_constructor_validation(sp):
    robot = _cfg_only(sp)
    with pytest.raises(ValueError, match="not consistent with robot dof"):
        sp.JointTask(robot, np.eye(6))
    S = np.zeros((2, 7))
    S[0, 1] = S[1, 1] = 1.0  # rank 1
    with pytest.raises(ValueError, match="not full rank"):
        sp.RobotController(robot, [sp.JointTask(robot, S, task_name="bad")])
    with pytest.raises(ValueError, match="cannot both be empty"):
        sp.MotionForceTask(robot, "end-effector", controlled_directions_translation=[], controlled_directions_rotation=[])
    with pytest.raises(ValueError, match="does not exist"):
        sp.RobotController(robot, [sp.MotionForceTask(robot, "no_such_link")])
    with pytest.raises(sp.SaipUnsupported, match="rank-1"):
        sp.RobotController(robot, [sp.MotionForceTask(robot, "end-effector", controlled_directions_translation=[[0, 0, 1]],
                                                      controlled_directions_rotation=None)])
    t = sp.MotionForceTask(robot, "end-effector")
    sp.RobotController(robot, [t])
    with pytest.raises(ValueError, match="positive or zero"):
        t.setPosControlGains(-1.0, 1.0, 0.0)
    with pytest.raises(ValueError, match="size 1 or 3"):
        t.setPosControlGains([1.0, 2.0], [1.0, 2.0], [0.0, 0.0])
    t.enableVelocitySaturation()            # defaults 0.3, pi/3 (MotionForceTask.h:63-64)
    t.enableVelocitySaturation(0.1, 0.5)
    with pytest.raises(ValueError, match="strictly positive"):
        t.enableVelocitySaturation(0.0, 0.5)
    with pytest.raises(ValueError, match="between 0 and 3"):
        t.parametrizeForceMotionSpaces(4, [0, 0, 1])
    with pytest.raises(ValueError, match="non singular vector"):
        t.parametrizeMomentRotMotionSpaces(1, [0, 0, 1e-3])
    assert t.parametrizeForceMotionSpaces(2, [1, 0, 0]) is True and t.parametrizeForceMotionSpaces(2, [3, 0, 0]) is False
    t.setClosedLoopForceControl(True)           # implemented (passivity controller off, its default)
    t.setClosedLoopMomentControl(True)
    t.setForceControlParameters(0.9, 0.8, 15.0, 5.0)
    t.enablePassivity()                         # POPC passivity observer / controller (MotionForceTask.h:630)
    t.disablePassivity()
    r2 = _cfg_only(sp)
    j2 = sp.JointTask(r2)
    sp.RobotController(r2, [j2])
    j2.enableVelocitySaturation(0.5)
    j2.enableVelocitySaturation(np.full(7, 0.5))
    with pytest.raises(ValueError, match="not consistent with task dof"):
        j2.enableVelocitySaturation(np.full(3, 0.5))
    with pytest.raises(ValueError, match="must be positive"):
        j2.enableVelocitySaturation(-1.0)


@pytest.mark.parametrize("cfg", [3, 7, 8])
def test_partial_task_projection_matches_reference_construction(sp, cfg):
    """P = blkdiag(Bt Bt^T, Br Br^T) from matrixRangeBasis of the controlled directions (MotionForceTask.cpp:55-87)
    and the engine's basis spans range(P)."""
    robot = _cfg_only(sp)
    spec = W.config_tasks(cfg)[0]
    t = sp.MotionForceTask(robot, spec["link"], spec["pos_in_link"], None, spec["dirs_trans"], spec["dirs_rot"])
    sp.RobotController(robot, [t])
    P, Bm = t.getTaskProjection()
    P_ref, B_ref, pr, orr = W.mf_projection(spec)
    assert np.allclose(P, P_ref, atol=1e-12)
    assert Bm.shape[1] == pr + orr
    assert np.allclose(Bm @ Bm.T, P_ref, atol=1e-12) and np.allclose(Bm.T @ Bm, np.eye(pr + orr), atol=1e-12)


def test_compute_entry_points_fail_loudly_without_a_device(sp):
    robot = _cfg_only(sp)
    tasks = [sp.MotionForceTask(robot, "end-effector", (0, 0, 0.07)), sp.JointTask(robot)]
    for t in tasks:
        t.disableInternalOtg()
    ctrl = sp.RobotController(robot, tasks)
    robot.setQ(np.zeros((4, 7)))
    with pytest.raises(sp.SaipNoDevice, match="no CPU path"):
        ctrl.updateControllerTaskModels()
    with pytest.raises(sp.SaipNoDevice):
        ctrl.computeControlTorques()
    with pytest.raises(sp.SaipNoDevice):
        ctrl.reinitializeTasks()
    with pytest.raises(sp.SaipNoDevice):
        ctrl.stepAsync()
    # direct C-ABI: creating a real batch without a device is refused
    L = sp.lib()
    h = C.c_void_p()
    if sp.device_count() == 0:
        st = L.saip_batch_create(robot._h, 4, 0, C.byref(h))
        assert st == 3 and b"no CPU path" in L.saip_last_error()


def test_product_package_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under sai-primitives_amd/ may reference it"""
    pkg = os.path.join(ROOT, "sai-primitives_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "saip_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_rejected_setter_leaves_the_replay_log_untouched(sp):
    """a configuration call the engine (or, for a task outside any batch, the Python side) rejects changes nothing: the log replayed into the
    batch the task joins keeps the last VALID call, getInternalOtgEnabled() keeps its value (the reference leaves its members untouched when a
    setter throws, JointTask.cpp:400-409), and buffers of replaced log entries are released"""
    robot = _cfg_only(sp)
    jt = sp.JointTask(robot, task_name="posture")
    jt.disableInternalOtg()
    jt.enableInternalOtgJerkLimited(1.0, 2.0, 3.0)
    good = [e for e in jt._log if e[0] == "saip_batch_set_otg_jerk_limited"]
    assert len(good) == 1 and jt.getInternalOtgEnabled()
    jt.disableInternalOtg()
    with pytest.raises(ValueError, match="jerk"):   # task outside any batch: rejected before anything is logged
        jt.enableInternalOtgJerkLimited(1.0, 2.0, 0.0)
    assert [e for e in jt._log if e[0] == "saip_batch_set_otg_jerk_limited"] == good and not jt.getInternalOtgEnabled()
    ctrl = sp.RobotController(robot, [jt])           # joins a (configuration-only) batch: the log replays without error
    with pytest.raises(ValueError):                  # task inside a batch: the engine rejects, the log still holds the valid call
        jt.enableInternalOtgJerkLimited(1.0, -2.0, 3.0)
    assert [e for e in jt._log if e[0] == "saip_batch_set_otg_jerk_limited"] == good and not jt.getInternalOtgEnabled()
    n0 = len(jt._keepalive)
    for i in range(50):                              # a caller that sets gains every cycle grows neither the log nor the keep-alive table
        jt.setGains(100.0 + i, 20.0, 0.0)
    assert len(jt._keepalive) <= n0 + 1 and len([e for e in jt._log if "gains" in e[0]]) == 1
    del ctrl


def test_xcd_block_remap_is_a_bijection():
    """csrc/saip_oct_common.h::saip_xcd_block (the wavefront-per-instance kernel takes instance saip_xcd_block(blockIdx, gridDim)): whatever the
    grid size, every instance is taken exactly once, and the workgroups that share an XCD (equal blockIdx % 8) take one contiguous range.
    The formula is restated here from the header's text; the -m gpu size tests (B = 1, 3, 65, 4096) exercise the device side."""
    src = open(os.path.join(ROOT, "sai-primitives_amd", "csrc", "saip_oct_common.h")).read()
    assert "(x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (orig >> 3)" in src   # the expression below is the header's

    def remap(orig, nwg):
        q, r, x = nwg >> 3, nwg & 7, orig & 7
        return (x * (q + 1) if x < r else r * (q + 1) + (x - r) * q) + (orig >> 3)

    for nwg in (1, 3, 7, 8, 9, 65, 511, 512, 4096, 4099):
        got = [remap(w, nwg) for w in range(nwg)]
        assert sorted(got) == list(range(nwg)), nwg
        for x in range(min(8, nwg)):
            mine = sorted(remap(w, nwg) for w in range(x, nwg, 8))
            assert mine == list(range(mine[0], mine[0] + len(mine))), (nwg, x)

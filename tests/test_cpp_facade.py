"""The header-only C++ facade (include/saip/SaiPrimitivesBatched.hpp) compiles with plain g++ against the C-ABI
library and keeps the reference's class/method names and error behaviour."""
import os
import subprocess

import numpy as np
import pytest

import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sai-primitives_amd")


def _build(tmp_path):
    import sai_primitives_amd as sp
    sp.build_library()
    exe = str(tmp_path / "facade_example")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "facade_example.cpp"),
                           "-L" + PKG, "-lsaip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    robot = tmp_path / "robot.txt"
    m = W.load_robot("panda_arm")
    jt = {"fixed": 0, "revolute": 1, "prismatic": 2}
    with open(robot, "w") as f:
        f.write(f"{len(m.links)}\n")
        for l in m.links:
            vals = l["origin_xyz"] + l["origin_rpy"] + l["axis"] + [l["mass"]] + l["com"] + l["inertia"] + \
                [l["q_lower"], l["q_upper"], l["velocity_limit"], l["effort_limit"]]
            f.write(f"{l['name']} {jt[l['joint_type']]} " + " ".join(repr(float(v)) for v in vals) + "\n")
    return exe, str(robot)


def test_cpp_facade_host_logic(tmp_path):
    exe, robot = _build(tmp_path)
    out = subprocess.run([exe, robot, "cfgonly"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "FACADE_CFG_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_facade_cycle_matches_oracle(tmp_path):
    from oracle import Oracle
    exe, robot = _build(tmp_path)
    B = 256
    d = W.make_inputs(2, B)
    blob = np.concatenate([d["q"].T, d["dq"].T, d["goals"][0].T, d["goals"][1].T], axis=0)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    np.ascontiguousarray(blob).tofile(inp)
    out = subprocess.run([exe, robot, "run", str(B), str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "FACADE_RUN_OK" in out.stdout, out.stdout + out.stderr
    tau = np.fromfile(outp).reshape(7, B).T
    ref, _ = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"])
    assert W.torque_error(tau, ref) < 1e-5


@pytest.mark.gpu
def test_cpp_facade_example04_manual_hierarchy(tmp_path):
    """the reference's per-task interface (TemplateTask.h:43-60) through the C++ facade: tasks without a RobotController, hierarchy
    built by hand like examples/04-task_and_redundancy/04-task_and_redundancy.cpp:141-206"""
    from oracle import Oracle
    exe, robot = _build(tmp_path)
    B = 192
    d = W.make_inputs(2, B)
    blob = np.concatenate([d["q"].T, d["dq"].T, d["goals"][0].T, d["goals"][1].T], axis=0)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    np.ascontiguousarray(blob).tofile(inp)
    out = subprocess.run([exe, robot, "ex04", str(B), str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "FACADE_EX04_OK" in out.stdout, out.stdout + out.stderr
    tau_mf, tau_jt, tau_jt_comp, tau_jt_dev, tau_ctrl = (x.T for x in np.fromfile(outp).reshape(5, 7, B))
    orc = Oracle(d["model"], d["tasks"])
    r_mf, _, Nt, st = orc.task_step(0, d["q"], d["dq"], d["goals"][0])
    assert (st & 1).sum() == 0
    r_jt, _, _, _ = orc.task_step(1, d["q"], d["dq"], d["goals"][1], N_prec=Nt)
    r_jtc, _, _, _ = orc.task_step(1, d["q"], d["dq"], d["goals"][1], N_prec=Nt, tau_prec=r_mf)
    assert W.torque_error(tau_mf, r_mf) < 1e-5 and W.torque_error(tau_jt, r_jt) < 1e-5 and W.torque_error(tau_jt_comp, r_jtc) < 1e-5
    assert np.array_equal(tau_jt_dev, tau_jt_comp)
    ref, _ = orc.step(d["q"], d["dq"], d["goals"])
    assert W.torque_error(tau_mf + tau_jt_comp, ref) < 1e-5 and W.torque_error(tau_ctrl, ref) < 1e-5

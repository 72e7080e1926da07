"""Native multi-GPU path of the engine (C++ on RCCL through the C-ABI, no Python in the data path): sai-primitives_amd/csrc/saip_comm.cpp.
On the one-GPU box the program asks for two shards on GPU 0; RCCL may refuse duplicate devices, in which case the single-shard path
(communicator of size 1, the same calls) is what runs.  The two-rank exchange itself is covered on CPU by tests/test_sharding_gloo.py."""
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
    exe = str(tmp_path / "multi_example")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "multi_example.cpp"),
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


def test_multi_example_compiles_against_the_c_abi(tmp_path):
    _build(tmp_path)   # plain g++ against libsaip.so: the comm entry points are exported (librccl itself is only loaded on first use)


@pytest.mark.gpu
@pytest.mark.parametrize("n_req", [1, 2])
def test_cpp_multi_device_shards_and_rccl_gather(tmp_path, n_req):
    from oracle import Oracle
    exe, robot = _build(tmp_path)
    B = 96
    d = W.make_inputs(2, n_req * B)
    blob = np.concatenate([d["q"].T, d["dq"].T, d["goals"][0].T, d["goals"][1].T], axis=0)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    np.ascontiguousarray(blob).tofile(inp)
    out = subprocess.run([exe, robot, str(B), str(n_req), str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "MULTI_OK" in out.stdout, out.stdout + out.stderr
    assert out.stdout.count("MULTI_TIMED") == 2     # saip_multi_time_steps in both gather modes (round 4)
    n = int(out.stdout.split("MULTI_DEVICES")[1].split()[0])
    print(out.stdout)
    first, last = np.fromfile(outp).reshape(2, n, 7, B)
    assert np.array_equal(first, last)                      # every device holds the whole result
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=4)
    assert st.sum() == 0
    for s in range(n):
        assert W.torque_error(first[s].T, ref[s * B:(s + 1) * B]) < 1e-5


@pytest.mark.gpu
def test_native_gather_one_rank_communicator():
    """what bench.py does per rank for N > 1, on the one rank a one-GPU box has: unique id -> communicator -> ncclAllGather of the torque
    slab on the engine stream (sharding.NativeGather); the gathered buffer then holds exactly the engine's torques"""
    import torch
    from sai_primitives_amd import sharding
    from sai_primitives_amd.controller import controller_from_specs
    B = 256
    d = W.make_inputs(2, B)
    robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    ctrl.updateControllerTaskModels()
    ctrl.setGoals(d["goals"])
    tau = ctrl.computeControlTorques()
    ld = ctrl.devicePointers()["ld"]
    out = torch.full((1, 7, ld), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    g = sharding.NativeGather(ctrl, 1, 0, 0, torch.device("cuda", 0))
    g.gather(out)
    ctrl.synchronize()
    g.close()
    assert np.array_equal(out[0, :, :B].cpu().numpy().T, tau)

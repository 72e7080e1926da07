"""Generates tests/golden/*.npz: seeded inputs + expected joint torques of the reference hot path,
computed by the independent NumPy restatement in oracle/restatement.py (NOT by the C oracle, NOT by the
product).  The reference itself cannot be built or imported here (C++ needing Eigen3/sai-model/RBDL,
SURVEY.md 8(c)), so these vectors pin the C oracle against a second restatement, not against the
reference binary ("parity unpinned", see oracle/saip_oracle.c header).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import workloads as W  # noqa: E402
import restatement as RS  # noqa: E402  oracle/restatement.py: the NumPy restatement of the reference path (test infrastructure)

# name -> (config id, instances, controller options)
SCENARIOS = {
    "cfg1_joint_pd": (1, 1, {}),
    "cfg2_mf_joint": (2, 48, {}),
    "cfg3_partial": (3, 48, {}),
    "cfg5_chain30": (5, 6, {}),
    "cfg6_sliding_base": (6, 32, {}),
    "cfg7_ori_only_full_decoupling": (7, 32, {}),
    "cfg8_planar_impedance": (8, 32, {}),
    "cfg2_gravity_saturation": (2, 32, dict(gravity_comp=True, torque_saturation=True)),
    "cfg9_hybrid_force_velsat": (9, 32, {}),
    "cfg10_full_force_space": (10, 32, {}),
    "cfg11_joint_limit_avoidance": (11, 48, dict(joint_limit_avoidance=True, torque_saturation=True)),
    "cfg11_jla_gravity": (11, 32, dict(joint_limit_avoidance=True, gravity_comp=True)),
    "cfg12_handling_disabled_truncated": (12, 48, {}),
    "cfg13_closed_loop_force": (13, 32, {}),
    "cfg14_blended_strategies": (14, 48, {}),
}


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (cfg, B, opts) in SCENARIOS.items():
        d = W.make_inputs(cfg, B)
        tau, status = RS.controller_step(d["model"], d["tasks"], d["q"], d["dq"], d["goals"], **opts)
        assert (status != 1).all(), name   # status 2 = task reduced to its non-singular subspace (config 12), 8 = blended strategies (14)
        arrs = dict(cfg=np.int64(cfg), q=d["q"], dq=d["dq"], tau=tau, status=status,
                    gravity_comp=np.bool_(opts.get("gravity_comp", False)),
                    torque_saturation=np.bool_(opts.get("torque_saturation", False)),
                    joint_limit_avoidance=np.bool_(opts.get("joint_limit_avoidance", False)))
        for i, g in enumerate(d["goals"]):
            arrs[f"goal{i}"] = g
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **{k: np.asarray(v, dtype=v.dtype.newbyteorder("<") if hasattr(v, "dtype") and v.dtype.kind == "f" else None) for k, v in arrs.items()})
        print(name, "B", B, "max|tau|", float(np.abs(tau).max()))


if __name__ == "__main__":
    main()

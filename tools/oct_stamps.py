"""diagnostic: phase timing of saip_cycle_oct, every workgroup (build the variant with tools/oct_variant.sh stamp -DSAIP_STAMP_OCT, run with
SAIP_LIB=sai-primitives_amd/libsaip_var_stamp.so python tools/oct_stamps.py [config])"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import workloads as W
from sai_primitives_amd import capi
from sai_primitives_amd.controller import controller_from_specs
CFG = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = W.make_inputs(CFG, B)
robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
ctrl.setKernel(3)
robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.setGoals(d["goals"])
NB = B // 8
runs = []
for rep in range(12):
    ctrl.stepAsync(); ctrl.synchronize()
    buf = (C.c_ulonglong * (1024 * 64))()
    capi.lib().saip_debug_stamps_oct(buf, 1024 * 64)
    full = np.array(buf[:], dtype=np.float64).reshape(1024, 2, 32)[:NB]
    runs.append(full)
s, sB = runs[-1][:, 0], runs[-1][:, 1]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 15]               # stamp 11 sits in the general joint task (not run by the headline stack); 12, 13 are not placed
names = ["input requests", "kinematics + Ic", "M column", "chol7 + M^-1 column", "Jf / vw", "-", "G + certificate",
         "T1 / A", "chol6 + solves", "law + g", "tau, row of N1, joint task (+ blended block)", "post + stores"]
dl = np.diff(s[:, idx], axis=1)
print(f"{'phase':48s} {'min':>8s} {'median':>8s} {'max':>8s}   (shader clocks, {NB} workgroups, last of 12 runs)")
for i, n_ in enumerate(names): print(f"{n_:48s} {dl[:, i].min():8.0f} {np.median(dl[:, i]):8.0f} {dl[:, i].max():8.0f}")
tot = s[:, 15] - s[:, 0]
print(f"{'workgroup total':48s} {tot.min():8.0f} {np.median(tot):8.0f} {tot.max():8.0f}")
print(f"first start -> last end: {s[:, 15].max() - s[:, 0].min():.0f}; start spread {s[:, 0].max() - s[:, 0].min():.0f}")
bl = s[:, 16:29]
has = bl[:, 0] > 0
if has.any() and (bl[has][:, -1] > bl[has][:, 0]).all():
    db = np.diff(bl[has], axis=1)
    print(f"blended block, {int(has.sum())} workgroups:")
    for i, n_ in enumerate(["G + Jacobi", "Ju / sigma", "decisions + pass-through", "state + control frame", "classification", "state update", "A_u / A_mod,u",
                            "block Cholesky x2", "forces", "solves + Schur", "Cholesky Pm + unit torques + tjs", "joint task behind"]):
        print(f"  {n_:46s} {db[:, i].min():8.0f} {np.median(db[:, i]):8.0f} {db[:, i].max():8.0f}")
if (sB[:, 29] > 0).all():
    print("two-wavefront form: wavefront B (shader clocks relative to its own start; A's clock is the same counter)")
    for lab, i0, i1 in [("B inputs", 0, 1), ("B kinematics", 1, 2), ("B Jacobian column", 2, 3), ("B Jf / vw", 4, 5), ("B G + certificate", 6, 7), ("B law", 9, 29)]:
        dd = sB[:, i1] - sB[:, i0]
        print(f"  {lab:46s} {dd.min():8.0f} {np.median(dd):8.0f} {dd.max():8.0f}")
    print(f"  B start -> barrier: median {np.median(sB[:, 29] - sB[:, 0]):.0f};  A start -> barrier passed: median {np.median(s[:, 30] - s[:, 0]):.0f};"
          f"  A waits at the barrier (stamp 9 -> 30): median {np.median(s[:, 30] - s[:, 9]):.0f};  B start - A start: median {np.median(sB[:, 0] - s[:, 0]):.0f}")
    if (s[:, 31] > 0).all():
        print(f"  second barrier: A arrives {np.median(s[:, 31] - s[:, 30]):.0f} after the first one, B {np.median(sB[:, 31] - s[:, 30]):.0f};  A: second barrier -> end of the tail {np.median(s[:, 14] - s[:, 31]):.0f}")
if (s[:, 11] > 0).all():  # general joint task (GJ != 0): the tail split at stamp 11
    print(f"general joint task: tau + column of N1 (10 -> 11) median {np.median(s[:, 11] - s[:, 10]):.0f}; joint task proper (11 -> 14) median {np.median(s[:, 14] - s[:, 11]):.0f}")

"""diagnostic: phase timing of saip_cycle_oct (build with SAIP_EXTRA_HIPCC_FLAGS=-DSAIP_STAMP_OCT into another library, run with SAIP_LIB)"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import workloads as W
from sai_primitives_amd import capi
from sai_primitives_amd.controller import controller_from_specs
d = W.make_inputs(2, 4096)
robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], 4096, device=0)
ctrl.setKernel(3)
robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.setGoals(d["goals"])
acc = None
for rep in range(20):
    ctrl.updateControllerTaskModels(); ctrl.computeControlTorques()
    buf = (C.c_ulonglong * 64)()
    capi.lib().saip_debug_stamps_oct(buf, 64)
    s = np.array(buf[:16], dtype=np.float64)
    dlt = np.diff(s)
    acc = dlt if acc is None else np.minimum(acc, dlt)
names = ["inputs", "kinematics+inertia+Ic", "M row + J column", "chol7 + solve", "Jf/vw", "-", "G + certificate", "T1/A", "chol6 + Lambda_mod", "law + g", "tau / N1 column", "range: norms", "range: pivot/deflate", "joint law + Lambda", "post"]
for n_, c in zip(names, acc): print(f"{n_:14s} {c:8.0f}")
print("total", acc.sum(), "clocks (100 MHz s_memtime ticks x ?)")

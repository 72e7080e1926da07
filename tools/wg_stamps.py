"""diagnostic: where one workgroup of the general kernel spends its clocks (build the variant with
SAIP_VARIANT_KERNEL=wg tools/oct_variant.sh wgstamp -DSAIP_STAMP_WG): s_memtime after every barrier of block 0, as a running log, so task
iterations are told apart.  Barriers inside saip_wg_linalg.h routines are not stamped: their time lands on the next kernel-file line.
   SAIP_LIB=sai-primitives_amd/libsaip_var_wgstamp.so python tools/wg_stamps.py [config] [batch]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import workloads as W
from sai_primitives_amd import capi
from sai_primitives_amd.controller import controller_from_specs
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = W.make_inputs(cfg, B)
robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
ctrl.setKernel(1)
robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.setGoals(d["goals"])
L = capi.lib()
L.saip_debug_stamps_wg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 4096)()
for rep in range(3):
    ctrl.stepAsync(); ctrl.synchronize()
    n = L.saip_debug_stamps_wg(buf, 4096)
a = np.array(buf[:2 * n], dtype=np.uint64).reshape(n, 2)
t = a[:, 0].astype(np.float64); ln = a[:, 1].astype(int)
tot = t[-1] - t[0]
print(f"{n} barriers of the kernel file, {tot:.0f} clocks")
acc = 0.0
for i in range(1, n):
    dt = t[i] - t[i - 1]
    if dt >= 0.004 * tot:
        print(f"  t = {t[i - 1] - t[0]:8.0f}  lines {ln[i - 1]:5d} -> {ln[i]:5d}: {dt:8.0f} clk  {100.0 * dt / tot:5.1f} %")
    else:
        acc += dt
print(f"  (intervals below 0.4 % each: {acc:.0f} clk, {100.0 * acc / tot:.1f} %)")

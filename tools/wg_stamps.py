"""diagnostic: where one workgroup of the general kernel spends its clocks (build saip_kernel_wg.hip with -DSAIP_STAMP_WG into another
library, run with SAIP_LIB): s_memtime after every barrier of block 0, printed as clocks per source-line interval.
   SAIP_LIB=... python tools/wg_stamps.py [config] [batch]"""
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
for rep in range(3):
    ctrl.updateControllerTaskModels(); ctrl.computeControlTorques()
buf = (C.c_ulonglong * 512)()
L = capi.lib()
L.saip_debug_stamps_wg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
L.saip_debug_stamps_wg(buf, 512)
a = np.array(buf[:], dtype=np.uint64).reshape(256, 2)
a = a[a[:, 0] > 0]
a = a[np.argsort(a[:, 0])]
t0 = a[0, 0]
prev_t, prev_l = t0, int(a[0, 1])
acc = {}
for t, l in a[1:]:
    key = (prev_l, int(l))
    acc[key] = acc.get(key, 0) + int(t - prev_t)
    prev_t, prev_l = t, int(l)
tot = int(a[-1, 0] - t0)
for (l0, l1), c in sorted(acc.items(), key=lambda kv: -kv[1])[:40]:
    print(f"lines {l0:5d} -> {l1:5d}: {c:9d} clk  {100.0 * c / tot:5.1f} %")
print("total", tot, "clk; static stamp sites hit:", len(a))

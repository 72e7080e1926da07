import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import workloads as W
from sai_primitives_amd.controller import controller_from_specs
import sai_primitives_amd as sp
cfg = int(os.environ.get("CFG", "2"))
d = W.make_inputs(cfg, 4096)
robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], 4096, device=0)
robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.setGoals(d["goals"])
ctrl.timeSteps(50, 0)
L = sp.lib()
buf = (C.c_ulonglong * 256)()
L.saip_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
L.saip_debug_stamps(buf, 256)
st = [(buf[2*i], buf[2*i+1]) for i in range(128) if buf[2*i+1] != 0]
st.sort()
t0 = st[0][0]
prev = t0
for t, line in st:
    print(f"line {line:5d}  t={t-t0:8d}  d={t-prev:7d}")
    prev = t

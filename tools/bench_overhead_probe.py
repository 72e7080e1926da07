"""where the fixed overhead of a short bench.py run goes (diagnostic; run on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as W
from sai_primitives_amd import sharding
from sai_primitives_amd.controller import controller_from_specs

B = 4096
d = W.make_inputs(2, B)
robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
n = robot.dof(); ld = ctrl.devicePointers()["ld"]
dev = torch.device("cuda", 0)
tau_t = torch.zeros((n, ld), dtype=torch.float64, device=dev)
out = sharding.gather_buffer(tau_t, 1)
torch.cuda.synchronize()
ctrl.bindTauDevice(tau_t.data_ptr())
robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.setGoals(d["goals"])
ctrl.timeSteps(5, 0)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    ev = ctrl.timeSteps(20, 0); t2 = time.perf_counter()
    g = sharding.gather_torques(tau_t, 1, out=out); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"sync {1e6*(t1-t0):.1f} us | timeSteps(20) wall {1e6*(t2-t1):.1f} us (events {ev*1e3:.1f} us) | gather issue {1e6*(t3-t2):.1f} us | sync {1e6*(t4-t3):.1f} us")
for rep in range(3):
    t1 = time.perf_counter(); ev = ctrl.timeSteps(20, 0); t2 = time.perf_counter()
    ctrl.synchronize(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"timeSteps(20) wall {1e6*(t2-t1):.1f} us | engine stream sync {1e6*(t3-t2):.1f} us | device sync {1e6*(t4-t3):.1f} us | again {1e6*(t5-t4):.1f} us")
for steps in (1, 20, 200):
    t1 = time.perf_counter(); ev = ctrl.timeSteps(steps, 0); t2 = time.perf_counter()
    print(f"timeSteps({steps}) wall {1e6*(t2-t1):.1f} us, events {ev*1e3:.1f} us")
import ctypes as C
from sai_primitives_amd import capi
L = capi.lib()
t1 = time.perf_counter()
for _ in range(20): L.saip_batch_step_async(ctrl._h)
t2 = time.perf_counter(); L.saip_batch_synchronize(ctrl._h); t3 = time.perf_counter()
print(f"20 x step_async issue {1e6*(t2-t1):.1f} us, then synchronize {1e6*(t3-t2):.1f} us")

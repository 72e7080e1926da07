"""diagnostic: where a wavefront of saip_cycle_wave spends its clocks.  Build the variant first:
   SAIP_VARIANT_KERNEL=wave SAIP_VARIANT_SCHED=default tools/oct_variant.sh wavestamp -DSAIP_STAMP_WAVE
   SAIP_LIB=sai-primitives_amd/libsaip_var_wavestamp.so python tools/wave_stamps.py [config] [batch]
s_memtime of lane 0 of the first 256 instances at the phase boundaries; medians over those wavefronts."""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import workloads as W
from sai_primitives_amd import capi
from sai_primitives_amd.controller import controller_from_specs
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = W.make_inputs(cfg, B)
robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
robot.setQ(d["q"]); robot.setDq(d["dq"]); robot.updateModel(); ctrl.setGoals(d["goals"])
L = capi.lib()
L.saip_debug_stamps_wave.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (256 * 64))()
for rep in range(3):
    ctrl.stepAsync(); ctrl.synchronize()
assert L.saip_debug_stamps_wave(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(256, 64).astype(np.float64)[:min(B, 256)]
names = {0: "start", 1: "kinematics (prefix product)", 2: "inertias, composite scan, S / F", 3: "M(q) build", 4: "M^-1 (sweep)", 5: "BIE: clamped entries, K^-1",
         60: "task loop end", 61: "epilogue"}
for t in range(5):
    for k, nm in enumerate(["task start", "MF: Jacobian + vw | JT: law, Jp", "MF: Jh, Jh^T | JT: Gram-Schmidt", "MF: certificate | JT: T1^T, A", "MF: T1, A, Lambda | JT: Lambda_mod, Lambda",
                            "MF: Lambda_mod | JT: g, tau", "MF: law, g, tau | JT: N update", "MF: N update"]):
        names[8 + 10 * t + k] = f"task {t}: {nm}"
ids = [i for i in sorted(names) if (a[:, i] > 0).all()]
tot = np.median(a[:, ids[-1]] - a[:, ids[0]])
print(f"kernel {ctrl.kernelName()}  config {cfg}  B {B}: median wavefront {tot:.0f} clocks (of the first {len(a)} instances)")
for p, c in zip(ids[:-1], ids[1:]):
    dt = np.median(a[:, c] - a[:, p])
    print(f"  -> {names[c]:55s} {dt:9.0f} clk {100 * dt / tot:5.1f} %")

"""diagnostic (GPU box): the in-kernel slow tail of the eight-lane kernels against the list launch it replaces when MANY instances leave the
non-singular branch -- config 3's stack (3-dof position task + partial joint task) with a given fraction of the batch near the elbow singularity,
evenly spread or packed into consecutive groups of eight (the tail's worst case: one wavefront recomputes all eight of its instances in turn).
   python tools/tail_load_probe.py            (run it twice: as is, and with SAIP_NO_SLOW_TAIL=1)"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
import workloads as W
from sai_primitives_amd.controller import controller_from_specs
B = 4096
d = W.make_inputs(3, B)
mode = "list launch" if os.environ.get("SAIP_NO_SLOW_TAIL") else "slow tail"
for frac, packed in [(0.0, False), (1 / 64, False), (1 / 64, True), (1 / 8, False), (1 / 8, True), (0.5, False)]:
    q = d["q"].copy()
    nsing = int(B * frac)
    idx = np.arange(nsing) if packed else (np.arange(nsing) * (B // max(nsing, 1))) % B
    q[idx, 3] = -0.07 - 0.2 * np.random.default_rng(1).uniform(size=nsing)   # elbow nearly straight: inside SingularityHandler's blending region
    robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
    robot.setQ(q); robot.setDq(d["dq"]); robot.updateModel(); ctrl.setGoals(d["goals"])
    ms = ctrl.timeSteps(50, 5)
    tau = ctrl.getTorques()
    nb = int(((ctrl.status & 8) != 0).sum())
    print(f"{mode:11s} singular fraction {frac:6.4f} {'packed' if packed else 'spread'}: {ms / 50 * 1e3:8.1f} us per step, blended {nb}, flagged {int((ctrl.status & 1).sum())}, finite {bool(np.isfinite(tau).all())}")

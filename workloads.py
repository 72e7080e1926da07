"""Harness-only module: robot descriptions, task specs, the synthetic workload generator of SURVEY.md 8(d) and
the roofline accounting (algorithmic flops / bytes per instance-cycle).

NOT part of the product path.  The product is the HIP engine behind include/saip.h; nothing in here is called by
it.  bench.py, __graft_entry__.smoke() and the tests import this file to synthesise inputs; the generator needs
forward kinematics, the Jacobian and the mass matrix only for the rejection rule "resample while s_min/s_0 < 0.10".
The NumPy restatement of the reference's control law (what the golden vectors are generated with) lives in
oracle/restatement.py, test infrastructure like the rest of oracle/.
"""
from __future__ import annotations

import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROBOT_DIR = os.path.join(_HERE, "sai-primitives_amd", "robots")

FULL_DYNAMIC_DECOUPLING = 0  # SaiPrimitivesCommonDefinitions.h:14-20
BOUNDED_INERTIA_ESTIMATES = 1
IMPEDANCE = 2


# --------------------------------------------------------------------------- PRNG
_MASK = (1 << 64) - 1


def _sm64_mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


class SplitMix64Streams:
    """One SplitMix64 stream per instance (instance-major generation).

    Stream b is seeded with mix(seed + (b+1)*GOLDEN); every draw advances only that
    instance's state, so rejection re-draws of one instance never shift another's."""

    GOLDEN = np.uint64(0x9E3779B97F4A7C15)

    def __init__(self, seed: int, batch: int):
        with np.errstate(over="ignore"):
            b = np.arange(1, batch + 1, dtype=np.uint64)
            self.state = _sm64_mix(np.uint64(seed & _MASK) + b * self.GOLDEN)

    def uniform(self, mask=None) -> np.ndarray:
        """next U[0,1) double per instance (only instances in `mask` advance)."""
        with np.errstate(over="ignore"):
            nxt = self.state + self.GOLDEN
            out = _sm64_mix(nxt)
        if mask is None:
            self.state = nxt
        else:
            self.state = np.where(mask, nxt, self.state)
        return (out >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def uniform_range(self, lo, hi, count, mask=None) -> np.ndarray:
        """(B, count) draws in [lo, hi)."""
        cols = [self.uniform(mask) for _ in range(count)]
        u = np.stack(cols, axis=1)
        return lo + (hi - lo) * u


# --------------------------------------------------------------------------- model
class RobotModel:
    def __init__(self, desc: dict):
        self.name = desc["name"]
        self.links = desc["links"]
        self.nl = len(self.links)
        self.link_names = [l["name"] for l in self.links]
        self.dof_of_link = []
        n = 0
        for l in self.links:
            if l["joint_type"] == "fixed":
                self.dof_of_link.append(-1)
            else:
                self.dof_of_link.append(n)
                n += 1
        self.dof = n
        mov = [l for l in self.links if l["joint_type"] != "fixed"]
        self.q_lower = np.array([l["q_lower"] for l in mov])
        self.q_upper = np.array([l["q_upper"] for l in mov])
        self.vel_limit = np.array([l["velocity_limit"] for l in mov])
        self.effort = np.array([l["effort_limit"] for l in mov])
        self.q_nominal = np.deg2rad(np.array([l.get("q_nominal_deg", 0.0) for l in mov]))

    def link_index(self, name: str) -> int:
        return self.link_names.index(name)


def load_robot(name: str) -> RobotModel:
    with open(os.path.join(ROBOT_DIR, name + ".json")) as f:
        return RobotModel(json.load(f))


def rot_rpy(rpy) -> np.ndarray:
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


def rodrigues(axis, q: np.ndarray) -> np.ndarray:
    """(B,3,3) rotation about unit `axis` by q (B,)"""
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    K = _skew(a)
    c = np.cos(q)[:, None, None]
    s = np.sin(q)[:, None, None]
    return np.eye(3)[None] + s * K[None] + (1 - c) * (K @ K)[None]


def fk(model: RobotModel, q: np.ndarray):
    """q (B,n) -> list of (R (B,3,3), o (B,3)) per link, world == base frame."""
    B = q.shape[0]
    R = np.broadcast_to(np.eye(3), (B, 3, 3)).copy()
    o = np.zeros((B, 3))
    out = []
    for li, l in enumerate(model.links):
        R0 = rot_rpy(l["origin_rpy"])
        o = o + R @ np.asarray(l["origin_xyz"], float)
        R = R @ R0
        d = model.dof_of_link[li]
        if l["joint_type"] == "revolute":
            R = R @ rodrigues(l["axis"], q[:, d])
        elif l["joint_type"] == "prismatic":
            a = np.asarray(l["axis"], float)
            o = o + (R @ a) * q[:, d][:, None]
        out.append((R, o))
    return out


def jacobian(model: RobotModel, frames, link: int, p_world: np.ndarray) -> np.ndarray:
    """(B,6,n) geometric Jacobian [Jv;Jw] of a point fixed in `link` (world frame)."""
    B = p_world.shape[0]
    J = np.zeros((B, 6, model.dof))
    for li in range(link + 1):
        d = model.dof_of_link[li]
        if d < 0:
            continue
        R, o = frames[li]
        z = R @ (np.asarray(model.links[li]["axis"], float) / np.linalg.norm(model.links[li]["axis"]))
        if model.links[li]["joint_type"] == "revolute":
            J[:, 0:3, d] = np.cross(z, p_world - o)
            J[:, 3:6, d] = z
        else:
            J[:, 0:3, d] = z
    return J


def mass_matrix(model: RobotModel, frames) -> np.ndarray:
    """M = sum_bodies m Jv^T Jv + Jw^T (R I R^T) Jw (Jacobian form; RBDL uses CRBA)."""
    B = frames[0][0].shape[0]
    n = model.dof
    M = np.zeros((B, n, n))
    for li, l in enumerate(model.links):
        R, o = frames[li]
        c = o + R @ np.asarray(l["com"], float)
        J = jacobian(model, frames, li, c)
        ixx, iyy, izz, ixy, ixz, iyz = l["inertia"]
        I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
        Iw = R @ I @ np.swapaxes(R, 1, 2)
        Jv, Jw = J[:, :3], J[:, 3:]
        M += l["mass"] * np.swapaxes(Jv, 1, 2) @ Jv + np.swapaxes(Jw, 1, 2) @ Iw @ Jw
    return M


# ----------------------------------------------------------- sai-model style helpers
def matrix_range_basis(A: np.ndarray, tol: float = 1e-3) -> np.ndarray:
    """SaiModel::matrixRangeBasis [RECALLED]: zeros(rows,1) if ~0, identity if full row
    rank, else the left singular vectors with sigma_i/sigma_0 >= tol."""
    rows, cols = A.shape
    if np.linalg.norm(A) < tol:
        return np.zeros((rows, 1))
    U, s, _ = np.linalg.svd(A, full_matrices=False)
    if s[0] < tol:
        return np.zeros((rows, 1))
    task_dof = min(rows, cols)
    for i in range(len(s) - 1, 0, -1):
        if s[i] / s[0] < tol:
            task_dof -= 1
        else:
            break
    if task_dof == rows:
        return np.eye(rows)
    return U[:, :task_dof]


# --------------------------------------------------------------------------- tasks
def motion_force_task(name, link, pos_in_link=(0, 0, 0), rot_in_link=None, dirs_trans=None,
                      dirs_rot=None, **kw):
    """Task spec; defaults = MotionForceTask.h:40-75 with OTG disabled."""
    t = dict(type="motion_force", name=name, link=link, pos_in_link=list(map(float, pos_in_link)),
             rot_in_link=(np.eye(3) if rot_in_link is None else np.asarray(rot_in_link, float)).tolist(),
             dirs_trans=dirs_trans, dirs_rot=dirs_rot,
             kp_pos=100.0, kv_pos=20.0, ki_pos=0.0, kp_ori=200.0, kv_ori=28.3, ki_ori=0.0,
             decoupling=BOUNDED_INERTIA_ESTIMATES, bie_threshold=0.1, s_min=6e-3, s_max=6e-2,
             dt=0.001,
             # MotionForceTask.h:50-66: force/moment gains (only kv acts in open loop), velocity saturation, force spaces
             kv_force=10.0, kv_moment=10.0, vel_sat=False, lin_sat=0.3, ang_sat=np.pi / 3,
             force_dim=0, force_axis=[0.0, 0.0, 1.0], moment_dim=0, moment_axis=[0.0, 0.0, 1.0],
             param_in_compliant_frame=False,
             singularity_handling=True,   # SingularityHandler::_enforce_handling_strategy, SingularityHandler.cpp:61
             singularity_strategies=True)  # the blended type-1 / type-2 strategies the reference always runs while the handling is enforced
                                           # (False = engine extra: instances outside the non-singular branch are flagged instead)
    t.update(kw)
    return t


def joint_task(name, S=None, **kw):
    """Task spec; defaults = JointTask.h:31-45 with OTG disabled. S = list of rows or None."""
    t = dict(type="joint", name=name, S=S, kp=50.0, kv=14.0, ki=0.0,
             decoupling=BOUNDED_INERTIA_ESTIMATES, bie_threshold=0.1, dt=0.001,
             vel_sat=False, sat=np.pi / 3)  # JointTask.h:43-44
    t.update(kw)
    return t


def mf_projection(task):
    """Partial task projector P (6x6) and task range basis (6xk), MotionForceTask.cpp:28, 55-87, 143-168."""
    if task["dirs_trans"] is None and task["dirs_rot"] is None:
        P = np.eye(6)
    else:
        bt = np.zeros((3, 1))
        br = np.zeros((3, 1))
        if task["dirs_trans"]:
            bt = matrix_range_basis(np.array(task["dirs_trans"], float).T)
        if task["dirs_rot"]:
            br = matrix_range_basis(np.array(task["dirs_rot"], float).T)
        P = np.zeros((6, 6))
        P[:3, :3] = bt @ bt.T
        P[3:, 3:] = br @ br.T
    rp = matrix_range_basis(P[:3, :3])
    ro = matrix_range_basis(P[3:, 3:])
    pr = 0 if np.linalg.norm(rp) == 0 else rp.shape[1]
    orr = 0 if np.linalg.norm(ro) == 0 else ro.shape[1]
    Bm = np.zeros((6, pr + orr))
    if pr:
        Bm[:3, :pr] = rp
    if orr:
        Bm[3:, pr:] = ro
    return P, Bm, pr, orr


def joint_selection(task, n):
    if task["S"] is None:
        return np.eye(n)
    S = np.asarray(task["S"], float)
    if S.ndim == 1:  # list of joint indices
        rows = [int(i) for i in S]
        S = np.zeros((len(rows), n))
        for r, i in enumerate(rows):
            S[r, i] = 1.0
    return S


def goal_size(task, n):
    if task["type"] == "motion_force":
        # x3, R9 (row-major), v3, w3, a3, alpha3 [, goal force 3, goal moment 3 when a force/moment space is active]
        if task.get("cl_force") or task.get("cl_moment"):
            return 36   # + sensed force 3, sensed moment 3 (sensor frame), MotionForceTask.cpp:805-828
        return 30 if (task.get("force_dim", 0) or task.get("moment_dim", 0)) else 24
    return 3 * joint_selection(task, n).shape[0]


# ------------------------------------------------------------------ workload synthesis
def _expm_so3(w: np.ndarray) -> np.ndarray:
    """(B,3) rotation vectors -> (B,3,3)"""
    th = np.linalg.norm(w, axis=1)
    out = np.empty((w.shape[0], 3, 3))
    for b in range(w.shape[0]):
        if th[b] < 1e-12:
            out[b] = np.eye(3)
        else:
            K = _skew(w[b] / th[b])
            out[b] = np.eye(3) + np.sin(th[b]) * K + (1 - np.cos(th[b])) * K @ K
    return out


CONFIGS = {
    1: dict(robot="panda_arm", batch=1, desc="single Panda JointTask PD (example 01 scenario)"),
    2: dict(robot="panda_arm", batch=4096, desc="Panda, full MotionForceTask + full JointTask"),
    3: dict(robot="panda_arm", batch=4096, desc="Panda, 3-dof position MotionForceTask + partial JointTask rows {0,2,4,6}"),
    4: dict(robot="panda_arm", batch=65536, desc="config-2 tasks, 8 shards"),
    5: dict(robot="chain30", batch=4096, desc="30-DoF chain, MotionForce(link30) + MotionForce(link15) + posture JointTask"),
    # extra parity scenarios (not BASELINE configs)
    6: dict(robot="panda_sliding_base", batch=256, desc="8-DoF sliding-base Panda, partial JointTask {0,7} above a full MotionForceTask (examples/06)"),
    7: dict(robot="panda_arm", batch=256, desc="Panda, orientation-only MotionForceTask (examples/09 style) + full JointTask, anisotropic gains, FULL_DYNAMIC_DECOUPLING"),
    8: dict(robot="panda_arm", batch=256, desc="Panda, planar (2 translation + 1 rotation dirs, non axis-aligned) MotionForceTask + JointTask, IMPEDANCE"),
    9: dict(robot="panda_arm", batch=256, desc="Panda, hybrid motion/force: force space dim 1 (compliant-frame axis) + moment space dim 2, open-loop force "
            "control with goal force/moment, velocity saturation on both tasks (examples/07-09 style)"),
    10: dict(robot="panda_arm", batch=256, desc="Panda, full force space (dim 3) + motion-only rotation, world-frame parametrisation, joint velocity saturation"),
    11: dict(robot="panda_arm", batch=256, desc="config-2 tasks with joint states pushed into the joint-limit-avoidance zones (q within 2% of the limits, "
             "|dq| up to the velocity limit): exercises RobotController's JLA wrap + torque saturation"),
    13: dict(robot="panda_arm", batch=256, desc="Panda, closed-loop force (1 axis) and moment (2 axes) control in the compliant frame with a rotated / offset "
             "force sensor, PI feedback with saturation, feed-forward gains; full JointTask in the nullspace"),
    14: dict(robot="panda_arm", batch=256, desc="config-2 tasks with the blended singularity strategies switched on; every third instance has the elbow nearly "
             "straight (type-1 singularity), every third the wrist nearly aligned next to the joint limit (type-2)"),
    15: dict(robot="panda_sliding_base", batch=256, desc="8-DoF sliding-base Panda, full MotionForceTask + full JointTask in its nullspace (rank 2): the headline stack on an 8-dof arm"),
    12: dict(robot="panda_arm", batch=256, desc="config-2 tasks with disableSingularityHandling(); every third instance has the elbow nearly straight "
             "(inside SingularityHandler's blending region): the motion-force task is reduced to its non-singular subspace"),
}


def config_tasks(cfg: int):
    if cfg == 1:
        return [joint_task("joint_task", kp=100.0, kv=20.0)]          # examples/01-joint_control:133
    if cfg in (2, 4, 11):
        return [motion_force_task("motion_force_task", "end-effector", (0, 0, 0.07)),  # examples/05:111-117
                joint_task("joint_task")]
    if cfg == 12:
        return [motion_force_task("motion_force_task", "end-effector", (0, 0, 0.07), singularity_handling=False),  # MotionForceTask.h:723
                joint_task("joint_task")]
    if cfg == 14:
        return [motion_force_task("motion_force_task", "end-effector", (0, 0, 0.07), singularity_strategies=True),
                joint_task("joint_task")]
    if cfg == 3:
        return [motion_force_task("motion_force_task", "end-effector", (0, 0, 0.07),
                                  dirs_trans=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], dirs_rot=None),  # examples/09:114-121
                joint_task("partial_joint_task", S=[0, 2, 4, 6])]
    if cfg == 5:
        return [motion_force_task("hand_task", "link30", (0, 0, 0.1)),
                motion_force_task("elbow_task", "link15", (0, 0, 0.1)),
                joint_task("posture_task")]
    if cfg == 6:
        return [joint_task("partial_joint_task", S=[0, 7]),                                   # examples/06:107-112
                motion_force_task("motion_force_task", "end-effector", (0, 0, 0.07))]
    if cfg == 15:
        return [motion_force_task("motion_force_task", "end-effector", (0, 0, 0.07)), joint_task("joint_task")]
    if cfg == 7:
        return [motion_force_task("ori_task", "end-effector", (0, 0, 0.07), dirs_trans=None,
                                  dirs_rot=[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                                  kp_ori=[150.0, 200.0, 250.0], kv_ori=[20.0, 28.3, 30.0], ki_ori=[0.0, 0.0, 0.0],
                                  decoupling=FULL_DYNAMIC_DECOUPLING),
                joint_task("joint_task", kp=[50, 60, 70, 80, 90, 100, 110], kv=[14, 15, 16, 17, 18, 19, 20],
                           ki=[0, 0, 0, 0, 0, 0, 0], decoupling=FULL_DYNAMIC_DECOUPLING)]
    if cfg == 8:
        return [motion_force_task("planar_task", "link7", (0.05, 0.0, 0.1),
                                  dirs_trans=[[1, 1, 0], [0, 0, 1], [2, 2, 1]], dirs_rot=[[1, -1, 0.5]],
                                  decoupling=IMPEDANCE),
                joint_task("joint_task", decoupling=IMPEDANCE)]
    if cfg == 9:
        return [motion_force_task("hybrid_task", "end-effector", (0, 0, 0.07), force_dim=1, force_axis=[0.1, -0.2, 1.0],
                                  moment_dim=2, moment_axis=[0.0, 1.0, 0.3], param_in_compliant_frame=True, vel_sat=True,
                                  lin_sat=0.02, ang_sat=0.1, kv_force=12.0, kv_moment=8.0, ki_pos=3.0, ki_ori=2.0),
                joint_task("joint_task", vel_sat=True, sat=0.15, ki=1.5)]
    if cfg == 10:
        return [motion_force_task("force_task", "end-effector", (0, 0, 0.07), force_dim=3, moment_dim=0),
                joint_task("joint_task", vel_sat=True, sat=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7])]
    if cfg == 13:
        c, s_ = np.cos(0.4), np.sin(0.4)
        return [motion_force_task("contact_task", "end-effector", (0, 0, 0.07), force_dim=1, force_axis=[0.0, 0.0, 1.0],
                                  moment_dim=2, moment_axis=[0.0, 0.0, 1.0], param_in_compliant_frame=True,
                                  cl_force=True, cl_moment=True, kp_force=0.9, kv_force=12.0, ki_force=1.7,
                                  kp_moment=0.6, kv_moment=8.0, ki_moment=1.1, kff_force=0.9, kff_moment=0.8,
                                  max_force_fb=4.0, max_moment_fb=0.5,                      # low enough to saturate some instances
                                  Rcs=[[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]], tcs=[0.01, -0.02, 0.05]),
                joint_task("joint_task")]
    raise ValueError(cfg)


def _task_sv_ratio(model, tasks, q):
    """min over MotionForce tasks of s_min/s_0 of the projected task Jacobian (batched)."""
    B, n = q.shape
    frames = fk(model, q)
    last_mf = max(i for i, t in enumerate(tasks) if t["type"] == "motion_force")
    need_N = last_mf > 0
    Minv = np.linalg.inv(mass_matrix(model, frames)) if need_N else None
    N_prec = np.broadcast_to(np.eye(n), (B, n, n))
    ratio = np.full(B, np.inf)
    for ti, t in enumerate(tasks[:last_mf + 1]):
        if t["type"] != "motion_force":
            # joint task above a motion-force task: selection rows are full row rank under N_prec here
            S = joint_selection(t, n)
            Jh = S[None] @ N_prec
            Lam = np.linalg.inv(Jh @ Minv @ np.swapaxes(Jh, 1, 2))
            N_prec = (np.eye(n)[None] - Minv @ np.swapaxes(Jh, 1, 2) @ Lam @ Jh) @ N_prec
            continue
        li = model.link_index(t["link"])
        R_l, o_l = frames[li]
        p = o_l + R_l @ np.asarray(t["pos_in_link"])
        _, Bm, pr, orr = mf_projection(t)
        Jh = np.einsum("ik,bin->bkn", Bm, jacobian(model, frames, li, p)) @ N_prec
        s = np.linalg.svd(Jh, compute_uv=False)
        ratio = np.minimum(ratio, s[:, -1] / s[:, 0])
        if need_N and ti < last_mf:
            Lam = np.linalg.inv(Jh @ Minv @ np.swapaxes(Jh, 1, 2))
            N = np.eye(n)[None] - Minv @ np.swapaxes(Jh, 1, 2) @ Lam @ Jh
            N_prec = N @ N_prec
    return ratio


def make_inputs(cfg: int, batch: int | None = None, *, shard: int = 0, min_ratio: float = 0.10):
    """Synthetic inputs of SURVEY.md 8(d). Returns dict(model, tasks, q, dq, goals).
    `shard` offsets the instance index so that rank r of a sharded run draws instances
    [r*batch, (r+1)*batch) of the same global stream."""
    model = load_robot(CONFIGS[cfg]["robot"])
    tasks = config_tasks(cfg)
    B = CONFIGS[cfg]["batch"] if batch is None else batch
    n = model.dof
    seed = 0x5A10000 + (2 if cfg == 4 else cfg)
    rng = SplitMix64Streams(seed, B * (shard + 1))
    rng.state = rng.state[shard * B:]
    if cfg == 1:
        q = model.q_nominal[None].copy()
        dq = np.zeros((1, n))
        qg = q + np.array([[0, 0, 0.4, -0.6, 0, 0, 0]])             # mirrors examples/01:158-161
        return dict(model=model, tasks=tasks, q=q, dq=dq,
                    goals=[np.concatenate([qg, np.zeros((1, n)), np.zeros((1, n))], axis=1)])
    rngq = model.q_upper - model.q_lower
    margin = 0.02 if cfg == 11 else 0.15
    lo, hi = model.q_lower + margin * rngq, model.q_upper - margin * rngq
    q = lo + (hi - lo) * rng.uniform_range(0.0, 1.0, n)
    has_mf = any(t["type"] == "motion_force" for t in tasks)
    if has_mf:
        for _ in range(5000):
            bad = _task_sv_ratio(model, tasks, q) < min_ratio
            if not bad.any():
                break
            qn = lo + (hi - lo) * rng.uniform_range(0.0, 1.0, n, mask=bad)
            q = np.where(bad[:, None], qn, q)
        else:
            raise RuntimeError("rejection sampling did not converge")
    if cfg == 12:
        near = (np.arange(B) % 3) == 0
        q4 = -0.07 - 0.28 * rng.uniform_range(0.0, 1.0, 1)[:, 0]     # elbow in [-0.35, -0.07] rad: sigma_5 / sigma_0 below s_max = 0.06
        q[:, 3] = np.where(near, q4, q[:, 3])
    if cfg == 14:
        u = rng.uniform_range(0.0, 1.0, 2)
        q[:, 3] = np.where((np.arange(B) % 3) == 0, -0.07 - 0.28 * u[:, 0], q[:, 3])   # elbow, as config 12
        q[:, 5] = np.where((np.arange(B) % 3) == 1, 0.005 + 0.03 * u[:, 1], q[:, 5])   # wrist: joints 5 and 7 nearly aligned
    dq = rng.uniform_range(-0.5, 0.5, n)
    if cfg == 11:
        dq = dq * 2.0 * model.vel_limit[None]    # up to +-vel_limit: velocity zones get entered
    frames = fk(model, q)
    goals = []
    for t in tasks:
        if t["type"] == "motion_force":
            li = model.link_index(t["link"])
            R_l, o_l = frames[li]
            x = o_l + R_l @ np.asarray(t["pos_in_link"])
            Rc = R_l @ np.asarray(t["rot_in_link"])
            xg = x + rng.uniform_range(-0.05, 0.05, 3)
            Rg = _expm_so3(rng.uniform_range(-0.2, 0.2, 3)) @ Rc
            vg = rng.uniform_range(-0.1, 0.1, 3)
            wg = rng.uniform_range(-0.1, 0.1, 3)
            ag = rng.uniform_range(-0.5, 0.5, 3)
            alg = rng.uniform_range(-0.5, 0.5, 3)
            blocks = [xg, Rg.reshape(B, 9), vg, wg, ag, alg]
            if goal_size(t, n) >= 30:
                blocks += [rng.uniform_range(-5.0, 5.0, 3), rng.uniform_range(-1.0, 1.0, 3)]  # goal force, goal moment
            if goal_size(t, n) == 36:
                blocks += [rng.uniform_range(-8.0, 8.0, 3), rng.uniform_range(-1.5, 1.5, 3)]  # sensed force, sensed moment (sensor frame)
            goals.append(np.concatenate(blocks, axis=1))
        else:
            S = joint_selection(t, n)
            m = S.shape[0]
            qg = q @ S.T + rng.uniform_range(-0.3, 0.3, m)
            dqg = rng.uniform_range(-0.1, 0.1, m)
            ddqg = rng.uniform_range(-0.5, 0.5, m)
            goals.append(np.concatenate([qg, dqg, ddqg], axis=1))
    return dict(model=model, tasks=tasks, q=q, dq=dq, goals=goals)


def torque_error(tau, tau_ref):
    """max over instances/joints of |tau-tau_ref| / max(|tau_ref|, 1e-3*||tau_ref||_inf)."""
    tau, tau_ref = np.asarray(tau), np.asarray(tau_ref)
    scale = np.maximum(np.abs(tau_ref), 1e-3 * np.max(np.abs(tau_ref), axis=-1, keepdims=True))
    return float(np.max(np.abs(tau - tau_ref) / scale))


# ------------------------------------------------------------------ roofline accounting (DESIGN.md "Algorithmic work")
def algorithmic_bytes(model, tasks) -> int:
    """minimal HBM I/O per instance-cycle: q, dq, goals in; tau out; FP64; integrators elided (ki = 0)."""
    n = model.dof
    words = 2 * n + n
    for t in tasks:
        if t["type"] == "motion_force":
            _, _, pr, orr = mf_projection(t)
            words += (9 if pr else 0) + (15 if orr else 0)  # x,v,a  /  R,w,alpha
        else:
            words += goal_size(t, n)
    return 8 * words


def algorithmic_flops(model, tasks) -> int:
    """Lean closed-form operation count per instance-cycle (mul and add count 1 each), by formula:
    sin/cos 40 each; FK 108/joint; body spatial inertia 125/body + suffix sums 10(n-1); M: 80 per upper-triangle entry;
    SPD inverse of a p x p matrix p^3; symmetric eigenvalues of a k x k Gram matrix 4/3 k^3 + 3 k^2 (tridiagonal QR);
    rank-revealing orthonormal range basis of an m x n matrix of rank r: 4 m n r; dense products at 2*m*k*p
    (symmetric results at half of that).  First task skips the product with N_prec = I; the last task skips the
    N_prec update.  This is the figure bench.py prices the kernel with (10.6 kflop for config 2; SURVEY's a-priori
    estimate was ~12 k)."""
    n = model.dof
    f = 40 * n + 108 * n + 125 * n + 10 * (n - 1) + 80 * n * (n + 1) // 2
    any_bie = any(t["decoupling"] == BOUNDED_INERTIA_ESTIMATES for t in tasks)
    f += n ** 3 * (2 if any_bie else 1)
    first = True
    for ti, t in enumerate(tasks):
        last = ti == len(tasks) - 1
        if t["type"] == "motion_force":
            _, _, pr, orr = mf_projection(t)
            k = pr + orr
            f += 9 * n + (0 if k == 6 else 2 * k * 6 * n)           # world Jacobian (+ basis projection)
            f += 0 if first else 2 * k * n * n                      # Jh = Jb N_prec
            f += k * (k + 1) // 2 * (2 * n - 1) + (4 * k ** 3) // 3 + 3 * k * k   # Gram + singular values
            r, law = k, 150 + 2 * 6 * k
        else:
            m = joint_selection(t, n).shape[0]
            r = min(m, n)
            f += 0 if (t["S"] is None or first) else 2 * m * n * n  # Jp = S N_prec
            law = 5 * m + 2 * n * n + 2 * m
            if not first:
                # rank of S N_prec: n minus the ranks of the tasks above, capped by m
                used = 0
                for u in tasks[:ti]:
                    used += (sum(mf_projection(u)[2:]) if u["type"] == "motion_force" else joint_selection(u, n).shape[0])
                r = max(1, min(m, n - used))
                f += 4 * m * n * r + 2 * r * m * n                  # range basis + Jh = U^T Jp
            law += 3 * 2 * r * m + 2 * 2 * r * r
        f += 2 * r * n * n + r * (r + 1) // 2 * (2 * n - 1) + r ** 3       # T1, A, Lambda
        if not last:
            f += 2 * r * r * n + 2 * n * n * r + n * n                      # Jbar^T, N
            f += 0 if first else 2 * n ** 3                                 # N_prec <- N N_prec
        if t["decoupling"] == BOUNDED_INERTIA_ESTIMATES:
            f += 2 * r * n * n + r * (r + 1) // 2 * (2 * n - 1) + r ** 3
        f += law + 2 * r * r + 2 * r * n                                    # control law, g, tau += Jh^T g
        first = False
    return int(f)

"""Harness-only module: robot descriptions, the synthetic workload generator of
SURVEY.md 8(d), and an independent NumPy restatement of the reference hot path.

NOT part of the product path.  The product is the HIP engine behind include/saip.h;
nothing in here is called by it.  This file is used by
  * tests/ and tests/golden/make_golden.py (second, independent restatement that the C
    oracle in oracle/ is cross-checked against),
  * bench.py / tests to synthesise inputs (needs FK + Jacobian singular values for the
    rejection sampling rule "resample while s_min/s_0 < 0.10").

The restatement follows, literally (SVD path, explicit inverses):
  /root/reference/src/RobotController.cpp:68-118
  /root/reference/src/tasks/MotionForceTask.cpp:247-509, 892-971
  /root/reference/src/tasks/SingularityHandler.cpp:75-228, 297-368
  /root/reference/src/tasks/JointTask.cpp:218-356
sai-model semantics (external, not in /root/reference) are restated from their algebraic
definitions, SURVEY.md Appendix B.
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


def gravity_vector(model: RobotModel, frames, g=(0.0, 0.0, -9.81)) -> np.ndarray:
    """jointGravityVector: tau_g = -sum_b m_b Jv_b^T g   (so that tau_g compensates gravity)."""
    B = frames[0][0].shape[0]
    tau = np.zeros((B, model.dof))
    gv = np.asarray(g, float)
    for li, l in enumerate(model.links):
        R, o = frames[li]
        c = o + R @ np.asarray(l["com"], float)
        J = jacobian(model, frames, li, c)
        tau -= l["mass"] * np.einsum("bij,i->bj", J[:, :3], gv)
    return tau


def popc_new():
    """POPCExplicitForceControl::reInitialize, POPCExplicitForceControl.cpp:10-22"""
    return dict(PO=0.0, Ecorr=0.0, stored=0.0, window=[], counter=50, Rc=1.0, vsum=0.0)


def popc_step(s, dt, kff, kv, fd, fs, vcl, vr, window=250, max_counter=50):
    """computePassivitySaturatedForce (enabled), POPCExplicitForceControl.cpp:35-95; returns the new Rc"""
    F_cmd = kff * fd + s["Rc"] * vcl - kv * vr
    vc2 = float(vcl @ vcl)
    p = float((fs - fd) @ vcl - F_cmd @ vr) * dt
    s["PO"] += p
    s["window"].append(p)
    if s["PO"] + s["stored"] + s["Ecorr"] > 0:
        while len(s["window"]) > window:
            if s["PO"] + s["Ecorr"] + s["stored"] > s["window"][0]:
                if s["window"][0] > 0:
                    s["PO"] -= s["window"][0]
                s["window"].pop(0)
            else:
                break
    if s["counter"] <= 0:
        s["counter"] = max_counter
        old = s["Rc"]
        if s["PO"] + s["stored"] + s["Ecorr"] < 0:
            with np.errstate(divide="ignore", invalid="ignore"):
                rc = 1 + np.float64(s["PO"] + s["stored"] + s["Ecorr"]) / np.float64(s["vsum"] * dt)
            if rc > 1:
                rc = 1.0
            if rc < 0:
                rc = 0.0
            s["Rc"] = float(rc)
        else:
            s["Rc"] = (1 + (0.1 * max_counter - 1) * s["Rc"]) / (0.1 * max_counter)
        s["Ecorr"] += (1 - old) * s["vsum"] * dt
        s["vsum"] = 0.0
    s["counter"] -= 1
    s["vsum"] += vc2
    return s["Rc"]


def forward_dynamics(model: RobotModel, q, dq, tau, g=(0.0, 0.0, -9.81), damping=0.0, eps=1e-5):
    """qdd (B,n) from the Lagrange equations, independent of any recursive Newton-Euler code:
    M qdd + Mdot dq - 1/2 grad_q(dq^T M dq) + dU/dq = tau - damping dq,  with dM/dq_k by central differences of mass_matrix.
    (test oracle for the resident integrator, csrc/saip_dynamics.hip; accuracy ~1e-9 relative)"""
    q, dq, tau = (np.asarray(x, float) for x in (q, dq, tau))
    B, n = q.shape
    M = mass_matrix(model, fk(model, q))
    dM = np.zeros((n, B, n, n))
    for k in range(n):
        e = np.zeros(n)
        e[k] = eps
        dM[k] = (mass_matrix(model, fk(model, q + e)) - mass_matrix(model, fk(model, q - e))) / (2 * eps)
    Mdot_dq = np.einsum("kbij,bk,bj->bi", dM, dq, dq)
    grad = 0.5 * np.einsum("kbij,bi,bj->bk", dM, dq, dq)
    grav = gravity_vector(model, fk(model, q), g)   # dU/dq
    rhs = tau - damping * dq - (Mdot_dq - grad) - grav
    return np.linalg.solve(M, rhs[..., None])[..., 0]


def total_energy(model: RobotModel, q, dq, g=(0.0, 0.0, -9.81)):
    """kinetic + potential energy (B,)"""
    fr = fk(model, q)
    M = mass_matrix(model, fr)
    T = 0.5 * np.einsum("bi,bij,bj->b", dq, M, dq)
    U = np.zeros(q.shape[0])
    gv = np.asarray(g, float)
    for li, l in enumerate(model.links):
        R, o = fr[li]
        c = o + R @ np.asarray(l["com"], float)
        U -= l["mass"] * (c @ gv)
    return T + U


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


def op_space_matrices(J, Minv):
    Lam = np.linalg.inv(J @ Minv @ J.T)
    Jbar = Minv @ J.T @ Lam
    N = np.eye(Minv.shape[0]) - Jbar @ J
    return Lam, Jbar, N


def orientation_error(Rd, Rc):
    """SaiModel::orientationError(desired, current) = -1/2 sum_i Rc[:,i] x Rd[:,i]"""
    e = np.zeros(3)
    for i in range(3):
        e += np.cross(Rc[:, i], Rd[:, i])
    return -0.5 * e


def bie_mass(M, thr):
    Mb = M.copy()
    for i in range(M.shape[0]):
        if Mb[i, i] < thr:
            Mb[i, i] = thr
    return Mb


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
             singularity_handling=True)   # SingularityHandler::_enforce_handling_strategy, SingularityHandler.cpp:61
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


def pinv_diag(k, tol=1e-6):
    """SaiModel::computePseudoInverse of a diagonal gain matrix (entries below tol are dropped)"""
    k = np.atleast_1d(np.asarray(k, float))
    return np.array([1.0 / x if abs(x) > tol else 0.0 for x in k])


def sigma_space(Pblock, dim, axis, rotation):
    """sigmaForce / sigmaMoment, MotionForceTask.cpp:892-925, 932-966"""
    if dim == 0:
        return np.zeros((3, 3))
    if dim == 3:
        return Pblock.copy()
    a = np.asarray(axis, float)
    a = rotation @ (a / np.linalg.norm(a))
    aa = np.outer(a, a)
    inner = aa if dim == 1 else np.eye(3) - aa
    return Pblock @ inner @ Pblock.T


# --------------------------------------------------- JointLimitAvoidanceTask (JointLimitAvoidanceTask.cpp:124-421)
JLA = dict(kv=20.0, pz1=9 * np.pi / 180, pz2=6 * np.pi / 180, vz1=0.5, vz2=0.3, ratio_pos=1.0, ratio_vel=0.05)  # .h:26-35
OFF, POS_Z1, POS_Z2, VEL_Z1, VEL_Z2 = range(5)


def jla_limit_status(model, q, dq):
    """updateLimitStatus, JointLimitAvoidanceTask.cpp:171-239 -> status, direction (+1/-1), limit value, torque limit per joint"""
    n = model.dof
    st, dr, lv, tl = [OFF] * n, [1] * n, [0.0] * n, [0.0] * n
    for i in range(n):
        lo, hi, vel, eff = model.q_lower[i], model.q_upper[i], model.vel_limit[i], model.effort[i]
        pos_valid = hi - lo > 2 * JLA["pz1"]
        vel_valid = vel > 2 * JLA["vz1"]
        if pos_valid:
            if q[i] > hi - JLA["pz1"]:
                dr[i], lv[i], tl[i], st[i] = 1, hi, eff, POS_Z1
            if q[i] > hi - JLA["pz2"]:
                st[i] = POS_Z2
            if q[i] < lo + JLA["pz1"]:
                dr[i], lv[i], tl[i], st[i] = -1, lo, eff, POS_Z1
            if q[i] < lo + JLA["pz2"]:
                st[i] = POS_Z2
        if vel_valid and (st[i] == OFF or dr[i] == -1):
            if dq[i] > vel - JLA["vz1"]:
                dr[i], lv[i], tl[i], st[i] = 1, vel, eff, VEL_Z1
            if dq[i] > vel - JLA["vz2"]:
                st[i] = VEL_Z2
        if vel_valid and (st[i] == OFF or dr[i] == 1):
            if dq[i] < -vel + JLA["vz1"]:
                dr[i], lv[i], tl[i], st[i] = -1, -vel, eff, VEL_Z1
            if dq[i] < -vel + JLA["vz2"]:
                st[i] = VEL_Z2
    return st, dr, lv, tl


def _blend(z, z1, z2, direction):
    """computeBlendingCoefficient, JointLimitAvoidanceTask.cpp:16-36"""
    if direction < 0:
        return 0.0 if z >= z1 else (1.0 if z <= z2 else (z1 - z) / (z1 - z2))
    return 0.0 if z <= z1 else (1.0 if z >= z2 else (z - z1) / (z2 - z1))


def jla_wrap(model, q, dq, Minv, tau_tasks):
    """RobotController.cpp:96-101: tau = JLA.computeTorques(tau) + N_constraints^T tau  (JLA model :124-169 with N_prec = I)"""
    n = model.dof
    st, dr, lv, tl = jla_limit_status(model, q, dq)
    act = [i for i in range(n) if st[i] != OFF]
    if not act:
        return tau_tasks.copy()
    kv, pz1, pz2, vz1, vz2, rp, rv = (JLA[k] for k in ("kv", "pz1", "pz2", "vz1", "vz2", "ratio_pos", "ratio_vel"))
    lat = []
    for i in act:
        t_i, d = tau_tasks[i], dr[i]
        clampv = lambda x: max(min(x, tl[i] * rv), -tl[i] * rv)
        if d > 0:
            if st[i] == POS_Z1:
                a = _blend(q[i], lv[i] - pz1, lv[i] - pz2, d)
                v = (1 - a) * t_i + a * (t_i - kv * dq[i])
            elif st[i] == POS_Z2:
                a = _blend(q[i], lv[i] - pz2, lv[i], d)
                v = (1 - a) * (t_i - kv * dq[i]) + a * (-tl[i] * rp - kv * dq[i])
            elif st[i] == VEL_Z1:
                a = _blend(dq[i], lv[i] - vz1, lv[i] - vz2, d)
                v = (1 - a) * t_i + a * (-kv * dq[i])
            else:
                a = _blend(dq[i], lv[i] - vz2, lv[i], d)
                v = (1 - a) * clampv(-kv * dq[i]) + a * (-a * tl[i] * rv)
        else:
            if st[i] == POS_Z1:
                a = _blend(q[i], lv[i] + pz1, lv[i] + pz2, d)
                v = a * t_i + (1 - a) * clampv(t_i - kv * dq[i])          # as written in the reference (:349-361)
            elif st[i] == POS_Z2:
                a = _blend(q[i], lv[i] + pz2, lv[i], d)
                v = (1 - a) * (t_i - kv * dq[i]) + a * (tl[i] * rp - kv * dq[i])
            elif st[i] == VEL_Z1:
                a = _blend(dq[i], lv[i] + vz1, lv[i] + vz2, d)
                v = (1 - a) * t_i + a * clampv(-kv * dq[i])
            else:
                a = _blend(dq[i], lv[i] + vz2, lv[i], d)
                v = (1 - a) * clampv(-kv * dq[i]) + a * (tl[i] * rv)
        lat.append(v)
    S = np.zeros((len(act), n))
    for r, i in enumerate(act):
        S[r, i] = 1.0
    Lam, _, N = op_space_matrices(S, Minv)      # range basis of S is the identity (rows of I: full row rank)
    return S.T @ np.array(lat) + N.T @ tau_tasks


# --------------------------------------------------- literal single-instance controller
class Singular(Exception):
    pass


# ---------------------------------------------------------------- SingularityHandler: blended type-1 / type-2 strategies
# Restated from SingularityHandler.cpp:75-368 with three documented choices where the reference is not well defined:
#  (1) classification (:252-273) perturbs q along the right singular vector, whose SIGN is an artefact of the SVD implementation;
#      here a direction is TYPE_1 when the larger of the two motions (q + 5 v, q - 5 v) exceeds the tolerance (sign-invariant);
#  (2) _type_2_torque_vector is built from _type_2_torque_ratio before that member is assigned (:48 vs :69); the intended
#      TYPE_2_TORQUE_RATIO = 1e-2 is used;
#  (3) everything else -- the torque formulas -- is invariant to the signs and to the basis inside each singular subspace.
SH_TYPE_1_TOL, SH_PERTURB, SH_BUFFER, SH_KP1, SH_KV1, SH_KV2 = 0.5, 5.0, 200, 50.0, 14.0, 5.0
SH_TYPE2_RATIO, SH_TYPE2_ANGLE = 1e-2, 5 * np.pi / 180


def sh_state_new(model):
    n = model.dof
    return dict(q_prior=0.5 * (model.q_lower + model.q_upper), c1=0, c2=0, history=[], dir=np.ones(n), ntypes=0)


def sh_blend(model, t, li, q, dq, frames1, M, Minv, N_prec, Jp, U, s, V, k, ns, st):
    """updateTaskModel in the blending / fully singular branches with the handling strategy enforced (:83-121, 146-228) + classifySingularity
    (:230-295).  Returns None when the task is fully non-singular (after updating the handler state like :237-243)."""
    n = model.dof
    if st is None:
        st = sh_state_new(model)
    if st["ntypes"] == 0 or st["c2"] > st["c1"]:                     # :232-235
        st["q_prior"] = q.copy()
    if ns == k:                                                      # no singular directions: :238-244
        st["ntypes"], st["history"], st["c1"], st["c2"] = 0, [], 0, 0
        return None
    U_s, V_s = U[:, ns:k], V[:, ns:k]
    alpha = 0.0 if ns == 0 else float(np.clip((s[ns] / s[0] - t["s_min"]) / (t["s_max"] - t["s_min"]), 0.0, 1.0))
    J_s = U_s.T @ Jp
    A_s = J_s @ Minv @ J_s.T
    Lam_s = np.linalg.pinv(A_s) if ns == 0 else np.linalg.inv(A_s)   # :97-98 / :117
    out = dict(blend=True, alpha=alpha, U_s=U_s, V_s=V_s, J_s=J_s, ns=ns)
    Mb_inv = np.linalg.inv(bie_mass(M, t["bie_threshold"]))
    dec = t["decoupling"]
    mod = lambda Jx, Lam: Lam if dec == FULL_DYNAMIC_DECOUPLING else (np.eye(Jx.shape[0]) if dec == IMPEDANCE else np.linalg.inv(Jx @ Mb_inv @ Jx.T))
    if ns == 0:                                                      # fully singular: :149-151 _N = N_prec, torques 0 (:316-317)
        out.update(N=N_prec.copy(), U_ns=None)
    else:
        U_ns = U[:, :ns]
        J_ns = U_ns.T @ Jp
        Lam_ns, _, N_ns = op_space_matrices(J_ns, Minv)
        Jpp = V_s.T @ N_ns @ N_prec                                  # :153
        Lam_js, _, N_js = op_space_matrices(Jpp, Minv)
        out.update(U_ns=U_ns, J_ns=J_ns, Lam_ns_mod=mod(J_ns, Lam_ns), Lam_s_mod=mod(J_s, Lam_s), Jpp=Jpp, Lam_js_mod=mod(Jpp, Lam_js),
                   N=N_js @ N_ns)
    # classifySingularity, :246-294
    R0, o0 = frames1[li]
    x0, Rc0 = o0 + R0 @ np.asarray(t["pos_in_link"]), R0 @ np.asarray(t["rot_in_link"])
    types = []
    for c in range(k - ns):
        motion = 0.0
        for sgn in (1.0, -1.0):
            fr = fk(model, (q + sgn * SH_PERTURB * V_s[:, c])[None])
            R1, o1 = fr[li][0][0], fr[li][1][0]
            dx = o1 + R1 @ np.asarray(t["pos_in_link"]) - x0
            dphi = orientation_error(R1 @ np.asarray(t["rot_in_link"]), Rc0)
            motion = max(motion, abs(np.concatenate([dx, dphi]) @ U_s[:, c]))
        types.append(1 if motion > SH_TYPE_1_TOL else 2)
    st["ntypes"] = len(types)
    if 1 in types:
        st["history"].append(1)
        st["c1"] += 1
    else:
        st["history"].append(2)
        st["c2"] += 1
    if len(st["history"]) > SH_BUFFER:
        if st["history"].pop(0) == 1:
            st["c1"] -= 1
        else:
            st["c2"] -= 1
    out["st"] = st
    return out


def sh_blend_torques(model, t, m, q, dq, F_um, F_f, st):
    """computeTorques in the singular branches, :310-367"""
    n = model.dof
    st = m["st"]
    if m["U_ns"] is None:
        return np.zeros(n)                                           # :316-317
    if t["decoupling"] == IMPEDANCE:                                 # :311-313
        return m["J_ns"].T @ (m["U_ns"].T @ F_um + m["U_ns"].T @ F_f)
    tau_ns = m["J_ns"].T @ (m["Lam_ns_mod"] @ m["U_ns"].T @ F_um + m["U_ns"].T @ F_f)
    V_s, Jpp = m["V_s"], m["Jpp"]
    kp1, kv1, kv2 = t.get("sh_gains", (SH_KP1, SH_KV1, SH_KV2))      # setSingularityHandlingGains, SingularityHandler.h:119
    if st["c1"] > st["c2"] or t.get("sh_all_type1", False):         # type-1 strategy: hold the entering posture, :329-336
        unit = -kp1 * (q - st["q_prior"]) - kv1 * dq
        tau_js = Jpp.T @ m["Lam_js_mod"] @ V_s.T @ unit
    else:                                                            # type-2 strategy, :337-355
        for i in range(n):
            if V_s[i, 0] != 0:
                if abs(q[i] - model.q_upper[i]) < SH_TYPE2_ANGLE:
                    st["dir"][i] = -1
                elif abs(q[i] - model.q_lower[i]) < SH_TYPE2_ANGLE:
                    st["dir"][i] = 1
        f = F_um + F_f
        fTd = (f / np.linalg.norm(f)) @ m["U_s"][:, 0]
        unit = st["dir"] * (abs(fTd) * SH_TYPE2_RATIO * model.effort)
        tau_js = Jpp.T @ V_s.T @ unit + Jpp.T @ m["Lam_js_mod"] @ V_s.T @ (-kv2 * dq)
    tau_s = m["J_s"].T @ (m["Lam_s_mod"] @ m["U_s"].T @ F_um + m["U_s"].T @ F_f)   # :358-359
    tau_s = np.where(np.isnan(tau_s), 0.0, np.clip(tau_s, -model.effort, model.effort))
    return tau_ns + m["alpha"] * tau_s + (1 - m["alpha"]) * tau_js


def controller_step_single(model, tasks, q, dq, goals, *, state=None, gravity_comp=False,
                           torque_saturation=False, joint_limit_avoidance=False, details=None, info=None):
    """One control cycle for ONE instance, literal reference path (SVDs, explicit inverses).
    goals[t]: flat vector (goal_size).  Returns tau (n,).  Raises Singular when an instance
    leaves the non-singular branch of SingularityHandler (alpha != 1)."""
    n = model.dof
    frames = fk(model, q[None])
    M = mass_matrix(model, frames)[0]
    Minv = np.linalg.inv(M)
    frames1 = [(R[0], o[0]) for R, o in frames]

    # ---- updateControllerTaskModels, RobotController.cpp:68-77
    N_prec = np.eye(n)
    tm = []
    for t in tasks:
        if t["type"] == "motion_force":
            li = model.link_index(t["link"]) if isinstance(t["link"], str) else t["link"]
            R_l, o_l = frames1[li]
            p = o_l + R_l @ np.asarray(t["pos_in_link"])
            P, Bm, pr, orr = mf_projection(t)
            k = pr + orr
            Jw = jacobian(model, frames, li, p[None])[0]
            J = P @ Jw                                              # MotionForceTask.cpp:261-263
            Jp = J @ N_prec                                         # :264
            U, s, Vt = np.linalg.svd(Jp, full_matrices=False)       # SingularityHandler.cpp:78-81
            handling = t.get("singularity_handling", True)           # _enforce_handling_strategy, :61
            strategies = handling and t.get("singularity_strategies", False)   # blended type-1 / type-2 strategies restated (see sh_blend)
            if k == 1:
                raise Singular("rank-1 motion force task unsupported (SURVEY App. C-4)")
            ns = k
            if s[0] < 1e-3:
                if handling and not strategies:
                    raise Singular("fully singular")
                ns = 0                                               # :83-98 + :316-317: the task is passed through
            else:
                for i in range(1, k):
                    if s[i] / s[0] < t["s_max"]:
                        if handling and not strategies:
                            raise Singular("blending region")       # :100-121: needs the blended strategies
                        ns = i                                       # handling disabled: _N = _N_ns, tau = tau_ns (:146-148, 319-330)
                        break
            if strategies:
                shs = None if state is None else state[len(tm)].setdefault("sh", sh_state_new(model))
                blend = sh_blend(model, t, li, q, dq, frames1, M, Minv, N_prec, Jp, U, s, Vt.T, k, ns, shs)
                if blend is not None:
                    if info is not None:
                        info["blended"] = True                       # reported as status 8
                    blend.update(J=J, Jp=Jp, N_prec=N_prec, P=P, R=R_l @ np.asarray(t["rot_in_link"]), x=p, sv=s)
                    tm.append(blend)
                    N_prec = blend["N"] @ N_prec
                    continue
            if ns < k and info is not None:
                info["truncated"] = True                             # reported as status 2
            if ns == 0:
                tm.append(dict(skip=True, N=np.eye(n), N_prec=N_prec, J=J, P=P, R=R_l @ np.asarray(t["rot_in_link"]), x=p, sv=s))
                continue
            k = ns
            U_ns = U[:, :k]                                          # :128 / :108
            J_ns = U_ns.T @ Jp
            Lam, Jbar, N = op_space_matrices(J_ns, Minv)             # :130-134
            if t["decoupling"] == FULL_DYNAMIC_DECOUPLING:
                Lam_mod = Lam
            elif t["decoupling"] == IMPEDANCE:
                Lam_mod = np.eye(k)
            else:
                Mb = bie_mass(M, t["bie_threshold"])                 # :175-193
                Lam_mod = np.linalg.inv(J_ns @ np.linalg.inv(Mb) @ J_ns.T)
            tm.append(dict(J=J, Jp=Jp, U_ns=U_ns, J_ns=J_ns, Lam=Lam, Lam_mod=Lam_mod, N=N,
                           N_prec=N_prec, P=P, R=R_l @ np.asarray(t["rot_in_link"]), x=p, sv=s))
            N_prec = N @ N_prec                                      # MotionForceTask.h:207-209
        else:
            S = joint_selection(t, n)
            Jp = S @ N_prec                                          # JointTask.cpp:231
            U = matrix_range_basis(Jp)                               # :233
            if np.linalg.norm(U) == 0:
                tm.append(dict(empty=True, N=np.eye(n), N_prec=N_prec))
                N_prec = N_prec.copy()
                continue
            Jh = U.T @ Jp
            Lam, _, N = op_space_matrices(Jh, Minv)                  # :241-245
            if t["decoupling"] == FULL_DYNAMIC_DECOUPLING:
                Lam_mod = Lam
            elif t["decoupling"] == IMPEDANCE:
                Lam_mod = np.eye(U.shape[1])
            else:
                Mb = bie_mass(M, t["bie_threshold"])
                Lam_mod = np.linalg.inv(U.T @ Jp @ np.linalg.inv(Mb) @ Jp.T @ U)  # :253-267
            tm.append(dict(empty=False, S=S, Jp=Jp, U=U, Lam=Lam, Lam_mod=Lam_mod, N=N, N_prec=N_prec))
            N_prec = N @ N_prec

    # ---- computeControlTorques, RobotController.cpp:79-118
    tau = np.zeros(n)
    for ti, (t, m) in enumerate(zip(tasks, tm)):
        g = np.asarray(goals[ti], float)
        st = None if state is None else state[ti]
        if t["type"] == "motion_force":
            xd, Rd = g[0:3], g[3:12].reshape(3, 3)
            vd, wd, ad, ald = g[12:15], g[15:18], g[18:21], g[21:24]
            J, P = m["J"], m["P"]
            v = J[:3] @ dq                                           # MotionForceTask.cpp:293-298
            w = J[3:] @ dq
            Rcur = m["R"]
            rot = Rcur if t["param_in_compliant_frame"] else np.eye(3)   # :894-897, 934-937, 755-769
            sig_f = sigma_space(P[:3, :3], t["force_dim"], t["force_axis"], rot)
            sig_m = sigma_space(P[3:, 3:], t["moment_dim"], t["moment_axis"], rot)
            sig_p = P[:3, :3] @ (np.eye(3) - sig_f) @ P[:3, :3].T    # :927-930
            sig_o = P[3:, 3:] @ (np.eye(3) - sig_m) @ P[3:, 3:].T    # :968-971
            goal_force = rot @ (g[24:27] if g.size >= 30 else np.zeros(3))
            goal_moment = rot @ (g[27:30] if g.size >= 30 else np.zeros(3))
            kp, kv, ki = (np.diag(np.broadcast_to(t[k_], 3)) for k_ in ("kp_pos", "kv_pos", "ki_pos"))
            kpo, kvo, kio = (np.diag(np.broadcast_to(t[k_], 3)) for k_ in ("kp_ori", "kv_ori", "ki_ori"))
            cl_f, cl_m = bool(t.get("cl_force")), bool(t.get("cl_moment"))
            if cl_f or cl_m:                                         # updateSensedForceAndMoment, :805-828
                Rcs, tcs = np.asarray(t.get("Rcs", np.eye(3)), float), np.asarray(t.get("tcs", np.zeros(3)), float)
                f_c = Rcs @ g[30:33]
                m_c = np.cross(tcs, f_c) + Rcs @ g[33:36]
                f_sw, m_sw = Rcur @ f_c, Rcur @ m_c
            ifo = np.zeros(3) if st is None else st.setdefault("int_force", np.zeros(3))
            imo = np.zeros(3) if st is None else st.setdefault("int_moment", np.zeros(3))
            if cl_f:                                                 # :327-349, passivity controller off (POPC...cpp:31-33)
                ifo = ifo + sig_f @ (f_sw - goal_force) * t["dt"]
                fb = sig_f @ (-t.get("kp_force", 0.7) * (f_sw - goal_force) - t.get("ki_force", 1.3) * ifo)
                if np.linalg.norm(fb) > t.get("max_force_fb", 20.0):
                    fb = fb * t.get("max_force_fb", 20.0) / np.linalg.norm(fb)
                rc = 1.0
                if t.get("passivity") and st is not None:
                    rc = popc_step(st.setdefault("popc", popc_new()), t["dt"], t.get("kff_force", 0.95), t["kv_force"],
                                   sig_f @ goal_force, sig_f @ f_sw, sig_f @ fb, sig_f @ v)   # :344-348
                F_force = rc * (sig_f @ fb) - t["kv_force"] * (sig_f @ v)
            else:
                F_force = sig_f @ (-t["kv_force"] * v)               # open loop, :350-354
            if cl_m:                                                 # :357-378
                imo = imo + sig_m @ (m_sw - goal_moment) * t["dt"]
                fbm = sig_m @ (-t.get("kp_moment", 0.7) * (m_sw - goal_moment) - t.get("ki_moment", 1.3) * imo)
                if np.linalg.norm(fbm) > t.get("max_moment_fb", 10.0):
                    fbm = fbm * t.get("max_moment_fb", 10.0) / np.linalg.norm(fbm)
                F_moment = sig_m @ (fbm - t["kv_moment"] * w)
            else:
                F_moment = sig_m @ (-t["kv_moment"] * w)             # :379-383
            if st is not None:
                st["int_force"], st["int_moment"] = ifo, imo
            kff_f = t.get("kff_force", 0.95) if cl_f else 1.0        # :484-487: BOTH feed-forward terms are scaled by the FORCE flag
            kff_m = t.get("kff_moment", 0.95) if cl_f else 1.0
            ipos = np.zeros(3) if st is None else st["int_pos"]
            iori = np.zeros(3) if st is None else st["int_ori"]
            ipos = ipos + sig_p @ (m["x"] - xd) * t["dt"]             # :411-413
            if t["vel_sat"]:                                         # :416-430
                kvi = np.diag(pinv_diag(np.diag(kv)))
                vdes = -kp @ kvi @ sig_p @ (m["x"] - xd) - ki @ kvi @ ipos
                if np.linalg.norm(vdes) > t["lin_sat"]:
                    vdes = vdes * t["lin_sat"] / np.linalg.norm(vdes)
                F_p = sig_p @ (ad - kv @ (v - vdes))
            else:
                F_p = sig_p @ (ad - kp @ (m["x"] - xd) - kv @ (v - vd) - ki @ ipos)   # :431-437
            dphi = sig_o @ orientation_error(Rd, Rcur)               # :441-443
            iori = iori + dphi * t["dt"]                             # :446
            if t["vel_sat"]:                                         # :449-462
                kvi = np.diag(pinv_diag(np.diag(kvo)))
                wdes = -kpo @ kvi @ dphi - kio @ kvi @ iori
                if np.linalg.norm(wdes) > t["ang_sat"]:
                    wdes = wdes * t["ang_sat"] / np.linalg.norm(wdes)
                F_o = sig_o @ (ald - kvo @ (w - wdes))
            else:
                F_o = sig_o @ (ald - kpo @ dphi - kvo @ (w - wd) - kio @ iori)        # :463-468
            if st is not None:
                st["int_pos"], st["int_ori"] = ipos, iori
            F_um = np.concatenate([F_p, F_o])                        # :475-478
            F_f = np.concatenate([F_force + kff_f * (sig_f @ goal_force), F_moment + kff_m * (sig_m @ goal_moment)])   # :471-487
            if m.get("blend"):
                tt = sh_blend_torques(model, t, m, q, dq, F_um, F_f, None if st is None else st.get("sh"))
            elif m.get("skip"):
                tt = np.zeros(n)
            else:
                tt = m["J_ns"].T @ (m["Lam_mod"] @ m["U_ns"].T @ F_um + m["U_ns"].T @ F_f)  # SH.cpp:307-309
            # computeTorques(tau_prec): _Lambda is never written => compensation == 0 (App. C-1)
            tau = tau + tt
            if details is not None:
                details.append(dict(F_um=F_um, tau=tt, **m))
        else:
            if m["empty"]:
                if details is not None:
                    details.append(dict(tau=np.zeros(n), **m))
                continue
            S, U, Jp = m["S"], m["U"], m["Jp"]
            mm = S.shape[0]
            qd, dqd, ddqd = g[0:mm], g[mm:2 * mm], g[2 * mm:3 * mm]
            kp, kv, ki = (np.diag(np.broadcast_to(t[k_], mm)) for k_ in ("kp", "kv", "ki"))
            e = S @ q - qd
            ie = np.zeros(mm) if st is None else st["int_err"]
            ie = ie + e * t["dt"]                                    # JointTask.cpp:323-324
            if st is not None:
                st["int_err"] = ie
            if t["vel_sat"]:                                         # :327-341 (clamp applied to the task dofs)
                kvi = np.diag(pinv_diag(np.diag(kv)))
                sat = np.broadcast_to(np.asarray(t["sat"], float), (mm,))
                vdes = np.clip(-kp @ kvi @ e - ki @ kvi @ ie, -sat, sat)
                f = -kv @ (S @ dq - vdes)
            else:
                f = -kp @ e - kv @ (S @ dq - dqd) - ki @ ie          # :342-345
            tr = m["Lam"] @ U.T @ ddqd + m["Lam_mod"] @ U.T @ f      # :348-351
            tt = Jp.T @ U @ tr                                       # :354-355
            comp = Jp.T @ U @ m["Lam"] @ U.T @ S @ Minv @ tau        # :287-291
            tt = tt - comp
            tau = tau + tt
            if details is not None:
                details.append(dict(tau=tt, **m))
    if torque_saturation:
        tau = np.clip(tau, -model.effort, model.effort)              # RobotController.cpp:86-94
    if joint_limit_avoidance:                                        # :96-112
        tau = jla_wrap(model, q, dq, Minv, tau)
        if torque_saturation:
            tau = np.clip(tau, -model.effort, model.effort)
    if gravity_comp:
        tau = tau + gravity_vector(model, frames)[0]                 # :114-116
    return tau


def controller_step(model, tasks, q, dq, goals, **kw):
    """Batched wrapper. q,dq (B,n); goals[t] (B, goal_size). Returns tau (B,n), status (B,) uint8."""
    B = q.shape[0]
    tau = np.full((B, model.dof), np.nan)
    status = np.zeros(B, np.uint8)
    for b in range(B):
        try:
            info = {}
            tau[b] = controller_step_single(model, tasks, q[b], dq[b], [g[b] for g in goals], info=info, **kw)
            if info.get("truncated"):
                status[b] = 2   # singularity handling disabled: a task was reduced to its non-singular subspace
            if info.get("blended"):
                status[b] |= 8  # a blended singularity strategy is active
        except Singular:
            status[b] = 1
    return tau, status


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

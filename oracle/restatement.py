"""Independent NumPy restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY (like everything under oracle/):
imported by tests/ and tests/golden/make_golden.py -- the second restatement the C oracle (oracle/saip_oracle.c) is
cross-checked against and the generator of the committed golden vectors.  Never imported by the
product package, bench.py's timed path or smoke().

The restatement follows, literally (SVD path, explicit inverses):
  /root/reference/src/RobotController.cpp:68-118
  /root/reference/src/tasks/MotionForceTask.cpp:247-509, 892-971
  /root/reference/src/tasks/SingularityHandler.cpp:75-228, 297-368
  /root/reference/src/tasks/JointTask.cpp:218-356
  /root/reference/src/tasks/JointLimitAvoidanceTask.cpp:124-421
  /root/reference/src/helper_modules/POPCExplicitForceControl.cpp:29-96
sai-model semantics (external, not in /root/reference) are restated from their algebraic definitions, SURVEY.md Appendix B.
Parity unpinned: the reference ships no tests or golden vectors for this path.
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workloads import (BOUNDED_INERTIA_ESTIMATES, FULL_DYNAMIC_DECOUPLING, IMPEDANCE, RobotModel, fk, jacobian,  # noqa: E402,F401
                       joint_selection, mass_matrix, matrix_range_basis, mf_projection)

# --------------------------------------------------- JointLimitAvoidanceTask (JointLimitAvoidanceTask.cpp:124-421)
JLA = dict(kv=20.0, pz1=9 * np.pi / 180, pz2=6 * np.pi / 180, vz1=0.5, vz2=0.3, ratio_pos=1.0, ratio_vel=0.05)  # .h:26-35
OFF, POS_Z1, POS_Z2, VEL_Z1, VEL_Z2 = range(5)


# ---------------------------------------------------------------- SingularityHandler: blended type-1 / type-2 strategies
# Restated from SingularityHandler.cpp:75-368 with three documented choices where the reference is not well defined:
#  (1) classification (:252-273) perturbs q along the right singular vector, whose SIGN is an artefact of the SVD implementation;
#      here a direction is TYPE_1 when the larger of the two motions (q + 5 v, q - 5 v) exceeds the tolerance (sign-invariant);
#  (2) _type_2_torque_vector is built from _type_2_torque_ratio before that member is assigned (:48 vs :69); the intended
#      TYPE_2_TORQUE_RATIO = 1e-2 is used;
#  (3) everything else -- the torque formulas -- is invariant to the signs and to the basis inside each singular subspace.
SH_TYPE_1_TOL, SH_PERTURB, SH_BUFFER, SH_KP1, SH_KV1, SH_KV2 = 0.5, 5.0, 200, 50.0, 14.0, 5.0
SH_TYPE2_RATIO, SH_TYPE2_ANGLE = 1e-2, 5 * np.pi / 180


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
            strategies = handling and t.get("singularity_strategies", True)   # blended type-1 / type-2 strategies restated (see sh_blend)
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
                        info["truncated" if ns == 0 else "blended"] = True   # status 8; a fully singular task is passed through: status 2
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

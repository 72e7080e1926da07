"""ctypes binding of the CPU oracle (oracle/saip_oracle.c).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from
the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_JT = {"fixed": 0, "revolute": 1, "prismatic": 2}


def build(force: bool = False) -> str:
    so = os.path.join(_DIR, "libsaip_oracle.so")
    src = os.path.join(_DIR, "saip_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-B", "libsaip_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, ip] + [dp] * 10
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_dof.argtypes = [C.c_void_p]
        L.orc_add_motion_force_task.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, C.c_int, dp, C.c_int]
        L.orc_add_joint_task.argtypes = [C.c_void_p, dp, C.c_int]
        L.orc_set_mf_gains.argtypes = [C.c_void_p, C.c_int] + [dp] * 6
        L.orc_set_joint_gains.argtypes = [C.c_void_p, C.c_int] + [dp] * 3
        L.orc_set_decoupling.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        L.orc_set_singularity_bounds.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.orc_set_timestep.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_set_singularity_handling.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_set_singularity_strategies.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_set_singularity_gains.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        L.orc_set_all_singularities_type1.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_set_type1_posture.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int]
        L.orc_set_options.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_goal_size.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_mf_force_spaces.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int, C.c_double, C.c_double]
        L.orc_set_closed_loop_force_control.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp, dp, dp]
        L.orc_set_passivity.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_set_velocity_saturation.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, dp]
        L.orc_reset_state.argtypes = [C.c_void_p]
        L.orc_set_lean.argtypes = [C.c_void_p, C.c_int]
        L.orc_step_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, C.POINTER(dp), dp, C.POINTER(C.c_ubyte), C.c_int]
        L.orc_step_details.argtypes = [C.c_void_p, dp, dp, C.POINTER(dp), dp, dp, dp, dp, dp, dp]
        L.orc_task_step.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp]
        _LIB = L
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _vec3(v):
    return np.broadcast_to(np.asarray(v, float), (3,)).copy()


class Oracle:
    """The oracle for one robot model + ordered task list (specs from workloads.py)."""

    def __init__(self, model, tasks, *, gravity_comp=False, torque_saturation=False, joint_limit_avoidance=False):
        L = lib()
        links = model.links
        nl = len(links)
        jt = (C.c_int * nl)(*[_JT[l["joint_type"]] for l in links])
        arrs = [np.array([l[k] for l in links], float).reshape(nl, -1) for k in
                ("origin_xyz", "origin_rpy", "axis")]
        mass = np.array([l["mass"] for l in links], float)
        com = np.array([l["com"] for l in links], float)
        ine = np.array([l["inertia"] for l in links], float)
        lim = [np.array([l[k] for l in links], float) for k in ("q_lower", "q_upper", "velocity_limit", "effort_limit")]
        keep = [_d(a) for a in arrs + [mass, com, ine] + lim]
        self._h = L.orc_create(nl, jt, *[k[1] for k in keep])
        if not self._h:
            raise ValueError("orc_create failed")
        self.model, self.tasks, self.n = model, tasks, model.dof
        assert L.orc_dof(self._h) == self.n
        self.goal_sizes = []
        for t in tasks:
            if t["type"] == "motion_force":
                li = model.link_index(t["link"]) if isinstance(t["link"], str) else int(t["link"])
                _, pp = _d(t["pos_in_link"])
                _, rp = _d(np.asarray(t["rot_in_link"], float).reshape(9))
                if t["dirs_trans"] is None and t["dirs_rot"] is None:
                    tid = L.orc_add_motion_force_task(self._h, li, pp, rp, None, -1, None, -1)
                else:
                    dt_ = np.asarray(t["dirs_trans"] or np.zeros((0, 3)), float).reshape(-1, 3)
                    dr_ = np.asarray(t["dirs_rot"] or np.zeros((0, 3)), float).reshape(-1, 3)
                    a1, p1 = _d(dt_)
                    a2, p2 = _d(dr_)
                    tid = L.orc_add_motion_force_task(self._h, li, pp, rp, p1, len(dt_), p2, len(dr_))
                if tid < 0:
                    raise ValueError("invalid motion force task")
                g = [_d(_vec3(t[k])) for k in ("kp_pos", "kv_pos", "ki_pos", "kp_ori", "kv_ori", "ki_ori")]
                L.orc_set_mf_gains(self._h, tid, *[x[1] for x in g])
                L.orc_set_singularity_bounds(self._h, tid, t["s_min"], t["s_max"])
                L.orc_set_singularity_handling(self._h, tid, int(bool(t.get("singularity_handling", True))))
                L.orc_set_singularity_strategies(self._h, tid, int(bool(t.get("singularity_strategies", True))))
                if "sh_gains" in t:
                    L.orc_set_singularity_gains(self._h, tid, *map(float, t["sh_gains"]))
                L.orc_set_all_singularities_type1(self._h, tid, int(bool(t.get("sh_all_type1", False))))
                fa, ma = _d(_vec3(t.get("force_axis", [0, 0, 1]))), _d(_vec3(t.get("moment_axis", [0, 0, 1])))
                L.orc_set_mf_force_spaces(self._h, tid, int(t.get("force_dim", 0)), fa[1], int(t.get("moment_dim", 0)), ma[1],
                                          int(bool(t.get("param_in_compliant_frame", False))), float(t.get("kv_force", 10.0)), float(t.get("kv_moment", 10.0)))
                L.orc_set_velocity_saturation(self._h, tid, int(bool(t.get("vel_sat", False))), float(t.get("lin_sat", 0.3)),
                                              float(t.get("ang_sat", np.pi / 3)), None)
                g8 = _d([t.get("kp_force", 0.7), t.get("ki_force", 1.3), t.get("kp_moment", 0.7), t.get("ki_moment", 1.3),
                         t.get("kff_force", 0.95), t.get("kff_moment", 0.95), t.get("max_force_fb", 20.0), t.get("max_moment_fb", 10.0)])
                rcs, tcs = _d(np.asarray(t.get("Rcs", np.eye(3)), float).reshape(9)), _d(np.asarray(t.get("tcs", np.zeros(3)), float))
                L.orc_set_passivity(self._h, tid, int(bool(t.get("passivity"))))
                L.orc_set_closed_loop_force_control(self._h, tid, int(bool(t.get("cl_force"))), int(bool(t.get("cl_moment"))), g8[1], rcs[1], tcs[1])
            else:
                import workloads as W
                S = W.joint_selection(t, self.n)
                if t["S"] is None:
                    tid = L.orc_add_joint_task(self._h, None, 0)
                else:
                    a, p = _d(S)
                    tid = L.orc_add_joint_task(self._h, p, S.shape[0])
                if tid < 0:
                    raise ValueError("invalid joint task")
                m = S.shape[0]
                g = [_d(np.broadcast_to(np.asarray(t[k], float), (m,)).copy()) for k in ("kp", "kv", "ki")]
                L.orc_set_joint_gains(self._h, tid, *[x[1] for x in g])
                sat = _d(np.broadcast_to(np.asarray(t.get("sat", np.pi / 3), float), (m,)).copy())
                L.orc_set_velocity_saturation(self._h, tid, int(bool(t.get("vel_sat", False))), 0.0, 0.0, sat[1])
            L.orc_set_decoupling(self._h, tid, int(t["decoupling"]), float(t["bie_threshold"]))
            L.orc_set_timestep(self._h, tid, float(t["dt"]))
            self.goal_sizes.append(L.orc_goal_size(self._h, tid))
        if L.orc_set_options(self._h, int(gravity_comp), int(joint_limit_avoidance), int(torque_saturation)) != 0:
            raise ValueError("unsupported option")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def reset_state(self):
        lib().orc_reset_state(self._h)

    def set_lean(self, lean: bool = True):
        """Lean CPU variant: the closed forms of App. A-4 / A-5 behind certificates, no duplicate Jacobians, no unused JLA model;
        the literal path still runs for every instance a certificate does not cover.  Same results to rounding (tests/test_oracle_lean.py)."""
        lib().orc_set_lean(self._h, int(bool(lean)))

    def set_type1_posture(self, task, q_des):
        """setType1Posture (SingularityHandler.h:140); q_des (n,) or (B,n)"""
        q = np.ascontiguousarray(q_des, dtype=np.float64)
        lib().orc_set_type1_posture(self._h, task, q.ctypes.data_as(C.POINTER(C.c_double)), int(q.ndim == 2))

    def step(self, q, dq, goals, nthreads: int = 1):
        """q,dq (B,n); goals[t] (B,goal_size). Returns tau (B,n), status (B,)."""
        B = q.shape[0]
        qa, qp = _d(np.asarray(q).T)
        dqa, dqp = _d(np.asarray(dq).T)
        gk = [_d(np.asarray(g).T) for g in goals]
        gp = (C.POINTER(C.c_double) * len(gk))(*[g[1] for g in gk])
        tau = np.empty((self.n, B))
        st = np.zeros(B, np.uint8)
        lib().orc_step_batch(self._h, B, qp, dqp, gp, tau.ctypes.data_as(C.POINTER(C.c_double)),
                             st.ctypes.data_as(C.POINTER(C.c_ubyte)), nthreads)
        return tau.T.copy(), st

    def step_soa(self, q_soa, dq_soa, goals_soa, nthreads: int = 1):
        """SoA variant without transposes (for timing): q_soa (n,B), goals_soa[t] (gs,B)."""
        B = q_soa.shape[1]
        gp = (C.POINTER(C.c_double) * len(goals_soa))(*[g.ctypes.data_as(C.POINTER(C.c_double)) for g in goals_soa])
        tau = np.empty((self.n, B))
        st = np.zeros(B, np.uint8)
        lib().orc_step_batch(self._h, B, q_soa.ctypes.data_as(C.POINTER(C.c_double)),
                             dq_soa.ctypes.data_as(C.POINTER(C.c_double)), gp,
                             tau.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_ubyte)), nthreads)
        return tau, st

    def task_step(self, task, q, dq, goal, N_prec=None, tau_prec=None):
        """The per-task interface (TemplateTask.h:43-60) for a batch, one instance at a time: updateTaskModel(N_prec) then
        computeTorques() / computeTorques(tau_prec).  q, dq (B,n); goal (B,goal_size); N_prec (B,n,n) or None (identity); tau_prec (B,n)
        or None.  Returns tau (B,n), N (B,n,n), N N_prec (B,n,n), status (B,)."""
        B, n = q.shape[0], self.n
        tau, N, Nt, st = np.empty((B, n)), np.empty((B, n, n)), np.empty((B, n, n)), np.zeros(B, np.int32)
        P = C.POINTER(C.c_double)
        for b in range(B):
            keep = [_d(q[b]), _d(dq[b]), _d(goal[b])]
            npk = _d(N_prec[b]) if N_prec is not None else (None, None)
            tpk = _d(tau_prec[b]) if tau_prec is not None else (None, None)
            t1, n1, n2 = np.empty(n), np.empty((n, n)), np.empty((n, n))
            st[b] = lib().orc_task_step(self._h, task, keep[0][1], keep[1][1], npk[1], tpk[1], keep[2][1], t1.ctypes.data_as(P),
                                        n1.ctypes.data_as(P), n2.ctypes.data_as(P))
            tau[b], N[b], Nt[b] = t1, n1, n2
        return tau, N, Nt, st

    def details(self, q, dq, goals):
        """single instance: dict(tau, N[t], Lambda[t] (MAXNxMAXN buffers), sv[t], M, task_tau[t], status)"""
        n, nt = self.n, len(self.tasks)
        qa, qp = _d(q)
        dqa, dqp = _d(dq)
        gk = [_d(g) for g in goals]
        gp = (C.POINTER(C.c_double) * nt)(*[g[1] for g in gk])
        tau = np.empty(n)
        N = np.zeros((nt, n, n))
        Lam = np.zeros((nt, 32 * 32))
        sv = np.zeros((nt, 6))
        M = np.zeros((n, n))
        tt = np.zeros((nt, n))
        P = C.POINTER(C.c_double)
        st = lib().orc_step_details(self._h, qp, dqp, gp, tau.ctypes.data_as(P), N.ctypes.data_as(P), Lam.ctypes.data_as(P),
                                    sv.ctypes.data_as(P), M.ctypes.data_as(P), tt.ctypes.data_as(P))
        return dict(tau=tau, N=N, Lambda=Lam, sv=sv, M=M, task_tau=tt, status=st)

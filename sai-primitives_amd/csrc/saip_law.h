// Task-space control laws shared by the cycle kernels (per-instance scalar code: one lane, or one thread of a workgroup).
// MotionForceTask::computeTorques, /root/reference/src/tasks/MotionForceTask.cpp:286-509 with OTG off and open-loop force
// control: sigma projectors (:892-971), open-loop force/moment damping + feed-forward (:350-354, :379-383, :471-487),
// position / orientation PID with optional velocity saturation (:411-468).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace saip {

#define LAW_UNR _Pragma("unroll")

__device__ __forceinline__ void law_mat3_vec(const double* A, const double* v, double* o) {
	o[0] = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
	o[1] = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
	o[2] = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
}

// sigmaForce / sigmaMoment for space dimension `dim` (0..3) and unit axis `axis`, expressed through `rot` (:892-925, :932-966):
// out = Pb * inner * Pb^T, inner = a a^T (dim 1) or I - a a^T (dim 2), a = rot * axis
template <typename CP>
__device__ __forceinline__ void law_sigma_space(CP Pb, const int dim, CP axis, const double* rot, double* out) {
	if (dim == 0) {
		LAW_UNR for (int i = 0; i < 9; i++) out[i] = 0.0;
		return;
	}
	if (dim == 3) {
		LAW_UNR for (int i = 0; i < 9; i++) out[i] = Pb[i];
		return;
	}
	const double ax[3] = {axis[0], axis[1], axis[2]};
	double a[3], in[9], T[9];
	law_mat3_vec(rot, ax, a);
	LAW_UNR for (int i = 0; i < 3; i++)
		LAW_UNR for (int j = 0; j < 3; j++) in[3 * i + j] = (dim == 1) ? a[i] * a[j] : ((i == j) ? 1.0 : 0.0) - a[i] * a[j];
	LAW_UNR for (int i = 0; i < 3; i++)
		LAW_UNR for (int j = 0; j < 3; j++) T[3 * i + j] = Pb[3 * i] * in[j] + Pb[3 * i + 1] * in[3 + j] + Pb[3 * i + 2] * in[6 + j];
	LAW_UNR for (int i = 0; i < 3; i++)
		LAW_UNR for (int j = 0; j < 3; j++) out[3 * i + j] = T[3 * i] * Pb[3 * j] + T[3 * i + 1] * Pb[3 * j + 1] + T[3 * i + 2] * Pb[3 * j + 2];
}

// General motion-force law.  tk: task constants (any address space); vw = Jw dq (unprojected world twist of the control point);
// xp, Rc: control point / control frame; G: goal block (36: x3 R9 v3 w3 a3 alpha3 f3 m3 + sensed force 3, sensed moment 3 in the
// sensor frame); ip, io: position / orientation integrators (in/out); ifm: force (0..2) and moment (3..5) integrators (in/out).
// Outputs F_um (unit-mass force, acts through Lambda_mod) and F_f (force-related terms, act directly).
// POPCExplicitForceControl::computePassivitySaturatedForce (POPCExplicitForceControl.cpp:29-96): windowed passivity observer
// (window 250 cycles) + passivity controller updated every 50 cycles; returns the scaling Rc of the feedback term.
// st = this instance's column of the task's POPC state ([7 + cap][ld]): PO, E_correction, Rc, sum |vcl|^2, counter, ring head, ring
// size, then the ring of per-cycle power samples (the reference's std::queue is unbounded; *overflow is set when cap is exceeded
// and the oldest sample is dropped).
__device__ inline double law_popc(double* st, int ld, int cap, double dt, double kff, double kv, const double* fd, const double* fs,
								  const double* vcl, const double* vr, bool* overflow) {
	const int PO_WINDOW = 250, PO_MAX_COUNTER = 50;
	double PO = st[0], Ecorr = st[(size_t)1 * ld], Rc = st[(size_t)2 * ld], vsum = st[(size_t)3 * ld];
	int counter = (int)st[(size_t)4 * ld], head = (int)st[(size_t)5 * ld], size = (int)st[(size_t)6 * ld];
	double* ring = st + (size_t)7 * ld;
	const double stored = 0.0;  // _stored_energy_PO is never written after reInitialize
	double Fcmd[3], vc2 = 0.0, p = 0.0;
	for (int i = 0; i < 3; i++) {
		Fcmd[i] = kff * fd[i] + Rc * vcl[i] - kv * vr[i];
		vc2 += vcl[i] * vcl[i];
		p += (fs[i] - fd[i]) * vcl[i] - Fcmd[i] * vr[i];
	}
	p *= dt;
	PO += p;
	if (size == cap) {  // window storage exhausted: drop the oldest sample without touching PO
		head = (head + 1) % cap;
		size--;
		*overflow = true;
	}
	ring[(size_t)((head + size) % cap) * ld] = p;
	size++;
	if (PO + stored + Ecorr > 0) {
		while (size > PO_WINDOW) {
			const double front = ring[(size_t)head * ld];
			if (PO + Ecorr + stored > front) {
				if (front > 0) PO -= front;
				head = (head + 1) % cap;
				size--;
			} else {
				break;
			}
		}
	}
	if (counter <= 0) {
		counter = PO_MAX_COUNTER;
		const double old_Rc = Rc;
		if (PO + stored + Ecorr < 0) {  // activity detected
			Rc = 1 + (PO + stored + Ecorr) / (vsum * dt);
			if (Rc > 1) Rc = 1;
			if (Rc < 0) Rc = 0;
		} else {
			Rc = (1 + (0.1 * PO_MAX_COUNTER - 1) * Rc) / (double)(0.1 * PO_MAX_COUNTER);
		}
		Ecorr += (1 - old_Rc) * vsum * dt;
		vsum = 0;
	}
	counter--;
	vsum += vc2;
	st[0] = PO;
	st[(size_t)1 * ld] = Ecorr;
	st[(size_t)2 * ld] = Rc;
	st[(size_t)3 * ld] = vsum;
	st[(size_t)4 * ld] = (double)counter;
	st[(size_t)5 * ld] = (double)head;
	st[(size_t)6 * ld] = (double)size;
	return Rc;
}

// CL = false compiles the closed-loop force / moment branch out (G then only needs 30 entries and ifm may be null).
// The goal block is taken by array reference so that its size is checked at compile time: the closed-loop branch reads entries 30..35
// (sensed force / moment).  A caller that handed in a 30-entry private array with CL = true made those reads run past its stack frame --
// the compiler may hoist such loads above the branch, so even instances that never take it read beyond the wavefront's scratch
// segment: the "memory aperture violation" of round 1.
template <bool CL, typename TK, int NG>
__device__ __forceinline__ void law_motion_force_general(TK tk, const double* vw, const double* xp, const double* Rc, const double (&G)[NG],
														 double* ip, double* io, double* ifm, double* Fum, double* Ff, double* popc_state = nullptr, int ld = 0,
														 bool* popc_overflow = nullptr) {
	static_assert(NG >= 30, "goal block: x3 R9 v3 w3 a3 alpha3 f3 m3");
	static_assert(!CL || NG >= 36, "closed-loop force / moment control reads the sensed force and moment: goal entries 30..35");
	const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	const double* rot = tk->param_in_compliant_frame ? Rc : I3;
	double sig_f[9], sig_m[9], sig_p[9], sig_o[9], T[9], ImS[9];
	law_sigma_space(tk->Ppos, tk->force_dim, tk->force_axis, rot, sig_f);
	law_sigma_space(tk->Pori, tk->moment_dim, tk->moment_axis, rot, sig_m);
	LAW_UNR for (int pass = 0; pass < 2; pass++) {  // sigmaPosition = Ppos (I - sigmaForce) Ppos^T, same for orientation (:927-930, :968-971)
		const double* sg = pass ? sig_m : sig_f;
		double* so = pass ? sig_o : sig_p;
		LAW_UNR for (int i = 0; i < 9; i++) ImS[i] = ((i % 4) == 0 ? 1.0 : 0.0) - sg[i];
		LAW_UNR for (int i = 0; i < 3; i++)
			LAW_UNR for (int j = 0; j < 3; j++) {
				const double p0 = pass ? tk->Pori[3 * i] : tk->Ppos[3 * i], p1 = pass ? tk->Pori[3 * i + 1] : tk->Ppos[3 * i + 1],
							 p2 = pass ? tk->Pori[3 * i + 2] : tk->Ppos[3 * i + 2];
				T[3 * i + j] = p0 * ImS[j] + p1 * ImS[3 + j] + p2 * ImS[6 + j];
			}
		LAW_UNR for (int i = 0; i < 3; i++)
			LAW_UNR for (int j = 0; j < 3; j++) {
				const double p0 = pass ? tk->Pori[3 * j] : tk->Ppos[3 * j], p1 = pass ? tk->Pori[3 * j + 1] : tk->Ppos[3 * j + 1],
							 p2 = pass ? tk->Pori[3 * j + 2] : tk->Ppos[3 * j + 2];
				so[3 * i + j] = T[3 * i] * p0 + T[3 * i + 1] * p1 + T[3 * i + 2] * p2;
			}
	}
	double v[3], w[3], tmp[3], e3[3], gf[3], gm[3];
	LAW_UNR for (int i = 0; i < 3; i++) {
		v[i] = tk->Ppos[3 * i] * vw[0] + tk->Ppos[3 * i + 1] * vw[1] + tk->Ppos[3 * i + 2] * vw[2];  // J = P Jw (:293-298)
		w[i] = tk->Pori[3 * i] * vw[3] + tk->Pori[3 * i + 1] * vw[4] + tk->Pori[3 * i + 2] * vw[5];
	}
	law_mat3_vec(rot, G + 24, gf);  // getGoalForce / getGoalMoment (:755-769)
	law_mat3_vec(rot, G + 27, gm);
	if (CL && (tk->cl_force || tk->cl_moment)) {
		// updateSensedForceAndMoment (:805-828): sensor frame -> control frame -> world
		double fc[3], mc[3], fsw[3], msw[3], t2[3], Rl[9];
		LAW_UNR for (int i = 0; i < 9; i++) Rl[i] = tk->Rcs[i];
		law_mat3_vec(Rl, G + 30, fc);
		law_mat3_vec(Rl, G + 33, mc);
		mc[0] += tk->tcs[1] * fc[2] - tk->tcs[2] * fc[1];
		mc[1] += tk->tcs[2] * fc[0] - tk->tcs[0] * fc[2];
		mc[2] += tk->tcs[0] * fc[1] - tk->tcs[1] * fc[0];
		law_mat3_vec(Rc, fc, fsw);
		law_mat3_vec(Rc, mc, msw);
		const double kff_f = tk->cl_force ? tk->kff_force : 1.0, kff_m = tk->cl_force ? tk->kff_moment : 1.0;  // :484-487: both scaled by the FORCE flag
		if (tk->cl_force) {  // :327-349 with the passivity controller disabled (POPCExplicitForceControl.cpp:31-33): vcl - kv vr
			double d[3], sd[3], fb[3], nrm = 0.0;
			LAW_UNR for (int i = 0; i < 3; i++) d[i] = fsw[i] - gf[i];
			law_mat3_vec(sig_f, d, sd);
			LAW_UNR for (int i = 0; i < 3; i++) ifm[i] += sd[i] * tk->dt;
			LAW_UNR for (int i = 0; i < 3; i++) t2[i] = -tk->kp_force * d[i] - tk->ki_force * ifm[i];
			law_mat3_vec(sig_f, t2, fb);
			LAW_UNR for (int i = 0; i < 3; i++) nrm += fb[i] * fb[i];
			nrm = sqrt(nrm);
			if (nrm > tk->max_force_fb) {
				const double sc = tk->max_force_fb / nrm;
				LAW_UNR for (int i = 0; i < 3; i++) fb[i] *= sc;
			}
			double sfb[3], sv[3], sg[3], rc = 1.0;
			law_mat3_vec(sig_f, fb, sfb);
			law_mat3_vec(sig_f, v, sv);
			law_mat3_vec(sig_f, gf, sg);
			if (tk->popc_enabled && popc_state) {  // enablePassivity(): :344-348
				double sfs[3];
				law_mat3_vec(sig_f, fsw, sfs);
				rc = law_popc(popc_state, ld, tk->popc_cap, tk->dt, tk->kff_force, tk->kv_force, sg, sfs, sfb, sv, popc_overflow);
			}
			LAW_UNR for (int i = 0; i < 3; i++) Ff[i] = rc * sfb[i] - tk->kv_force * sv[i] + kff_f * sg[i];
		} else {
			double sv[3], sg[3];
			law_mat3_vec(sig_f, v, sv);
			law_mat3_vec(sig_f, gf, sg);
			LAW_UNR for (int i = 0; i < 3; i++) Ff[i] = -tk->kv_force * sv[i] + kff_f * sg[i];
		}
		if (tk->cl_moment) {  // :357-378
			double d[3], sd[3], fb[3], nrm = 0.0;
			LAW_UNR for (int i = 0; i < 3; i++) d[i] = msw[i] - gm[i];
			law_mat3_vec(sig_m, d, sd);
			LAW_UNR for (int i = 0; i < 3; i++) ifm[3 + i] += sd[i] * tk->dt;
			LAW_UNR for (int i = 0; i < 3; i++) t2[i] = -tk->kp_moment * d[i] - tk->ki_moment * ifm[3 + i];
			law_mat3_vec(sig_m, t2, fb);
			LAW_UNR for (int i = 0; i < 3; i++) nrm += fb[i] * fb[i];
			nrm = sqrt(nrm);
			if (nrm > tk->max_moment_fb) {
				const double sc = tk->max_moment_fb / nrm;
				LAW_UNR for (int i = 0; i < 3; i++) fb[i] *= sc;
			}
			double sg[3];
			LAW_UNR for (int i = 0; i < 3; i++) t2[i] = fb[i] - tk->kv_moment * w[i];
			law_mat3_vec(sig_m, t2, Ff + 3);
			law_mat3_vec(sig_m, gm, sg);
			LAW_UNR for (int i = 0; i < 3; i++) Ff[3 + i] += kff_m * sg[i];
		} else {
			double sg[3];
			LAW_UNR for (int i = 0; i < 3; i++) t2[i] = -tk->kv_moment * w[i];
			law_mat3_vec(sig_m, t2, Ff + 3);
			law_mat3_vec(sig_m, gm, sg);
			LAW_UNR for (int i = 0; i < 3; i++) Ff[3 + i] += kff_m * sg[i];
		}
	} else {
		// open-loop force / moment terms + feed-forward (:350-354, :379-383, :471-487; no kff scaling in open loop)
		LAW_UNR for (int i = 0; i < 3; i++) tmp[i] = gf[i] - tk->kv_force * v[i];
		law_mat3_vec(sig_f, tmp, Ff);
		LAW_UNR for (int i = 0; i < 3; i++) tmp[i] = gm[i] - tk->kv_moment * w[i];
		law_mat3_vec(sig_m, tmp, Ff + 3);
	}
	// linear motion (:411-437)
	double se[3];
	LAW_UNR for (int i = 0; i < 3; i++) e3[i] = xp[i] - G[i];
	law_mat3_vec(sig_p, e3, se);
	LAW_UNR for (int i = 0; i < 3; i++) ip[i] += se[i] * tk->dt;
	if (tk->vel_sat) {
		double vdes[3], nrm = 0.0;
		LAW_UNR for (int i = 0; i < 3; i++) {
			vdes[i] = -tk->kp_pos[i] * tk->kvinv_pos[i] * se[i] - tk->ki_pos[i] * tk->kvinv_pos[i] * ip[i];
			nrm += vdes[i] * vdes[i];
		}
		nrm = sqrt(nrm);
		const double sc = (nrm > tk->lin_sat) ? tk->lin_sat / nrm : 1.0;
		LAW_UNR for (int i = 0; i < 3; i++) tmp[i] = G[18 + i] - tk->kv_pos[i] * (v[i] - sc * vdes[i]);
	} else {
		LAW_UNR for (int i = 0; i < 3; i++) tmp[i] = G[18 + i] - tk->kp_pos[i] * e3[i] - tk->kv_pos[i] * (v[i] - G[12 + i]) - tk->ki_pos[i] * ip[i];
	}
	law_mat3_vec(sig_p, tmp, Fum);
	// angular motion (:441-468); orientationError(desired, current) = -1/2 sum_c Rc[:,c] x Rd[:,c]
	double oe[3] = {0, 0, 0}, dphi[3];
	LAW_UNR for (int c = 0; c < 3; c++) {
		const double a0 = Rc[c], a1 = Rc[3 + c], a2 = Rc[6 + c], b0 = G[3 + c], b1 = G[6 + c], b2 = G[9 + c];
		oe[0] -= 0.5 * (a1 * b2 - a2 * b1);
		oe[1] -= 0.5 * (a2 * b0 - a0 * b2);
		oe[2] -= 0.5 * (a0 * b1 - a1 * b0);
	}
	law_mat3_vec(sig_o, oe, dphi);
	LAW_UNR for (int i = 0; i < 3; i++) io[i] += dphi[i] * tk->dt;
	if (tk->vel_sat) {
		double wdes[3], nrm = 0.0;
		LAW_UNR for (int i = 0; i < 3; i++) {
			wdes[i] = -tk->kp_ori[i] * tk->kvinv_ori[i] * dphi[i] - tk->ki_ori[i] * tk->kvinv_ori[i] * io[i];
			nrm += wdes[i] * wdes[i];
		}
		nrm = sqrt(nrm);
		const double sc = (nrm > tk->ang_sat) ? tk->ang_sat / nrm : 1.0;
		LAW_UNR for (int i = 0; i < 3; i++) tmp[i] = G[21 + i] - tk->kv_ori[i] * (w[i] - sc * wdes[i]);
	} else {
		LAW_UNR for (int i = 0; i < 3; i++) tmp[i] = G[21 + i] - tk->kp_ori[i] * dphi[i] - tk->kv_ori[i] * (w[i] - G[15 + i]) - tk->ki_ori[i] * io[i];
	}
	law_mat3_vec(sig_o, tmp, Fum + 3);
}

// JointLimitAvoidanceTask for ONE joint: limit status (updateLimitStatus, JointLimitAvoidanceTask.cpp:171-239) and the zone-blended
// constraint torque (computeTorques :278-412) with the task defaults of JointLimitAvoidanceTask.h:26-35 (RobotController owns the
// task privately: they cannot be changed through the controller).  Returns the constraint torque; *active = limit status != OFF.
__device__ __forceinline__ double jla_joint(const double q, const double dq, const double lo, const double hi, const double vel,
											 const double eff, const double t, bool* active) {
	const double KV = 20.0, PZ1 = 9.0 * M_PI / 180.0, PZ2 = 6.0 * M_PI / 180.0, VZ1 = 0.5, VZ2 = 0.3, RP = 1.0, RV = 0.05;
	int st = 0, dir = 1;  // 0 OFF, 1 POS_Z1, 2 POS_Z2, 3 VEL_Z1, 4 VEL_Z2
	double lv = 0.0;
	const bool pos_valid = (hi - lo > 2 * PZ1), vel_valid = (vel > 2 * VZ1);
	if (pos_valid) {
		if (q > hi - PZ1) { dir = 1; lv = hi; st = 1; }
		if (q > hi - PZ2) st = 2;
		if (q < lo + PZ1) { dir = -1; lv = lo; st = 1; }
		if (q < lo + PZ2) st = 2;
	}
	if (vel_valid && (st == 0 || dir == -1)) {
		if (dq > vel - VZ1) { dir = 1; lv = vel; st = 3; }
		if (dq > vel - VZ2) st = 4;
	}
	if (vel_valid && (st == 0 || dir == 1)) {
		if (dq < -vel + VZ1) { dir = -1; lv = -vel; st = 3; }
		if (dq < -vel + VZ2) st = 4;
	}
	*active = (st != 0);
	if (st == 0) return 0.0;
	const double sgn = (double)dir;
	const bool is_pos = (st <= 2);
	const double z = is_pos ? q : dq;
	const double za = (st == 1) ? PZ1 : (st == 2) ? PZ2 : (st == 3) ? VZ1 : VZ2;   // distance of the zone start to the limit
	const double zb = (st == 1) ? PZ2 : (st == 3) ? VZ2 : 0.0;                        // distance of the zone end to the limit
	const double z1 = lv - sgn * za, z2 = lv - sgn * zb;
	double alpha;  // computeBlendingCoefficient :16-36
	if (dir < 0) alpha = (z >= z1) ? 0.0 : (z <= z2) ? 1.0 : (z1 - z) / (z1 - z2);
	else alpha = (z <= z1) ? 0.0 : (z >= z2) ? 1.0 : (z - z1) / (z2 - z1);
	const double cl = eff * RV;
	const double damp = -KV * dq, damp_c = fmax(fmin(damp, cl), -cl), td = t + damp, td_c = fmax(fmin(td, cl), -cl);
	if (dir > 0) {
		if (st == 1) return (1 - alpha) * t + alpha * td;
		if (st == 2) return (1 - alpha) * td + alpha * (-eff * RP + damp);
		if (st == 3) return (1 - alpha) * t + alpha * damp;
		return (1 - alpha) * damp_c + alpha * (-alpha * cl);
	}
	if (st == 1) return alpha * t + (1 - alpha) * td_c;  // as written in the reference (:349-361)
	if (st == 2) return (1 - alpha) * td + alpha * (eff * RP + damp);
	if (st == 3) return (1 - alpha) * t + alpha * damp_c;
	return (1 - alpha) * damp_c + alpha * cl;
}

}  // namespace saip

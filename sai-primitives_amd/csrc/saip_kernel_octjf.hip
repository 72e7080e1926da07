// Control-cycle kernel for small batches of 7- and 8-dof chains whose hierarchy starts with a JOINT task: { partial JointTask, full
// MotionForceTask in its nullspace } -- the stack of the reference's examples/06 (8-dof sliding-base Panda: the joint task holds the
// base joint and the last joint, the 6-dof motion-force task acts in what is left; BASELINE config 6).
//
// Same distribution as saip_kernel_oct.hip: EIGHT LANES PER ROBOT INSTANCE, lane r owns joint r / row r of every matrix, eight instances
// per wavefront, one wavefront per workgroup, a small LDS block per instance, scans along the chain by DPP, the inherently serial
// factorisations redundantly in every lane.  With 8 dof no lane is padding (lane 7 is a real joint); with 7 dof lane 7 pads M with an
// identity row.  The lane-per-instance kernel (the path of this stack so far) fills 64 of the chip's 1024 SIMDs at 4096 instances.
//
// What the order of the two tasks changes against the { MotionForceTask, JointTask } kernel:
//   JointTask first (JointTask.cpp:218-283 with N_prec = I): Jp = S.  The rows of S are distinct unit vectors (engine: octjf_eligible),
//     so S has full row rank and matrixRangeBasis returns the identity; Lambda_j = (S M^-1 S^T)^-1 is the inverse of an m x m block of
//     M^-1 (m <= 4), N_j = I - W S with W = M^-1 S^T Lambda_j (8 x m), tau_j = S^T (Lambda_j qdd_d + Lambda_mod,j f), and nothing is
//     compensated (tau_prec = 0, :285-292).
//   MotionForceTask second (MotionForceTask.cpp:247-268): Jp = J N_j = J - (J W) S differs from J only in the m selected columns.  The
//     branch predicate of SingularityHandler.cpp:78-121 is taken on G = Jp Jp^T by the certificates of the eight-lane kernel (an exact
//     Gram eigen-solve on one lane of the instance in the band between them); outside the fully non-singular branch the instance is
//     flagged (status 1) and recomputed by the general kernel launched behind over the flag list when the task can handle it there.
//     A 6-dof task needs no basis and, being the last task, no nullspace and no Lambda: only
//     Lambda_mod = (Jp M_x^-1 Jp^T)^-1 with M_x = M, M_BIE or nothing (IMPEDANCE): tau_mf = Jp^T (Lambda_mod F_um + F_f).
//   RobotController::computeControlTorques (RobotController.cpp:79-118): tau = tau_j + tau_mf, torque saturation, gravity compensation
//     (joint limit avoidance keeps the stack on the lane kernel).
#include <hip/hip_runtime.h>
#include <math.h>

#include "saip_device.h"
#include "saip_fk.h"
#include "saip_law.h"
#include "saip_oct_common.h"
// the general kernel's body as this kernel's slow tail (see saip_kernel_oct.hip: oct_slow_tail): one wavefront plays the 64-thread workgroup
#define SAIP_WG_SYNC()                                          \
	do {                                                        \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
		__builtin_amdgcn_wave_barrier();                        \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
	} while (0)
#include "saip_wg_cycle.h"

namespace saip {

struct JfInst {
	union {  // the frames are dead once the Jacobian and the control frame exist
		double X[8][12];  // world frame per joint: R (9, row-major) + o (3)
		struct {
			double T1[6][8];  // Jp M_x^-1
			double A[6][8];   // Jp M_x^-1 Jp^T (rows 0..5, columns 0..5)
		};
	};
	double zo[8][6];   // joint motion vector about the world origin: angular part w (3), linear part v (3)
	double M[8][8];    // lower triangle of M(q), row r by lane r (7 dof: row 7 = e_7)
	double Mi[8][8];   // M^-1
	double Mb[8][8];   // M_BIE^-1 (= M^-1 when nothing is clamped)
	double J[6][8];    // world Jacobian of the control point [Jv; Jw]
	double Jp[6][8];   // J N_j
	double W[8][4];    // M^-1 S^T Lambda_j
	double vec[8][8];  // 0 q, 1 dq, 2..4 joint goal (position, velocity, acceleration rows), 5 flags
	double goal[32];   // motion-force goal: x3 R9 v3 w3 a3 alpha3 (+ f3 m3 of the general laws)
	double vw[8], xp[4], Rc[12], Fum[8], Ff[8];
	double ist[10];    // integrator state, fetched with the inputs: 0..5 motion-force task, 6..9 joint task rows; the control laws leave the advanced
	                   // values for the epilogue (motion-force: here; joint: in the owning lane).  (Also the padding that makes the instance stride = 2
	                   // (mod 32) doubles: the eight instances of a wavefront hit different LDS banks.)
};
static_assert(sizeof(JfInst) % 256 == 16, "instance stride must be 2 (mod 32) doubles");
static_assert(sizeof(JfInst) * 8 * 4 <= 160 * 1024, "four wavefronts per CU");

__device__ __forceinline__ void jf_sync() {  // one wavefront per workgroup: its LDS operations execute in order, a wavefront-scope fence is enough
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ROLE 0: the whole cycle in one wavefront.  ROLE 1 / 2: wavefronts A / B of the two-wavefront form (batches that fit the chip in one
// round: <= 2 workgroups per CU): B runs what depends on the frames only -- the motion-force task's control point, the task velocity
// vw = J dq and the control law (one lane per instance: 150 instructions under a one-lane exec mask) -- and hands (F_um, F_f) over through
// A's LDS block at the ONE workgroup barrier of the kernel, which A reaches only when it needs the force (behind the factor of Lambda_mod^-1).
// B keeps its frames in its own small block `bx`; every other synchronisation point is wavefront-local.
template <int ROLE>
__device__ __forceinline__ bool jf_body(const CycleParams& P, JfInst* smem, double (*bx)[8][12]) {
	constexpr bool roleA = ROLE != 2, roleB = ROLE != 1;
	constexpr int N = 8;
	const int lane = threadIdx.x & 63;
	const int r = octl_r(lane), grp = octl_grp(lane);  // (interleaved lane layout, saip_oct_common.h)
	const int n = P.n;  // 7 or 8
	const int rr = r < n ? r : n - 1;  // (7 dof) the padding lane mirrors joint 6
	const bool act = r < n;
	JfInst& sm = smem[grp];
	const int b_raw = (int)blockIdx.x * 8 + grp;
	const bool live = b_raw < P.B;
	const int b = live ? b_raw : P.B - 1;
	const size_t ld = P.ld;
	const ModelDev& md = *P.model;
	const TaskDev& jt = P.tasks[0];
	const TaskDev& mf = P.tasks[1];
	const int m = jt.m;  // <= 4
	int idx[4] = {0, 0, 0, 0};  // the joint each task row selects (batch-uniform: scalar loads)
	for (int a = 0; a < 4; a++)
		for (int l = 0; l < n; l++)
			if (a < m && jt.S[a * n + l] != 0.0) idx[a] = l;

	// ---------------------------------------------------------------- inputs
	const double q_r = P.q[(size_t)rr * ld + b];
	const double dq_r = P.dq[(size_t)rr * ld + b];
	// this joint's constants, requested together with the state.  (Left where they are used, the loads of R0 / p0 / axis sat INSIDE the branch on the
	// joint type -- a second memory round trip behind the one that fetched the type -- and those of the body's inertia behind the frames: two
	// exposed round trips at the start of every wavefront.)
	const int jtype_r = md.jtype[rr];
	double R0v[9], p0v[3], axv[3];
	UNR for (int e = 0; e < 9; e++) R0v[e] = md.R0[rr][e];
	UNR for (int e = 0; e < 3; e++) {
		p0v[e] = md.p0[rr][e];
		axv[e] = md.axis[rr][e];
	}
	double comv[3] = {0, 0, 0}, inev[6] = {0, 0, 0, 0, 0, 0}, mass_r = 0.0;
	if (roleA) {
		UNR for (int e = 0; e < 3; e++) comv[e] = md.com[rr][e];
		UNR for (int e = 0; e < 6; e++) inev[e] = md.inertia[rr][e];
		mass_r = md.mass[rr];
	}
	const int ra = r < m ? r : 0;
	double jg_q = 0.0, jg_dq = 0.0, jg_ddq = 0.0;
	if (roleA) {
		jg_q = saip_ldg(P.goal2[0], (size_t)ra * ld + b);
		jg_dq = saip_ldg(P.goal2[0], (size_t)(m + ra) * ld + b);
		jg_ddq = saip_ldg(P.goal2[0], (size_t)(2 * m + ra) * ld + b);
	}
	const bool track_mf = mf.has_ki || P.integ_always, track_jt = jt.has_ki || P.integ_always;
	double mfi_pre = 0.0, jti_pre = 0.0, jt_ie_new = 0.0;  // integrator state: requested with the inputs, committed from LDS / registers in the epilogue
	if (roleB && track_mf && r < 6) mfi_pre = saip_ldg(P.integ2[1], (size_t)r * ld + b);
	if (roleA && track_jt) jti_pre = saip_ldg(P.integ2[0], (size_t)ra * ld + b);
	const int gc = mf.goal_comps < 32 ? mf.goal_comps : 32;
	double gl[4] = {0.0, 0.0, 0.0, 0.0};
	if (roleB) {
		UNR for (int c = 0; c < 4; c++) gl[c] = (8 * c + r < gc) ? saip_ldg(P.goal2[1], (size_t)(8 * c + r) * ld + b) : 0.0;
	}

	// ---------------------------------------------------------------- updateModel(): kinematics (log-step prefix product over the chain)
	double Tw[12];
	{
		const double* R0 = R0v;
		const double ax = axv[0], ay = axv[1], az = axv[2];
		if (jtype_r == 1) {
			double s, c;
			sincos_joint(q_r, &s, &c);
			if (md.all_axis_z) {  // (batch-uniform) rotation about the local z axis: R0 Rz(q) mixes the first two columns of R0 only
				UNR for (int i = 0; i < 3; i++) {
					Tw[3 * i] = R0[3 * i] * c + R0[3 * i + 1] * s;
					Tw[3 * i + 1] = R0[3 * i + 1] * c - R0[3 * i] * s;
					Tw[3 * i + 2] = R0[3 * i + 2];
				}
			} else {
				const double v = 1.0 - c;
				const double Rq[9] = {c + ax * ax * v,      ax * ay * v - az * s, ax * az * v + ay * s,
									  ay * ax * v + az * s, c + ay * ay * v,      ay * az * v - ax * s,
									  az * ax * v - ay * s, az * ay * v + ax * s, c + az * az * v};
				UNR for (int i = 0; i < 3; i++)
					UNR for (int j = 0; j < 3; j++) Tw[3 * i + j] = R0[3 * i] * Rq[j] + R0[3 * i + 1] * Rq[3 + j] + R0[3 * i + 2] * Rq[6 + j];
			}
			UNR for (int e = 0; e < 3; e++) Tw[9 + e] = p0v[e];
		} else {
			UNR for (int e = 0; e < 9; e++) Tw[e] = R0[e];
			const double a[3] = {ax, ay, az};
			double d[3];
			oct_mat3_vec(R0, a, d);
			UNR for (int e = 0; e < 3; e++) Tw[9 + e] = p0v[e] + d[e] * q_r;
		}
	}
	if (roleA) {
		sm.vec[0][r] = q_r;
		sm.vec[1][r] = dq_r;
		sm.vec[2][r] = jg_q;
		sm.vec[3][r] = jg_dq;
		sm.vec[4][r] = jg_ddq;
		if (r < 4) sm.ist[6 + r] = jti_pre;
	}
	if (roleB) {
		UNR for (int c = 0; c < 4; c++) sm.goal[8 * c + r] = gl[c];
		if (r < 6) sm.ist[r] = mfi_pre;
	}
	OCT_FRAME_STEP(1, Tw, r)
	OCT_FRAME_STEP(2, Tw, r)
	OCT_FRAME_STEP(4, Tw, r)
	double (*Xf)[12] = (ROLE == 2) ? bx[grp] : sm.X;  // the frames of this wavefront (B: its own block)
	UNR for (int e = 0; e < 12; e++) Xf[r][e] = Tw[e];
	const bool rev = jtype_r == 1;
	double z[3], o[3];
	{
		const double a[3] = {axv[0], axv[1], axv[2]};
		oct_mat3_vec(Tw, a, z);
		UNR for (int e = 0; e < 3; e++) o[e] = Tw[9 + e];
	}
	// this joint's motion vector about the world origin, S_r = (w, v): revolute (z, o x z), prismatic (0, z) -- what M_ir = S_i . (Ic_r S_r) needs
	// of the other joints (LDS stores are not guarded: the padding lane owns slot 7 of every row and nobody reads it)
	const double sw[3] = {rev ? z[0] : 0.0, rev ? z[1] : 0.0, rev ? z[2] : 0.0};
	const double sv[3] = {rev ? (o[1] * z[2] - o[2] * z[1]) : z[0], rev ? (o[2] * z[0] - o[0] * z[2]) : z[1], rev ? (o[0] * z[1] - o[1] * z[0]) : z[2]};
	if (roleA) {
		UNR for (int e = 0; e < 3; e++) {
			sm.zo[r][e] = sw[e];
			sm.zo[r][3 + e] = sv[e];
		}
	}
	// per-body spatial inertia about the world origin, then the composite inertias of the subtrees (suffix sums by DPP row_shl)
	double Ic[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	if (roleA) {
		double c[3], Bd[10];
		const double cl[3] = {comv[0], comv[1], comv[2]};
		oct_mat3_vec(Tw, cl, c);
		UNR for (int e = 0; e < 3; e++) c[e] += o[e];
		const double* I6 = inev;
		const double Il[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
		double RI[9], Iw[9];
		UNR for (int i = 0; i < 3; i++)
			UNR for (int j = 0; j < 3; j++) RI[3 * i + j] = Tw[3 * i] * Il[j] + Tw[3 * i + 1] * Il[3 + j] + Tw[3 * i + 2] * Il[6 + j];
		UNR for (int i = 0; i < 3; i++)
			UNR for (int j = 0; j < 3; j++) Iw[3 * i + j] = RI[3 * i] * Tw[3 * j] + RI[3 * i + 1] * Tw[3 * j + 1] + RI[3 * i + 2] * Tw[3 * j + 2];
		const double ms = mass_r, cc2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
		Bd[0] = ms;
		Bd[1] = ms * c[0];
		Bd[2] = ms * c[1];
		Bd[3] = ms * c[2];
		Bd[4] = Iw[0] + ms * (cc2 - c[0] * c[0]);
		Bd[5] = Iw[4] + ms * (cc2 - c[1] * c[1]);
		Bd[6] = Iw[8] + ms * (cc2 - c[2] * c[2]);
		Bd[7] = Iw[1] - ms * c[0] * c[1];
		Bd[8] = Iw[2] - ms * c[0] * c[2];
		Bd[9] = Iw[5] - ms * c[1] * c[2];
		UNR for (int e = 0; e < 10; e++) Ic[e] = act ? Bd[e] : 0.0;
	}
	// (selects, not multiplications by 0: a neighbouring instance with non-finite state must not leak in)
	if (roleA) {
		UNR for (int e = 0; e < 10; e++) {
			const double up = OCT_ILV ? octl_shl<1>(Ic[e]) : oct_dpp<DPP_ROW_SHL + 1>(Ic[e]);
			Ic[e] += (OCT_ILV || r + 1 < 8) ? up : 0.0;
		}
		UNR for (int e = 0; e < 10; e++) {
			const double up = OCT_ILV ? octl_shl<2>(Ic[e]) : oct_dpp<DPP_ROW_SHL + 2>(Ic[e]);
			Ic[e] += (OCT_ILV || r + 2 < 8) ? up : 0.0;
		}
		UNR for (int e = 0; e < 10; e++) {
			const double up = OCT_ILV ? octl_shl<4>(Ic[e]) : oct_dpp<DPP_ROW_SHL + 4>(Ic[e]);
			Ic[e] += (OCT_ILV || r + 4 < 8) ? up : 0.0;
		}
	}
	jf_sync();
	// ---------------------------------------------------------------- M(q): row r of the lower triangle, M_ir = S_i . (Ic_r S_r), i <= r
	if (roleA) {
		const double* w = sw;
		const double* v = sv;
		const double ms = Ic[0], hx = Ic[1], hy = Ic[2], hz = Ic[3];
		const double f[3] = {ms * v[0] + (w[1] * hz - w[2] * hy), ms * v[1] + (w[2] * hx - w[0] * hz), ms * v[2] + (w[0] * hy - w[1] * hx)};
		const double nO[3] = {Ic[4] * w[0] + Ic[7] * w[1] + Ic[8] * w[2] + (hy * v[2] - hz * v[1]),
							  Ic[7] * w[0] + Ic[5] * w[1] + Ic[9] * w[2] + (hz * v[0] - hx * v[2]),
							  Ic[8] * w[0] + Ic[9] * w[1] + Ic[6] * w[2] + (hx * v[1] - hy * v[0])};
		UNR for (int i = 0; i < N; i++) {  // S_i = (w_i, v_i) from LDS; entries beyond the diagonal are written too but never read
			const double mri = sm.zo[i][0] * nO[0] + sm.zo[i][1] * nO[1] + sm.zo[i][2] * nO[2] + sm.zo[i][3] * f[0] + sm.zo[i][4] * f[1] + sm.zo[i][5] * f[2];
			sm.M[r][i] = act ? mri : ((i == r) ? 1.0 : 0.0);  // 7 dof: row 7 of the padded matrix is e_7
		}
	}
	// ---------------------------------------------------------------- MotionForceTask: this joint's column of the world Jacobian, control point, control frame
	double pw[3], Rc[9], jw[6];
	{
		const int bd = mf.body;
		double Rb[12];
		UNR for (int e = 0; e < 12; e++) Rb[e] = Xf[bd][e];
		oct_mat3_vec(Rb, mf.pos, pw);
		UNR for (int e = 0; e < 3; e++) pw[e] += Rb[9 + e];
		UNR for (int i = 0; i < 3; i++)
			UNR for (int j = 0; j < 3; j++) Rc[3 * i + j] = Rb[3 * i] * mf.rot[j] + Rb[3 * i + 1] * mf.rot[3 + j] + Rb[3 * i + 2] * mf.rot[6 + j];
		const bool on = act && rr <= bd;
		const double rx = pw[0] - o[0], ry = pw[1] - o[1], rz = pw[2] - o[2];
		const double c0 = z[1] * rz - z[2] * ry, c1 = z[2] * rx - z[0] * rz, c2 = z[0] * ry - z[1] * rx;
		jw[0] = on ? (rev ? c0 : z[0]) : 0.0;
		jw[1] = on ? (rev ? c1 : z[1]) : 0.0;
		jw[2] = on ? (rev ? c2 : z[2]) : 0.0;
		jw[3] = (on && rev) ? z[0] : 0.0;
		jw[4] = (on && rev) ? z[1] : 0.0;
		jw[5] = (on && rev) ? z[2] : 0.0;
		if (roleA) {
			UNR for (int a = 0; a < 6; a++) sm.J[a][r] = jw[a];
		}
		if (roleB) {
			// the task velocities use the UNprojected Jacobian (MotionForceTask.cpp:293-298): vw = J dq
			UNR for (int a = 0; a < 6; a++) {
				const double s = octl_sum(jw[a] * dq_r);
				if (r == 0) sm.vw[a] = s;
			}
			if (r == 0) {
				UNR for (int e = 0; e < 3; e++) sm.xp[e] = pw[e];
				UNR for (int e = 0; e < 9; e++) sm.Rc[e] = Rc[e];
			}
		}
	}
	jf_sync();
	// ---------------------------------------------------------------- motion-force control law, one lane per instance (MotionForceTask.cpp:286-509)
	if (roleB && r == 0) {
		double ip[3] = {0, 0, 0}, io[3] = {0, 0, 0};
		if (track_mf)
			for (int i = 0; i < 3; i++) {
				ip[i] = sm.ist[i];
				io[i] = sm.ist[3 + i];
			}
		if (mf.general_law) {  // velocity saturation, force / moment spaces with open-loop force control
			const double(&G)[32] = sm.goal;
			law_motion_force_general<false>(&mf, sm.vw, sm.xp, sm.Rc, G, ip, io, nullptr, sm.Fum, sm.Ff);
		} else {  // the defaults (sigma_force = sigma_moment = 0, no velocity saturation): :411-468 with the host's sigma matrices
			const double* G = sm.goal;
			const double* vw = sm.vw;
			double v[3], w[3], e3[3], tmp[3], oe[3] = {0, 0, 0}, dphi[3];
			UNR for (int i = 0; i < 3; i++) {
				v[i] = mf.Ppos[3 * i] * vw[0] + mf.Ppos[3 * i + 1] * vw[1] + mf.Ppos[3 * i + 2] * vw[2];
				w[i] = mf.Pori[3 * i] * vw[3] + mf.Pori[3 * i + 1] * vw[4] + mf.Pori[3 * i + 2] * vw[5];
				e3[i] = pw[i] - G[i];
			}
			UNR for (int i = 0; i < 3; i++) ip[i] += (mf.sig_p[3 * i] * e3[0] + mf.sig_p[3 * i + 1] * e3[1] + mf.sig_p[3 * i + 2] * e3[2]) * mf.dt;
			UNR for (int i = 0; i < 3; i++) tmp[i] = G[18 + i] - mf.kp_pos[i] * e3[i] - mf.kv_pos[i] * (v[i] - G[12 + i]) - mf.ki_pos[i] * ip[i];
			UNR for (int i = 0; i < 3; i++) sm.Fum[i] = mf.sig_p[3 * i] * tmp[0] + mf.sig_p[3 * i + 1] * tmp[1] + mf.sig_p[3 * i + 2] * tmp[2];
			UNR for (int c = 0; c < 3; c++) {  // orientationError(desired, current) = -1/2 sum_c Rc[:,c] x Rd[:,c]
				const double a0 = Rc[c], a1 = Rc[3 + c], a2 = Rc[6 + c], b0 = G[3 + c], b1 = G[6 + c], b2 = G[9 + c];
				oe[0] -= 0.5 * (a1 * b2 - a2 * b1);
				oe[1] -= 0.5 * (a2 * b0 - a0 * b2);
				oe[2] -= 0.5 * (a0 * b1 - a1 * b0);
			}
			UNR for (int i = 0; i < 3; i++) dphi[i] = mf.sig_o[3 * i] * oe[0] + mf.sig_o[3 * i + 1] * oe[1] + mf.sig_o[3 * i + 2] * oe[2];
			UNR for (int i = 0; i < 3; i++) io[i] += dphi[i] * mf.dt;
			UNR for (int i = 0; i < 3; i++) tmp[i] = G[21 + i] - mf.kp_ori[i] * dphi[i] - mf.kv_ori[i] * (w[i] - G[15 + i]) - mf.ki_ori[i] * io[i];
			UNR for (int i = 0; i < 3; i++) sm.Fum[3 + i] = mf.sig_o[3 * i] * tmp[0] + mf.sig_o[3 * i + 1] * tmp[1] + mf.sig_o[3 * i + 2] * tmp[2];
			UNR for (int i = 0; i < 6; i++) sm.Ff[i] = 0.0;
		}
		if (track_mf)
			for (int i = 0; i < 3; i++) {
				sm.ist[i] = ip[i];
				sm.ist[3 + i] = io[i];
			}
	}
	if (ROLE == 2) {  // B is done: (F_um, F_f) are in A's block (and the advanced integrators in HBM) behind this barrier
		__syncthreads();
		return false;
	}
	// ---------------------------------------------------------------- M^-1 (and M_BIE^-1): Cholesky factor in every lane, own column of the inverse
	// M_BIE = M with the diagonal clamped from below (SingularityHandler.cpp:176-182, JointTask.cpp:253-259); factored a second time only
	// when some instance of the wavefront has a clamped entry
	double mi[N], mb[N];
	{
		double L[N][N], dinv[N];
		UNR for (int i = 0; i < N; i++)
			UNR for (int j = 0; j <= i; j++) L[i][j] = sm.M[i][j];
		bool clamp = false;
		if (P.any_bie) {
			UNR for (int e = 0; e < N; e++) clamp = clamp || (e < n && P.bie_thr - L[e][e] > 0.0);
		}
		if (__any(clamp)) {
			double Lb[N][N], dinvb[N];
			UNR for (int i = 0; i < N; i++)
				UNR for (int j = 0; j <= i; j++) Lb[i][j] = (i == j && i < n) ? fmax(L[i][j], P.bie_thr) : L[i][j];
			oct_cholesky<N>(L, dinv);
			oct_cholesky<N>(Lb, dinvb);
			oct_solve_unit<N>(L, dinv, r, mi);
			oct_solve_unit<N>(Lb, dinvb, r, mb);
		} else {
			oct_cholesky<N>(L, dinv);
			oct_solve_unit<N>(L, dinv, r, mi);
			UNR for (int j = 0; j < N; j++) mb[j] = mi[j];
		}
	}
	UNR for (int j = 0; j < N; j++) {
		sm.Mi[r][j] = mi[j];
		sm.Mb[r][j] = mb[j];
	}
	jf_sync();
	// ---------------------------------------------------------------- JointTask (first in the hierarchy): m x m algebra, every lane alike
	double tau_j = 0.0;
	{
		double Aj[4][4], Lamj[4][4], Lmodj[4][4];
		UNR for (int a = 0; a < 4; a++)
			UNR for (int c = 0; c < 4; c++) Aj[a][c] = (a < m && c < m) ? sm.Mi[idx[a]][idx[c]] : ((a == c) ? 1.0 : 0.0);
		oct_spd_inverse_n<4>(Aj, Lamj);
		if (jt.decoupling == DEC_BIE) {  // JointTask.cpp:253-267
			UNR for (int a = 0; a < 4; a++)
				UNR for (int c = 0; c < 4; c++) Aj[a][c] = (a < m && c < m) ? sm.Mb[idx[a]][idx[c]] : ((a == c) ? 1.0 : 0.0);
			oct_spd_inverse_n<4>(Aj, Lmodj);
		} else {
			UNR for (int a = 0; a < 4; a++)
				UNR for (int c = 0; c < 4; c++) Lmodj[a][c] = (jt.decoupling == DEC_IMPEDANCE) ? ((a == c) ? 1.0 : 0.0) : Lamj[a][c];
		}
		// W = M^-1 S^T Lambda_j: row r by lane r (N_j = I - W S)
		UNR for (int a = 0; a < 4; a++) {
			double s = 0.0;
			UNR for (int c = 0; c < 4; c++) s = fma((c < m) ? sm.Mi[r][idx[c]] : 0.0, Lamj[c][a], s);
			sm.W[r][a] = (a < m) ? s : 0.0;
		}
		// control law of the task rows (JointTask.cpp:294-356); tau_prec = 0, nothing to compensate (:285-292)
		double ai[4], fi[4];
		UNR for (int a = 0; a < 4; a++) {
			const bool row = a < m;
			const double cur = sm.vec[0][idx[a]], vel = sm.vec[1][idx[a]];
			const double gq = sm.vec[2][a], gdq = sm.vec[3][a], gddq = sm.vec[4][a];
			const double e = cur - gq;
			double ie = (track_jt && row) ? sm.ist[6 + a] : 0.0;
			ie += e * jt.dt;  // :323-324
			jt_ie_new = (r == a) ? ie : jt_ie_new;  // lane a keeps row a for the epilogue
			double f;
			if (jt.vel_sat) {  // :327-341
				double vdes = -jt.kp[a] * jt.kvinv[a] * e - jt.ki[a] * jt.kvinv[a] * ie;
				vdes = fmin(fmax(vdes, -jt.sat[a]), jt.sat[a]);
				f = -jt.kv[a] * (vel - vdes);
			} else {
				f = -jt.kp[a] * e - jt.kv[a] * (vel - gdq) - jt.ki[a] * ie;  // :342-345
			}
			ai[a] = row ? gddq : 0.0;
			fi[a] = row ? f : 0.0;
		}
		UNR for (int a = 0; a < 4; a++) {
			double g = 0.0;
			UNR for (int c = 0; c < 4; c++) g = fma(Lamj[a][c], ai[c], fma(Lmodj[a][c], fi[c], g));  // :348-351
			tau_j = (a < m && idx[a] == r) ? g : tau_j;  // tau_j = S^T g
		}
	}
	jf_sync();
	// ---------------------------------------------------------------- MotionForceTask: Jp = J N_j = J - (J W) S, column r by lane r
	double jp[6];
	{
		int a_own = -1;
		UNR for (int a = 0; a < 4; a++) a_own = (a < m && idx[a] == r) ? a : a_own;
		const int ao = a_own < 0 ? 0 : a_own;
		UNR for (int x = 0; x < 6; x++) {
			double s = 0.0;
			UNR for (int i = 0; i < N; i++) s = fma(sm.J[x][i], sm.W[i][ao], s);
			jp[x] = (a_own >= 0) ? jw[x] - s : jw[x];
			sm.Jp[x][r] = jp[x];
		}
	}
	jf_sync();
	double Jf[6][N];
	UNR for (int a = 0; a < 6; a++)
		UNR for (int j = 0; j < N; j++) Jf[a][j] = sm.Jp[a][j];
	// ---- SingularityHandler branch predicate on G = Jp Jp^T (SingularityHandler.cpp:78-121), every lane of the instance alike
	bool singular = false;
	{
		double G[6][6], tr = 0.0;
		UNR for (int a = 0; a < 6; a++)
			UNR for (int c = 0; c <= a; c++) {
				double s = 0.0;
				UNR for (int j = 0; j < N; j++) s = fma(Jf[a][j], Jf[c][j], s);
				G[a][c] = s;
				G[c][a] = s;
				if (a == c) tr += s;
			}
		const double itr = oct_rcp(tr);
		const double smax2 = mf.s_max * mf.s_max, tol2 = mf.s_abs_tol * mf.s_abs_tol;
		auto ldl_positive = [&](const double shift) {
			double Gs[6][6];
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c <= a; c++) Gs[a][c] = G[a][c] * itr;
			bool pos = true;
			UNR for (int a = 0; a < 6; a++) Gs[a][a] -= shift;
			UNR for (int kk = 0; kk < 6; kk++) {
				const double d = Gs[kk][kk];
				pos = pos && (d > 1e-13);
				const double id = oct_rcp(d);
				UNR for (int i = kk + 1; i < 6; i++) {
					const double lik = Gs[i][kk] * id;
					UNR for (int j = kk + 1; j <= i; j++) Gs[i][j] = fma(-lik, Gs[j][kk], Gs[i][j]);
				}
			}
			return pos;
		};
		// stage 1: lambda_max(Gs) <= trace(Gs) = 1 and lambda_max(G) >= trace(G) / 6, so a positive definite Gs - s_max^2 I and
		// trace(G) >= 6 s_abs_tol^2 certify the fully non-singular branch
		bool need_exact = !((tr > 0.0) && (tr >= 6.0 * tol2) && ldl_positive(smax2));
#if defined(SAIP_OCT_FORCE_EXACT)  // test-only build (libsaip_octexact.so): every instance takes the exact eigen fallback
		need_exact = true;
#endif
		if (__any(need_exact)) {
			// stage 2: the trace-power bound u = trace(Gs^8)^(1/8) in [lambda_max, 6^(1/8) lambda_max] tightens both tests
			double G2[6][6], Gt[6][6];
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c < 6; c++) G2[a][c] = G[a][c] * itr;
			UNR for (int rep = 0; rep < 2; rep++) {
				UNR for (int a = 0; a < 6; a++)
					UNR for (int c = 0; c <= a; c++) {
						double s = 0.0;
						UNR for (int l = 0; l < 6; l++) s = fma(G2[a][l], G2[l][c], s);
						Gt[a][c] = s;
					}
				UNR for (int a = 0; a < 6; a++)
					UNR for (int c = 0; c <= a; c++) {
						G2[a][c] = Gt[a][c];
						G2[c][a] = Gt[a][c];
					}
			}
			double t8 = 0.0;
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c < 6; c++) t8 = fma(G2[a][c], G2[a][c], t8);
			const double u = sqrt(sqrt(sqrt(t8)));
			const bool ok_s0 = (u * mf.cert_kroot * tr >= tol2);
			const bool pos = ldl_positive(smax2 * u);
			need_exact = need_exact && !(ok_s0 && pos && tr > 0.0);
			if (need_exact) {
				// stage 3, the opposite certificate: with l = max_i Gs_ii <= lambda_max(Gs), a NEGATIVE LDL^T pivot of Gs - s_max^2 l I (all earlier
				// pivots positive: Sylvester) proves lambda_min < s_max^2 lambda_max, i.e. the instance is outside the non-singular branch for
				// sure; so does trace(G) < s_abs_tol^2.  A batch with many near-singular postures does not queue up for the one-lane eigen-solve.
				double lmx = 0.0;
				UNR for (int a = 0; a < 6; a++) lmx = fmax(lmx, G[a][a] * itr);
				double Gs[6][6];
				UNR for (int a = 0; a < 6; a++)
					UNR for (int c = 0; c <= a; c++) Gs[a][c] = G[a][c] * itr;
				UNR for (int a = 0; a < 6; a++) Gs[a][a] -= smax2 * lmx;
				bool neg = false, clean = true;
				UNR for (int kk = 0; kk < 6; kk++) {
					const double dpv = Gs[kk][kk];
					neg = neg || (clean && dpv < -1e-12);
					clean = clean && (dpv > 1e-13);
					const double id = oct_rcp(dpv);
					UNR for (int i = kk + 1; i < 6; i++) {
						const double lik = Gs[i][kk] * id;
						UNR for (int j = kk + 1; j <= i; j++) Gs[i][j] = fma(-lik, Gs[j][kk], Gs[i][j]);
					}
				}
				if ((tr > 0.0 && neg) || tr < tol2) {
					singular = true;
					need_exact = false;
				}
			}
#if defined(SAIP_OCT_FORCE_EXACT)
			need_exact = true;
			singular = false;
#endif
			if (__any(need_exact)) {  // exact (rare): singular values from the Gram eigenvalues, one lane per instance
				if (need_exact && r == 0) {
					for (int a = 0; a < 6; a++)
						for (int c = 0; c < 6; c++) sm.A[a][c] = G[a][c];  // (T1 / A are not written yet; the frames are dead)
					oct_serial_jacobi(&sm.A[0][0], nullptr, 6, 8);
					double lmax = 0.0;
					for (int i = 0; i < 6; i++) lmax = fmax(lmax, sm.A[i][i]);
					const double s0 = sqrt(lmax);
					bool sing = !(s0 >= mf.s_abs_tol);
					for (int i = 0; i < 6; i++)
						if (sqrt(fmax(sm.A[i][i], 0.0)) / s0 < mf.s_max) sing = true;
					sm.vec[5][0] = sing ? 1.0 : 0.0;
				}
				jf_sync();
				if (need_exact) singular = sm.vec[5][0] != 0.0;
				jf_sync();
			}
		}
	}
	// ---- Lambda_mod = (Jp M_x^-1 Jp^T)^-1 by its Cholesky factor; tau_mf = Jp^T (Lambda_mod F_um + F_f)  (SingularityHandler.cpp:175-193, 307-309)
	if (ROLE == 1) __syncthreads();  // the one workgroup barrier: B's force (it has been waiting here for a while)
	double Fum[6], g[6];
	UNR for (int a = 0; a < 6; a++) Fum[a] = sm.Fum[a];
	if (mf.decoupling == DEC_IMPEDANCE) {
		UNR for (int a = 0; a < 6; a++) g[a] = Fum[a];
	} else {
		const bool bie = mf.decoupling == DEC_BIE;
		UNR for (int x = 0; x < 6; x++) {  // column r of T1 = Jp M_x^-1 from this lane's row of the (symmetric) inverse
			double s = 0.0;
			UNR for (int j = 0; j < N; j++) s = fma(Jf[x][j], bie ? mb[j] : mi[j], s);
			sm.T1[x][r] = s;
		}
		jf_sync();
		{  // row c of A = T1 Jp^T by lane c (lanes 6, 7 mirror row 5)
			const int c6 = r < 6 ? r : 5;
			double t1r[N];
			UNR for (int j = 0; j < N; j++) t1r[j] = sm.T1[c6][j];
			UNR for (int y = 0; y < 6; y++) {
				double s = 0.0;
				UNR for (int j = 0; j < N; j++) s = fma(t1r[j], Jf[y][j], s);
				if (r < 6) sm.A[r][y] = s;
			}
		}
		jf_sync();
		double L6[6][6], d6[6];
		UNR for (int i = 0; i < 6; i++)
			UNR for (int j = 0; j <= i; j++) L6[i][j] = sm.A[i][j];
		oct_cholesky<6>(L6, d6);
		oct_solve<6>(L6, d6, Fum, g);
	}
	double tau_mf = 0.0;
	UNR for (int x = 0; x < 6; x++) tau_mf = fma(jp[x], g[x] + sm.Ff[x], tau_mf);
	double tv = tau_j + tau_mf;
	// ---------------------------------------------------------------- post-processing, RobotController.cpp:86-116
	if (P.torque_sat) {
		const double lim = md.effort[rr];
		tv = tv > lim ? lim : (tv < -lim ? -lim : tv);
	}
	if (P.gravity_comp) {
		const double* gv = md.gravity;
		const double ax_ = Ic[1] - Ic[0] * o[0], ay_ = Ic[2] - Ic[0] * o[1], az_ = Ic[3] - Ic[0] * o[2];  // m (c - o)
		const double cx = z[1] * az_ - z[2] * ay_, cy = z[2] * ax_ - z[0] * az_, cz = z[0] * ay_ - z[1] * ax_;
		const double tgr = -(cx * gv[0] + cy * gv[1] + cz * gv[2]);
		const double tgp = -Ic[0] * (z[0] * gv[0] + z[1] * gv[1] + z[2] * gv[2]);
		tv += rev ? tgr : tgp;
	}
	if (!(tv == tv)) singular = true;  // non-finite inputs: refuse rather than write NaN as a valid torque
	singular = octl_votes(__builtin_amdgcn_ballot_w64(singular && act), lane) != 0u;  // (one verdict per instance)
	// a flagged instance holds its last valid torques (or gets them from the slow path launched behind) unless NaN was asked for
	// (with the slow tail on, the tail decides torques and status of a flagged instance: nothing is written for it here)
	const bool to_tail = P.slow_tail && singular;
	if (live && act && !singular) P.tau[(size_t)r * ld + b] = tv;
	if (live && act && singular && P.flag_nan && !to_tail) P.tau[(size_t)r * ld + b] = __builtin_nan("");
	if (live && r == 0 && !to_tail) P.status[b] = singular ? 1 : 0;
	if (live && r == 0 && singular && P.flag_list) {
		const int slot = atomicAdd(P.flag_count, 1);
		if (slot < P.B) P.flag_list[slot] = b;
	}  // work list of the device-side slow path
	if (P.flag_list && blockIdx.x == 0 && lane == 0) *P.flag_count_next = 0;  // the other list: its last readers finished a cycle ago
	// integrators: commit what the control laws advanced them to unless the instance ended flagged
	if (live && !singular) {
		if (track_mf && r < 6) P.integ2[1][(size_t)r * ld + b] = sm.ist[r];
		if (track_jt && r < m) P.integ2[0][(size_t)r * ld + b] = jt_ie_new;
	}
	return live && to_tail;
}

// Slow tail (round 4; the same as saip_kernel_oct.hip's): the instances of this wavefront whose motion-force task left the non-singular branch
// are recomputed here by the general kernel's body on this wavefront's own LDS block -- no list, no second launch behind the kernel.
__device__ __forceinline__ void jf_slow_tail(const CycleParams& P, const bool flagged, void* lds) {
	const unsigned long long votes = __ballot(flagged);
	if (__builtin_expect(votes == 0ull, 1)) return;  // wave-uniform: the usual case
	static_assert(sizeof(WgSmem<8>) <= 8 * sizeof(JfInst), "the general kernel's block fits the eight instance blocks of a wavefront");
	WgSmem<8>& wsm = *reinterpret_cast<WgSmem<8>*>(lds);
	for (int g = 0; g < 8; g++) {
		const int l0 = ((g >> 1) << 4) | (g & 1);  // the lane of joint 0 of instance g (interleaved layout)
		if (((votes >> l0) & 1ull) == 0ull) continue;
		SAIP_WG_SYNC();
		wg_cycle<8, 64>(P, (int)blockIdx.x * 8 + g, wsm);
	}
}

template <bool DUO>
__global__ void __launch_bounds__(DUO ? 128 : 64) saip_cycle_octjf(const CycleParams P) {
	__shared__ JfInst smem[8];
	__shared__ double bx[DUO ? 8 : 1][8][12];  // wavefront B's frames
	if (DUO) {
		const bool wave_b = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1;
		if (wave_b) jf_body<2>(P, smem, bx);
		else jf_slow_tail(P, jf_body<1>(P, smem, bx), smem);
	} else {
		jf_slow_tail(P, jf_body<0>(P, smem, bx), smem);
	}
}

// two wavefronts per group of eight instances while the launch fits the chip in one round (<= 2 workgroups per CU), as in saip_kernel_oct.hip
hipError_t launch_cycle_octjf(const CycleParams& P, hipStream_t stream) {
	const dim3 grid((P.B + 7) / 8);
	static const bool duo_on = getenv("SAIP_OCT_NO_DUO") == nullptr;
	static int cus = 0;
	if (cus == 0) {
		int dev = 0, n = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
		cus = n;
	}
	if (duo_on && (int)grid.x <= 2 * cus) hipLaunchKernelGGL(saip_cycle_octjf<true>, grid, dim3(128), 0, stream, P);
	else hipLaunchKernelGGL(saip_cycle_octjf<false>, grid, dim3(64), 0, stream, P);
	return hipGetLastError();
}

}  // namespace saip

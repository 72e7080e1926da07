// Batched forward dynamics + semi-implicit Euler step: the "step after the path" of SURVEY.md 8(f) f4, so that closed-loop
// rollouts (OTG -> control cycle -> dynamics) stay on the device.  It stands in for what the reference's examples do with the
// external physics engine: sim->setJointTorques(...); sim->integrate()   (/root/reference/examples/05-using_robot_controller/
// 05-using_robot_controller.cpp:225-231; sai-simulation is not part of the reference tree, so there is no reference arithmetic to
// match -- the oracle is the Lagrangian restatement in oracle/restatement.forward_dynamics, see tests/test_gpu_dynamics.py).
//
//   M(q) qdd + b(q, dq) + g(q) = tau        b: Coriolis / centrifugal, g: gravity
//   dq <- dq + dt qdd ;  q <- q + dt dq     (semi-implicit Euler, `substeps` times per call with the torque held)
//
// One lane per instance (7-dof chains run saip_dynamics_oct.hip instead: eight lanes per instance).  Bias forces by one recursive Newton-Euler pass in world coordinates, M(q) from composite rigid bodies
// (spatial inertias about the world origin, suffix sums along the chain), Cholesky solve; fully unrolled for 7-dof arms.  Per-body arrays are lane-private
// (scratch for NMAX = 32, mostly registers for NMAX = 8); the kernel is FP64-latency bound like the cycle kernels and is not on the
// benchmarked path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "saip_device.h"

namespace saip {

namespace {

struct V3 {
	double x, y, z;
};
__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 mulR(const double* R, V3 v) {
	return V3{R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z};
}
__device__ __forceinline__ V3 mulRt(const double* R, V3 v) {
	return V3{R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z, R[2] * v.x + R[5] * v.y + R[8] * v.z};
}

template <int NMAX>
struct Chain {  // world-frame kinematics of the movable bodies of one instance
	double R[NMAX][9];
	V3 o[NMAX], z[NMAX], c[NMAX];  // joint origin, joint axis, centre of mass
};

template <int NMAX>
__device__ void chain_fk(const ModelDev& md, int n, const double* q, Chain<NMAX>& K) {
	double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	V3 o = v3(0, 0, 0);
#pragma unroll
	for (int j = 0; j < n; j++) {
		o = o + mulR(R, v3(md.p0[j][0], md.p0[j][1], md.p0[j][2]));
		double Rt[9];
		for (int r = 0; r < 3; r++)
			for (int c = 0; c < 3; c++) Rt[3 * r + c] = R[3 * r] * md.R0[j][c] + R[3 * r + 1] * md.R0[j][3 + c] + R[3 * r + 2] * md.R0[j][6 + c];
		const double ax = md.axis[j][0], ay = md.axis[j][1], az = md.axis[j][2];
		if (md.jtype[j] == 1) {
			double s, c;
			sincos(q[j], &s, &c);
			const double v = 1.0 - c;
			const double Rq[9] = {c + ax * ax * v,      ax * ay * v - az * s, ax * az * v + ay * s,
								  ay * ax * v + az * s, c + ay * ay * v,      ay * az * v - ax * s,
								  az * ax * v - ay * s, az * ay * v + ax * s, c + az * az * v};
			for (int r = 0; r < 3; r++)
				for (int c2 = 0; c2 < 3; c2++) R[3 * r + c2] = Rt[3 * r] * Rq[c2] + Rt[3 * r + 1] * Rq[3 + c2] + Rt[3 * r + 2] * Rq[6 + c2];
		} else {
			for (int e = 0; e < 9; e++) R[e] = Rt[e];
			o = o + q[j] * mulR(R, v3(ax, ay, az));
		}
		for (int e = 0; e < 9; e++) K.R[j][e] = R[e];
		K.o[j] = o;
		K.z[j] = mulR(R, v3(ax, ay, az));
		K.c[j] = o + mulR(R, v3(md.com[j][0], md.com[j][1], md.com[j][2]));
	}
}

// world inertia times vector: R I R^T w
__device__ __forceinline__ V3 inertia_mul(const double* R, const double* I6, V3 w) {
	const V3 l = mulRt(R, w);
	const V3 Il = v3(I6[0] * l.x + I6[3] * l.y + I6[4] * l.z, I6[3] * l.x + I6[1] * l.y + I6[5] * l.z, I6[4] * l.x + I6[5] * l.y + I6[2] * l.z);
	return mulR(R, Il);
}

// Recursive Newton-Euler in world coordinates: joint torques for (dq, ddq) with base acceleration a0 (= -gravity).
// with_velocity = false drops every velocity-product term (used for the columns of M).
template <int NMAX>
__device__ void rnea(const ModelDev& md, int n, const Chain<NMAX>& K, const double* dq, const double* ddq, V3 a0, bool with_velocity, double* tau) {
	V3 f[NMAX], nn[NMAX];  // net force on body j, net moment about its centre of mass
	V3 w = v3(0, 0, 0), al = v3(0, 0, 0), a = a0, op = v3(0, 0, 0);
#pragma unroll
	for (int j = 0; j < n; j++) {
		const V3 r = K.o[j] - op;  // from the previous joint origin (rigidly attached to body j-1) to this one
		a = a + cross(al, r);
		if (with_velocity) a = a + cross(w, cross(w, r));
		const V3 z = K.z[j];
		if (md.jtype[j] == 1) {
			if (with_velocity) al = al + dq[j] * cross(w, z);
			al = al + ddq[j] * z;
			if (with_velocity) w = w + dq[j] * z;
		} else {
			if (with_velocity) a = a + 2.0 * dq[j] * cross(w, z);
			a = a + ddq[j] * z;
		}
		const V3 rc = K.c[j] - K.o[j];
		V3 ac = a + cross(al, rc);
		if (with_velocity) ac = ac + cross(w, cross(w, rc));
		f[j] = md.mass[j] * ac;
		nn[j] = inertia_mul(K.R[j], md.inertia[j], al);
		if (with_velocity) nn[j] = nn[j] + cross(w, inertia_mul(K.R[j], md.inertia[j], w));
		op = K.o[j];
	}
	V3 F = v3(0, 0, 0), N = v3(0, 0, 0);  // force / moment (about o_j) transmitted through joint j
#pragma unroll
	for (int jj = 0; jj < n; jj++) {
		const int j = n - 1 - jj;
		if (j < n - 1) N = N + cross(K.o[j + 1] - K.o[j], F);  // shift the child's wrench from o_{j+1} to o_j
		F = F + f[j];
		N = N + nn[j] + cross(K.c[j] - K.o[j], f[j]);
		tau[j] = md.jtype[j] == 1 ? dot(K.z[j], N) : dot(K.z[j], F);
	}
}

}  // namespace

// Joint-space inertia from composite rigid bodies in world coordinates (serial chain: every later body is a descendant).
// Spatial inertia about the world origin O: mass m, first moment hm = m c, rotational inertia Io = R I R^T + m (c.c 1 - c c^T).
// Column j: momentum of the composite body j.. under the unit motion of joint j, (p, L_O) = Ic_j s_j with s_j = (z_j, o_j x z_j) for a
// revolute and (0, z_j) for a prismatic joint; M_ij = s_i . (L_O, p) for i <= j.
template <int NMAX, bool EXACT>
__device__ __forceinline__ void mass_matrix_crb(const ModelDev& md, int n_rt, const Chain<NMAX>& K, double (&M)[NMAX][NMAX]) {
	const int n = EXACT ? NMAX : n_rt;
	double cm = 0.0;
	V3 ch = v3(0, 0, 0);
	double cI[6] = {0, 0, 0, 0, 0, 0};  // xx yy zz xy xz yz
#pragma unroll
	for (int jj = 0; jj < n; jj++) {
		const int j = n - 1 - jj;
		// add body j to the composite
		const double m = md.mass[j];
		const V3 c = K.c[j];
		const double* R = K.R[j];
		const double* I6 = md.inertia[j];
		// R I R^T
		double RI[9];
#pragma unroll
		for (int r = 0; r < 3; r++) {
			RI[3 * r + 0] = R[3 * r] * I6[0] + R[3 * r + 1] * I6[3] + R[3 * r + 2] * I6[4];
			RI[3 * r + 1] = R[3 * r] * I6[3] + R[3 * r + 1] * I6[1] + R[3 * r + 2] * I6[5];
			RI[3 * r + 2] = R[3 * r] * I6[4] + R[3 * r + 1] * I6[5] + R[3 * r + 2] * I6[2];
		}
		const double cc = dot(c, c);
		cI[0] += RI[0] * R[0] + RI[1] * R[1] + RI[2] * R[2] + m * (cc - c.x * c.x);
		cI[1] += RI[3] * R[3] + RI[4] * R[4] + RI[5] * R[5] + m * (cc - c.y * c.y);
		cI[2] += RI[6] * R[6] + RI[7] * R[7] + RI[8] * R[8] + m * (cc - c.z * c.z);
		cI[3] += RI[0] * R[3] + RI[1] * R[4] + RI[2] * R[5] - m * c.x * c.y;
		cI[4] += RI[0] * R[6] + RI[1] * R[7] + RI[2] * R[8] - m * c.x * c.z;
		cI[5] += RI[3] * R[6] + RI[4] * R[7] + RI[5] * R[8] - m * c.y * c.z;
		cm += m;
		ch = ch + m * c;
		// unit motion of joint j
		const bool rev = md.jtype[j] == 1;
		const V3 wj = rev ? K.z[j] : v3(0, 0, 0);
		const V3 vj = rev ? cross(K.o[j], K.z[j]) : K.z[j];
		const V3 p = cm * vj + cross(wj, ch);
		const V3 L = v3(cI[0] * wj.x + cI[3] * wj.y + cI[4] * wj.z, cI[3] * wj.x + cI[1] * wj.y + cI[5] * wj.z, cI[4] * wj.x + cI[5] * wj.y + cI[2] * wj.z) +
					 cross(ch, vj);
#pragma unroll
		for (int i = 0; i < n; i++) {
			if (i > j) continue;
			const bool ri = md.jtype[i] == 1;
			const V3 wi = ri ? K.z[i] : v3(0, 0, 0);
			const V3 vi = ri ? cross(K.o[i], K.z[i]) : K.z[i];
			const double v = dot(wi, L) + dot(vi, p);
			M[i][j] = v;
			M[j][i] = v;
		}
	}
}

template <int NMAX, bool EXACT>
__global__ void __launch_bounds__(64) saip_integrate_kernel(const SimParams S) {
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= S.B) return;
	const ModelDev& md = *S.model;
	const int n = EXACT ? NMAX : S.n;
	double q[NMAX], dq[NMAX], tau[NMAX], h[NMAX], ddq[NMAX], e[NMAX];
	double M[NMAX][NMAX];
#pragma unroll
	for (int j = 0; j < n; j++) {
		q[j] = S.q[(size_t)j * S.ld + b];
		dq[j] = S.dq[(size_t)j * S.ld + b];
		tau[j] = S.tau[(size_t)j * S.ld + b];
		if (!(tau[j] == tau[j])) tau[j] = 0.0;  // flagged instances (NaN torques) coast
		e[j] = 0.0;
		ddq[j] = 0.0;
	}
	const V3 a0 = v3(-S.gravity[0], -S.gravity[1], -S.gravity[2]);
	Chain<NMAX> K;
	for (int step = 0; step < S.substeps; step++) {
		chain_fk<NMAX>(md, n, q, K);
		rnea<NMAX>(md, n, K, dq, e, a0, true, h);  // b(q, dq) + g(q)
		mass_matrix_crb<NMAX, EXACT>(md, n, K, M);
		// Cholesky M = L L^T (lower, in place), then two triangular solves
#pragma unroll
		for (int k = 0; k < n; k++) {
			double d = M[k][k];
#pragma unroll
			for (int l = 0; l < n; l++)
				if (l < k) d -= M[k][l] * M[k][l];
			d = sqrt(d);
			M[k][k] = d;
			const double rd = 1.0 / d;
#pragma unroll
			for (int i = 0; i < n; i++) {
				if (i <= k) continue;
				double s = M[i][k];
#pragma unroll
				for (int l = 0; l < n; l++)
					if (l < k) s -= M[i][l] * M[k][l];
				M[i][k] = s * rd;
			}
		}
#pragma unroll
		for (int i = 0; i < n; i++) {
			double s = tau[i] - h[i] - S.damping * dq[i];
#pragma unroll
			for (int l = 0; l < n; l++)
				if (l < i) s -= M[i][l] * ddq[l];
			ddq[i] = s / M[i][i];
		}
#pragma unroll
		for (int ii = 0; ii < n; ii++) {
			const int i = n - 1 - ii;
			double s = ddq[i];
#pragma unroll
			for (int l = 0; l < n; l++)
				if (l > i) s -= M[l][i] * ddq[l];
			ddq[i] = s / M[i][i];
		}
#pragma unroll
		for (int j = 0; j < n; j++) {
			dq[j] += S.dt * ddq[j];
			q[j] += S.dt * dq[j];
		}
	}
#pragma unroll
	for (int j = 0; j < n; j++) {
		S.q[(size_t)j * S.ld + b] = q[j];
		S.dq[(size_t)j * S.ld + b] = dq[j];
		if (S.ddq) S.ddq[(size_t)j * S.ld + b] = ddq[j];
	}
}

hipError_t launch_integrate_oct(const SimParams& S, hipStream_t stream);  // saip_dynamics_oct.hip: eight lanes per instance, 7-dof chains

hipError_t launch_integrate(const SimParams& S, hipStream_t stream) {
	// 7-dof chains: the eight-lanes-per-instance kernel is faster at every batch size measured (8.7 vs 33.4 us at 4096, 46.8 vs 81.1 us at
	// 65 536, 166 vs 218 us at 262 144 per two substeps); SAIP_OCT_DYN_MAX_BATCH=0 selects the lane-per-instance kernel for comparisons
	static const char* oct_env = getenv("SAIP_OCT_DYN_MAX_BATCH");
	if (S.n == 7 && (!oct_env || S.B <= atoi(oct_env))) return launch_integrate_oct(S, stream);
	const int grid = (S.B + 63) / 64;
	if (S.n == 7) hipLaunchKernelGGL((saip_integrate_kernel<7, true>), dim3(grid), dim3(64), 0, stream, S);   // Panda-class arms: unrolled
	else if (S.n <= 8) hipLaunchKernelGGL((saip_integrate_kernel<8, false>), dim3(grid), dim3(64), 0, stream, S);
	else hipLaunchKernelGGL((saip_integrate_kernel<32, false>), dim3(grid), dim3(64), 0, stream, S);
	return hipGetLastError();
}

}  // namespace saip

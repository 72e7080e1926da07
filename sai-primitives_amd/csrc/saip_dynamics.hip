// Batched forward dynamics + semi-implicit Euler step: the "step after the path" of SURVEY.md 8(f) f4, so that closed-loop
// rollouts (OTG -> control cycle -> dynamics) stay on the device.  It stands in for what the reference's examples do with the
// external physics engine: sim->setJointTorques(...); sim->integrate()   (/root/reference/examples/05-using_robot_controller/
// 05-using_robot_controller.cpp:225-231; sai-simulation is not part of the reference tree, so there is no reference arithmetic to
// match -- the oracle is the Lagrangian restatement in workloads.forward_dynamics, see tests/test_gpu_dynamics.py).
//
//   M(q) qdd + b(q, dq) + g(q) = tau        b: Coriolis / centrifugal, g: gravity
//   dq <- dq + dt qdd ;  q <- q + dt dq     (semi-implicit Euler, `substeps` times per call with the torque held)
//
// One lane per instance.  Bias forces by the recursive Newton-Euler pass in world coordinates, M column by column from the same
// backward pass with unit joint accelerations (O(n^2) per instance, n <= 32), Cholesky solve.  Per-body arrays are lane-private
// (scratch for NMAX = 32, mostly registers for NMAX = 8); the kernel is FP64-latency bound like the cycle kernels and is not on the
// benchmarked path.
#include <hip/hip_runtime.h>
#include <math.h>

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
	for (int j = n - 1; j >= 0; j--) {
		if (j < n - 1) N = N + cross(K.o[j + 1] - K.o[j], F);  // shift the child's wrench from o_{j+1} to o_j
		F = F + f[j];
		N = N + nn[j] + cross(K.c[j] - K.o[j], f[j]);
		tau[j] = md.jtype[j] == 1 ? dot(K.z[j], N) : dot(K.z[j], F);
	}
}

}  // namespace

template <int NMAX>
__global__ void __launch_bounds__(64) saip_integrate_kernel(const SimParams S) {
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= S.B) return;
	const ModelDev& md = *S.model;
	const int n = S.n;
	double q[NMAX], dq[NMAX], tau[NMAX], h[NMAX], ddq[NMAX], e[NMAX], col[NMAX];
	double M[NMAX][NMAX];
	for (int j = 0; j < n; j++) {
		q[j] = S.q[(size_t)j * S.ld + b];
		dq[j] = S.dq[(size_t)j * S.ld + b];
		tau[j] = S.tau[(size_t)j * S.ld + b];
		if (!(tau[j] == tau[j])) tau[j] = 0.0;  // flagged instances (NaN torques) coast
	}
	const V3 a0 = v3(-S.gravity[0], -S.gravity[1], -S.gravity[2]);
	Chain<NMAX> K;
	for (int step = 0; step < S.substeps; step++) {
		chain_fk<NMAX>(md, n, q, K);
		for (int j = 0; j < n; j++) e[j] = 0.0;
		rnea<NMAX>(md, n, K, dq, e, a0, true, h);  // b(q, dq) + g(q)
		for (int i = 0; i < n; i++) {              // M[:, i] = RNEA(ddq = e_i) without velocity and gravity terms
			e[i] = 1.0;
			rnea<NMAX>(md, n, K, dq, e, v3(0, 0, 0), false, col);
			e[i] = 0.0;
			for (int j = 0; j < n; j++) M[j][i] = col[j];
		}
		// Cholesky M = L L^T (lower, in place), then two triangular solves
		for (int k = 0; k < n; k++) {
			double d = M[k][k];
			for (int l = 0; l < k; l++) d -= M[k][l] * M[k][l];
			d = sqrt(d);
			M[k][k] = d;
			for (int i = k + 1; i < n; i++) {
				double s = M[i][k];
				for (int l = 0; l < k; l++) s -= M[i][l] * M[k][l];
				M[i][k] = s / d;
			}
		}
		for (int i = 0; i < n; i++) {
			double s = tau[i] - h[i] - S.damping * dq[i];
			for (int l = 0; l < i; l++) s -= M[i][l] * ddq[l];
			ddq[i] = s / M[i][i];
		}
		for (int i = n - 1; i >= 0; i--) {
			double s = ddq[i];
			for (int l = i + 1; l < n; l++) s -= M[l][i] * ddq[l];
			ddq[i] = s / M[i][i];
		}
		for (int j = 0; j < n; j++) {
			dq[j] += S.dt * ddq[j];
			q[j] += S.dt * dq[j];
		}
	}
	for (int j = 0; j < n; j++) {
		S.q[(size_t)j * S.ld + b] = q[j];
		S.dq[(size_t)j * S.ld + b] = dq[j];
		if (S.ddq) S.ddq[(size_t)j * S.ld + b] = ddq[j];
	}
}

hipError_t launch_integrate(const SimParams& S, hipStream_t stream) {
	const int grid = (S.B + 63) / 64;
	if (S.n <= 8) hipLaunchKernelGGL((saip_integrate_kernel<8>), dim3(grid), dim3(64), 0, stream, S);
	else hipLaunchKernelGGL((saip_integrate_kernel<32>), dim3(grid), dim3(64), 0, stream, S);
	return hipGetLastError();
}

}  // namespace saip

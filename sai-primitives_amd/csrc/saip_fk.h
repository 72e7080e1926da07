// Serial-chain forward kinematics of one control frame for one instance (straight walk over the chain; used by the
// re-initialisation paths only -- the cycle kernels have their own fused kinematics).
// SaiModel::positionInWorld / rotationInWorld call sites: MotionForceTask.cpp:212-216, 286-289.
#pragma once
#include "saip_device.h"

namespace saip {

// sin/cos for joint angles: Cody-Waite reduction by pi/2 (two FMA terms, exact for |x| < 1e5) + fdlibm kernel polynomials on
// [-pi/4, pi/4] (errors < 1 ulp); the rare |x| >= 1e5 takes the library path.  ~35 instructions instead of ~110.
__device__ __forceinline__ void sincos_joint(const double x, double* sn, double* cs) {
	if (!(fabs(x) < 1.0e5)) {
		sincos(x, sn, cs);
		return;
	}
	const double k = rint(x * 6.36619772367581382433e-01);
	double r = fma(-k, 1.57079632673412561417e+00, x);
	r = fma(-k, 6.07710050650619224932e-11, r);
	const double z = r * r;
	const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03);
	const double s0 = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
	const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
	const double c0 = 1.0 - fma(0.5, z, -z * pc);
	const int q = (int)k & 3;
	const double sa = (q & 1) ? c0 : s0, ca = (q & 1) ? s0 : c0;
	*sn = (q & 2) ? -sa : sa;
	*cs = ((q + 1) & 2) ? -ca : ca;
}

__device__ __forceinline__ void fk_mat3_vec(const double* R, const double* v, double* o) {
	o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
	o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
	o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}

// world position of the control point and world rotation of the control frame of motion-force task tk
__device__ inline void fk_control_frame(const ModelDev& md, const TaskDev& tk, const double* q, int ld, int b, double pos[3], double Rc[9]) {
	double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
	for (int j = 0; j <= tk.body; j++) {
		double t3[3], Rn[9], Rt[9];
		fk_mat3_vec(R, md.p0[j], t3);
		for (int e = 0; e < 3; e++) o[e] += t3[e];
		for (int r = 0; r < 3; r++)
			for (int c = 0; c < 3; c++) Rt[3 * r + c] = R[3 * r] * md.R0[j][c] + R[3 * r + 1] * md.R0[j][3 + c] + R[3 * r + 2] * md.R0[j][6 + c];
		const double qj = q[(size_t)j * ld + b];
		const double ax = md.axis[j][0], ay = md.axis[j][1], az = md.axis[j][2];
		if (md.jtype[j] == 1) {
			double s, c;
			sincos(qj, &s, &c);  // library sin/cos on purpose: this pose seeds the OTG state, whose knife-edge decisions (collinearity
			                     // within 2^-52) are compared against reference fixtures generated with the same arithmetic
			const double v = 1.0 - c;
			const double Rq[9] = {c + ax * ax * v,      ax * ay * v - az * s, ax * az * v + ay * s,
								  ay * ax * v + az * s, c + ay * ay * v,      ay * az * v - ax * s,
								  az * ax * v - ay * s, az * ay * v + ax * s, c + az * az * v};
			for (int r = 0; r < 3; r++)
				for (int c2 = 0; c2 < 3; c2++) Rn[3 * r + c2] = Rt[3 * r] * Rq[c2] + Rt[3 * r + 1] * Rq[3 + c2] + Rt[3 * r + 2] * Rq[6 + c2];
		} else {
			const double a[3] = {ax, ay, az};
			double d[3];
			fk_mat3_vec(Rt, a, d);
			for (int e = 0; e < 3; e++) o[e] += d[e] * qj;
			for (int e = 0; e < 9; e++) Rn[e] = Rt[e];
		}
		for (int e = 0; e < 9; e++) R[e] = Rn[e];
	}
	double p[3];
	fk_mat3_vec(R, tk.pos, p);
	for (int e = 0; e < 3; e++) pos[e] = o[e] + p[e];
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++) Rc[3 * r + c] = R[3 * r] * tk.rot[c] + R[3 * r + 1] * tk.rot[3 + c] + R[3 * r + 2] * tk.rot[6 + c];
}

}  // namespace saip

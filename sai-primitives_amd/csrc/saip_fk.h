// Serial-chain forward kinematics of one control frame for one instance (straight walk over the chain; used by the
// re-initialisation paths only -- the cycle kernels have their own fused kinematics).
// SaiModel::positionInWorld / rotationInWorld call sites: MotionForceTask.cpp:212-216, 286-289.
#pragma once
#include "saip_device.h"

namespace saip {

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
			sincos(qj, &s, &c);
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

// ---------------------------------------------------------------------------------------------------------------------------------
// THIRD-PARTY NOTICE.  The closed-form profile expressions, root solvers and case structure in this file are derived from Ruckig 0.10.1
// (https://github.com/pantor/ruckig, vendored by the reference under ruckig/), which is distributed under the MIT License:
//
//   MIT License -- Copyright (c) 2021 Lars Berscheid
//
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated documentation files
//   (the "Software"), to deal in the Software without restriction, including without limitation the rights to use, copy, modify, merge,
//   publish, distribute, sublicense, and/or sell copies of the Software, and to permit persons to whom the Software is furnished to do
//   so, subject to the following conditions:
//
//   The above copyright notice and this permission notice shall be included in all copies or substantial portions of the Software.
//
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF
//   MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT.  IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE
//   FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN
//   CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
//
// See THIRD_PARTY.md at the repository root.
// ---------------------------------------------------------------------------------------------------------------------------------
// Jerk-limited (third-order) online trajectory generation for ONE degree of freedom: the closed forms of the reference's vendored
// Ruckig 0.10.1 position interface, restated as plain per-lane functions (no containers, no exceptions, no heap) so that a GPU lane can
// evaluate them.  What each piece follows:
//   o3_check            Profile::check / check_with_timing          /root/reference/ruckig/include/ruckig/profile.hpp:186-282
//   o3_solve_cub / o3_solve_quart / o3_shrink_interval / poly helpers                    ruckig/include/ruckig/roots.hpp:59-397
//   o3_brake            BrakeProfile::get_position_brake_trajectory + finalize            ruckig/src/ruckig/brake.cpp:5-80, brake.hpp:41-62
//   O3Step1             PositionThirdOrderStep1 (extremal profiles of one dof)             ruckig/src/ruckig/position-third-step1.cpp
//   o3_block            Block::calculate_block (up to two blocked intervals)               ruckig/include/ruckig/block.hpp:60-134
//   O3Step2             PositionThirdOrderStep2 (profile of a given duration)              ruckig/src/ruckig/position-third-step2.cpp
//   o3_sample           Trajectory::at_time for one dof                                    ruckig/include/ruckig/trajectory.hpp:65-143
// The wrappers that reach this mode: JointTask::enableInternalOtgJerkLimited (/root/reference/src/tasks/JointTask.cpp:383-410),
// MotionForceTask::enableInternalOtgJerkLimited (MotionForceTask.cpp:525-545), OTG_joints::setMaxJerk (OTG_joints.cpp:73-86).
//
// The expressions are the reference's, term for term (these are solutions of polynomial systems: there is no second way to write
// them), evaluated in the same order with FMA contraction off, so the many threshold decisions (1e-8 ... DBL_EPSILON) fall the same
// way.  The file compiles for the device (saip_otg.hip) and, unchanged, for the host: oracle/ref_ruckig/otg3_compare.cpp runs every
// function here against the reference's own classes on random inputs (test infrastructure; `make -C oracle ref`).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define O3_FN __device__ __forceinline__
#define O3_NI __device__ __noinline__
#else
#define O3_FN inline
#define O3_NI inline
#endif

namespace saip {
namespace otg3 {

constexpr double O3_EPS = 2.220446049250313e-16;  // DBL_EPSILON
constexpr double O3_INF = __builtin_huge_val();
constexpr double O3_TOL = 1e-14;                  // roots::tolerance

enum { UDDU = 0, UDUD = 1 };
enum { L_ACC0_ACC1_VEL = 0, L_VEL, L_ACC0, L_ACC1, L_ACC0_ACC1, L_ACC0_VEL, L_ACC1_VEL, L_NONE };

O3_FN double o3_abs(double x) { return __builtin_fabs(x); }
O3_FN double o3_min(double a, double b) { return (b < a) ? b : a; }  // std::min
O3_FN double o3_max(double a, double b) { return (a < b) ? b : a;  }  // std::max
O3_FN double o3_pow2(double v) { return v * v; }

// one dof's profile: seven segments of constant jerk
struct P3 {
	double t[7], j[7], a[8], v[8], p[8];
	double tsum;        // t_sum.back()
	double pf, vf, af;  // target state
	int dir;            // 0 UP, 1 DOWN
	int cs, lim;        // control signs, reached limits
};

O3_FN void o3_zero_t(P3& pr) {
	for (int i = 0; i < 7; i++) pr.t[i] = 0.0;
}

// Profile::check<control_signs, limits, set_limits>(jf, vMax, vMin, aMax, aMin), profile.hpp:186-270
template <int CS, int LIM, bool SETLIM>
O3_FN bool o3_check(P3& pr, double jf, double vMax, double vMin, double aMax, double aMin) {
	if (pr.t[0] < 0) return false;
	double ts = pr.t[0];
	for (int i = 0; i < 6; i++) {
		if (pr.t[i + 1] < 0) return false;
		ts = ts + pr.t[i + 1];
	}
	pr.tsum = ts;
	if (LIM == L_ACC0_ACC1_VEL || LIM == L_ACC0_VEL || LIM == L_ACC1_VEL || LIM == L_VEL) {
		if (pr.t[3] < O3_EPS) return false;
	}
	if (LIM == L_ACC0 || LIM == L_ACC0_ACC1) {
		if (pr.t[1] < O3_EPS) return false;
	}
	if (LIM == L_ACC1 || LIM == L_ACC0_ACC1) {
		if (pr.t[5] < O3_EPS) return false;
	}
	if (ts > 1e12) return false;
	pr.j[0] = pr.t[0] > 0 ? jf : 0.0;
	pr.j[1] = 0.0;
	pr.j[2] = pr.t[2] > 0 ? -jf : 0.0;
	pr.j[3] = 0.0;
	pr.j[5] = 0.0;
	if (CS == UDDU) {
		pr.j[4] = pr.t[4] > 0 ? -jf : 0.0;
		pr.j[6] = pr.t[6] > 0 ? jf : 0.0;
	} else {
		pr.j[4] = pr.t[4] > 0 ? jf : 0.0;
		pr.j[6] = pr.t[6] > 0 ? -jf : 0.0;
	}
	pr.dir = (vMax > 0) ? 0 : 1;
	const double vUppLim = (pr.dir == 0 ? vMax : vMin) + 1e-12;
	const double vLowLim = (pr.dir == 0 ? vMin : vMax) - 1e-12;
	for (int i = 0; i < 7; i++) {
		pr.a[i + 1] = pr.a[i] + pr.t[i] * pr.j[i];
		pr.v[i + 1] = pr.v[i] + pr.t[i] * (pr.a[i] + pr.t[i] * pr.j[i] / 2);
		pr.p[i + 1] = pr.p[i] + pr.t[i] * (pr.v[i] + pr.t[i] * (pr.a[i] / 2 + pr.t[i] * pr.j[i] / 6));
		if (LIM == L_ACC0_ACC1_VEL || LIM == L_ACC0_ACC1 || LIM == L_ACC0_VEL || LIM == L_ACC1_VEL || LIM == L_VEL) {
			if (i == 2) pr.a[3] = 0.0;
		}
		if (SETLIM) {
			if (LIM == L_ACC1) {
				if (i == 2) pr.a[3] = aMin;
			}
			if (LIM == L_ACC0_ACC1) {
				if (i == 0) pr.a[1] = aMax;
				if (i == 4) pr.a[5] = aMin;
			}
		}
		if (i > 1 && pr.a[i + 1] * pr.a[i] < -O3_EPS) {
			const double v_a_zero = pr.v[i] - (pr.a[i] * pr.a[i]) / (2 * pr.j[i]);
			if (v_a_zero > vUppLim || v_a_zero < vLowLim) return false;
		}
	}
	pr.cs = CS;
	pr.lim = LIM;
	const double aUppLim = (pr.dir == 0 ? aMax : aMin) + 1e-12;
	const double aLowLim = (pr.dir == 0 ? aMin : aMax) - 1e-12;
	return o3_abs(pr.p[7] - pr.pf) < 1e-8 && o3_abs(pr.v[7] - pr.vf) < 1e-8 && o3_abs(pr.a[7] - pr.af) < 1e-10 && pr.a[1] >= aLowLim &&
		   pr.a[3] >= aLowLim && pr.a[5] >= aLowLim && pr.a[1] <= aUppLim && pr.a[3] <= aUppLim && pr.a[5] <= aUppLim && pr.v[3] <= vUppLim &&
		   pr.v[4] <= vUppLim && pr.v[5] <= vUppLim && pr.v[6] <= vUppLim && pr.v[3] >= vLowLim && pr.v[4] >= vLowLim && pr.v[5] >= vLowLim &&
		   pr.v[6] >= vLowLim;
}
// check_with_timing(tf, jf, ..., jMax): profile.hpp:272-282 (the time itself is not checked there)
template <int CS, int LIM>
O3_FN bool o3_check_jf(P3& pr, double jf, double vMax, double vMin, double aMax, double aMin, double jMax) {
	return (o3_abs(jf) < o3_abs(jMax) + 1e-12) && o3_check<CS, LIM, false>(pr, jf, vMax, vMin, aMax, aMin);
}

// ---------------------------------------------------------------- roots.hpp
struct Roots {  // PositiveSet<double, 4>: values >= 0, sorted on first iteration
	double x[4];
	int n;
	O3_FN void clear() { n = 0; }
	O3_FN void insert(double v) {
		if (v >= 0) {
			x[n] = v;
			n++;
		}
	}
	O3_FN void sort() {  // std::sort on at most four values
		for (int i = 1; i < n; i++)
			for (int k = i; k > 0 && x[k] < x[k - 1]; k--) {
				const double tmp = x[k];
				x[k] = x[k - 1];
				x[k - 1] = tmp;
			}
	}
};

// all non-negative roots of a x^3 + b x^2 + c x + d, roots.hpp:59-146
O3_FN Roots o3_solve_cub(double a, double b, double c, double d) {
	Roots r;
	r.clear();
	if (o3_abs(d) < O3_EPS) {
		r.insert(0.0);
		d = c;
		c = b;
		b = a;
		a = 0.0;
	}
	if (o3_abs(a) < O3_EPS) {
		if (o3_abs(b) < O3_EPS) {
			if (o3_abs(c) > O3_EPS) r.insert(-d / c);
		} else {
			const double discriminant = c * c - 4 * b * d;
			if (discriminant >= 0) {
				const double inv2b = 1.0 / (2 * b);
				const double y = sqrt(discriminant);
				r.insert((-c + y) * inv2b);
				r.insert((-c - y) * inv2b);
			}
		}
	} else {
		const double inva = 1.0 / a;
		const double invaa = inva * inva;
		const double bb = b * b;
		const double bover3a = b * inva / 3;
		const double p = (a * c - bb / 3) * invaa;
		const double halfq = (2 * bb * b - 9 * a * b * c + 27 * a * a * d) / 54 * invaa * inva;
		const double yy = p * p * p / 27 + halfq * halfq;
		const double cos120 = -0.50, sin120 = 0.866025403784438646764;
		if (yy > O3_EPS) {
			const double y = sqrt(yy);
			const double uuu = -halfq + y;
			const double vvv = -halfq - y;
			const double www = o3_abs(uuu) > o3_abs(vvv) ? uuu : vvv;
			const double w = cbrt(www);
			r.insert(w - p / (3 * w) - bover3a);
		} else if (yy < -O3_EPS) {
			const double x = -halfq;
			const double y = sqrt(-yy);
			double theta, rr;
			if (o3_abs(x) > O3_EPS) {
				theta = (x > 0.0) ? atan(y / x) : (atan(y / x) + M_PI);
				rr = sqrt(x * x - yy);
			} else {
				theta = M_PI / 2;
				rr = y;
			}
			theta /= 3;
			rr = 2 * cbrt(rr);
			const double ux = cos(theta) * rr;
			const double uyi = sin(theta) * rr;
			r.insert(ux - bover3a);
			r.insert(ux * cos120 - uyi * sin120 - bover3a);
			r.insert(ux * cos120 + uyi * sin120 - bover3a);
		} else {
			const double www = -halfq;
			const double w = 2 * cbrt(www);
			r.insert(w - bover3a);
			r.insert(w * cos120 - bover3a);
		}
	}
	r.sort();
	return r;
}

// solveResolvent, roots.hpp:151-190
O3_FN int o3_solve_resolvent(double x[3], double a, double b, double c) {
	const double cos120 = -0.50, sin120 = 0.866025403784438646764;
	a /= 3;
	const double a2 = a * a;
	double q = a2 - b / 3;
	const double r = (a * (2 * a2 - b) + c) / 2;
	const double r2 = r * r;
	const double q3 = q * q * q;
	if (r2 < q3) {
		const double qsqrt = sqrt(q);
		const double t = o3_min(o3_max(r / (q * qsqrt), -1.0), 1.0);
		q = -2 * qsqrt;
		const double theta = acos(t) / 3;
		const double ux = cos(theta) * q;
		const double uyi = sin(theta) * q;
		x[0] = ux - a;
		x[1] = ux * cos120 - uyi * sin120 - a;
		x[2] = ux * cos120 + uyi * sin120 - a;
		return 3;
	}
	double A = -cbrt(o3_abs(r) + sqrt(r2 - q3));
	if (r < 0.0) A = -A;
	const double B = (0.0 == A ? 0.0 : q / A);
	x[0] = (A + B) - a;
	x[1] = -(A + B) / 2 - a;
	x[2] = sqrt(3.0) * (A - B) / 2;
	if (o3_abs(x[2]) < O3_EPS) {
		x[2] = x[1];
		return 2;
	}
	return 1;
}

// all non-negative roots of x^4 + a x^3 + b x^2 + c x + d, roots.hpp:193-282
O3_NI Roots o3_solve_quart(double a, double b, double c, double d) {
	Roots r;
	r.clear();
	if (o3_abs(d) < O3_EPS) {
		if (o3_abs(c) < O3_EPS) {
			r.insert(0.0);
			const double D = a * a - 4 * b;
			if (o3_abs(D) < O3_EPS) {
				r.insert(-a / 2);
			} else if (D > 0.0) {
				const double sqrtD = sqrt(D);
				r.insert((-a - sqrtD) / 2);
				r.insert((-a + sqrtD) / 2);
			}
			r.sort();
			return r;
		}
		if (o3_abs(a) < O3_EPS && o3_abs(b) < O3_EPS) {
			r.insert(0.0);
			r.insert(-cbrt(c));
			r.sort();
			return r;
		}
	}
	const double a3 = -b;
	const double b3 = a * c - 4 * d;
	const double c3 = -a * a * d - c * c + 4 * b * d;
	double x3[3];
	const int number_zeroes = o3_solve_resolvent(x3, a3, b3, c3);
	double y = x3[0];
	if (number_zeroes != 1) {
		if (o3_abs(x3[1]) > o3_abs(y)) y = x3[1];
		if (o3_abs(x3[2]) > o3_abs(y)) y = x3[2];
	}
	double q1, q2, p1, p2;
	double D = y * y - 4 * d;
	if (o3_abs(D) < O3_EPS) {
		q1 = q2 = y / 2;
		D = a * a - 4 * (b - y);
		if (o3_abs(D) < O3_EPS) {
			p1 = p2 = a / 2;
		} else {
			const double sqrtD = sqrt(D);
			p1 = (a + sqrtD) / 2;
			p2 = (a - sqrtD) / 2;
		}
	} else {
		const double sqrtD = sqrt(D);
		q1 = (y + sqrtD) / 2;
		q2 = (y - sqrtD) / 2;
		p1 = (a * q1 - c) / (q1 - q2);
		p2 = (c - a * q2) / (q1 - q2);
	}
	const double eps = 16 * O3_EPS;
	D = p1 * p1 - 4 * q1;
	if (o3_abs(D) < eps) {
		r.insert(-p1 / 2);
	} else if (D > 0.0) {
		const double sqrtD = sqrt(D);
		r.insert((-p1 - sqrtD) / 2);
		r.insert((-p1 + sqrtD) / 2);
	}
	D = p2 * p2 - 4 * q2;
	if (o3_abs(D) < eps) {
		r.insert(-p2 / 2);
	} else if (D > 0.0) {
		const double sqrtD = sqrt(D);
		r.insert((-p2 - sqrtD) / 2);
		r.insert((-p2 + sqrtD) / 2);
	}
	r.sort();
	return r;
}

// polyEval, roots.hpp:292-318 (coefficients highest order first)
template <int N>
O3_FN double o3_poly_eval(const double (&p)[N], double x) {
	double ret = 0.0;
	if (o3_abs(x) < O3_EPS) {
		ret = p[N - 1];
	} else if (x == 1.0) {
		for (int i = N - 1; i >= 0; i--) ret += p[i];
	} else {
		double xn = 1.0;
		for (int i = N - 1; i >= 0; i--) {
			ret += p[i] * xn;
			xn *= x;
		}
	}
	return ret;
}
template <int N>
O3_FN void o3_poly_deri(const double (&c)[N], double (&d)[N - 1]) {  // polyDeri, :321-328
	for (int i = 0; i < N - 1; i++) d[i] = (N - 1 - i) * c[i];
}
template <int N>
O3_FN void o3_poly_monic_deri(const double (&c)[N], double (&d)[N - 1]) {  // polyMonicDeri, :330-338
	d[0] = 1.0;
	for (int i = 1; i < N - 1; i++) d[i] = (N - 1 - i) * c[i] / (N - 1);
}
// shrinkInterval, roots.hpp:345-397: one zero of p inside [l, h] by safeguarded Newton
template <int N>
O3_FN double o3_shrink_interval(const double (&p)[N], double l, double h) {
	const double fl = o3_poly_eval(p, l);
	const double fh = o3_poly_eval(p, h);
	if (fl == 0.0) return l;
	if (fh == 0.0) return h;
	if (fl > 0.0) {
		const double tmp = l;
		l = h;
		h = tmp;
	}
	double rts = (l + h) / 2;
	double dxold = o3_abs(h - l);
	double dx = dxold;
	double deriv[N - 1];
	o3_poly_deri(p, deriv);
	double f = o3_poly_eval(p, rts);
	double df = o3_poly_eval(deriv, rts);
	double temp;
	for (int it = 0; it < 128; it++) {
		if ((((rts - h) * df - f) * ((rts - l) * df - f) > 0.0) || (o3_abs(2 * f) > o3_abs(dxold * df))) {
			dxold = dx;
			dx = (h - l) / 2;
			rts = l + dx;
			if (l == rts) break;
		} else {
			dxold = dx;
			dx = f / df;
			temp = rts;
			rts -= dx;
			if (temp == rts) break;
		}
		if (o3_abs(dx) < O3_TOL) break;
		f = o3_poly_eval(p, rts);
		df = o3_poly_eval(deriv, rts);
		if (f < 0.0) l = rts;
		else h = rts;
	}
	return rts;
}

// ---------------------------------------------------------------- brake pre-trajectory (third order), brake.cpp:5-80, brake.hpp:41-62
struct Brake3 {
	double t[2], j[2], a[2], v[2], p[2];
	double duration;
};
O3_FN double o3_v_at_t(double v0, double a0, double j, double t) { return v0 + t * (a0 + j * t / 2); }
O3_FN double o3_v_at_a_zero(double v0, double a0, double j) { return v0 + (a0 * a0) / (2 * j); }
O3_FN void o3_velocity_brake(Brake3& b, double v0, double a0, double vMax, double vMin, double, double aMin, double jMax) {
	const double eps = 2.2e-14;
	b.j[0] = -jMax;
	const double t_to_a_min = (a0 - aMin) / jMax;
	const double t_to_v_max = a0 / jMax + sqrt(a0 * a0 + 2 * jMax * (v0 - vMax)) / o3_abs(jMax);
	const double t_to_v_min = a0 / jMax + sqrt(a0 * a0 / 2 + jMax * (v0 - vMin)) / o3_abs(jMax);
	const double t_min_to_v_max = o3_min(t_to_v_max, t_to_v_min);
	if (t_to_a_min < t_min_to_v_max) {
		const double v_at_a_min = o3_v_at_t(v0, a0, -jMax, t_to_a_min);
		const double t_to_v_max_with_constant = -(v_at_a_min - vMax) / aMin;
		const double t_to_v_min_with_constant = aMin / (2 * jMax) - (v_at_a_min - vMin) / aMin;
		b.t[0] = o3_max(t_to_a_min - eps, 0.0);
		b.t[1] = o3_max(o3_min(t_to_v_max_with_constant, t_to_v_min_with_constant), 0.0);
	} else {
		b.t[0] = o3_max(t_min_to_v_max - eps, 0.0);
	}
}
O3_FN void o3_acceleration_brake(Brake3& b, double v0, double a0, double vMax, double vMin, double aMax, double aMin, double jMax) {
	const double eps = 2.2e-14;
	b.j[0] = -jMax;
	const double t_to_a_max = (a0 - aMax) / jMax;
	const double t_to_a_zero = a0 / jMax;
	const double v_at_a_max = o3_v_at_t(v0, a0, -jMax, t_to_a_max);
	const double v_at_a_zero = o3_v_at_t(v0, a0, -jMax, t_to_a_zero);
	if ((v_at_a_zero > vMax && jMax > 0) || (v_at_a_zero < vMax && jMax < 0)) {
		o3_velocity_brake(b, v0, a0, vMax, vMin, aMax, aMin, jMax);
	} else if ((v_at_a_max < vMin && jMax > 0) || (v_at_a_max > vMin && jMax < 0)) {
		const double t_to_v_min = -(v_at_a_max - vMin) / aMax;
		const double t_to_v_max = -aMax / (2 * jMax) - (v_at_a_max - vMax) / aMax;
		b.t[0] = t_to_a_max + eps;
		b.t[1] = o3_max(o3_min(t_to_v_min, t_to_v_max - eps), 0.0);
	} else {
		b.t[0] = t_to_a_max + eps;
	}
}
// get_position_brake_trajectory + finalize: ps, vs, as enter as the current state and leave as the state behind the brake
O3_FN void o3_brake(Brake3& b, double& ps, double& vs, double& as, double vMax, double vMin, double aMax, double aMin, double jMax) {
	b.t[0] = b.t[1] = 0.0;
	b.j[0] = b.j[1] = 0.0;
	b.a[0] = b.a[1] = b.v[0] = b.v[1] = b.p[0] = b.p[1] = 0.0;
	b.duration = 0.0;
	const double v0 = vs, a0 = as;
	if (a0 > aMax) {
		o3_acceleration_brake(b, v0, a0, vMax, vMin, aMax, aMin, jMax);
	} else if (a0 < aMin) {
		o3_acceleration_brake(b, v0, a0, vMin, vMax, aMin, aMax, -jMax);
	} else if ((v0 > vMax && o3_v_at_a_zero(v0, a0, -jMax) > vMin) || (a0 > 0 && o3_v_at_a_zero(v0, a0, jMax) > vMax)) {
		o3_velocity_brake(b, v0, a0, vMax, vMin, aMax, aMin, jMax);
	} else if ((v0 < vMin && o3_v_at_a_zero(v0, a0, jMax) < vMax) || (a0 < 0 && o3_v_at_a_zero(v0, a0, -jMax) < vMin)) {
		o3_velocity_brake(b, v0, a0, vMin, vMax, aMin, aMax, -jMax);
	}
	if (b.t[0] <= 0.0 && b.t[1] <= 0.0) {
		b.duration = 0.0;
		return;
	}
	b.duration = b.t[0];
	b.p[0] = ps;
	b.v[0] = vs;
	b.a[0] = as;
	{
		const double t = b.t[0], j = b.j[0];
		const double pn = ps + t * (vs + t * (as / 2 + t * j / 6)), vn = vs + t * (as + t * j / 2), an = as + t * j;  // utils.hpp:43-49
		ps = pn;
		vs = vn;
		as = an;
	}
	if (b.t[1] > 0.0) {
		b.duration += b.t[1];
		b.p[1] = ps;
		b.v[1] = vs;
		b.a[1] = as;
		const double t = b.t[1], j = b.j[1];
		const double pn = ps + t * (vs + t * (as / 2 + t * j / 6)), vn = vs + t * (as + t * j / 2), an = as + t * j;
		ps = pn;
		vs = vn;
		as = an;
	}
}

// ---------------------------------------------------------------- blocked intervals, block.hpp
struct Block3 {
	P3 pmin, pa, pb;  // fastest profile; the profiles that end the blocked intervals a and b
	double t_min, a_left, a_right, b_left, b_right;
	bool has_a, has_b;
};
O3_FN void o3_interval(const P3& L, const P3& R, double brake_dur, double& left, double& right, P3& prof) {  // Block::Interval(profile_left, profile_right), block.hpp:30-42
	const double ld = L.tsum + brake_dur + 0.0, rd = R.tsum + brake_dur + 0.0;
	if (ld < rd) {
		left = ld;
		right = rd;
		prof = R;
	} else {
		left = rd;
		right = ld;
		prof = L;
	}
}
O3_FN void o3_set_min(Block3& blk, const P3& pr, double brake_dur) {  // block.hpp:45-50
	blk.pmin = pr;
	blk.t_min = pr.tsum + brake_dur + 0.0;
	blk.has_a = blk.has_b = false;
	blk.a_left = blk.a_right = blk.b_left = blk.b_right = O3_INF;
}
// Block::calculate_block<6, true>, block.hpp:60-134
O3_NI bool o3_block(Block3& blk, P3 (&vp)[6], int cnt, double brake_dur) {
	if (cnt == 1) {
		o3_set_min(blk, vp[0], brake_dur);
		return true;
	} else if (cnt == 2) {
		if (o3_abs(vp[0].tsum - vp[1].tsum) < 8 * O3_EPS) {
			o3_set_min(blk, vp[0], brake_dur);
			return true;
		}
		const int idx_min = (vp[0].tsum < vp[1].tsum) ? 0 : 1;
		const int idx_else_1 = (idx_min + 1) % 2;
		o3_set_min(blk, vp[idx_min], brake_dur);
		blk.has_a = true;
		o3_interval(vp[idx_min], vp[idx_else_1], brake_dur, blk.a_left, blk.a_right, blk.pa);
		return true;
	} else if (cnt == 4) {
		int rem = -1;
		if (o3_abs(vp[0].tsum - vp[1].tsum) < 32 * O3_EPS && vp[0].dir != vp[1].dir) rem = 1;
		else if (o3_abs(vp[2].tsum - vp[3].tsum) < 256 * O3_EPS && vp[2].dir != vp[3].dir) rem = 3;
		else if (o3_abs(vp[0].tsum - vp[3].tsum) < 256 * O3_EPS && vp[0].dir != vp[3].dir) rem = 3;
		else return false;
		for (int i = rem; i < cnt - 1; i++) vp[i] = vp[i + 1];
		cnt -= 1;
	} else if (cnt % 2 == 0) {
		return false;
	}
	int idx_min = 0;  // std::min_element: the first minimal
	for (int i = 1; i < cnt; i++)
		if (vp[i].tsum < vp[idx_min].tsum) idx_min = i;
	o3_set_min(blk, vp[idx_min], brake_dur);
	if (cnt == 3) {
		const int e1 = (idx_min + 1) % 3, e2 = (idx_min + 2) % 3;
		blk.has_a = true;
		o3_interval(vp[e1], vp[e2], brake_dur, blk.a_left, blk.a_right, blk.pa);
		return true;
	} else if (cnt == 5) {
		const int e1 = (idx_min + 1) % 5, e2 = (idx_min + 2) % 5, e3 = (idx_min + 3) % 5, e4 = (idx_min + 4) % 5;
		blk.has_a = blk.has_b = true;
		if (vp[e1].dir == vp[e2].dir) {
			o3_interval(vp[e1], vp[e2], brake_dur, blk.a_left, blk.a_right, blk.pa);
			o3_interval(vp[e3], vp[e4], brake_dur, blk.b_left, blk.b_right, blk.pb);
		} else {
			o3_interval(vp[e1], vp[e4], brake_dur, blk.a_left, blk.a_right, blk.pa);
			o3_interval(vp[e2], vp[e3], brake_dur, blk.b_left, blk.b_right, blk.pb);
		}
		return true;
	}
	return false;
}

}  // namespace otg3
}  // namespace saip

#include "saip_otg3_step1.h"
#include "saip_otg3_step2.h"

namespace saip {
namespace otg3 {

// Trajectory::state_to_integrate_from + integrate for one dof (trajectory.hpp:65-143, utils.hpp:43-49), one section.
// pr: the stored profile (t, j, a[0..7], p[0], v[0]; p[], v[] rebuilt exactly as check() left them).
O3_FN void o3_sample(double time, double duration, const P3& pr, const Brake3& br, double& pn, double& vn, double& an) {
	double p[8], v[8], ts[7];
	p[0] = pr.p[0];
	v[0] = pr.v[0];
	for (int i = 0; i < 7; i++) {
		v[i + 1] = v[i] + pr.t[i] * (pr.a[i] + pr.t[i] * pr.j[i] / 2);
		p[i + 1] = p[i] + pr.t[i] * (v[i] + pr.t[i] * (pr.a[i] / 2 + pr.t[i] * pr.j[i] / 6));
		ts[i] = (i == 0) ? pr.t[0] : ts[i - 1] + pr.t[i];
	}
	double td, ps, vs, as, js;
	if (time >= duration) {
		td = time - (br.duration + ts[6]);  // t_pre = profile.t_sum.back() + brake.duration (+ accel)
		ps = p[7];
		vs = v[7];
		as = pr.a[7];
		js = 0.0;
	} else {
		td = time;
		bool done = false;
		if (br.duration > 0) {
			if (td < br.duration) {
				const int index = (td < br.t[0]) ? 0 : 1;
				if (index > 0) td = td - br.t[index - 1];
				ps = br.p[index];
				vs = br.v[index];
				as = br.a[index];
				js = br.j[index];
				done = true;
			} else {
				td = td - br.duration;
			}
		}
		if (!done) {
			if (td >= ts[6]) {
				td = td - ts[6];
				ps = p[7];
				vs = v[7];
				as = pr.a[7];
				js = 0.0;
			} else {
				int idx = 0;  // std::upper_bound(t_sum, td): first i with t_sum[i] > td
				for (int i = 6; i >= 0; i--)
					if (ts[i] > td) idx = i;
				if (idx > 0) td = td - ts[idx - 1];
				ps = p[idx];
				vs = v[idx];
				as = pr.a[idx];
				js = pr.j[idx];
			}
		}
	}
	pn = ps + td * (vs + td * (as / 2 + td * js / 6));
	vn = vs + td * (as + td * js / 2);
	an = as + td * js;
}

}  // namespace otg3
}  // namespace saip

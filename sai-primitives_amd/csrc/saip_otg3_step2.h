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
// PositionThirdOrderStep2: a profile of one dof with a GIVEN duration tf (time synchronisation with the limiting dof) --
// /root/reference/ruckig/src/ruckig/position-third-step2.cpp.  Included by saip_otg3.h.
#pragma once

namespace saip {
namespace otg3 {

struct O3Step2 {
	double v0, a0, tf, vf, af, _vMax, _vMin, _aMax, _aMin, _jMax;
	double pd, tf_tf, tf_p3, tf_p4, vd, vd_vd, ad, ad_ad, v0_v0, vf_vf, a0_a0, a0_p3, a0_p4, a0_p5, a0_p6, af_af, af_p3, af_p4, af_p5, af_p6, jMax_jMax, g1, g2;

	O3_FN void init(double tf_, double p0, double v0_, double a0_, double pf, double vf_, double af_, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :10-41
		v0 = v0_;
		a0 = a0_;
		tf = tf_;
		vf = vf_;
		af = af_;
		_vMax = vMax;
		_vMin = vMin;
		_aMax = aMax;
		_aMin = aMin;
		_jMax = jMax;
		pd = pf - p0;
		tf_tf = tf * tf;
		tf_p3 = tf_tf * tf;
		tf_p4 = tf_tf * tf_tf;
		vd = vf - v0;
		vd_vd = vd * vd;
		v0_v0 = v0 * v0;
		vf_vf = vf * vf;
		ad = af - a0;
		ad_ad = ad * ad;
		a0_a0 = a0 * a0;
		af_af = af * af;
		a0_p3 = a0 * a0_a0;
		a0_p4 = a0_a0 * a0_a0;
		a0_p5 = a0_p3 * a0_a0;
		a0_p6 = a0_p4 * a0_a0;
		af_p3 = af * af_af;
		af_p4 = af_af * af_af;
		af_p5 = af_p3 * af_af;
		af_p6 = af_p4 * af_af;
		jMax_jMax = jMax * jMax;
		g1 = -pd + tf * v0;
		g2 = -2 * pd + tf * (v0 + vf);
	}

	O3_NI bool time_acc0_acc1_vel(P3& w, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :43-81
		// UDDU, solution 1
		if ((2 * (aMax - aMin) + ad) / jMax < tf) {
			const double h1 = sqrt((a0_p4 + af_p4 - 4 * a0_p3 * (2 * aMax + aMin) / 3 - 4 * af_p3 * (aMax + 2 * aMin) / 3 + 2 * (a0_a0 - af_af) * aMax * aMax +
									(4 * a0 * aMax - 2 * a0_a0) * (af_af - 2 * af * aMin + (aMin - aMax) * aMin + 2 * jMax * (aMin * tf - vd)) +
									2 * af_af * (aMin * aMin + 2 * jMax * (aMax * tf - vd)) +
									4 * jMax * (2 * aMin * (af * vd + jMax * g1) + (aMax * aMax - aMin * aMin) * vd + jMax * vd_vd) + 8 * aMax * jMax_jMax * (pd - tf * vf)) /
									   (aMax * aMin) +
								   4 * af_af + 2 * a0_a0 + (4 * af + aMax - aMin) * (aMax - aMin) + 4 * jMax * (aMin - aMax + jMax * tf - 2 * af) * tf) *
							  o3_abs(jMax) / jMax;
			w.t[0] = (-a0 + aMax) / jMax;
			w.t[1] = (-(af_af - a0_a0 + 2 * aMax * aMax + aMin * (aMin - 2 * ad - 3 * aMax) + 2 * jMax * (aMin * tf - vd)) + aMin * h1) / (2 * (aMax - aMin) * jMax);
			w.t[2] = aMax / jMax;
			w.t[3] = (aMin - aMax + h1) / (2 * jMax);
			w.t[4] = -aMin / jMax;
			w.t[5] = tf - (w.t[0] + w.t[1] + w.t[2] + w.t[3] + 2 * w.t[4] + af / jMax);
			w.t[6] = w.t[4] + af / jMax;
			if (o3_check<UDDU, L_ACC0_ACC1_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		// UDUD
		if ((-a0 + 4 * aMax - af) / jMax < tf) {
			w.t[0] = (-a0 + aMax) / jMax;
			w.t[1] = (3 * (a0_p4 + af_p4) - 4 * (a0_p3 + af_p3) * aMax - 4 * af_p3 * aMax + 24 * (a0 + af) * aMax * aMax * aMax - 6 * (af_af + a0_a0) * (aMax * aMax - 2 * jMax * vd) +
					  6 * a0_a0 * (af_af - 2 * af * aMax - 2 * aMax * jMax * tf) - 12 * aMax * aMax * (2 * aMax * aMax - 2 * aMax * jMax * tf + jMax * vd) - 24 * af * aMax * jMax * vd +
					  12 * jMax_jMax * (2 * aMax * g1 + vd_vd)) /
					 (12 * aMax * jMax * (a0_a0 + af_af - 2 * (a0 + af) * aMax + 2 * (aMax * aMax - aMax * jMax * tf + jMax * vd)));
			w.t[2] = aMax / jMax;
			w.t[3] = (-a0_a0 - af_af + 2 * aMax * (a0 + af - 2 * aMax) - 2 * jMax * vd) / (2 * aMax * jMax) + tf;
			w.t[4] = w.t[2];
			w.t[5] = tf - (w.t[0] + w.t[1] + w.t[2] + w.t[3] + 2 * w.t[4] - af / jMax);
			w.t[6] = w.t[4] - af / jMax;
			if (o3_check<UDUD, L_ACC0_ACC1_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		return false;
	}

	O3_NI bool time_acc1_vel(P3& w, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :83-172
		// UDDU
		{
			const double ph1 = a0_a0 + af_af - aMin * (a0 + 2 * af - aMin) - 2 * jMax * (vd - aMin * tf);
			const double ph2 = 2 * aMin * (jMax * g1 + af * vd) - aMin * aMin * vd + jMax * vd_vd;
			const double ph3 = af_af + aMin * (aMin - 2 * af) - 2 * jMax * (vd - aMin * tf);
			double polynom[4];
			polynom[0] = (2 * (2 * a0 - aMin)) / jMax;
			polynom[1] = (4 * a0_a0 + ph1 - 3 * a0 * aMin) / jMax_jMax;
			polynom[2] = (2 * a0 * ph1) / (jMax_jMax * jMax);
			polynom[3] = (3 * (a0_p4 + af_p4) - 4 * (a0_p3 + 2 * af_p3) * aMin + 6 * af_af * (aMin * aMin - 2 * jMax * vd) + 12 * jMax * ph2 + 6 * a0_a0 * ph3) / (12 * jMax_jMax * jMax_jMax);
			const double t_min = -a0 / jMax;
			const double t_max = o3_min((tf + 2 * aMin / jMax - (a0 + af) / jMax) / 2, (aMax - a0) / jMax);
			const Roots roots = o3_solve_quart(polynom[0], polynom[1], polynom[2], polynom[3]);
			for (int ri = 0; ri < roots.n; ri++) {
				double t = roots.x[ri];
				if (t < t_min || t > t_max) continue;
				// single Newton step (regarding pd)
				if (o3_abs(a0 + jMax * t) > 16 * O3_EPS) {
					const double h0 = jMax * t * t;
					const double orig = -pd +
										(3 * (a0_p4 + af_p4) - 8 * af_p3 * aMin - 4 * a0_p3 * aMin + 6 * af_af * (aMin * aMin + 2 * jMax * (h0 - vd)) +
										 6 * a0_a0 * (af_af - 2 * af * aMin + aMin * aMin + 2 * aMin * jMax * (-2 * t + tf) + 2 * jMax * (5 * h0 - vd)) +
										 24 * a0 * jMax * t * (a0_a0 + af_af - 2 * af * aMin + aMin * aMin + 2 * jMax * (aMin * (-t + tf) + h0 - vd)) - 24 * af * aMin * jMax * (h0 - vd) +
										 12 * jMax * (aMin * aMin * (h0 - vd) + jMax * (h0 - vd) * (h0 - vd))) /
											(24 * aMin * jMax_jMax) +
										h0 * (tf - t) + tf * v0;
					const double deriv = (a0 + jMax * t) * ((a0_a0 + af_af) / (aMin * jMax) + (aMin - a0 - 2 * af) / jMax + (4 * a0 * t + 2 * h0 - 2 * vd) / aMin + 2 * tf - 3 * t);
					t -= orig / deriv;
				}
				const double h1 = -((a0_a0 + af_af) / 2 + jMax * (-vd + 2 * a0 * t + jMax * t * t)) / aMin;
				w.t[0] = t;
				w.t[1] = 0;
				w.t[2] = a0 / jMax + t;
				w.t[3] = tf - (h1 - aMin + a0 + af) / jMax - 2 * t;
				w.t[4] = -aMin / jMax;
				w.t[5] = (h1 + aMin) / jMax;
				w.t[6] = w.t[4] + af / jMax;
				if (o3_check<UDDU, L_ACC1_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
			}
		}
		// UDUD
		{
			const double ph1 = a0_a0 - af_af + (2 * af - a0) * aMax - aMax * aMax - 2 * jMax * (vd - aMax * tf);
			const double ph2 = aMax * aMax + 2 * jMax * vd;
			const double ph3 = af_af + ph2 - 2 * aMax * (af + jMax * tf);
			const double ph4 = 2 * aMax * jMax * g1 + aMax * aMax * vd + jMax * vd_vd;
			double polynom[4];
			polynom[0] = (4 * a0 - 2 * aMax) / jMax;
			polynom[1] = (4 * a0_a0 - 3 * a0 * aMax + ph1) / jMax_jMax;
			polynom[2] = (2 * a0 * ph1) / (jMax_jMax * jMax);
			polynom[3] = (3 * (a0_p4 + af_p4) - 4 * (a0_p3 + 2 * af_p3) * aMax - 24 * af * aMax * jMax * vd + 12 * jMax * ph4 - 6 * a0_a0 * ph3 + 6 * af_af * ph2) / (12 * jMax_jMax * jMax_jMax);
			const double t_min = -a0 / jMax;
			const double t_max = o3_min((tf + ad / jMax - 2 * aMax / jMax) / 2, (aMax - a0) / jMax);
			const Roots roots = o3_solve_quart(polynom[0], polynom[1], polynom[2], polynom[3]);
			for (int ri = 0; ri < roots.n; ri++) {
				const double t = roots.x[ri];
				if (t > t_max || t < t_min) continue;
				const double h1 = ((a0_a0 - af_af) / 2 + jMax_jMax * t * t - jMax * (vd - 2 * a0 * t)) / aMax;
				w.t[0] = t;
				w.t[1] = 0;
				w.t[2] = t + a0 / jMax;
				w.t[3] = tf + (h1 + ad - aMax) / jMax - 2 * t;
				w.t[4] = aMax / jMax;
				w.t[5] = -(h1 + aMax) / jMax;
				w.t[6] = w.t[4] - af / jMax;
				if (o3_check<UDUD, L_ACC1_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
			}
		}
		return false;
	}

	O3_NI bool time_acc0_vel(P3& w, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :174-264
		if (tf < o3_max((-a0 + aMax) / jMax, 0.0) + o3_max(aMax / jMax, 0.0)) return false;
		const double ph1 = 12 * jMax * (-aMax * aMax * vd - jMax * vd_vd + 2 * aMax * jMax * (-pd + tf * vf));
		// UDDU
		{
			double polynom[4];
			polynom[0] = (2 * aMax) / jMax;
			polynom[1] = (a0_a0 - af_af + 2 * ad * aMax + aMax * aMax + 2 * jMax * (vd - aMax * tf)) / jMax_jMax;
			polynom[2] = 0;
			polynom[3] = -(-3 * (a0_p4 + af_p4) + 4 * (af_p3 + 2 * a0_p3) * aMax - 12 * a0 * aMax * (af_af - 2 * jMax * vd) + 6 * a0_a0 * (af_af - aMax * aMax - 2 * jMax * vd) +
						   6 * af_af * (aMax * aMax - 2 * aMax * jMax * tf + 2 * jMax * vd) + ph1) /
						 (12 * jMax_jMax * jMax_jMax);
			const double t_min = -af / jMax;
			const double t_max = o3_min(tf - (2 * aMax - a0) / jMax, -aMin / jMax);
			const Roots roots = o3_solve_quart(polynom[0], polynom[1], polynom[2], polynom[3]);
			for (int ri = 0; ri < roots.n; ri++) {
				double t = roots.x[ri];
				if (t < t_min || t > t_max) continue;
				// single Newton step (regarding pd)
				if (t > O3_EPS) {
					const double h1 = jMax * t * t + vd;
					const double orig = (-3 * (a0_p4 + af_p4) + 4 * (af_p3 + 2 * a0_p3) * aMax - 24 * af * aMax * jMax_jMax * t * t - 12 * a0 * aMax * (af_af - 2 * jMax * h1) +
										 6 * a0_a0 * (af_af - aMax * aMax - 2 * jMax * h1) + 6 * af_af * (aMax * aMax - 2 * aMax * jMax * tf + 2 * jMax * h1) -
										 12 * jMax * (aMax * aMax * h1 + jMax * h1 * h1 + 2 * aMax * jMax * (pd + jMax * t * t * (t - tf) - tf * vf))) /
										(24 * aMax * jMax_jMax);
					const double deriv = -t * (a0_a0 - af_af + 2 * aMax * (ad - jMax * tf) + aMax * aMax + 3 * aMax * jMax * t + 2 * jMax * h1) / aMax;
					t -= orig / deriv;
				}
				const double h1 = ((a0_a0 - af_af) / 2 + jMax * (jMax * t * t + vd)) / aMax;
				w.t[0] = (-a0 + aMax) / jMax;
				w.t[1] = (h1 - aMax) / jMax;
				w.t[2] = aMax / jMax;
				w.t[3] = tf - (h1 + ad + aMax) / jMax - 2 * t;
				w.t[4] = t;
				w.t[5] = 0;
				w.t[6] = af / jMax + t;
				if (o3_check<UDDU, L_ACC0_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
			}
		}
		// UDUD
		{
			double polynom[4];
			polynom[0] = (-2 * aMax) / jMax;
			polynom[1] = -(a0_a0 + af_af - 2 * (a0 + af) * aMax + aMax * aMax + 2 * jMax * (vd - aMax * tf)) / jMax_jMax;
			polynom[2] = 0;
			polynom[3] = (3 * (a0_p4 + af_p4) - 4 * (af_p3 + 2 * a0_p3) * aMax + 6 * a0_a0 * (af_af + aMax * aMax + 2 * jMax * vd) - 12 * a0 * aMax * (af_af + 2 * jMax * vd) +
						  6 * af_af * (aMax * aMax - 2 * aMax * jMax * tf + 2 * jMax * vd) - ph1) /
						 (12 * jMax_jMax * jMax_jMax);
			const double t_min = af / jMax;
			const double t_max = o3_min(tf - aMax / jMax, aMax / jMax);
			const Roots roots = o3_solve_quart(polynom[0], polynom[1], polynom[2], polynom[3]);
			for (int ri = 0; ri < roots.n; ri++) {
				double t = roots.x[ri];
				if (t < t_min || t > t_max) continue;
				// single Newton step (regarding pd)
				{
					const double h1 = jMax * t * t - vd;
					const double orig = -(3 * (a0_p4 + af_p4) - 4 * (2 * a0_p3 + af_p3) * aMax + 24 * af * aMax * jMax_jMax * t * t - 12 * a0 * aMax * (af_af - 2 * jMax * h1) +
										  6 * a0_a0 * (af_af + aMax * aMax - 2 * jMax * h1) + 6 * af_af * (aMax * aMax - 2 * jMax * (tf * aMax + h1)) +
										  12 * jMax * (-aMax * aMax * h1 + jMax * h1 * h1 - 2 * aMax * jMax * (-pd + jMax * t * t * (t - tf) + tf * vf))) /
										(24 * aMax * jMax_jMax);
					const double deriv = t * (a0_a0 + af_af - 2 * jMax * h1 - 2 * (a0 + af + jMax * tf) * aMax + aMax * aMax + 3 * aMax * jMax * t) / aMax;
					t -= orig / deriv;
				}
				const double h1 = ((a0_a0 + af_af) / 2 + jMax * (vd - jMax * t * t)) / aMax;
				w.t[0] = (-a0 + aMax) / jMax;
				w.t[1] = (h1 - aMax) / jMax;
				w.t[2] = aMax / jMax;
				w.t[3] = tf - (h1 - a0 - af + aMax) / jMax - 2 * t;
				w.t[4] = t;
				w.t[5] = 0;
				w.t[6] = -(af / jMax) + t;
				if (o3_check<UDUD, L_ACC0_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
			}
		}
		return false;
	}

	// helpers of time_vel (the reference's local lambdas, :318-347 and :443-475)
	O3_FN bool vel_check_root_uddu(P3& w, double t, double vMax, double vMin, double aMax, double aMin, double jMax) {
		// single Newton step (regarding pd)
		{
			const double h1 = sqrt((a0_a0 + af_af) / (2 * jMax_jMax) + (2 * a0 * t + jMax * t * t - vd) / jMax);
			const double orig = -pd - (2 * a0_p3 + 4 * af_p3 + 24 * a0 * jMax * t * (af + jMax * (h1 + t - tf)) + 6 * a0_a0 * (af + jMax * (2 * t - tf)) + 6 * (a0_a0 + af_af) * jMax * h1 +
									   12 * af * jMax * (jMax * t * t - vd) + 12 * jMax_jMax * (jMax * t * t * (h1 + t - tf) - tf * v0 - h1 * vd)) /
										  (12 * jMax_jMax);
			const double deriv_newton = -(a0 + jMax * t) * (3 * (h1 + t) - 2 * tf + (a0 + 2 * af) / jMax);
			if (!(orig != orig) && !(deriv_newton != deriv_newton) && o3_abs(deriv_newton) > O3_EPS) t -= orig / deriv_newton;
		}
		if (t > tf || (t != t)) return false;
		const double h1 = sqrt((a0_a0 + af_af) / (2 * jMax_jMax) + (t * (2 * a0 + jMax * t) - vd) / jMax);
		w.t[0] = t;
		w.t[1] = 0;
		w.t[2] = t + a0 / jMax;
		w.t[3] = tf - 2 * (t + h1) - (a0 + af) / jMax;
		w.t[4] = h1;
		w.t[5] = 0;
		w.t[6] = h1 + af / jMax;
		return o3_check<UDDU, L_VEL, false>(w, jMax, vMax, vMin, aMax, aMin);
	}
	O3_FN bool vel_check_root_udud(P3& w, double t, double vMax, double vMin, double aMax, double aMin, double jMax) {
		// double Newton step (regarding pd)
		{
			double h1 = sqrt((af_af - a0_a0) / (2 * jMax_jMax) - ((2 * a0 + jMax * t) * t - vd) / jMax);
			double orig = -pd + (af_p3 - a0_p3 + 3 * a0_a0 * jMax * (tf - 2 * t)) / (6 * jMax_jMax) + (2 * a0 + jMax * t) * t * (tf - t) + (jMax * h1 - af) * h1 * h1 + tf * v0;
			double deriv_newton = (a0 + jMax * t) * (2 * (af + jMax * tf) - 3 * jMax * (h1 + t) - a0) / jMax;
			t -= orig / deriv_newton;
			h1 = sqrt((af_af - a0_a0) / (2 * jMax_jMax) - ((2 * a0 + jMax * t) * t - vd) / jMax);
			orig = -pd + (af_p3 - a0_p3 + 3 * a0_a0 * jMax * (tf - 2 * t)) / (6 * jMax_jMax) + (2 * a0 + jMax * t) * t * (tf - t) + (jMax * h1 - af) * h1 * h1 + tf * v0;
			if (o3_abs(orig) > 1e-9) {
				deriv_newton = (a0 + jMax * t) * (2 * (af + jMax * tf) - 3 * jMax * (h1 + t) - a0) / jMax;
				t -= orig / deriv_newton;
			}
		}
		const double h1 = sqrt((af_af - a0_a0) / (2 * jMax_jMax) - ((2 * a0 + jMax * t) * t - vd) / jMax);
		w.t[0] = t;
		w.t[1] = 0;
		w.t[2] = t + a0 / jMax;
		w.t[3] = tf - 2 * (t + h1) + ad / jMax;
		w.t[4] = h1;
		w.t[5] = 0;
		w.t[6] = h1 - af / jMax;
		return o3_check<UDUD, L_VEL, false>(w, jMax, vMax, vMin, aMax, aMin);
	}

	O3_NI bool time_vel(P3& w, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :266-505
		const double tz_min = o3_max(0.0, -a0 / jMax);
		const double tz_max = o3_min((tf - a0 / jMax) / 2, (aMax - a0) / jMax);
		// UDDU
		if (o3_abs(v0) < O3_EPS && o3_abs(a0) < O3_EPS && o3_abs(vf) < O3_EPS && o3_abs(af) < O3_EPS) {
			const Roots roots = o3_solve_cub(1, -tf / 2, 0, pd / (2 * jMax));
			for (int ri = 0; ri < roots.n; ri++) {
				double t = roots.x[ri];
				if (t > tf / 4) continue;
				// single Newton step (regarding pd)
				if (t > O3_EPS) {
					const double orig = -pd + jMax * t * t * (tf - 2 * t);
					const double deriv = 2 * jMax * t * (tf - 3 * t);
					t -= orig / deriv;
				}
				w.t[0] = t;
				w.t[1] = 0;
				w.t[2] = t;
				w.t[3] = tf - 4 * t;
				w.t[4] = t;
				w.t[5] = 0;
				w.t[6] = t;
				if (o3_check<UDDU, L_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
			}
		} else {
			const double p1 = af_af - 2 * jMax * (-2 * af * tf + jMax * tf_tf + 3 * vd);
			const double ph1 = af_p3 - 3 * jMax_jMax * g1 - 3 * af * jMax * vd;
			const double ph2 = af_p4 + 8 * af_p3 * jMax * tf + 12 * jMax * (3 * jMax * vd_vd - af_af * vd + 2 * af * jMax * (g1 - tf * vd) - 2 * jMax_jMax * tf * g1);
			const double ph3 = a0 * (af - jMax * tf);
			const double ph4 = jMax * (-ad + jMax * tf);
			// root of a 5th order polynomial
			double polynom[6];
			polynom[0] = 1.0;
			polynom[1] = (15 * a0_a0 + af_af + 4 * af * jMax * tf - 16 * ph3 - 2 * jMax * (jMax * tf_tf + 3 * vd)) / (4 * ph4);
			polynom[2] = (29 * a0_p3 - 2 * af_p3 - 33 * a0 * ph3 + 6 * jMax_jMax * g1 + 6 * af * jMax * vd + 6 * a0 * p1) / (6 * jMax * ph4);
			polynom[3] = (61 * a0_p4 - 76 * a0_a0 * ph3 - 16 * a0 * ph1 + 30 * a0_a0 * p1 + ph2) / (24 * jMax_jMax * ph4);
			polynom[4] = (a0 * (7 * a0_p4 - 10 * a0_a0 * ph3 - 4 * a0 * ph1 + 6 * a0_a0 * p1 + ph2)) / (12 * jMax_jMax * jMax * ph4);
			polynom[5] = (7 * a0_p6 + af_p6 - 12 * a0_p4 * ph3 + 48 * af_p3 * jMax_jMax * g1 - 8 * a0_p3 * ph1 - 72 * jMax_jMax * jMax * (jMax * g1 * g1 + vd_vd * vd + 2 * af * g1 * vd) -
						  6 * af_p4 * jMax * vd + 36 * af_af * jMax_jMax * vd_vd + 9 * a0_p4 * p1 + 3 * a0_a0 * ph2) /
						 (144 * jMax_jMax * jMax_jMax * ph4);
			double deriv[5], dderiv[4];
			o3_poly_monic_deri(polynom, deriv);
			o3_poly_deri(deriv, dderiv);
			// the extrema: 4th order derivative solved analytically
			const Roots d_extremas = o3_solve_quart(deriv[1], deriv[2], deriv[3], deriv[4]);
			double tz_current = tz_min;
			for (int ri = 0; ri < d_extremas.n; ri++) {
				double tz = d_extremas.x[ri];
				if (tz >= tz_max) continue;
				const double orig = o3_poly_eval(deriv, tz);
				if (o3_abs(orig) > O3_TOL) tz -= orig / o3_poly_eval(dderiv, tz);
				const double val_new = o3_poly_eval(polynom, tz);
				if (o3_abs(val_new) < 64 * o3_abs(o3_poly_eval(dderiv, tz)) * O3_TOL) {
					if (vel_check_root_uddu(w, tz, vMax, vMin, aMax, aMin, jMax)) return true;
				} else if (o3_poly_eval(polynom, tz_current) * val_new < 0) {
					if (vel_check_root_uddu(w, o3_shrink_interval(polynom, tz_current, tz), vMax, vMin, aMax, aMin, jMax)) return true;
				}
				tz_current = tz;
			}
			const double val_max = o3_poly_eval(polynom, tz_max);
			if (o3_poly_eval(polynom, tz_current) * val_max < 0) {
				if (vel_check_root_uddu(w, o3_shrink_interval(polynom, tz_current, tz_max), vMax, vMin, aMax, aMin, jMax)) return true;
			} else if (o3_abs(val_max) < 8 * O3_EPS) {
				if (vel_check_root_uddu(w, tz_max, vMax, vMin, aMax, aMin, jMax)) return true;
			}
		}
		// UDUD
		{
			const double ph1 = af_af - 2 * jMax * (2 * af * tf + jMax * tf_tf - 3 * vd);
			const double ph2 = af_p3 - 3 * jMax_jMax * g1 + 3 * af * jMax * vd;
			const double ph3 = 2 * jMax * tf * g1 + 3 * vd_vd;
			const double ph4 = af_p4 - 8 * af_p3 * jMax * tf + 12 * jMax * (jMax * ph3 + af_af * vd + 2 * af * jMax * (g1 - tf * vd));
			const double ph5 = af + jMax * tf;
			// root of a 6th order polynomial
			double polynom[7];
			polynom[0] = 1.0;
			polynom[1] = (5 * a0 - ph5) / jMax;
			polynom[2] = (39 * a0_a0 - ph1 - 16 * a0 * ph5) / (4 * jMax_jMax);
			polynom[3] = (55 * a0_p3 - 33 * a0_a0 * ph5 - 6 * a0 * ph1 + 2 * ph2) / (6 * jMax_jMax * jMax);
			polynom[4] = (101 * a0_p4 + ph4 - 76 * a0_p3 * ph5 - 30 * a0_a0 * ph1 + 16 * a0 * ph2) / (24 * jMax_jMax * jMax_jMax);
			polynom[5] = (a0 * (11 * a0_p4 + ph4 - 10 * a0_p3 * ph5 - 6 * a0_a0 * ph1 + 4 * a0 * ph2)) / (12 * jMax_jMax * jMax_jMax * jMax);
			polynom[6] = (11 * a0_p6 - af_p6 - 12 * a0_p5 * ph5 - 48 * af_p3 * jMax_jMax * g1 - 9 * a0_p4 * ph1 + 72 * jMax_jMax * jMax * (jMax * g1 * g1 - vd_vd * vd - 2 * af * g1 * vd) -
						  6 * af_p4 * jMax * vd - 36 * af_af * jMax_jMax * vd_vd + 8 * a0_p3 * ph2 + 3 * a0_a0 * ph4) /
						 (144 * jMax_jMax * jMax_jMax * jMax_jMax);
			double deriv[6], dderiv[5], ddderiv[4];
			o3_poly_monic_deri(polynom, deriv);
			o3_poly_monic_deri(deriv, dderiv);
			o3_poly_deri(dderiv, ddderiv);
			double dd_tz_current = tz_min;
			double iv_l[6], iv_r[6];  // roots::Set<std::pair<double, double>, 6>: sorted by (first, second) when iterated
			int niv = 0;
			const Roots dd_extremas = o3_solve_quart(dderiv[1], dderiv[2], dderiv[3], dderiv[4]);
			for (int ri = 0; ri < dd_extremas.n; ri++) {
				double tz = dd_extremas.x[ri];
				if (tz >= tz_max) continue;
				const double orig = o3_poly_eval(dderiv, tz);
				if (o3_abs(orig) > O3_TOL) tz -= orig / o3_poly_eval(ddderiv, tz);
				if (o3_poly_eval(deriv, dd_tz_current) * o3_poly_eval(deriv, tz) < 0) {
					iv_l[niv] = dd_tz_current;
					iv_r[niv] = tz;
					niv++;
				}
				dd_tz_current = tz;
			}
			if (o3_poly_eval(deriv, dd_tz_current) * o3_poly_eval(deriv, tz_max) < 0) {
				iv_l[niv] = dd_tz_current;
				iv_r[niv] = tz_max;
				niv++;
			}
			for (int i = 1; i < niv; i++)  // std::sort of the pairs
				for (int k = i; k > 0 && (iv_l[k] < iv_l[k - 1] || (iv_l[k] == iv_l[k - 1] && iv_r[k] < iv_r[k - 1])); k--) {
					const double tl = iv_l[k], tr = iv_r[k];
					iv_l[k] = iv_l[k - 1];
					iv_r[k] = iv_r[k - 1];
					iv_l[k - 1] = tl;
					iv_r[k - 1] = tr;
				}
			double tz_current = tz_min;
			for (int i = 0; i < niv; i++) {
				const double tz = o3_shrink_interval(deriv, iv_l[i], iv_r[i]);
				if (tz >= tz_max) continue;
				const double p_val = o3_poly_eval(polynom, tz);
				if (o3_abs(p_val) < 64 * o3_abs(o3_poly_eval(dderiv, tz)) * O3_TOL) {
					if (vel_check_root_udud(w, tz, vMax, vMin, aMax, aMin, jMax)) return true;
				} else if (o3_poly_eval(polynom, tz_current) * p_val < 0) {
					if (vel_check_root_udud(w, o3_shrink_interval(polynom, tz_current, tz), vMax, vMin, aMax, aMin, jMax)) return true;
				}
				tz_current = tz;
			}
			if (o3_poly_eval(polynom, tz_current) * o3_poly_eval(polynom, tz_max) < 0) {
				if (vel_check_root_udud(w, o3_shrink_interval(polynom, tz_current, tz_max), vMax, vMin, aMax, aMin, jMax)) return true;
			}
		}
		return false;
	}

	O3_NI bool time_acc0_acc1(P3& w, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :507-546
		if (o3_abs(a0) < O3_EPS && o3_abs(af) < O3_EPS) {
			const double h1 = 2 * aMin * g1 + vd_vd + aMax * (2 * pd + aMin * tf_tf - 2 * tf * vf);
			const double h2 = ((aMax - aMin) * (-aMin * vd + aMax * (aMin * tf - vd)));
			const double jf = h2 / h1;
			w.t[0] = aMax / jf;
			w.t[1] = (-2 * aMax * h1 + aMin * aMin * g2) / h2;
			w.t[2] = w.t[0];
			w.t[3] = 0;
			w.t[4] = -aMin / jf;
			w.t[5] = tf - (2 * w.t[0] + w.t[1] + 2 * w.t[4]);
			w.t[6] = w.t[4];
			return o3_check_jf<UDDU, L_ACC0_ACC1>(w, jf, vMax, vMin, aMax, aMin, jMax);
		}
		// UDDU
		{
			const double h1 = sqrt(144 * o3_pow2((aMax - aMin) * (-aMin * vd + aMax * (aMin * tf - vd)) - af_af * (aMax * tf - vd) + 2 * af * aMin * (aMax * tf - vd) +
												 a0_a0 * (aMin * tf + v0 - vf) - 2 * a0 * aMax * (aMin * tf - vd)) +
								   48 * ad *
									   (3 * a0_p3 - 3 * af_p3 + 12 * aMax * aMin * (-aMax + aMin) + 4 * af_af * (aMax + 2 * aMin) +
										a0 * (-3 * af_af + 8 * af * (aMin - aMax) + 6 * (aMax * aMax + 2 * aMax * aMin - aMin * aMin)) + 6 * af * (aMax * aMax - 2 * aMax * aMin - aMin * aMin) +
										a0_a0 * (3 * af - 4 * (2 * aMax + aMin))) *
									   (2 * aMin * g1 + vd * vd + aMax * (2 * pd + aMin * tf * tf - 2 * tf * vf)));
			const double jf = -(3 * af_af * aMax * tf - 3 * a0_a0 * aMin * tf - 6 * ad * aMax * aMin * tf + 3 * aMax * aMin * (aMin - aMax) * tf + 3 * (a0_a0 - af_af) * vd +
								6 * vd * (af * aMin - a0 * aMax) + 3 * (aMax * aMax - aMin * aMin) * vd + h1 / 4) /
							  (6 * (2 * aMin * g1 + vd * vd + aMax * (2 * pd + aMin * tf_tf - 2 * tf * vf)));
			w.t[0] = (aMax - a0) / jf;
			w.t[1] = (a0_a0 - af_af + 2 * ad * aMin - 2 * (aMax * aMax - 2 * aMax * aMin + aMin * aMin + aMin * jf * tf - jf * vd)) / (2 * (aMax - aMin) * jf);
			w.t[2] = aMax / jf;
			w.t[3] = 0;
			w.t[4] = -aMin / jf;
			w.t[5] = tf - (w.t[0] + w.t[1] + w.t[2] + 2 * w.t[4] + af / jf);
			w.t[6] = w.t[4] + af / jf;
			if (o3_check_jf<UDDU, L_ACC0_ACC1>(w, jf, vMax, vMin, aMax, aMin, jMax)) return true;
		}
		return false;
	}

	O3_NI bool time_acc1(P3& w, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :548-630
		// a3 != 0, UDDU
		{
			const double h0 = sqrt(jMax_jMax * (a0_p4 + af_p4 - 4 * af_p3 * jMax * tf + 6 * af_af * jMax_jMax * tf_tf - 4 * a0_p3 * (af - jMax * tf) + 6 * a0_a0 * (af - jMax * tf) * (af - jMax * tf) +
												24 * af * jMax_jMax * g1 - 4 * a0 * (af_p3 - 3 * af_af * jMax * tf + 6 * jMax_jMax * (-pd + tf * vf)) - 12 * jMax_jMax * (-vd_vd + jMax * tf * g2)) /
								   3) /
							  jMax;
			const double h1 = sqrt((a0_a0 + af_af - 2 * a0 * af - 2 * ad * jMax * tf + 2 * h0) / jMax_jMax + tf_tf);
			w.t[0] = -(a0_a0 + af_af + 2 * a0 * (jMax * tf - af) - 2 * jMax * vd + h0) / (2 * jMax * (-ad + jMax * tf));
			w.t[1] = 0;
			w.t[2] = (tf - h1) / 2 - ad / (2 * jMax);
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = h1;
			w.t[6] = tf - (w.t[0] + w.t[2] + w.t[5]);
			if (o3_check<UDDU, L_ACC1, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		// UDUD
		{
			const double h0 = sqrt(jMax_jMax * (a0_p4 + af_p4 + 4 * (af_p3 - a0_p3) * jMax * tf + 6 * af_af * jMax_jMax * tf_tf + 6 * a0_a0 * (af + jMax * tf) * (af + jMax * tf) + 24 * af * jMax_jMax * g1 -
												4 * a0 * (a0_a0 * af + af_p3 + 3 * af_af * jMax * tf + 6 * jMax_jMax * (-pd + tf * vf)) + 12 * jMax_jMax * (vd_vd + jMax * tf * g2)) /
								   3) /
							  jMax;
			const double h1 = sqrt((a0_a0 + af_af - 2 * a0 * af + 2 * ad * jMax * tf + 2 * h0) / jMax_jMax + tf_tf);
			w.t[0] = 0;
			w.t[1] = 0;
			w.t[2] = -(a0_a0 + af_af - 2 * a0 * af + 2 * jMax * (vd - a0 * tf) + h0) / (2 * jMax * (ad + jMax * tf));
			w.t[3] = 0;
			w.t[4] = ad / (2 * jMax) + (tf - h1) / 2;
			w.t[5] = h1;
			w.t[6] = tf - (w.t[5] + w.t[4] + w.t[2]);
			if (o3_check<UDUD, L_ACC1, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		// UDDU, solution 2
		{
			const double h0a = a0_p3 - af_p3 - 3 * a0_a0 * aMin + 3 * aMin * aMin * (a0 + jMax * tf) + 3 * af * aMin * (-aMin - 2 * jMax * tf) - 3 * af_af * (-aMin - jMax * tf) -
							   3 * jMax_jMax * (-2 * pd - aMin * tf_tf + 2 * tf * vf);
			const double h0b = a0_a0 + af_af - 2 * (a0 + af) * aMin + 2 * (aMin * aMin - jMax * (-aMin * tf + vd));
			const double h0c = a0_p4 + 3 * af_p4 - 4 * (a0_p3 + 2 * af_p3) * aMin + 6 * a0_a0 * aMin * aMin + 6 * af_af * (aMin * aMin - 2 * jMax * vd) +
							   12 * jMax * (2 * aMin * jMax * g1 - aMin * aMin * vd + jMax * vd_vd) + 24 * af * aMin * jMax * vd -
							   4 * a0 * (af_p3 - 3 * af * aMin * (-aMin - 2 * jMax * tf) + 3 * af_af * (-aMin - jMax * tf) + 3 * jMax * (-aMin * aMin * tf + jMax * (-2 * pd - aMin * tf_tf + 2 * tf * vf)));
			const double h1 = o3_abs(jMax) / jMax * sqrt(4 * h0a * h0a - 6 * h0b * h0c);
			const double h2 = 6 * jMax * h0b;
			w.t[0] = 0;
			w.t[1] = 0;
			w.t[2] = (2 * h0a + h1) / h2;
			w.t[3] = -(a0_a0 + af_af - 2 * (a0 + af) * aMin + 2 * (aMin * aMin + aMin * jMax * tf - jMax * vd)) / (2 * jMax * (a0 - aMin - jMax * w.t[2]));
			w.t[4] = (a0 - aMin) / jMax - w.t[2];
			w.t[5] = tf - (w.t[2] + w.t[3] + w.t[4] + (af - aMin) / jMax);
			w.t[6] = (af - aMin) / jMax;
			if (o3_check<UDDU, L_ACC1, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		// UDUD, solution 1
		{
			const double h0a = -a0_p3 + af_p3 + 3 * (a0_a0 - af_af) * aMax - 3 * ad * aMax * aMax - 6 * af * aMax * jMax * tf + 3 * af_af * jMax * tf +
							   3 * jMax * (aMax * aMax * tf + jMax * (-2 * pd - aMax * tf_tf + 2 * tf * vf));
			const double h0b = a0_a0 - af_af + 2 * ad * aMax + 2 * jMax * (aMax * tf - vd);
			const double h0c = a0_p4 + 3 * af_p4 - 4 * (a0_p3 + 2 * af_p3) * aMax + 6 * a0_a0 * aMax * aMax - 24 * af * aMax * jMax * vd + 12 * jMax * (2 * aMax * jMax * g1 + jMax * vd_vd + aMax * aMax * vd) +
							   6 * af_af * (aMax * aMax + 2 * jMax * vd) -
							   4 * a0 * (af_p3 + 3 * af * aMax * (aMax - 2 * jMax * tf) - 3 * af_af * (aMax - jMax * tf) + 3 * jMax * (aMax * aMax * tf + jMax * (-2 * pd - aMax * tf_tf + 2 * tf * vf)));
			const double h1 = o3_abs(jMax) / jMax * sqrt(4 * h0a * h0a - 6 * h0b * h0c);
			const double h2 = 6 * jMax * h0b;
			w.t[0] = 0;
			w.t[1] = 0;
			w.t[2] = -(2 * h0a + h1) / h2;
			w.t[3] = 2 * h1 / h2;
			w.t[4] = (aMax - a0) / jMax + w.t[2];
			w.t[5] = tf - (w.t[2] + w.t[3] + w.t[4] + (-af + aMax) / jMax);
			w.t[6] = (-af + aMax) / jMax;
			if (o3_check<UDUD, L_ACC1, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		return false;
	}

	O3_NI bool time_acc0(P3& w, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :632-698
		// UDUD
		{
			const double h1 = sqrt(ad_ad / (2 * jMax_jMax) - ad * (aMax - a0) / (jMax_jMax) + (aMax * tf - vd) / jMax);
			w.t[0] = (aMax - a0) / jMax;
			w.t[1] = tf - ad / jMax - 2 * h1;
			w.t[2] = h1;
			w.t[3] = 0;
			w.t[4] = (af - aMax) / jMax + h1;
			w.t[5] = 0;
			w.t[6] = 0;
			if (o3_check<UDUD, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		// UDUD
		{
			const double h0a = -a0_a0 + af_af - 2 * ad * aMax + 2 * jMax * (aMax * tf - vd);
			const double h0b = a0_p3 + 2 * af_p3 - 6 * af_af * aMax - 3 * a0_a0 * (af - jMax * tf) - 3 * a0 * aMax * (aMax - 2 * af + 2 * jMax * tf) -
							   3 * jMax * (jMax * (-2 * pd + aMax * tf_tf + 2 * tf * v0) + aMax * (aMax * tf - 2 * vd)) + 3 * af * (aMax * aMax + 2 * aMax * jMax * tf - 2 * jMax * vd);
			const double h0 = o3_abs(jMax) * sqrt(4 * h0b * h0b - 18 * h0a * h0a * h0a);
			const double h1 = 3 * jMax * h0a;
			w.t[0] = (-a0 + aMax) / jMax;
			w.t[1] = (-a0_p3 + af_p3 + af_af * (-6 * aMax + 3 * jMax * tf) + a0_a0 * (-3 * af + 6 * aMax + 3 * jMax * tf) + 6 * af * (aMax * aMax - jMax * vd) +
					  3 * a0 * (af_af - 2 * (aMax * aMax + jMax * vd)) - 6 * jMax * (aMax * (aMax * tf - 2 * vd) + jMax * g2)) /
					 h1;
			w.t[2] = -(ad + h0 / h1) / (2 * jMax) + tf / 2 - w.t[1] / 2;
			w.t[3] = h0 / (jMax * h1);
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = tf - (w.t[0] + w.t[1] + w.t[2] + w.t[3]);
			if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		// a3 != 0, UDDU solution 1
		{
			const double h0a = a0_p3 + 2 * af_p3 - 6 * (af_af + aMax * aMax) * aMax - 6 * (a0 + af) * aMax * jMax * tf + 9 * aMax * aMax * (af + jMax * tf) + 3 * a0 * aMax * (-2 * af + 3 * aMax) +
							   3 * a0_a0 * (af - 2 * aMax + jMax * tf) - 6 * jMax_jMax * g1 + 6 * (af - aMax) * jMax * vd - 3 * aMax * jMax_jMax * tf_tf;
			const double h0b = a0_a0 + af_af + 2 * (aMax * aMax - (a0 + af) * aMax + jMax * (vd - aMax * tf));
			const double h1 = o3_abs(jMax) / jMax * sqrt(4 * h0a * h0a - 18 * h0b * h0b * h0b);
			const double h2 = 6 * jMax * h0b;
			w.t[0] = (-a0 + aMax) / jMax;
			w.t[1] = ad / jMax - 2 * w.t[0] - (2 * h0a - h1) / h2 + tf;
			w.t[2] = -(2 * h0a + h1) / h2;
			w.t[3] = (2 * h0a - h1) / h2;
			w.t[4] = tf - (w.t[0] + w.t[1] + w.t[2] + w.t[3]);
			w.t[5] = 0;
			w.t[6] = 0;
			if (o3_check<UDDU, L_ACC0, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		return false;
	}

	O3_NI bool time_none(P3& w, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :700-1000
		if (o3_abs(v0) < O3_EPS && o3_abs(a0) < O3_EPS && o3_abs(af) < O3_EPS) {
			const double h1 = sqrt(tf_tf * vf_vf + o3_pow2(4 * pd - tf * vf));
			const double jf = 4 * (4 * pd - 2 * tf * vf + h1) / tf_p3;
			w.t[0] = tf / 4;
			w.t[1] = 0;
			w.t[2] = 2 * w.t[0];
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = w.t[0];
			if (o3_check_jf<UDDU, L_NONE>(w, jf, vMax, vMin, aMax, aMin, jMax)) return true;
		}
		if (o3_abs(a0) < O3_EPS && o3_abs(af) < O3_EPS) {
			// profiles with a3 != 0, UDDU: first acc, then constant
			double polynom[4];
			polynom[0] = -2 * tf;
			polynom[1] = 2 * vd / jMax + tf_tf;
			polynom[2] = 4 * (pd - tf * vf) / jMax;
			polynom[3] = (vd_vd + jMax * tf * g2) / (jMax_jMax);
			const Roots roots = o3_solve_quart(polynom[0], polynom[1], polynom[2], polynom[3]);
			for (int ri = 0; ri < roots.n; ri++) {
				double t = roots.x[ri];
				if (t > tf / 2 || t > (aMax - a0) / jMax) continue;
				// single Newton step (regarding pd)
				{
					const double h1 = (jMax * t * (t - tf) + vd) / (jMax * (2 * t - tf));
					const double h2 = (2 * jMax * t * (t - tf) + jMax * tf_tf - 2 * vd) / (jMax * (2 * t - tf) * (2 * t - tf));
					const double orig = (-2 * pd + 2 * tf * v0 + h1 * h1 * jMax * (tf - 2 * t) + jMax * tf * (2 * h1 * t - t * t - (h1 - t) * tf)) / 2;
					const double deriv = (jMax * tf * (2 * t - tf) * (h2 - 1)) / 2 + h1 * jMax * (tf - (2 * t - tf) * h2 - h1);
					t -= orig / deriv;
				}
				w.t[0] = t;
				w.t[1] = 0;
				w.t[2] = (jMax * t * (t - tf) + vd) / (jMax * (2 * t - tf));
				w.t[3] = tf - 2 * t;
				w.t[4] = t - w.t[2];
				w.t[5] = 0;
				w.t[6] = 0;
				if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
			}
		}
		// UDUD T 0246
		{
			const double h0 = sqrt(2 * jMax_jMax *
								   (2 * o3_pow2(a0_p3 - af_p3 - 3 * af_af * jMax * tf + 9 * af * jMax_jMax * tf_tf - 3 * a0_a0 * (af + jMax * tf) + 3 * a0 * o3_pow2(af + jMax * tf) +
												3 * jMax_jMax * (8 * pd + jMax * tf_tf * tf - 8 * tf * vf)) -
									3 * (a0_a0 + af_af - 2 * af * jMax * tf - 2 * a0 * (af + jMax * tf) - jMax * (jMax * tf_tf + 4 * v0 - 4 * vf)) *
										(a0_p4 + af_p4 + 4 * af_p3 * jMax * tf + 6 * af_af * jMax_jMax * tf_tf - 3 * jMax_jMax * jMax_jMax * tf_tf * tf_tf - 4 * a0_p3 * (af + jMax * tf) +
										 6 * a0_a0 * o3_pow2(af + jMax * tf) - 12 * af * jMax_jMax * (8 * pd + jMax * tf_tf * tf - 8 * tf * v0) + 48 * jMax_jMax * vd_vd + 48 * jMax_jMax * jMax * tf * g2 -
										 4 * a0 * (af_p3 + 3 * af_af * jMax * tf - 9 * af * jMax_jMax * tf_tf - 3 * jMax_jMax * (8 * pd + jMax * tf_tf * tf - 8 * tf * vf))))) /
							  jMax;
			const double h1 = 12 * jMax * (-a0_a0 - af_af + 2 * af * jMax * tf + 2 * a0 * (af + jMax * tf) + jMax * (jMax * tf_tf + 4 * v0 - 4 * vf));
			const double h2 = -4 * a0_p3 + 4 * af_p3 + 12 * a0_a0 * af - 12 * a0 * af_af + 48 * jMax_jMax * pd + 12 * (a0_a0 - af_af) * jMax * tf - 24 * jMax_jMax * tf * (v0 + vf) + 24 * ad * jMax * vd;
			const double h3 = 2 * a0_p3 - 2 * af_p3 - 6 * a0_a0 * af + 6 * a0 * af_af;
			w.t[0] = (h3 - 48 * jMax_jMax * (tf * vf - pd) - 6 * (a0_a0 + af_af) * jMax * tf + 12 * a0 * af * jMax * tf + 6 * (a0 + 3 * af + jMax * tf) * tf_tf * jMax_jMax - h0) / h1;
			w.t[1] = 0;
			w.t[2] = (h2 + h0) / h1;
			w.t[3] = 0;
			w.t[4] = (-h2 + h0) / h1;
			w.t[5] = 0;
			w.t[6] = (-h3 + 48 * jMax_jMax * (tf * v0 - pd) - 6 * (a0_a0 + af_af) * jMax * tf + 12 * a0 * af * jMax * tf + 6 * (af + 3 * a0 + jMax * tf) * tf_tf * jMax_jMax - h0) / h1;
			if (o3_check<UDUD, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		// profiles with a3 != 0, UDDU
		{
			// T 0234
			{
				const double ph1 = af + jMax * tf;
				double polynom[4];
				polynom[0] = -2 * (ad + jMax * tf) / jMax;
				polynom[1] = 2 * (a0_a0 + af_af + jMax * (af * tf + vd) - 2 * a0 * ph1) / jMax_jMax + tf_tf;
				polynom[2] = 2 * (a0_p3 - af_p3 - 3 * af_af * jMax * tf + 3 * a0 * ph1 * (ph1 - a0) - 6 * jMax_jMax * (-pd + tf * vf)) / (3 * jMax_jMax * jMax);
				polynom[3] = (a0_p4 + af_p4 + 4 * af_p3 * jMax * tf - 4 * a0_p3 * ph1 + 6 * a0_a0 * ph1 * ph1 + 24 * jMax_jMax * af * g1 -
							  4 * a0 * (af_p3 + 3 * af_af * jMax * tf + 6 * jMax_jMax * (-pd + tf * vf)) + 6 * jMax_jMax * af_af * tf_tf + 12 * jMax_jMax * (vd_vd + jMax * tf * g2)) /
							 (12 * jMax_jMax * jMax_jMax);
				const double t_min = ad / jMax;
				const double t_max = o3_min((aMax - a0) / jMax, (ad / jMax + tf) / 2);
				const Roots roots = o3_solve_quart(polynom[0], polynom[1], polynom[2], polynom[3]);
				for (int ri = 0; ri < roots.n; ri++) {
					double t = roots.x[ri];
					if (t < t_min || t > t_max) continue;
					// single Newton step (regarding pd)
					{
						const double h0 = jMax * (2 * t - tf) - ad;
						const double h1 = (ad_ad - 2 * af * jMax * t + 2 * a0 * jMax * (t - tf) + 2 * jMax * (jMax * t * (t - tf) + vd)) / (2 * jMax * h0);
						const double h2 = (-ad_ad + 2 * jMax_jMax * (tf_tf + t * (t - tf)) + (a0 + af) * jMax * tf - ad * h0 - 2 * jMax * vd) / (h0 * h0);
						const double orig = (-a0_p3 + af_p3 + 3 * ad_ad * jMax * (h1 - t) + 3 * ad * jMax_jMax * (h1 - t) * (h1 - t) - 3 * a0 * af * ad +
											 3 * jMax_jMax * (a0 * tf_tf - 2 * pd + 2 * tf * v0 + h1 * h1 * jMax * (tf - 2 * t) + jMax * tf * (2 * h1 * t - t * t - (h1 - t) * tf))) /
											(6 * jMax_jMax);
						const double deriv = (h0 * (-ad + jMax * tf) * (h2 - 1)) / (2 * jMax) + h1 * (-ad + jMax * (tf - h1) - h0 * h2);
						t -= orig / deriv;
					}
					w.t[0] = t;
					w.t[1] = 0;
					w.t[2] = (ad_ad + 2 * jMax * (-a0 * tf - ad * t + jMax * t * (t - tf) + vd)) / (2 * jMax * (-ad + jMax * (2 * t - tf)));
					w.t[3] = ad / jMax + tf - 2 * t;
					w.t[4] = tf - (t + w.t[2] + w.t[3]);
					w.t[5] = 0;
					w.t[6] = 0;
					if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
				}
			}
			// T 3456
			{
				const double h1 = 3 * jMax * (ad_ad + 2 * jMax * (a0 * tf - vd));
				const double h2 = ad_ad + 2 * jMax * (a0 * tf - vd);
				const double h0 = sqrt(4 * o3_pow2(2 * (a0_p3 - af_p3) - 6 * a0_a0 * (af - jMax * tf) + 6 * jMax_jMax * g1 + 3 * a0 * (2 * af_af - 2 * jMax * af * tf + jMax_jMax * tf_tf) + 6 * ad * jMax * vd) -
									   18 * h2 * h2 * h2) /
								  h1 * o3_abs(jMax) / jMax;
				w.t[0] = 0;
				w.t[1] = 0;
				w.t[2] = 0;
				w.t[3] = (af_p3 - a0_p3 + 3 * (af_af - a0_a0) * jMax * tf - 3 * ad * (a0 * af + 2 * jMax * vd) - 6 * jMax_jMax * g2) / h1;
				w.t[4] = (tf - w.t[3] - h0) / 2 - ad / (2 * jMax);
				w.t[5] = h0;
				w.t[6] = (tf - w.t[3] + ad / jMax - h0) / 2;
				if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
			}
			// T 2346
			{
				const double ph1 = ad_ad + 2 * (af + a0) * jMax * tf - jMax * (jMax * tf_tf + 4 * vd);
				const double ph2 = jMax * tf_tf * g1 - vd * (-2 * pd - tf * v0 + 3 * tf * vf);
				const double ph3 = 5 * af_af - 8 * af * jMax * tf + 2 * jMax * (2 * jMax * tf_tf - vd);
				const double ph4 = jMax_jMax * tf_p4 - 2 * vd_vd + 8 * jMax * tf * (-pd + tf * vf);
				const double ph5 = (5 * af_p4 - 8 * af_p3 * jMax * tf - 12 * af_af * jMax * (jMax * tf_tf + vd) + 24 * af * jMax_jMax * (-2 * pd + jMax * tf_p3 + 2 * tf * vf) - 6 * jMax_jMax * ph4);
				const double ph6 = -vd_vd + jMax * tf * (-2 * pd + 3 * tf * v0 - tf * vf) - af * g2;
				double polynom[4];
				polynom[0] = -(4 * (a0_p3 - af_p3) - 12 * a0_a0 * (af - jMax * tf) + 6 * a0 * (2 * af_af - 2 * af * jMax * tf + jMax * (jMax * tf_tf - 2 * vd)) + 6 * af * jMax * (3 * jMax * tf_tf + 2 * vd) -
							   6 * jMax_jMax * (-4 * pd + jMax * tf_p3 - 2 * tf * v0 + 6 * tf * vf)) /
							 (3 * jMax * ph1);
				polynom[1] = -(-a0_p4 - af_p4 + 4 * a0_p3 * (af - jMax * tf) + a0_a0 * (-6 * af_af + 8 * af * jMax * tf - 4 * jMax * (jMax * tf_tf - vd)) + 2 * af_af * jMax * (jMax * tf_tf + 2 * vd) -
							   4 * af * jMax_jMax * (-3 * pd + jMax * tf_p3 + 2 * tf * v0 + tf * vf) + jMax_jMax * (jMax_jMax * tf_p4 - 8 * vd_vd + 4 * jMax * tf * (-3 * pd + tf * v0 + 2 * tf * vf)) +
							   2 * a0 * (2 * af_p3 - 2 * af_af * jMax * tf + af * jMax * (-3 * jMax * tf_tf - 4 * vd) + jMax_jMax * (-6 * pd + jMax * tf_p3 - 4 * tf * v0 + 10 * tf * vf))) /
							 (jMax_jMax * ph1);
				polynom[2] = -(a0_p5 - af_p5 + af_p4 * jMax * tf - 5 * a0_p4 * (af - jMax * tf) + 2 * a0_p3 * ph3 + 4 * af_p3 * jMax * (jMax * tf_tf + vd) + 12 * jMax_jMax * af * ph6 -
							   2 * a0_a0 * (5 * af_p3 - 9 * af_af * jMax * tf - 6 * af * jMax * vd + 6 * jMax_jMax * (-2 * pd - tf * v0 + 3 * tf * vf)) - 12 * jMax_jMax * jMax * ph2 + a0 * ph5) /
							 (3 * jMax_jMax * jMax * ph1);
				polynom[3] = -(-a0_p6 - af_p6 + 6 * a0_p5 * (af - jMax * tf) - 48 * af_p3 * jMax_jMax * g1 + 72 * jMax_jMax * jMax * (jMax * g1 * g1 + vd_vd * vd + 2 * af * g1 * vd) - 3 * a0_p4 * ph3 -
							   36 * af_af * jMax_jMax * vd_vd + 6 * af_p4 * jMax * vd + 4 * a0_p3 * (5 * af_p3 - 9 * af_af * jMax * tf - 6 * af * jMax * vd + 6 * jMax_jMax * (-2 * pd - tf * v0 + 3 * tf * vf)) -
							   3 * a0_a0 * ph5 + 6 * a0 * (af_p5 - af_p4 * jMax * tf - 4 * af_p3 * jMax * (jMax * tf_tf + vd) + 12 * jMax_jMax * (-af * ph6 + jMax * ph2))) /
							 (18 * jMax_jMax * jMax_jMax * ph1);
				const double t_max = (a0 - aMin) / jMax;
				const Roots roots = o3_solve_quart(polynom[0], polynom[1], polynom[2], polynom[3]);
				for (int ri = 0; ri < roots.n; ri++) {
					double t = roots.x[ri];
					if (t > t_max) continue;
					// single Newton step (regarding pd)
					{
						const double h1 = ad_ad / 2 + jMax * (af * t + (jMax * t - a0) * (t - tf) - vd);
						const double h2 = -ad + jMax * (tf - 2 * t);
						const double h3 = sqrt(h1);
						const double orig = (af_p3 - a0_p3 + 3 * af * jMax * t * (af + jMax * t) + 3 * a0_a0 * (af + jMax * t) - 3 * a0 * (af_af + 2 * af * jMax * t + jMax_jMax * (t * t - tf_tf)) +
											 3 * jMax_jMax * (-2 * pd + jMax * t * (t - tf) * tf + 2 * tf * v0)) /
												(6 * jMax_jMax) -
											h3 * h3 * h3 / (jMax * o3_abs(jMax)) + ((-ad - jMax * t) * h1) / (jMax_jMax);
						const double deriv = (6 * jMax * h2 * h3 / o3_abs(jMax) + 2 * (-ad - jMax * tf) * h2 -
											  2 * (3 * ad_ad + af * jMax * (8 * t - 2 * tf) + 4 * a0 * jMax * (-2 * t + tf) + 2 * jMax * (jMax * t * (3 * t - 2 * tf) - vd))) /
											 (4 * jMax);
						t -= orig / deriv;
					}
					const double h1 = sqrt(2 * ad_ad + 4 * jMax * (ad * t + a0 * tf + jMax * t * (t - tf) - vd)) / o3_abs(jMax);
					// solution 2 with aPlat
					w.t[0] = 0;
					w.t[1] = 0;
					w.t[2] = t;
					w.t[3] = tf - 2 * t - ad / jMax - h1;
					w.t[4] = h1 / 2;
					w.t[5] = 0;
					w.t[6] = tf - (t + w.t[3] + w.t[4]);
					if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
				}
			}
		}
		// profiles with a3 != 0, UDUD
		{
			// T 0124
			{
				const double ph0 = -2 * pd - tf * v0 + 3 * tf * vf;
				const double ph1 = -ad + jMax * tf;
				const double ph2 = jMax * tf_tf * g1 - vd * ph0;
				const double ph3 = 5 * af_af + 2 * jMax * (2 * jMax * tf_tf - vd - 4 * af * tf);
				const double ph4 = jMax_jMax * tf_p4 - 2 * vd_vd + 8 * jMax * tf * (-pd + tf * vf);
				const double ph5 = (5 * af_p4 - 8 * af_p3 * jMax * tf - 12 * af_af * jMax * (jMax * tf_tf + vd) + 24 * af * jMax_jMax * (-2 * pd + jMax * tf_p3 + 2 * tf * vf) - 6 * jMax_jMax * ph4);
				const double ph6 = -vd_vd + jMax * tf * (-2 * pd + 3 * tf * v0 - tf * vf);
				const double ph7 = 3 * jMax_jMax * ph1 * ph1;
				double polynom[4];
				polynom[0] = (4 * af * tf - 2 * jMax * tf_tf - 4 * vd) / ph1;
				polynom[1] = (-2 * (a0_p4 + af_p4) + 8 * af_p3 * jMax * tf + 6 * af_af * jMax_jMax * tf_tf + 8 * a0_p3 * (af - jMax * tf) - 12 * a0_a0 * (af - jMax * tf) * (af - jMax * tf) -
							  12 * af * jMax_jMax * (-pd + jMax * tf_p3 - 2 * tf * v0 + 3 * tf * vf) +
							  2 * a0 * (4 * af_p3 - 12 * af_af * jMax * tf + 9 * af * jMax_jMax * tf_tf - 3 * jMax_jMax * (2 * pd + jMax * tf_p3 - 2 * tf * vf)) +
							  3 * jMax_jMax * (jMax_jMax * tf_p4 + 4 * vd_vd - 4 * jMax * tf * (pd + tf * v0 - 2 * tf * vf))) /
							 ph7;
				polynom[2] = (-a0_p5 + af_p5 - af_p4 * jMax * tf + 5 * a0_p4 * (af - jMax * tf) - 2 * a0_p3 * ph3 - 4 * af_p3 * jMax * (jMax * tf_tf + vd) + 12 * af_af * jMax_jMax * g2 - 12 * af * jMax_jMax * ph6 +
							  2 * a0_a0 * (5 * af_p3 - 9 * af_af * jMax * tf - 6 * af * jMax * vd + 6 * jMax_jMax * ph0) + 12 * jMax_jMax * jMax * ph2 +
							  a0 * (-5 * af_p4 + 8 * af_p3 * jMax * tf + 12 * af_af * jMax * (jMax * tf_tf + vd) - 24 * af * jMax_jMax * (-2 * pd + jMax * tf_p3 + 2 * tf * vf) + 6 * jMax_jMax * ph4)) /
							 (jMax * ph7);
				polynom[3] = -(a0_p6 + af_p6 - 6 * a0_p5 * (af - jMax * tf) + 48 * af_p3 * jMax_jMax * g1 - 72 * jMax_jMax * jMax * (jMax * g1 * g1 + vd_vd * vd + 2 * af * g1 * vd) + 3 * a0_p4 * ph3 -
							   6 * af_p4 * jMax * vd + 36 * af_af * jMax_jMax * vd_vd - 4 * a0_p3 * (5 * af_p3 - 9 * af_af * jMax * tf - 6 * af * jMax * vd + 6 * jMax_jMax * ph0) + 3 * a0_a0 * ph5 -
							   6 * a0 * (af_p5 - af_p4 * jMax * tf - 4 * af_p3 * jMax * (jMax * tf_tf + vd) + 12 * jMax_jMax * (af_af * g2 - af * ph6 + jMax * ph2))) /
							 (6 * jMax_jMax * ph7);
				const Roots roots = o3_solve_quart(polynom[0], polynom[1], polynom[2], polynom[3]);
				for (int ri = 0; ri < roots.n; ri++) {
					const double t = roots.x[ri];
					if (t > tf || t > (aMax - a0) / jMax) continue;
					const double h1 = sqrt(ad_ad / (2 * jMax_jMax) + (a0 * (t + tf) - af * t + jMax * t * tf - vd) / jMax);
					w.t[0] = t;
					w.t[1] = tf - ad / jMax - 2 * h1;
					w.t[2] = h1;
					w.t[3] = 0;
					w.t[4] = ad / jMax + h1 - t;
					w.t[5] = 0;
					w.t[6] = 0;
					if (o3_check<UDUD, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
				}
			}
		}
		// 3 step profile (UZD), sometimes missed because of numerical errors: T 012
		{
			const double h1 = sqrt(-ad_ad + jMax * (2 * (a0 + af) * tf - 4 * vd + jMax * tf_tf)) / o3_abs(jMax);
			w.t[0] = (tf - h1 + ad / jMax) / 2;
			w.t[1] = h1;
			w.t[2] = (tf - h1 - ad / jMax) / 2;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = 0;
			if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		// 3 step profile (UZU)
		{
			double polynom[4];
			polynom[0] = ad_ad;
			polynom[1] = ad_ad * tf;
			polynom[2] = (a0_a0 + af_af + 10 * a0 * af) * tf_tf + 24 * (tf * (af * v0 - a0 * vf) - pd * ad) + 12 * vd_vd;
			polynom[3] = -3 * tf * ((a0_a0 + af_af + 2 * a0 * af) * tf_tf - 4 * vd * (a0 + af) * tf + 4 * vd_vd);
			const Roots roots = o3_solve_cub(polynom[0], polynom[1], polynom[2], polynom[3]);
			for (int ri = 0; ri < roots.n; ri++) {
				const double t = roots.x[ri];
				if (t > tf) continue;
				const double jf = ad / (tf - t);
				w.t[0] = (2 * (vd - a0 * tf) + ad * (t - tf)) / (2 * jf * t);
				w.t[1] = t;
				w.t[2] = 0;
				w.t[3] = 0;
				w.t[4] = 0;
				w.t[5] = 0;
				w.t[6] = tf - (w.t[0] + w.t[1]);
				if (o3_check_jf<UDDU, L_NONE>(w, jf, vMax, vMin, aMax, aMin, jMax)) return true;
			}
		}
		// 3 step profile (UDU)
		{
			w.t[0] = (ad_ad / jMax + 2 * (a0 + af) * tf - jMax * tf_tf - 4 * vd) / (4 * (ad - jMax * tf));
			w.t[1] = 0;
			w.t[2] = -ad / (2 * jMax) + tf / 2;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = tf - (w.t[0] + w.t[2]);
			if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) return true;
		}
		return false;
	}

	// get_profile, :1113-1145 (minimize_jerk is off in the reference's wrappers: time_none_smooth is never tried)
	O3_NI bool run(P3& w) {
		const bool up_first = (pd > tf * v0);
		const double vMax = up_first ? _vMax : _vMin;
		const double vMin = up_first ? _vMin : _vMax;
		const double aMax = up_first ? _aMax : _aMin;
		const double aMin = up_first ? _aMin : _aMax;
		const double jMax = up_first ? _jMax : -_jMax;
		return time_acc0_acc1_vel(w, vMax, vMin, aMax, aMin, jMax) || time_vel(w, vMax, vMin, aMax, aMin, jMax) || time_acc0_vel(w, vMax, vMin, aMax, aMin, jMax) ||
			   time_acc1_vel(w, vMax, vMin, aMax, aMin, jMax) || time_acc0_acc1_vel(w, vMin, vMax, aMin, aMax, -jMax) || time_vel(w, vMin, vMax, aMin, aMax, -jMax) ||
			   time_acc0_vel(w, vMin, vMax, aMin, aMax, -jMax) || time_acc1_vel(w, vMin, vMax, aMin, aMax, -jMax) || time_acc0_acc1(w, vMax, vMin, aMax, aMin, jMax) ||
			   time_acc0(w, vMax, vMin, aMax, aMin, jMax) || time_acc1(w, vMax, vMin, aMax, aMin, jMax) || time_none(w, vMax, vMin, aMax, aMin, jMax) ||
			   time_acc0_acc1(w, vMin, vMax, aMin, aMax, -jMax) || time_acc0(w, vMin, vMax, aMin, aMax, -jMax) || time_acc1(w, vMin, vMax, aMin, aMax, -jMax) ||
			   time_none(w, vMin, vMax, aMin, aMax, -jMax);
	}
};

}  // namespace otg3
}  // namespace saip

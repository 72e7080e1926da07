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
// PositionThirdOrderStep1: the extremal (time-optimal and interval-ending) profiles of one dof -- /root/reference/ruckig/src/ruckig/position-third-step1.cpp.
// Included by saip_otg3.h (types and helpers there).
#pragma once

namespace saip {
namespace otg3 {

struct O3Step1 {
	double v0, a0, vf, af, _vMax, _vMin, _aMax, _aMin, _jMax;
	double pd, v0_v0, vf_vf, a0_a0, a0_p3, a0_p4, af_af, af_p3, af_p4, jMax_jMax;
	P3 w;        // candidate under construction (carries the boundary state)
	P3 vp[6];    // valid_profiles
	int cnt;
	bool overflow;

	O3_FN void init(double p0, double v0_, double a0_, double pf, double vf_, double af_, double vMax, double vMin, double aMax, double aMin, double jMax) {  // :6-22
		v0 = v0_;
		a0 = a0_;
		vf = vf_;
		af = af_;
		_vMax = vMax;
		_vMin = vMin;
		_aMax = aMax;
		_aMin = aMin;
		_jMax = jMax;
		pd = pf - p0;
		v0_v0 = v0 * v0;
		vf_vf = vf * vf;
		a0_a0 = a0 * a0;
		af_af = af * af;
		a0_p3 = a0 * a0_a0;
		a0_p4 = a0_a0 * a0_a0;
		af_p3 = af * af_af;
		af_p4 = af_af * af_af;
		jMax_jMax = jMax * jMax;
		w.p[0] = p0;
		w.v[0] = v0;
		w.a[0] = a0;
		w.pf = pf;
		w.vf = vf;
		w.af = af;
		w.tsum = 0.0;
		w.dir = w.cs = w.lim = 0;
		for (int i = 0; i < 7; i++) w.t[i] = w.j[i] = 0.0;
		for (int i = 1; i < 8; i++) w.a[i] = w.v[i] = w.p[i] = 0.0;
		cnt = 0;
		overflow = false;
	}
	O3_FN void add() {  // add_profile, position.hpp
		if (cnt < 6) vp[cnt] = w;
		else overflow = true;
		cnt++;
	}

	O3_NI void time_all_vel(double vMax, double vMin, double aMax, double aMin, double jMax) {  // :24-92
		// ACC0_ACC1_VEL
		w.t[0] = (-a0 + aMax) / jMax;
		w.t[1] = (a0_a0 / 2 - aMax * aMax - jMax * (v0 - vMax)) / (aMax * jMax);
		w.t[2] = aMax / jMax;
		w.t[3] = (3 * (a0_p4 * aMin - af_p4 * aMax) + 8 * aMax * aMin * (af_p3 - a0_p3 + 3 * jMax * (a0 * v0 - af * vf)) + 6 * a0_a0 * aMin * (aMax * aMax - 2 * jMax * v0) -
				  6 * af_af * aMax * (aMin * aMin - 2 * jMax * vf) -
				  12 * jMax * (aMax * aMin * (aMax * (v0 + vMax) - aMin * (vf + vMax) - 2 * jMax * pd) + (aMin - aMax) * jMax * vMax * vMax + jMax * (aMax * vf_vf - aMin * v0_v0))) /
				 (24 * aMax * aMin * jMax_jMax * vMax);
		w.t[4] = -aMin / jMax;
		w.t[5] = -(af_af / 2 - aMin * aMin - jMax * (vf - vMax)) / (aMin * jMax);
		w.t[6] = w.t[4] + af / jMax;
		if (o3_check<UDDU, L_ACC0_ACC1_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) {
			add();
			return;
		}
		// ACC1_VEL
		const double t_acc0 = sqrt(a0_a0 / (2 * jMax_jMax) + (vMax - v0) / jMax);
		w.t[0] = t_acc0 - a0 / jMax;
		w.t[1] = 0;
		w.t[2] = t_acc0;
		w.t[3] = -(3 * af_p4 - 8 * aMin * (af_p3 - a0_p3) - 24 * aMin * jMax * (a0 * v0 - af * vf) + 6 * af_af * (aMin * aMin - 2 * jMax * vf) -
				   12 * jMax * (2 * aMin * jMax * pd + aMin * aMin * (vf + vMax) + jMax * (vMax * vMax - vf_vf) + aMin * t_acc0 * (a0_a0 - 2 * jMax * (v0 + vMax)))) /
				 (24 * aMin * jMax_jMax * vMax);
		if (o3_check<UDDU, L_ACC1_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) {
			add();
			return;
		}
		// ACC0_VEL
		const double t_acc1 = sqrt(af_af / (2 * jMax_jMax) + (vMax - vf) / jMax);
		w.t[0] = (-a0 + aMax) / jMax;
		w.t[1] = (a0_a0 / 2 - aMax * aMax - jMax * (v0 - vMax)) / (aMax * jMax);
		w.t[2] = aMax / jMax;
		w.t[3] = (3 * a0_p4 + 8 * aMax * (af_p3 - a0_p3) + 24 * aMax * jMax * (a0 * v0 - af * vf) + 6 * a0_a0 * (aMax * aMax - 2 * jMax * v0) -
				  12 * jMax * (-2 * aMax * jMax * pd + aMax * aMax * (v0 + vMax) + jMax * (vMax * vMax - v0_v0) + aMax * t_acc1 * (-af_af + 2 * (vf + vMax) * jMax))) /
				 (24 * aMax * jMax_jMax * vMax);
		w.t[4] = t_acc1;
		w.t[5] = 0;
		w.t[6] = t_acc1 + af / jMax;
		if (o3_check<UDDU, L_ACC0_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) {
			add();
			return;
		}
		// VEL
		w.t[0] = t_acc0 - a0 / jMax;
		w.t[1] = 0;
		w.t[2] = t_acc0;
		w.t[3] = (af_p3 - a0_p3) / (3 * jMax_jMax * vMax) + (a0 * v0 - af * vf + (af_af * t_acc1 + a0_a0 * t_acc0) / 2) / (jMax * vMax) - (v0 / vMax + 1.0) * t_acc0 -
				 (vf / vMax + 1.0) * t_acc1 + pd / vMax;
		if (o3_check<UDDU, L_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) add();
	}

	O3_NI void time_acc0_acc1(double vMax, double vMin, double aMax, double aMin, double jMax, bool return_after_found) {  // :94-135
		double h1 = (3 * (af_p4 * aMax - a0_p4 * aMin) + aMax * aMin * (8 * (a0_p3 - af_p3) + 3 * aMax * aMin * (aMax - aMin) + 6 * aMin * af_af - 6 * aMax * a0_a0) +
					 12 * jMax * (aMax * aMin * ((aMax - 2 * a0) * v0 - (aMin - 2 * af) * vf) + aMin * a0_a0 * v0 - aMax * af_af * vf)) /
						(3 * (aMax - aMin) * jMax_jMax) +
					4 * (aMax * vf_vf - aMin * v0_v0 - 2 * aMin * aMax * pd) / (aMax - aMin);
		if (h1 >= 0) {
			h1 = sqrt(h1) / 2;
			const double h2 = a0_a0 / (2 * aMax * jMax) + (aMin - 2 * aMax) / (2 * jMax) - v0 / aMax;
			const double h3 = -af_af / (2 * aMin * jMax) - (aMax - 2 * aMin) / (2 * jMax) + vf / aMin;
			// UDDU: Solution 2
			if (h2 > h1 / aMax && h3 > -h1 / aMin) {
				w.t[0] = (-a0 + aMax) / jMax;
				w.t[1] = h2 - h1 / aMax;
				w.t[2] = aMax / jMax;
				w.t[3] = 0;
				w.t[4] = -aMin / jMax;
				w.t[5] = h3 + h1 / aMin;
				w.t[6] = w.t[4] + af / jMax;
				if (o3_check<UDDU, L_ACC0_ACC1, true>(w, jMax, vMax, vMin, aMax, aMin)) {
					add();
					if (return_after_found) return;
				}
			}
			// UDDU: Solution 1
			if (h2 > -h1 / aMax && h3 > h1 / aMin) {
				w.t[0] = (-a0 + aMax) / jMax;
				w.t[1] = h2 + h1 / aMax;
				w.t[2] = aMax / jMax;
				w.t[3] = 0;
				w.t[4] = -aMin / jMax;
				w.t[5] = h3 - h1 / aMin;
				w.t[6] = w.t[4] + af / jMax;
				if (o3_check<UDDU, L_ACC0_ACC1, true>(w, jMax, vMax, vMin, aMax, aMin)) add();
			}
		}
	}

	O3_NI void time_all_none_acc0_acc1(double vMax, double vMin, double aMax, double aMin, double jMax, bool return_after_found) {  // :137-307
		// NONE UDDU / UDUD strategy: t7 == 0
		const double h2_none = (a0_a0 - af_af) / (2 * jMax) + (vf - v0);
		const double h2_h2 = h2_none * h2_none;
		const double t_min_none = (a0 - af) / jMax;
		const double t_max_none = (aMax - aMin) / jMax;
		double polynom_none[4];
		polynom_none[0] = 0;
		polynom_none[1] = -2 * (a0_a0 + af_af - 2 * jMax * (v0 + vf)) / jMax_jMax;
		polynom_none[2] = 4 * (a0_p3 - af_p3 + 3 * jMax * (af * vf - a0 * v0)) / (3 * jMax * jMax_jMax) - 4 * pd / jMax;
		polynom_none[3] = -h2_h2 / jMax_jMax;
		// ACC0
		const double h3_acc0 = (a0_a0 - af_af) / (2 * aMax * jMax) + (vf - v0) / aMax;
		const double t_min_acc0 = (aMax - af) / jMax;
		const double t_max_acc0 = (aMax - aMin) / jMax;
		const double h0_acc0 = 3 * (af_p4 - a0_p4) + 8 * (a0_p3 - af_p3) * aMax + 24 * aMax * jMax * (af * vf - a0 * v0) - 6 * a0_a0 * (aMax * aMax - 2 * jMax * v0) +
							   6 * af_af * (aMax * aMax - 2 * jMax * vf) + 12 * jMax * (jMax * (vf_vf - v0_v0 - 2 * aMax * pd) - aMax * aMax * (vf - v0));
		const double h2_acc0 = -af_af + aMax * aMax + 2 * jMax * vf;
		double polynom_acc0[4];
		polynom_acc0[0] = -2 * aMax / jMax;
		polynom_acc0[1] = h2_acc0 / jMax_jMax;
		polynom_acc0[2] = 0;
		polynom_acc0[3] = h0_acc0 / (12 * jMax_jMax * jMax_jMax);
		// ACC1
		const double h3_acc1 = -(a0_a0 + af_af) / (2 * jMax * aMin) + aMin / jMax + (vf - v0) / aMin;
		const double t_min_acc1 = (aMin - a0) / jMax;
		const double t_max_acc1 = (aMax - a0) / jMax;
		const double h0_acc1 = (a0_p4 - af_p4) / 4 + 2 * (af_p3 - a0_p3) * aMin / 3 + (a0_a0 - af_af) * aMin * aMin / 2 +
							   jMax * (af_af * vf + a0_a0 * v0 + 2 * aMin * (jMax * pd - a0 * v0 - af * vf) + aMin * aMin * (v0 + vf) + jMax * (v0_v0 - vf_vf));
		const double h2_acc1 = a0_a0 - a0 * aMin + 2 * jMax * v0;
		double polynom_acc1[4];
		polynom_acc1[0] = 2 * (2 * a0 - aMin) / jMax;
		polynom_acc1[1] = (5 * a0_a0 + aMin * (aMin - 6 * a0) + 2 * jMax * v0) / jMax_jMax;
		polynom_acc1[2] = 2 * (a0 - aMin) * h2_acc1 / (jMax_jMax * jMax);
		polynom_acc1[3] = h0_acc1 / (jMax_jMax * jMax_jMax);
		// Budan's theorem on the shifted polynomial
		double polynom_acc0_min[4] = {polynom_acc0[0], polynom_acc0[1], polynom_acc0[2], polynom_acc0[3]};
		polynom_acc0_min[0] += 4 * t_min_acc0;
		polynom_acc0_min[1] += (3 * polynom_acc0[0] + 6 * t_min_acc0) * t_min_acc0;
		polynom_acc0_min[2] += (2 * polynom_acc0[1] + (3 * polynom_acc0[0] + 4 * t_min_acc0) * t_min_acc0) * t_min_acc0;
		polynom_acc0_min[3] += (polynom_acc0[2] + (polynom_acc0[1] + (polynom_acc0[0] + t_min_acc0) * t_min_acc0) * t_min_acc0) * t_min_acc0;
		const bool polynom_acc0_has_solution = (polynom_acc0_min[0] < 0.0) || (polynom_acc0_min[1] < 0.0) || (polynom_acc0_min[2] < 0.0) || (polynom_acc0_min[3] <= 0.0);
		const bool polynom_acc1_has_solution = (polynom_acc1[0] < 0.0) || (polynom_acc1[1] < 0.0) || (polynom_acc1[2] < 0.0) || (polynom_acc1[3] <= 0.0);

		Roots roots_none = o3_solve_quart(polynom_none[0], polynom_none[1], polynom_none[2], polynom_none[3]);
		Roots roots_acc0, roots_acc1;
		roots_acc0.clear();
		roots_acc1.clear();
		if (polynom_acc0_has_solution) roots_acc0 = o3_solve_quart(polynom_acc0[0], polynom_acc0[1], polynom_acc0[2], polynom_acc0[3]);
		if (polynom_acc1_has_solution) roots_acc1 = o3_solve_quart(polynom_acc1[0], polynom_acc1[1], polynom_acc1[2], polynom_acc1[3]);

		for (int ri = 0; ri < roots_none.n; ri++) {
			double t = roots_none.x[ri];
			if (t < t_min_none || t > t_max_none) continue;
			// single Newton step (regarding pd)
			if (t > O3_EPS) {
				const double h1 = jMax * t * t;
				const double orig = -h2_h2 / (4 * jMax * t) + h2_none * (af / jMax + t) +
									(4 * a0_p3 + 2 * af_p3 - 6 * a0_a0 * (af + 2 * jMax * t) + 12 * (af - a0) * jMax * v0 + 3 * jMax_jMax * (-4 * pd + (h1 + 8 * v0) * t)) / (12 * jMax_jMax);
				const double deriv = h2_none + 2 * v0 - a0_a0 / jMax + h2_h2 / (4 * h1) + (3 * h1) / 4;
				t -= orig / deriv;
			}
			const double h0 = h2_none / (2 * jMax * t);
			w.t[0] = h0 + t / 2 - a0 / jMax;
			w.t[1] = 0;
			w.t[2] = t;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = -h0 + t / 2 + af / jMax;
			if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				if (return_after_found) return;
			}
		}
		for (int ri = 0; ri < roots_acc0.n; ri++) {
			double t = roots_acc0.x[ri];
			if (t < t_min_acc0 || t > t_max_acc0) continue;
			if (t > O3_EPS) {
				const double h1 = jMax * t;
				const double orig = h0_acc0 / (12 * jMax_jMax * t) + t * (h2_acc0 + h1 * (h1 - 2 * aMax));
				const double deriv = 2 * (h2_acc0 + h1 * (2 * h1 - 3 * aMax));
				t -= orig / deriv;
			}
			w.t[0] = (-a0 + aMax) / jMax;
			w.t[1] = h3_acc0 - 2 * t + jMax / aMax * t * t;
			w.t[2] = t;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = (af - aMax) / jMax + t;
			if (o3_check<UDDU, L_ACC0, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				if (return_after_found) return;
			}
		}
		for (int ri = 0; ri < roots_acc1.n; ri++) {
			double t = roots_acc1.x[ri];
			if (t < t_min_acc1 || t > t_max_acc1) continue;
			// double Newton step (regarding pd)
			if (t > O3_EPS) {
				const double h5 = a0_p3 + 2 * jMax * a0 * v0;
				double h1 = jMax * t;
				double orig = -(h0_acc1 / 2 + h1 * (h5 + a0 * (aMin - 2 * h1) * (aMin - h1) + a0_a0 * (5 * h1 / 2 - 2 * aMin) + aMin * aMin * h1 / 2 + jMax * (h1 / 2 - aMin) * (h1 * t + 2 * v0))) / jMax;
				double deriv = (aMin - a0 - h1) * (h2_acc1 + h1 * (4 * a0 - aMin + 2 * h1));
				t -= o3_min(orig / deriv, t);
				h1 = jMax * t;
				orig = -(h0_acc1 / 2 + h1 * (h5 + a0 * (aMin - 2 * h1) * (aMin - h1) + a0_a0 * (5 * h1 / 2 - 2 * aMin) + aMin * aMin * h1 / 2 + jMax * (h1 / 2 - aMin) * (h1 * t + 2 * v0))) / jMax;
				if (o3_abs(orig) > 1e-9) {
					deriv = (aMin - a0 - h1) * (h2_acc1 + h1 * (4 * a0 - aMin + 2 * h1));
					t -= orig / deriv;
					h1 = jMax * t;
					orig = -(h0_acc1 / 2 + h1 * (h5 + a0 * (aMin - 2 * h1) * (aMin - h1) + a0_a0 * (5 * h1 / 2 - 2 * aMin) + aMin * aMin * h1 / 2 + jMax * (h1 / 2 - aMin) * (h1 * t + 2 * v0))) / jMax;
					if (o3_abs(orig) > 1e-9) {
						deriv = (aMin - a0 - h1) * (h2_acc1 + h1 * (4 * a0 - aMin + 2 * h1));
						t -= orig / deriv;
					}
				}
			}
			w.t[0] = t;
			w.t[1] = 0;
			w.t[2] = (a0 - aMin) / jMax + t;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = h3_acc1 - (2 * a0 + jMax * t) * t / aMin;
			w.t[6] = (af - aMin) / jMax;
			if (o3_check<UDDU, L_ACC1, true>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				if (return_after_found) return;
			}
		}
	}

	O3_NI void time_acc1_vel_two_step(double vMax, double vMin, double aMax, double aMin, double jMax) {  // :310-322
		w.t[0] = 0;
		w.t[1] = 0;
		w.t[2] = a0 / jMax;
		w.t[3] = -(3 * af_p4 - 8 * aMin * (af_p3 - a0_p3) - 24 * aMin * jMax * (a0 * v0 - af * vf) + 6 * af_af * (aMin * aMin - 2 * jMax * vf) -
				   12 * jMax * (2 * aMin * jMax * pd + aMin * aMin * (vf + vMax) + jMax * (vMax * vMax - vf_vf) + aMin * a0 * (a0_a0 - 2 * jMax * (v0 + vMax)) / jMax)) /
				 (24 * aMin * jMax_jMax * vMax);
		w.t[4] = -aMin / jMax;
		w.t[5] = -(af_af / 2 - aMin * aMin + jMax * (vMax - vf)) / (aMin * jMax);
		w.t[6] = w.t[4] + af / jMax;
		if (o3_check<UDDU, L_ACC1_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) add();
	}

	O3_NI void time_acc0_two_step(double vMax, double vMin, double aMax, double aMin, double jMax) {  // :324-397
		// two step
		{
			w.t[0] = 0;
			w.t[1] = (af_af - a0_a0 + 2 * jMax * (vf - v0)) / (2 * a0 * jMax);
			w.t[2] = (a0 - af) / jMax;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = 0;
			if (o3_check<UDDU, L_ACC0, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				return;
			}
		}
		// three step - removed pf
		{
			w.t[0] = (-a0 + aMax) / jMax;
			w.t[1] = (a0_a0 + af_af - 2 * aMax * aMax + 2 * jMax * (vf - v0)) / (2 * aMax * jMax);
			w.t[2] = (-af + aMax) / jMax;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = 0;
			if (o3_check<UDDU, L_ACC0, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				return;
			}
		}
		// three step - removed aMax
		{
			const double h0 = 3 * (af_af - a0_a0 + 2 * jMax * (v0 + vf));
			const double h2 = a0_p3 + 2 * af_p3 + 6 * jMax_jMax * pd + 6 * (af - a0) * jMax * vf - 3 * a0 * af_af;
			const double h1 = sqrt(2 * (2 * h2 * h2 + h0 * (a0_p4 - 6 * a0_a0 * (af_af + 2 * jMax * vf) + 8 * a0 * (af_p3 + 3 * jMax_jMax * pd + 3 * af * jMax * vf) -
															 3 * (af_p4 + 4 * af_af * jMax * vf + 4 * jMax_jMax * (vf_vf - v0_v0))))) *
							  o3_abs(jMax) / jMax;
			w.t[0] = (4 * af_p3 + 2 * a0_p3 - 6 * a0 * af_af + 12 * jMax_jMax * pd + 12 * (af - a0) * jMax * vf + h1) / (2 * jMax * h0);
			w.t[1] = -h1 / (jMax * h0);
			w.t[2] = (-4 * a0_p3 - 2 * af_p3 + 6 * a0_a0 * af + 12 * jMax_jMax * pd - 12 * (af - a0) * jMax * v0 + h1) / (2 * jMax * h0);
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = 0;
			if (o3_check<UDDU, L_ACC0, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				return;
			}
		}
		// three step - t = (aMax - aMin) / jMax
		{
			const double t = (aMax - aMin) / jMax;
			w.t[0] = (-a0 + aMax) / jMax;
			w.t[1] = (a0_a0 - af_af) / (2 * aMax * jMax) + (vf - v0 + jMax * t * t) / aMax - 2 * t;
			w.t[2] = t;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = (af - aMin) / jMax;
			if (o3_check<UDDU, L_ACC0, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				return;
			}
		}
	}

	O3_NI void time_vel_two_step(double vMax, double vMin, double aMax, double aMin, double jMax) {  // :399-436
		const double h1 = sqrt(af_af / (2 * jMax_jMax) + (vMax - vf) / jMax);
		// four step
		{
			w.t[0] = -a0 / jMax;
			w.t[1] = 0;
			w.t[2] = 0;
			w.t[3] = (af_p3 - a0_p3) / (3 * jMax_jMax * vMax) + (a0 * v0 - af * vf + (af_af * h1) / 2) / (jMax * vMax) - (vf / vMax + 1.0) * h1 + pd / vMax;
			w.t[4] = h1;
			w.t[5] = 0;
			w.t[6] = h1 + af / jMax;
			if (o3_check<UDDU, L_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				return;
			}
		}
		// four step
		{
			w.t[0] = 0;
			w.t[1] = 0;
			w.t[2] = a0 / jMax;
			w.t[3] = (af_p3 - a0_p3) / (3 * jMax_jMax * vMax) + (a0 * v0 - af * vf + (af_af * h1 + a0_p3 / jMax) / 2) / (jMax * vMax) - (v0 / vMax + 1.0) * a0 / jMax -
					 (vf / vMax + 1.0) * h1 + pd / vMax;
			w.t[4] = h1;
			w.t[5] = 0;
			w.t[6] = h1 + af / jMax;
			if (o3_check<UDDU, L_VEL, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				return;
			}
		}
	}

	O3_NI void time_none_two_step(double vMax, double vMin, double aMax, double aMin, double jMax) {  // :438-472
		// two step
		{
			const double h0 = sqrt((a0_a0 + af_af) / 2 + jMax * (vf - v0)) * o3_abs(jMax) / jMax;
			w.t[0] = (h0 - a0) / jMax;
			w.t[1] = 0;
			w.t[2] = (h0 - af) / jMax;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = 0;
			if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				return;
			}
		}
		// single step
		{
			w.t[0] = (af - a0) / jMax;
			w.t[1] = 0;
			w.t[2] = 0;
			w.t[3] = 0;
			w.t[4] = 0;
			w.t[5] = 0;
			w.t[6] = 0;
			if (o3_check<UDDU, L_NONE, false>(w, jMax, vMax, vMin, aMax, aMin)) {
				add();
				return;
			}
		}
	}

	// get_profile, :522-585 (the zero-limits special case :524-537 cannot occur: the wrappers reject limits <= 0)
	O3_NI bool run(Block3& blk, double brake_dur) {
		if (o3_abs(vf) < O3_EPS && o3_abs(af) < O3_EPS) {
			const bool up = (pd >= 0);
			const double vMax = up ? _vMax : _vMin, vMin = up ? _vMin : _vMax, aMax = up ? _aMax : _aMin, aMin = up ? _aMin : _aMax, jMax = up ? _jMax : -_jMax;
			if (o3_abs(v0) < O3_EPS && o3_abs(a0) < O3_EPS && o3_abs(pd) < O3_EPS) {
				time_all_none_acc0_acc1(vMax, vMin, aMax, aMin, jMax, true);
			} else {
				// there is no blocked interval when vf == 0 && af == 0: return after the first profile found
				time_all_vel(vMax, vMin, aMax, aMin, jMax);
				if (cnt == 0) time_all_none_acc0_acc1(vMax, vMin, aMax, aMin, jMax, true);
				if (cnt == 0) time_acc0_acc1(vMax, vMin, aMax, aMin, jMax, true);
				if (cnt == 0) time_all_vel(vMin, vMax, aMin, aMax, -jMax);
				if (cnt == 0) time_all_none_acc0_acc1(vMin, vMax, aMin, aMax, -jMax, true);
				if (cnt == 0) time_acc0_acc1(vMin, vMax, aMin, aMax, -jMax, true);
			}
		} else {
			time_all_none_acc0_acc1(_vMax, _vMin, _aMax, _aMin, _jMax, false);
			time_all_none_acc0_acc1(_vMin, _vMax, _aMin, _aMax, -_jMax, false);
			time_acc0_acc1(_vMax, _vMin, _aMax, _aMin, _jMax, false);
			time_acc0_acc1(_vMin, _vMax, _aMin, _aMax, -_jMax, false);
			time_all_vel(_vMax, _vMin, _aMax, _aMin, _jMax);
			time_all_vel(_vMin, _vMax, _aMin, _aMax, -_jMax);
		}
		if (cnt == 0) {
			time_none_two_step(_vMax, _vMin, _aMax, _aMin, _jMax);
			if (cnt == 0) time_none_two_step(_vMin, _vMax, _aMin, _aMax, -_jMax);
			if (cnt == 0) time_acc0_two_step(_vMax, _vMin, _aMax, _aMin, _jMax);
			if (cnt == 0) time_acc0_two_step(_vMin, _vMax, _aMin, _aMax, -_jMax);
			if (cnt == 0) time_vel_two_step(_vMax, _vMin, _aMax, _aMin, _jMax);
			if (cnt == 0) time_vel_two_step(_vMin, _vMax, _aMin, _aMax, -_jMax);
			if (cnt == 0) time_acc1_vel_two_step(_vMax, _vMin, _aMax, _aMin, _jMax);
			if (cnt == 0) time_acc1_vel_two_step(_vMin, _vMax, _aMin, _aMax, -_jMax);
		}
		if (overflow) return false;
		return o3_block(blk, vp, cnt, brake_dur);
	}
};

}  // namespace otg3
}  // namespace saip

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
// Batched online trajectory generation (internal OTG of JointTask, acceleration-limited mode -- the reference default).
//
// Reference call stack, one control cycle of one robot (JointTask.cpp:313-319):
//   OTG_joints::setGoalPositionAndVelocity   /root/reference/src/helper_modules/OTG_joints.cpp:98-114
//   OTG_joints::update                       OTG_joints.cpp:116-150
//     ruckig::Ruckig::update                 /root/reference/ruckig/include/ruckig/ruckig.hpp:184-221
//       TargetCalculator::calculate          ruckig/include/ruckig/calculator_target.hpp:228-524   (only when the input changed)
//         BrakeProfile (second order)        ruckig/src/ruckig/brake.cpp:82-102, include/ruckig/brake.hpp:65-75
//         PositionSecondOrderStep1           ruckig/src/ruckig/position-second-step1.cpp:11-136, block.hpp:60-134
//         synchronize                        calculator_target.hpp:128-207
//         phase synchronization              calculator_target.hpp:44-126, 381-455
//         PositionSecondOrderStep2           ruckig/src/ruckig/position-second-step2.cpp:13-117
//       Trajectory::at_time                  ruckig/include/ruckig/trajectory.hpp:65-143
//
// MI355X mapping: ONE LANE PER (instance, task dof).  A group of GS = 8 (dof <= 8) or 32 lanes is one robot, so a wavefront
// carries 8 (or 2) robots; the per-dof extremal profiles (step 1) and the time-synchronised profiles (step 2) are computed
// independently in every lane, and the three cross-dof steps of the algorithm -- the choice of the synchronisation time among the
// 3*dof candidate durations, the collinearity test of phase synchronisation, and the wrapper's norm tests -- are ballots and
// xor-butterflies inside the lane group.  No LDS, no barriers, no divergence between robots except where their trajectories
// really differ.  State lives in HBM lane-major ([field][instance*GS + dof]) so that every load and store of the cycle is one
// fully coalesced 512-byte access per wavefront.
//
// Arithmetic: the expressions below are evaluated in the same order as the reference's, with FMA contraction switched off, so
// branch decisions that compare against 1e-8 ... 2^-52 thresholds see (almost always bit-) identical numbers.
#include <hip/hip_runtime.h>
#include <math.h>

#include "saip_device.h"
#include "saip_dynamics_oct_body.h"
#include "saip_fk.h"

#pragma clang fp contract(off)

#include "saip_otg3.h"  // jerk-limited (third-order) profiles: brake, step 1, step 2, sampling -- per-lane functions

namespace saip {

namespace {

constexpr double OTG_DBL_EPS = 2.220446049250313e-16;  // std::numeric_limits<double>::epsilon() / DBL_EPSILON
constexpr double OTG_INF = __builtin_huge_val();

// ruckig::Profile restricted to what the second-order position interface touches: segment durations t[0..6] and accelerations
// a[0..6] (a[7] = af = 0, jerk = 0), the direction flag, and t_sum.back().  p[0], v[0], pf, vf are held by the caller.
struct Prof {
	double t[7], a[7];
	double tsum;  // t_sum.back()
	int dir;      // 0 UP, 1 DOWN
};

__device__ __forceinline__ double dabs(double x) { return __builtin_fabs(x); }

// Profile::check_for_second_order<UDDU, *>, profile.hpp:308-350.  Fills a[], dir, tsum; optionally returns p.back(), v.back().
__device__ __forceinline__ bool prof_check(Prof& pr, double p0, double v0, double pf, double vf, double aUp, double aDown, double vMax, double vMin,
											double* p_back = nullptr, double* v_back = nullptr) {
	if (pr.t[0] < 0) return false;
	double ts = pr.t[0];
#pragma unroll
	for (int i = 0; i < 6; i++) {
		if (pr.t[i + 1] < 0) return false;
		ts = ts + pr.t[i + 1];
	}
	pr.tsum = ts;
	if (ts > 1e12) return false;
	pr.a[0] = pr.t[0] > 0 ? aUp : 0.0;
	pr.a[1] = 0.0;
	pr.a[2] = pr.t[2] > 0 ? aDown : 0.0;
	pr.a[3] = 0.0;
	pr.a[4] = pr.t[4] > 0 ? aDown : 0.0;
	pr.a[5] = 0.0;
	pr.a[6] = pr.t[6] > 0 ? aUp : 0.0;
	pr.dir = (vMax > 0) ? 0 : 1;
	const double vUppLim = (pr.dir == 0 ? vMax : vMin) + 1e-12;
	const double vLowLim = (pr.dir == 0 ? vMin : vMax) - 1e-12;
	double v[8], p = p0;
	v[0] = v0;
#pragma unroll
	for (int i = 0; i < 7; i++) {
		v[i + 1] = v[i] + pr.t[i] * pr.a[i];
		p = p + pr.t[i] * (v[i] + pr.t[i] * pr.a[i] / 2);
	}
	if (p_back) *p_back = p;
	if (v_back) *v_back = v[7];
	return dabs(p - pf) < 1e-8 && dabs(v[7] - vf) < 1e-8 && v[2] <= vUppLim && v[3] <= vUppLim && v[4] <= vUppLim && v[5] <= vUppLim &&
		   v[6] <= vUppLim && v[2] >= vLowLim && v[3] >= vLowLim && v[4] >= vLowLim && v[5] >= vLowLim && v[6] >= vLowLim;
}

__device__ __forceinline__ void prof_zero_t(Prof& pr) {
#pragma unroll
	for (int i = 0; i < 7; i++) pr.t[i] = 0.0;
}

// Block (block.hpp:14-134) for at most three valid profiles: the fastest profile and at most one blocked interval
struct Block {
	Prof pmin, aprof;
	double t_min, a_left, a_right;
	bool has_a;
};

struct Step1 {  // PositionSecondOrderStep1, position-second-step1.cpp
	double p0, v0, pf, vf, pd, brake_dur;
	Prof cand[3];
	int cnt;
	bool overflow;

	__device__ __forceinline__ void add(const Prof& pr) {
		if (cnt < 3) cand[cnt] = pr;
		else overflow = true;  // the reference's std::array<Profile, 3> would be overrun
		cnt++;
	}
	__device__ __forceinline__ void time_acc0(double vMax, double vMin, double aMax, double aMin) {  // :11-25
		Prof pr;
		prof_zero_t(pr);
		pr.t[0] = (-v0 + vMax) / aMax;
		pr.t[1] = (aMin * v0 * v0 - aMax * vf * vf) / (2 * aMax * aMin * vMax) + vMax * (aMax - aMin) / (2 * aMax * aMin) + pd / vMax;
		pr.t[2] = (vf - vMax) / aMin;
		if (prof_check(pr, p0, v0, pf, vf, aMax, aMin, vMax, vMin)) add(pr);
	}
	__device__ __forceinline__ void time_none(double vMax, double vMin, double aMax, double aMin, bool return_after_found) {  // :27-63
		double h1 = (aMax * vf * vf - aMin * v0 * v0 - 2 * aMax * aMin * pd) / (aMax - aMin);
		if (h1 >= 0.0) {
			h1 = sqrt(h1);
			Prof pr;
			prof_zero_t(pr);
			pr.t[0] = -(v0 + h1) / aMax;
			pr.t[2] = (vf + h1) / aMin;
			if (prof_check(pr, p0, v0, pf, vf, aMax, aMin, vMax, vMin)) {
				add(pr);
				if (return_after_found) return;
			}
			pr.t[0] = (-v0 + h1) / aMax;
			pr.t[1] = 0.0;
			pr.t[2] = (vf - h1) / aMin;
			if (prof_check(pr, p0, v0, pf, vf, aMax, aMin, vMax, vMin)) add(pr);
		}
	}
	// get_profile, :99-136 (the zero-velocity-limit special case :101-114 cannot occur: the wrappers reject limits <= 0)
	__device__ __forceinline__ bool run(double _vMax, double _vMin, double _aMax, double _aMin, Block& blk) {
		cnt = 0;
		overflow = false;
		pd = pf - p0;
		if (dabs(vf) < OTG_DBL_EPS) {
			const bool up = (pd >= 0);
			const double vMax = up ? _vMax : _vMin, vMin = up ? _vMin : _vMax, aMax = up ? _aMax : _aMin, aMin = up ? _aMin : _aMax;
			time_none(vMax, vMin, aMax, aMin, true);
			if (cnt == 0) time_acc0(vMax, vMin, aMax, aMin);
			if (cnt == 0) time_none(vMin, vMax, aMin, aMax, true);
			if (cnt == 0) time_acc0(vMin, vMax, aMin, aMax);
		} else {
			time_none(_vMax, _vMin, _aMax, _aMin, false);
			time_none(_vMin, _vMax, _aMin, _aMax, false);
			time_acc0(_vMax, _vMin, _aMax, _aMin);
			time_acc0(_vMin, _vMax, _aMin, _aMax);
		}
		if (overflow) return false;
		return calculate_block(blk);
	}
	__device__ __forceinline__ Prof pick(int i) const { return i == 0 ? cand[0] : (i == 1 ? cand[1] : cand[2]); }
	__device__ __forceinline__ void interval(Block& blk, int il, int ir) const {  // Block::Interval(profile_left, profile_right), block.hpp:30-42
		const Prof L = pick(il), R = pick(ir);
		const double ld = L.tsum + brake_dur + 0.0, rd = R.tsum + brake_dur + 0.0;
		blk.has_a = true;
		if (ld < rd) {
			blk.a_left = ld;
			blk.a_right = rd;
			blk.aprof = R;
		} else {
			blk.a_left = rd;
			blk.a_right = ld;
			blk.aprof = L;
		}
	}
	__device__ __forceinline__ void set_min(Block& blk, int i) const {  // block.hpp:45-50
		blk.pmin = pick(i);
		blk.t_min = blk.pmin.tsum + brake_dur + 0.0;
		blk.has_a = false;
		blk.a_left = blk.a_right = OTG_INF;
	}
	__device__ __forceinline__ bool calculate_block(Block& blk) const {  // block.hpp:60-134 for counter <= 3
		if (cnt == 1) {
			set_min(blk, 0);
			return true;
		}
		if (cnt == 2) {
			if (dabs(cand[0].tsum - cand[1].tsum) < 8 * OTG_DBL_EPS) {
				set_min(blk, 0);
				return true;
			}
			const int idx_min = (cand[0].tsum < cand[1].tsum) ? 0 : 1;
			const int idx_else = (idx_min + 1) % 2;
			set_min(blk, idx_min);
			interval(blk, idx_min, idx_else);
			return true;
		}
		if (cnt == 3) {
			int idx_min = 0;  // std::min_element: first minimal
			if (cand[1].tsum < cand[idx_min].tsum) idx_min = 1;
			if (cand[2].tsum < (idx_min == 0 ? cand[0].tsum : cand[1].tsum)) idx_min = 2;
			set_min(blk, idx_min);
			interval(blk, (idx_min + 1) % 3, (idx_min + 2) % 3);
			return true;
		}
		return false;
	}
};

struct Step2 {  // PositionSecondOrderStep2, position-second-step2.cpp
	double tf, p0, v0, pf, vf, pd, vd;

	__device__ __forceinline__ bool time_acc0(Prof& pr, double vMax, double vMin, double aMax, double aMin) const {  // :13-66
		{
			const double h1 = sqrt((2 * aMax * (pd - tf * vf) - 2 * aMin * (pd - tf * v0) + vd * vd) / (aMax * aMin) + tf * tf);
			prof_zero_t(pr);
			pr.t[0] = (aMax * vd - aMax * aMin * (tf - h1)) / (aMax * (aMax - aMin));
			pr.t[1] = h1;
			pr.t[2] = tf - (pr.t[0] + h1);
			if (prof_check(pr, p0, v0, pf, vf, aMax, aMin, vMax, vMin)) return true;
		}
		{
			const double h1 = (-vd + aMax * tf);
			prof_zero_t(pr);
			pr.t[0] = -vd * vd / (2 * aMax * h1) + (pd - v0 * tf) / h1;
			pr.t[1] = -vd / aMax + tf;
			pr.t[6] = tf - (pr.t[0] + pr.t[1]);
			if (prof_check(pr, p0, v0, pf, vf, aMax, aMin, vMax, vMin)) return true;
		}
		{
			prof_zero_t(pr);
			pr.t[1] = -vd / aMax + tf;
			pr.t[6] = vd / aMax;
			if (prof_check(pr, p0, v0, pf, vf, aMax, aMin, vMax, vMin)) return true;
		}
		return false;
	}
	__device__ __forceinline__ bool time_none(Prof& pr, double vMax, double vMin, double aMax, double aMin) const {  // :68-106
		if (dabs(v0) < OTG_DBL_EPS && dabs(vf) < OTG_DBL_EPS && dabs(pd) < OTG_DBL_EPS) {
			prof_zero_t(pr);
			pr.t[1] = tf;
			if (prof_check(pr, p0, v0, pf, vf, aMax, aMin, vMax, vMin)) return true;
		}
		{
			const double h1 = 2 * (vf * tf - pd);
			prof_zero_t(pr);
			pr.t[0] = h1 / vd;
			pr.t[1] = tf - pr.t[0];
			const double af = vd * vd / h1;
			if ((aMin - 1e-12 < af) && (af < aMax + 1e-12) && prof_check(pr, p0, v0, pf, vf, af, -af, vMax, vMin)) return true;
		}
		return false;
	}
	__device__ __forceinline__ bool check_all(Prof& pr, double vMax, double vMin, double aMax, double aMin) const {
		return time_acc0(pr, vMax, vMin, aMax, aMin) || time_none(pr, vMax, vMin, aMax, aMin);
	}
	__device__ __forceinline__ bool run(Prof& pr, double _vMax, double _vMin, double _aMax, double _aMin) {  // :108-117
		pd = pf - p0;
		vd = vf - v0;
		if (pd > 0) return check_all(pr, _vMax, _vMin, _aMax, _aMin) || check_all(pr, _vMin, _vMax, _aMin, _aMax);
		return check_all(pr, _vMin, _vMax, _aMin, _aMax) || check_all(pr, _vMax, _vMin, _aMax, _aMin);
	}
};

// ---------------------------------------------------------------- lane-group collectives (GS lanes = one robot)
template <int GS>
struct Grp {
	int lane, base, dof;
	__device__ __forceinline__ Grp() {
		lane = threadIdx.x & 63;
		dof = lane & (GS - 1);
		base = lane - dof;
	}
	__device__ __forceinline__ unsigned long long bits(bool pred) const {
		const unsigned long long m = __ballot(pred ? 1 : 0);
		const unsigned long long gm = (GS == 64) ? ~0ull : ((1ull << GS) - 1ull);
		return (m >> base) & gm;
	}
	__device__ __forceinline__ bool any(bool pred) const { return bits(pred) != 0ull; }
	__device__ __forceinline__ double sum(double x) const {
#pragma unroll
		for (int off = 1; off < GS; off <<= 1) x = x + __shfl_xor(x, off);
		return x;
	}
	__device__ __forceinline__ double bcast(double x, int src_dof) const { return __shfl(x, base + src_dof); }
	__device__ __forceinline__ int bcast(int x, int src_dof) const { return __shfl(x, base + src_dof); }
};

// state arena fields, each [ngroups * GS] doubles, lane-major
enum {
	F_IN_P, F_IN_V, F_IN_A,      // wrapper _input.current_*          (OTG_joints.h:187)
	F_TG_P, F_TG_V,              // wrapper _input.target_*
	F_CI_P, F_CI_V, F_CI_A,      // ruckig current_input.current_*    (ruckig.hpp:31)
	F_CI_TP, F_CI_TV,            // ruckig current_input.target_*
	F_OUT_P, F_OUT_V, F_OUT_A,   // wrapper _output.new_*
	F_T0, F_T1, F_T2, F_T6,      // chosen profile: the segment durations the second-order interface can make non-zero
	F_A0, F_A2, F_A6,            //                 and their accelerations
	F_P0, F_V0,                  // profile start (after the brake pre-trajectory)
	F_BR_T, F_BR_A, F_BR_P, F_BR_V,  // brake pre-trajectory: duration, acceleration, start state
	// jerk-limited mode only (the third-order profile has seven live segments, a jerk per segment and a two-section brake pre-trajectory)
	F3_T3, F3_T4, F3_T5,
	F3_A1, F3_A3, F3_A4, F3_A5, F3_A7,
	F3_J0, F3_J1, F3_J2, F3_J3, F3_J4, F3_J5, F3_J6,
	F3_BR_T1, F3_BR_J0, F3_BR_J1, F3_BR_P1, F3_BR_V1, F3_BR_A1,
	F_COUNT
};

struct Sample {
	double p, v, a;
};
// Trajectory::state_to_integrate_from + integrate for one dof (trajectory.hpp:65-143, utils.hpp:43-49), single section, jerk 0
__device__ __forceinline__ Sample sample_profile(double time, double duration, const double t[7], const double a[7], double p0, double v0, double br_t,
												  double br_a, double br_p, double br_v) {
	// p[i], v[i], t_sum[i] exactly as check_for_second_order left them
	double p[8], v[8], ts[7];
	p[0] = p0;
	v[0] = v0;
#pragma unroll
	for (int i = 0; i < 7; i++) {
		v[i + 1] = v[i] + t[i] * a[i];
		p[i + 1] = p[i] + t[i] * (v[i] + t[i] * a[i] / 2);
		ts[i] = (i == 0) ? t[0] : ts[i - 1] + t[i];
	}
	double td, ps, vs, as;
	if (time >= duration) {
		td = time - (br_t + ts[6]);
		ps = p[7];
		vs = v[7];
		as = 0.0;  // a.back() = af
	} else {
		td = time;
		bool in_brake = false;
		if (br_t > 0) {
			if (td < br_t) in_brake = true;
			else td = td - br_t;
		}
		if (in_brake) {
			ps = br_p;
			vs = br_v;
			as = br_a;
		} else if (td >= ts[6]) {
			td = td - ts[6];
			ps = p[7];
			vs = v[7];
			as = 0.0;
		} else {
			int idx = 0;  // std::upper_bound(t_sum, td): first i with t_sum[i] > td
#pragma unroll
			for (int i = 6; i >= 0; i--)
				if (ts[i] > td) idx = i;
			double tprev = 0.0;
			ps = p[0];
			vs = v[0];
			as = a[0];
#pragma unroll
			for (int i = 1; i < 7; i++)
				if (idx == i) {
					tprev = ts[i - 1];
					ps = p[i];
					vs = v[i];
					as = a[i];
				}
			if (idx > 0) td = td - tprev;
		}
	}
	Sample s;
	s.p = ps + td * (vs + td * (as / 2 + td * 0.0 / 6));
	s.v = vs + td * (as + td * 0.0 / 2);
	s.a = as + td * 0.0;
	return s;
}


// ---------------------------------------------------------------- per-lane state and the shared Ruckig::update
struct Lane {
	double in_p, in_v, in_a, tg_p, tg_v, out_p, out_v, out_a;  // wrapper _input.current/target, _output.new
	double ci_p, ci_v, ci_a, ci_tp, ci_tv;                      // ruckig current_input
	double t[7], a[7], P0, V0, br_t, br_a, br_p, br_v;          // stored trajectory of this dof
	// jerk-limited mode: a[] holds a[0..6] of the third-order profile, a7 its last entry; br_t / br_a / br_p / br_v are section 0 of the brake
	double j[7], a7, br_t1, br_j0, br_j1, br_p1, br_v1, br_a1;
};

#define ST(f) S[(size_t)(f) * stride + li]
template <bool JERK = false>
__device__ __forceinline__ void lane_load(Lane& L, const double* S, size_t stride, size_t li, bool act) {
	L.in_p = L.in_v = L.in_a = L.tg_p = L.tg_v = L.out_p = L.out_v = L.out_a = 0.0;
	L.ci_p = L.ci_v = L.ci_a = L.ci_tp = L.ci_tv = 0.0;
#pragma unroll
	for (int i = 0; i < 7; i++) L.t[i] = L.a[i] = 0.0;
	L.P0 = L.V0 = L.br_t = L.br_a = L.br_p = L.br_v = 0.0;
#pragma unroll
	for (int i = 0; i < 7; i++) L.j[i] = 0.0;
	L.a7 = L.br_t1 = L.br_j0 = L.br_j1 = L.br_p1 = L.br_v1 = L.br_a1 = 0.0;
	if (!act) return;
	L.in_p = ST(F_IN_P);
	L.in_v = ST(F_IN_V);
	L.in_a = ST(F_IN_A);
	L.tg_p = ST(F_TG_P);
	L.tg_v = ST(F_TG_V);
	L.out_p = ST(F_OUT_P);
	L.out_v = ST(F_OUT_V);
	L.out_a = ST(F_OUT_A);
	L.ci_p = ST(F_CI_P);
	L.ci_v = ST(F_CI_V);
	L.ci_a = ST(F_CI_A);
	L.ci_tp = ST(F_CI_TP);
	L.ci_tv = ST(F_CI_TV);
	L.t[0] = ST(F_T0);
	L.t[1] = ST(F_T1);
	L.t[2] = ST(F_T2);
	L.t[6] = ST(F_T6);
	L.a[0] = ST(F_A0);
	L.a[2] = ST(F_A2);
	L.a[6] = ST(F_A6);
	L.P0 = ST(F_P0);
	L.V0 = ST(F_V0);
	L.br_t = ST(F_BR_T);
	L.br_a = ST(F_BR_A);
	L.br_p = ST(F_BR_P);
	L.br_v = ST(F_BR_V);
	if (JERK) {
		L.t[3] = ST(F3_T3);
		L.t[4] = ST(F3_T4);
		L.t[5] = ST(F3_T5);
		L.a[1] = ST(F3_A1);
		L.a[3] = ST(F3_A3);
		L.a[4] = ST(F3_A4);
		L.a[5] = ST(F3_A5);
		L.a7 = ST(F3_A7);
#pragma unroll
		for (int i = 0; i < 7; i++) L.j[i] = ST(F3_J0 + i);
		L.br_t1 = ST(F3_BR_T1);
		L.br_j0 = ST(F3_BR_J0);
		L.br_j1 = ST(F3_BR_J1);
		L.br_p1 = ST(F3_BR_P1);
		L.br_v1 = ST(F3_BR_V1);
		L.br_a1 = ST(F3_BR_A1);
	}
}
// new_traj: a trajectory was (re)calculated this cycle; plain sampling cycles leave the profile fields alone
template <bool JERK = false>
__device__ __forceinline__ void lane_store(const Lane& L, double* S, size_t stride, size_t li, bool new_traj) {
	ST(F_IN_P) = L.in_p;
	ST(F_IN_V) = L.in_v;
	ST(F_IN_A) = L.in_a;
	ST(F_TG_P) = L.tg_p;
	ST(F_TG_V) = L.tg_v;
	ST(F_OUT_P) = L.out_p;
	ST(F_OUT_V) = L.out_v;
	ST(F_OUT_A) = L.out_a;
	ST(F_CI_P) = L.ci_p;
	ST(F_CI_V) = L.ci_v;
	ST(F_CI_A) = L.ci_a;
	if (new_traj) {
		ST(F_CI_TP) = L.ci_tp;
		ST(F_CI_TV) = L.ci_tv;
		ST(F_T0) = L.t[0];
		ST(F_T1) = L.t[1];
		ST(F_T2) = L.t[2];
		ST(F_T6) = L.t[6];
		ST(F_A0) = L.a[0];
		ST(F_A2) = L.a[2];
		ST(F_A6) = L.a[6];
		ST(F_P0) = L.P0;
		ST(F_V0) = L.V0;
		ST(F_BR_T) = L.br_t;
		ST(F_BR_A) = L.br_a;
		ST(F_BR_P) = L.br_p;
		ST(F_BR_V) = L.br_v;
		if (JERK) {
			ST(F3_T3) = L.t[3];
			ST(F3_T4) = L.t[4];
			ST(F3_T5) = L.t[5];
			ST(F3_A1) = L.a[1];
			ST(F3_A3) = L.a[3];
			ST(F3_A4) = L.a[4];
			ST(F3_A5) = L.a[5];
			ST(F3_A7) = L.a7;
#pragma unroll
			for (int i = 0; i < 7; i++) ST(F3_J0 + i) = L.j[i];
			ST(F3_BR_T1) = L.br_t1;
			ST(F3_BR_J0) = L.br_j0;
			ST(F3_BR_J1) = L.br_j1;
			ST(F3_BR_P1) = L.br_p1;
			ST(F3_BR_V1) = L.br_v1;
			ST(F3_BR_A1) = L.br_a1;
		}
	}
}
#undef ST

// ruckig::Ruckig::update (ruckig.hpp:184-221) for the robots of this wavefront whose `run` is set (group-uniform).
// Returns the ruckig::Result (0 Working, 1 Finished, < 0 error); *new_traj tells whether a trajectory was calculated and stored.
template <int GS>
__device__ __forceinline__ int ruckig_update(const Grp<GS>& g, bool act, bool run, int m, double vMax, double aMax, bool new_limits, double dt, Lane& L,
											 int& flags, double& time, double& duration, bool* new_traj) {
	const int dof = g.dof;
	const double vMin = -vMax, aMin = -aMax;
	int result = 0;
	*new_traj = false;
	const bool differs = act && !(L.in_p == L.ci_p && L.in_v == L.ci_v && L.in_a == L.ci_a && L.tg_p == L.ci_tp && L.tg_v == L.ci_tv);
	const bool need_calc = run && (g.any(differs) || !(flags & OTG_FLAG_RUCKIG_INIT) || new_limits);

	if (g.any(need_calc)) {
		// validate_input (input_parameter.hpp:155-330, check_target_state_within_limits only): NaNs and |vf| beyond the limits
		bool bad = act && need_calc && (isnan(L.in_p) || isnan(L.in_v) || isnan(L.in_a) || isnan(L.tg_p) || isnan(L.tg_v) || L.tg_v > vMax || L.tg_v < vMin);
		int err = g.any(bad) ? -100 : 0;

		// ---------------------------------------------------- per-dof: brake pre-trajectory + step 1 (calculator_target.hpp:235-318)
		Block blk;
		blk.t_min = 0.0;
		blk.has_a = false;
		blk.a_left = blk.a_right = OTG_INF;
		double nb_t = 0.0, nb_a = 0.0, np0 = L.in_p, nv0 = L.in_v;
		{
			if (L.in_v > vMax) {  // brake.cpp:82-102
				nb_a = aMin;
				nb_t = (vMax - L.in_v) / aMin + 2.2e-14;
			} else if (L.in_v < vMin) {
				nb_a = aMax;
				nb_t = (vMin - L.in_v) / aMax + 2.2e-14;
			}
			if (nb_t > 0.0) {  // finalize_second_order, brake.hpp:65-75
				np0 = L.in_p + nb_t * (L.in_v + nb_t * (nb_a / 2 + nb_t * 0.0 / 6));
				nv0 = L.in_v + nb_t * (nb_a + nb_t * 0.0 / 2);
			} else {
				nb_t = 0.0;
			}
			Step1 s1;
			s1.p0 = np0;
			s1.v0 = nv0;
			s1.pf = L.tg_p;
			s1.vf = L.tg_v;
			s1.brake_dur = nb_t;
			const bool found = s1.run(vMax, vMin, aMax, aMin, blk);
			if (err == 0 && g.any(act && need_calc && !found)) err = -110;  // ErrorExecutionTimeCalculation
		}

		// ---------------------------------------------------- synchronize (calculator_target.hpp:128-207): the smallest candidate
		// duration {t_min, a.right} of any dof that no dof blocks; ties resolved towards the lower candidate index
		double t_sync = OTG_INF;
		int lim_idx = -1;  // candidate index = quot * m + dof
		for (int c = 0; c < 2; c++) {
			const double mine = (c == 0) ? blk.t_min : (blk.has_a ? blk.a_right : OTG_INF);
			for (int s = 0; s < m; s++) {
				const double cand = g.bcast(mine, s);
				const bool blocked = act && ((cand < blk.t_min) || (blk.has_a && blk.a_left < cand && cand < blk.a_right));
				const bool is_blocked = g.any(blocked);
				if (!is_blocked && !isinf(cand) && !(cand < 0.0) && cand < t_sync) {
					t_sync = cand;
					lim_idx = c * m + s;
				}
			}
		}
		if (err == 0 && lim_idx < 0) err = -111;  // ErrorSynchronizationCalculation
		const int lim_dof = lim_idx < 0 ? 0 : lim_idx % m;
		const int lim_quot = lim_idx < 0 ? 0 : lim_idx / m;
		if (err == 0 && t_sync > 7.6e3) err = -101;  // ErrorTrajectoryDuration, :354-358

		Prof mine = blk.pmin;  // profile of this dof
		if (dof == lim_dof && lim_quot == 1) mine = blk.aprof;
		bool done = (t_sync == 0.0);  // :360-366: every dof keeps its p_min

		// ---------------------------------------------------- phase synchronisation (calculator_target.hpp:44-126, 373-455)
		if (err == 0 && !done) {
			const double pdv = L.tg_p - L.in_p;
			// the scale dof: the first dof with a non-zero pd, else v0, else a0, else vf (af = 0)
			int which = 0;  // 1 pd, 2 v0, 3 a0, 4 vf
			if (dabs(pdv) > OTG_DBL_EPS) which = 1;
			else if (dabs(L.in_v) > OTG_DBL_EPS) which = 2;
			else if (dabs(L.in_a) > OTG_DBL_EPS) which = 3;
			else if (dabs(L.tg_v) > OTG_DBL_EPS) which = 4;
			const unsigned long long cand_bits = g.bits(act && which != 0);
			bool collinear = cand_bits != 0ull;
			const int sdof = collinear ? __builtin_ctzll(cand_bits) : 0;
			const int swhich = g.bcast(which, sdof);
			const double sv = swhich == 1 ? pdv : (swhich == 2 ? L.in_v : (swhich == 3 ? L.in_a : L.tg_v));  // scale_vector[dof]
			const double scale = g.bcast(sv, sdof);
			const double pd_scale = g.bcast(pdv, sdof) / scale, v0_scale = g.bcast(L.in_v, sdof) / scale, vf_scale = g.bcast(L.tg_v, sdof) / scale;
			const double a0_scale = g.bcast(L.in_a, sdof) / scale, af_scale = 0.0 / scale;
			const double scale_limiting = g.bcast(sv, lim_dof);
			const int lim_dir = g.bcast(mine.dir, lim_dof);
			const double control_limiting = (lim_dir == 0) ? g.bcast(aMax, lim_dof) : g.bcast(aMin, lim_dof);
			const bool off = act && (dabs(pdv - pd_scale * sv) > OTG_DBL_EPS || dabs(L.in_v - v0_scale * sv) > OTG_DBL_EPS ||
									 dabs(L.in_a - a0_scale * sv) > OTG_DBL_EPS || dabs(L.tg_v - vf_scale * sv) > OTG_DBL_EPS ||
									 dabs(0.0 - af_scale * sv) > OTG_DBL_EPS);
			if (g.any(off)) collinear = false;
			const double phase_control = control_limiting * sv / scale_limiting;
			if (collinear) {
				Prof ph;
#pragma unroll
				for (int i = 0; i < 7; i++) ph.t[i] = g.bcast(mine.t[i], lim_dof);  // p.t = p_limiting.t
				bool okp = true;
				if (dof != lim_dof) {
					const double aUp = phase_control, aDown = -phase_control;
					okp = (aMin - 1e-12 < aUp) && (aUp < aMax + 1e-12) && (aMin - 1e-12 < aDown) && (aDown < aMax + 1e-12) &&
						  prof_check(ph, np0, nv0, L.tg_p, L.tg_v, aUp, aDown, vMax, vMin);
				}
				if (!g.any(act && !okp)) {
					if (dof != lim_dof) mine = ph;
					done = true;
				}
			}
		}
		// ---------------------------------------------------- time synchronisation (calculator_target.hpp:457-521)
		if (err == 0 && !done) {
			bool ok2 = true;
			if (dof != lim_dof) {
				const double t_profile = t_sync - nb_t - 0.0;
				if (dabs(t_profile - blk.t_min) < 2 * OTG_DBL_EPS) mine = blk.pmin;
				else if (blk.has_a && dabs(t_profile - blk.a_right) < 2 * OTG_DBL_EPS) mine = blk.aprof;
				else {
					Step2 s2;
					s2.tf = t_profile;
					s2.p0 = np0;
					s2.v0 = nv0;
					s2.pf = L.tg_p;
					s2.vf = L.tg_v;
					ok2 = s2.run(mine, vMax, vMin, aMax, aMin);
				}
			}
			if (g.any(act && !ok2)) err = -111;
		}

		if (need_calc) {
			if (err != 0) {
				result = err;  // ruckig.hpp:199-201: returns before touching current_input / output
			} else {
#pragma unroll
				for (int i = 0; i < 7; i++) {
					L.t[i] = mine.t[i];
					L.a[i] = mine.a[i];
				}
				L.P0 = np0;
				L.V0 = nv0;
				L.br_t = nb_t;
				L.br_a = nb_a;
				L.br_p = L.in_p;
				L.br_v = L.in_v;
				duration = t_sync;
				time = 0.0;
				L.ci_p = L.in_p;  // current_input = input, ruckig.hpp:203
				L.ci_v = L.in_v;
				L.ci_a = L.in_a;
				L.ci_tp = L.tg_p;
				L.ci_tv = L.tg_v;
				flags |= OTG_FLAG_RUCKIG_INIT;
				*new_traj = true;
			}
		}
	}

	if (run && result == 0) {
		// ruckig.hpp:209-220: advance one control period along the stored trajectory
		time = time + dt;
		const Sample s = sample_profile(time, duration, L.t, L.a, L.P0, L.V0, L.br_t, L.br_a, L.br_p, L.br_v);
		L.out_p = s.p;
		L.out_v = s.v;
		L.out_a = s.a;
		L.ci_p = L.out_p;  // output.pass_to_input(current_input)
		L.ci_v = L.out_v;
		L.ci_a = L.out_a;
		if (time > duration) result = 1;  // Finished
	}
	return result;
}


// ---------------------------------------------------------------- jerk-limited mode (JointTask::enableInternalOtgJerkLimited, JointTask.cpp:383-410;
// MotionForceTask.cpp:525-545; OTG_joints::setMaxJerk, OTG_joints.cpp:73-86): Ruckig::update with the third-order position interface.
// Same skeleton as ruckig_update above -- brake pre-trajectory, step 1 per dof, synchronisation over the lane group (now with up to two
// blocked intervals per dof: 3 m candidate durations), phase synchronisation (the control is the jerk), step 2 -- with the per-dof closed
// forms of saip_otg3.h.  The trajectory calculation keeps its profiles in private memory (it runs only when a goal or a limit changed);
// the per-cycle sampling does not touch them.
__device__ __noinline__ void o3_phase_check(otg3::P3& ph, const int cs, const double jf, const double vMax, const double vMin, const double aMax, const double aMin,
											const double jMax, bool* ok) {
	*ok = (cs == otg3::UDDU) ? otg3::o3_check_jf<otg3::UDDU, otg3::L_NONE>(ph, jf, vMax, vMin, aMax, aMin, jMax)
							 : otg3::o3_check_jf<otg3::UDUD, otg3::L_NONE>(ph, jf, vMax, vMin, aMax, aMin, jMax);
}
// the trajectory calculation (only lane groups whose input changed get here): out of line, on a COPY of the lane state -- a state struct whose
// address escapes into a call cannot be kept in registers, and the per-cycle sampling path around this call must not pay for that
template <int GS>
__device__ __noinline__ int ruckig_calc3(const Grp<GS>& g, bool act, bool need_calc, int m, double vMax, double aMax, double jMax, Lane& L, int& flags, double& time,
										 double& duration, bool* new_traj) {
	using namespace otg3;
	const int dof = g.dof;
	const double vMin = -vMax, aMin = -aMax;
	int result = 0;
	{
		// validate_input (input_parameter.hpp:155-330, check_target_state_within_limits only; the target acceleration is always 0 here)
		bool bad = act && need_calc && (isnan(L.in_p) || isnan(L.in_v) || isnan(L.in_a) || isnan(L.tg_p) || isnan(L.tg_v) || L.tg_v > vMax || L.tg_v < vMin);
		int err = g.any(bad) ? -100 : 0;

		// ---------------------------------------------------- per-dof: brake pre-trajectory + step 1 (calculator_target.hpp:235-318)
		Block3 blk;
		Brake3 br;
		double np0 = L.in_p, nv0 = L.in_v, na0 = L.in_a;
		{
			o3_brake(br, np0, nv0, na0, vMax, vMin, aMax, aMin, jMax);
			O3Step1 s1;
			s1.init(np0, nv0, na0, L.tg_p, L.tg_v, 0.0, vMax, vMin, aMax, aMin, jMax);
			bool found = false;
			if (act && need_calc && err == 0) found = s1.run(blk, br.duration);
			if (!found) {
				blk.t_min = 0.0;
				blk.has_a = blk.has_b = false;
				blk.a_left = blk.a_right = blk.b_left = blk.b_right = OTG_INF;
				blk.pmin = s1.w;
			}
			if (err == 0 && g.any(act && need_calc && !found)) err = -110;  // ErrorExecutionTimeCalculation
		}

		// ---------------------------------------------------- synchronize (calculator_target.hpp:128-207): the smallest candidate duration
		// {t_min, a.right, b.right} of any dof that no dof blocks; ties towards the lower candidate index
		double t_sync = OTG_INF;
		int lim_idx = -1;  // candidate index = quot * m + dof
		if (m == 1) {      // :320-325: a single dof keeps its fastest profile
			t_sync = blk.t_min;
			lim_idx = 0;
		} else {
			for (int c = 0; c < 3; c++) {
				const double mine = (c == 0) ? blk.t_min : (c == 1 ? (blk.has_a ? blk.a_right : OTG_INF) : (blk.has_b ? blk.b_right : OTG_INF));
				for (int s = 0; s < m; s++) {
					const double cand = g.bcast(mine, s);
					const bool blocked = act && ((cand < blk.t_min) || (blk.has_a && blk.a_left < cand && cand < blk.a_right) || (blk.has_b && blk.b_left < cand && cand < blk.b_right));
					const bool is_blocked = g.any(blocked);
					if (!is_blocked && !isinf(cand) && !(cand < 0.0) && cand < t_sync) {
						t_sync = cand;
						lim_idx = c * m + s;
					}
				}
			}
		}
		if (err == 0 && lim_idx < 0) err = -111;  // ErrorSynchronizationCalculation
		const int lim_dof = lim_idx < 0 ? 0 : lim_idx % m;
		const int lim_quot = lim_idx < 0 ? 0 : lim_idx / m;
		if (err == 0 && m > 1 && t_sync > 7.6e3) err = -101;  // ErrorTrajectoryDuration, :354-358

		P3 mine = blk.pmin;  // profile of this dof
		if (dof == lim_dof && lim_quot == 1) mine = blk.pa;
		if (dof == lim_dof && lim_quot == 2) mine = blk.pb;
		bool done = (t_sync == 0.0) || (m == 1);  // :360-366: every dof keeps its p_min

		// ---------------------------------------------------- phase synchronisation (calculator_target.hpp:44-126, 373-455)
		if (err == 0 && !done) {
			const double pdv = L.tg_p - L.in_p;
			int which = 0;  // 1 pd, 2 v0, 3 a0, 4 vf (af = 0)
			if (dabs(pdv) > OTG_DBL_EPS) which = 1;
			else if (dabs(L.in_v) > OTG_DBL_EPS) which = 2;
			else if (dabs(L.in_a) > OTG_DBL_EPS) which = 3;
			else if (dabs(L.tg_v) > OTG_DBL_EPS) which = 4;
			const unsigned long long cand_bits = g.bits(act && which != 0);
			bool collinear = cand_bits != 0ull;
			const int sdof = collinear ? __builtin_ctzll(cand_bits) : 0;
			const int swhich = g.bcast(which, sdof);
			const double sv = swhich == 1 ? pdv : (swhich == 2 ? L.in_v : (swhich == 3 ? L.in_a : L.tg_v));  // scale_vector[dof]
			const double scale = g.bcast(sv, sdof);
			const double pd_scale = g.bcast(pdv, sdof) / scale, v0_scale = g.bcast(L.in_v, sdof) / scale, vf_scale = g.bcast(L.tg_v, sdof) / scale;
			const double a0_scale = g.bcast(L.in_a, sdof) / scale, af_scale = 0.0 / scale;
			const double scale_limiting = g.bcast(sv, lim_dof);
			const int lim_dir = g.bcast(mine.dir, lim_dof);
			const int lim_cs = g.bcast(mine.cs, lim_dof);
			const double control_limiting = (lim_dir == 0) ? g.bcast(jMax, lim_dof) : -g.bcast(jMax, lim_dof);
			const bool off = act && (dabs(pdv - pd_scale * sv) > OTG_DBL_EPS || dabs(L.in_v - v0_scale * sv) > OTG_DBL_EPS ||
									 dabs(L.in_a - a0_scale * sv) > OTG_DBL_EPS || dabs(L.tg_v - vf_scale * sv) > OTG_DBL_EPS ||
									 dabs(0.0 - af_scale * sv) > OTG_DBL_EPS);
			if (g.any(off)) collinear = false;
			const double phase_control = control_limiting * sv / scale_limiting;
			if (collinear) {
				P3 ph = blk.pmin;  // boundary state of this dof
#pragma unroll
				for (int i = 0; i < 7; i++) ph.t[i] = g.bcast(mine.t[i], lim_dof);  // p.t = p_limiting.t
				bool okp = true;
				if (act && dof != lim_dof) o3_phase_check(ph, lim_cs, phase_control, vMax, vMin, aMax, aMin, jMax, &okp);
				if (!g.any(act && !okp)) {
					if (dof != lim_dof) mine = ph;
					done = true;
				}
			}
		}
		// ---------------------------------------------------- time synchronisation (calculator_target.hpp:457-521)
		if (err == 0 && !done) {
			bool ok2 = true;
			if (act && dof != lim_dof) {
				const double t_profile = t_sync - br.duration - 0.0;
				if (dabs(t_profile - blk.t_min) < 2 * OTG_DBL_EPS) mine = blk.pmin;
				else if (blk.has_a && dabs(t_profile - blk.a_right) < 2 * OTG_DBL_EPS) mine = blk.pa;
				else if (blk.has_b && dabs(t_profile - blk.b_right) < 2 * OTG_DBL_EPS) mine = blk.pb;
				else {
					O3Step2 s2;
					s2.init(t_profile, np0, nv0, na0, L.tg_p, L.tg_v, 0.0, vMax, vMin, aMax, aMin, jMax);
					mine = blk.pmin;  // boundary state
					ok2 = s2.run(mine);
				}
			}
			if (g.any(act && !ok2)) err = -111;
		}

		if (need_calc) {
			if (err != 0) {
				result = err;  // ruckig.hpp:199-201: returns before touching current_input / output
			} else {
#pragma unroll
				for (int i = 0; i < 7; i++) {
					L.t[i] = mine.t[i];
					L.a[i] = mine.a[i];
					L.j[i] = mine.j[i];
				}
				L.a7 = mine.a[7];
				L.P0 = mine.p[0];
				L.V0 = mine.v[0];
				L.br_t = br.t[0];
				L.br_t1 = (br.duration > 0 && br.t[1] > 0.0) ? br.t[1] : 0.0;
				L.br_j0 = br.j[0];
				L.br_j1 = br.j[1];
				L.br_p = br.p[0];
				L.br_v = br.v[0];
				L.br_a = br.a[0];
				L.br_p1 = br.p[1];
				L.br_v1 = br.v[1];
				L.br_a1 = br.a[1];
				if (!(br.duration > 0)) L.br_t = 0.0;
				duration = t_sync;
				time = 0.0;
				L.ci_p = L.in_p;  // current_input = input, ruckig.hpp:203
				L.ci_v = L.in_v;
				L.ci_a = L.in_a;
				L.ci_tp = L.tg_p;
				L.ci_tv = L.tg_v;
				flags |= OTG_FLAG_RUCKIG_INIT;
				*new_traj = true;
			}
		}
	}
	return result;
}

template <int GS>
__device__ __forceinline__ int ruckig_update3(const Grp<GS>& g, bool act, bool run, int m, double vMax, double aMax, double jMax, bool new_limits, double dt, Lane& L,
											  int& flags, double& time, double& duration, bool* new_traj) {
	using namespace otg3;
	int result = 0;
	*new_traj = false;
	const bool differs = act && !(L.in_p == L.ci_p && L.in_v == L.ci_v && L.in_a == L.ci_a && L.tg_p == L.ci_tp && L.tg_v == L.ci_tv);
	const bool need_calc = run && (g.any(differs) || !(flags & OTG_FLAG_RUCKIG_INIT) || new_limits);
	if (g.any(need_calc)) {
		Lane Lc = L;
		int fl = flags;
		double tm = time, du = duration;
		bool nt = false;
		result = ruckig_calc3<GS>(g, act, need_calc, m, vMax, aMax, jMax, Lc, fl, tm, du, &nt);
		L = Lc;
		flags = fl;
		time = tm;
		duration = du;
		*new_traj = nt;
	}

	if (run && result == 0) {
		// ruckig.hpp:209-220: advance one control period along the stored trajectory
		time = time + dt;
		P3 pr;
		Brake3 br;
#pragma unroll
		for (int i = 0; i < 7; i++) {
			pr.t[i] = L.t[i];
			pr.a[i] = L.a[i];
			pr.j[i] = L.j[i];
		}
		pr.a[7] = L.a7;
		pr.p[0] = L.P0;
		pr.v[0] = L.V0;
		br.t[0] = L.br_t;
		br.t[1] = L.br_t1;
		br.duration = L.br_t + L.br_t1;
		br.j[0] = L.br_j0;
		br.j[1] = L.br_j1;
		br.p[0] = L.br_p;
		br.v[0] = L.br_v;
		br.a[0] = L.br_a;
		br.p[1] = L.br_p1;
		br.v[1] = L.br_v1;
		br.a[1] = L.br_a1;
		double pn, vn, an;
		o3_sample(time, duration, pr, br, pn, vn, an);
		L.out_p = pn;
		L.out_v = vn;
		L.out_a = an;
		L.ci_p = L.out_p;  // output.pass_to_input(current_input)
		L.ci_v = L.out_v;
		L.ci_a = L.out_a;
		if (time > duration) result = 1;  // Finished
	}
	return result;
}

}  // namespace

// One control cycle of the internal OTG of one joint task for the whole batch.  mode 0: cycle (JointTask.cpp:313-319);
// mode 1: OTG_joints::reInitialize(S q) (JointTask.cpp:95-106, OTG_joints.cpp:28-42).
template <int GS, bool JERK = false>
__device__ __forceinline__ void otg_joints_body(const OtgDev& O, const int B, const int ld, const int mode, const int block) {
	const Grp<GS> g;
	const int gtid = block * blockDim.x + threadIdx.x;
	const int inst = gtid / GS;
	const int dof = g.dof;
	const int m = O.m;
	const bool inst_ok = inst < B;
	const bool act = inst_ok && dof < m;
	const size_t stride = (size_t)O.lanes;  // B * GS
	const size_t li = (size_t)gtid;         // lane-major slot (only touched when act)
	double* S = O.state;
#define ST(f) S[(size_t)(f) * stride + li]
	const size_t gi = (size_t)(act ? dof : 0) * ld + (inst_ok ? inst : 0);  // row `dof` of the task's goal / desired arrays

	if (mode == 1) {
		// current task position S q (JointTask.cpp:98), then setGoalPosition(initial): OTG_joints.cpp:36, 98-114
		double cur = 0.0;
		if (act) {
			const TaskDev& tk = *O.task;
			if (tk.s_identity) cur = O.q[(size_t)dof * ld + inst];
			else
				for (int j = 0; j < O.n; j++) cur = fma(tk.S[dof * O.n + j], O.q[(size_t)j * ld + inst], cur);
		}
		const double tp = act ? ST(F_TG_P) : 0.0, tv = act ? ST(F_TG_V) : 0.0;
		const double dp2 = g.sum((cur - tp) * (cur - tp)), gp2 = g.sum(cur * cur), tp2 = g.sum(tp * tp);
		const double tv2 = g.sum(tv * tv);
		const bool same = (dp2 <= 1e-24 * fmin(gp2, tp2)) && (tv2 <= 0.0);
		if (act) {
			if (!same) {
				ST(F_TG_P) = cur;
				ST(F_TG_V) = 0.0;
			}
			ST(F_OUT_P) = cur;  // :38-41
			ST(F_OUT_V) = 0.0;
			ST(F_OUT_A) = 0.0;
			ST(F_IN_P) = cur;
			ST(F_IN_V) = 0.0;
			ST(F_IN_A) = 0.0;
			O.desired[gi] = cur;  // JointTask.cpp:100-105
			O.desired[gi + (size_t)m * ld] = 0.0;
			O.desired[gi + (size_t)2 * m * ld] = 0.0;
			if (dof == 0) {
				int fl = O.flags[inst] & ~(OTG_FLAG_ERROR | OTG_FLAG_FINISHED_MOVING);
				if (!same) fl &= ~OTG_FLAG_GOAL_REACHED;
				O.flags[inst] = fl | OTG_FLAG_STATE_VALID;
				O.result[inst] = 1;
			}
		}
		return;
	}
#undef ST

	const double goal_p = act ? O.goal[gi] : 0.0;
	const double goal_v = act ? O.goal[gi + (size_t)m * ld] : 0.0;
	Lane L;
	lane_load<JERK>(L, S, stride, li, act);
	int flags = inst_ok ? O.flags[inst] : 0;
	const double vMax = act ? O.limits[dof] : 1.0, aMax = act ? O.limits[SAIP_MAXN + dof] : 1.0, jMax = (JERK && act) ? O.limits[2 * SAIP_MAXN + dof] : 1.0;
	// new limits since the last cycle: enableInternalOtgAccelerationLimited -> disableJerkLimits() zeroes the current acceleration
	// of the wrapper's input (OTG_joints.cpp:90-93); the changed limits themselves force a new trajectory (ruckig.hpp:197)
	const bool new_limits = inst_ok && (O.epoch != O.seen_epoch[inst]);
	if (new_limits && !JERK) L.in_a = 0.0;

	// setGoalPositionAndVelocity, OTG_joints.cpp:98-114 (Eigen isApprox: ||a-b||^2 <= 1e-24 min(||a||^2, ||b||^2))
	{
		const double dp2 = g.sum((goal_p - L.tg_p) * (goal_p - L.tg_p)), gp2 = g.sum(goal_p * goal_p), tp2 = g.sum(L.tg_p * L.tg_p);
		const double dv2 = g.sum((goal_v - L.tg_v) * (goal_v - L.tg_v)), gv2 = g.sum(goal_v * goal_v), tv2 = g.sum(L.tg_v * L.tg_v);
		const bool same = (dp2 <= 1e-24 * fmin(gp2, tp2)) && (dv2 <= 1e-24 * fmin(gv2, tv2));
		if (!same) {
			flags &= ~OTG_FLAG_GOAL_REACHED;
			L.tg_p = goal_p;
			L.tg_v = goal_v;
		}
	}

	int result = 0;
	bool new_traj = false;
	const bool run = inst_ok && !(flags & OTG_FLAG_GOAL_REACHED);  // OTG_joints.cpp:117-119
	if (g.any(run)) {
		double time = inst_ok ? O.time[inst] : 0.0, duration = inst_ok ? O.duration[inst] : 0.0;
		if (JERK) result = ruckig_update3<GS>(g, act, run, m, vMax, aMax, jMax, new_limits, O.dt, L, flags, time, duration, &new_traj);
		else result = ruckig_update<GS>(g, act, run, m, vMax, aMax, new_limits, O.dt, L, flags, time, duration, &new_traj);
		// OTG_joints::update after the ruckig call, OTG_joints.cpp:125-149
		if (run) {
			if (result == 1) {
				const double vn2 = g.sum(act ? L.out_v * L.out_v : 0.0);
				if (sqrt(vn2) < 1e-3) {
					flags |= OTG_FLAG_GOAL_REACHED;
				} else {
					// :131-133 would throw in the reference (a never-assigned member is passed on); the documented intent and the 6-dof
					// wrapper (OTG_6dof_cartesian.cpp:209-210) are "same target position, zero target velocity".  Counted in the flags.
					flags |= OTG_FLAG_FINISHED_MOVING;
					const double tv2 = g.sum(act ? L.tg_v * L.tg_v : 0.0);
					if (!(tv2 <= 0.0)) {
						flags &= ~OTG_FLAG_GOAL_REACHED;
						L.tg_v = 0.0;
					}
				}
			} else if (result == 0) {
				L.in_p = L.out_p;  // _output.pass_to_input(_input)
				L.in_v = L.out_v;
				L.in_a = L.out_a;
			} else {
				L.in_v = 0.0;  // error: keep the previous output (never overwritten), zero the input velocity and acceleration
				L.in_a = 0.0;
				flags |= OTG_FLAG_ERROR;
			}
			if (act && dof == 0) {
				O.time[inst] = time;
				O.duration[inst] = duration;
			}
		}
	}
	if (act) {
		lane_store<JERK>(L, S, stride, li, new_traj);
		// JointTask.cpp:317-319: the desired state the control law tracks
		O.desired[gi] = L.out_p;
		O.desired[gi + (size_t)m * ld] = L.out_v;
		O.desired[gi + (size_t)2 * m * ld] = L.out_a;
		if (dof == 0) {
			O.flags[inst] = flags;
			O.result[inst] = run ? result : 1;
			if (new_limits && (run ? result == 0 || result == 1 : true)) O.seen_epoch[inst] = O.epoch;
		}
	}
}


// ================================================================ Cartesian OTG of a motion-force task
// /root/reference/src/helper_modules/OTG_6dof_cartesian.cpp: six Ruckig dofs = position (3) + the rotation vector of the
// orientation relative to a reference frame that is re-based on every new orientation goal (:155-191).  Lane group of 8, dofs
// 0-2 linear, 3-5 angular; the 3x3 frame algebra is evaluated redundantly by every lane of the group (it is tiny next to the
// state traffic) and each lane keeps its own component.
namespace {

struct M3 {
	double m[9];
};
__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B) {
	M3 C;
#pragma unroll
	for (int r = 0; r < 3; r++)
#pragma unroll
		for (int c = 0; c < 3; c++) C.m[3 * r + c] = A.m[3 * r] * B.m[c] + A.m[3 * r + 1] * B.m[3 + c] + A.m[3 * r + 2] * B.m[6 + c];
	return C;
}
__device__ __forceinline__ M3 m3_tmul(const M3& A, const M3& B) {  // A^T B
	M3 C;
#pragma unroll
	for (int r = 0; r < 3; r++)
#pragma unroll
		for (int c = 0; c < 3; c++) C.m[3 * r + c] = A.m[r] * B.m[c] + A.m[3 + r] * B.m[3 + c] + A.m[6 + r] * B.m[6 + c];
	return C;
}
__device__ __forceinline__ void m3_vec(const M3& A, const double v[3], double o[3]) {
#pragma unroll
	for (int r = 0; r < 3; r++) o[r] = A.m[3 * r] * v[0] + A.m[3 * r + 1] * v[1] + A.m[3 * r + 2] * v[2];
}
__device__ __forceinline__ void m3_tvec(const M3& A, const double v[3], double o[3]) {  // A^T v
#pragma unroll
	for (int r = 0; r < 3; r++) o[r] = A.m[r] * v[0] + A.m[3 + r] * v[1] + A.m[6 + r] * v[2];
}
__device__ __forceinline__ double m3_sqnorm(const M3& A) {
	double s = 0.0;
#pragma unroll
	for (int e = 0; e < 9; e++) s += A.m[e] * A.m[e];
	return s;
}
__device__ __forceinline__ bool m3_approx(const M3& A, const M3& B, double prec) {  // Eigen isApprox on matrices (Frobenius)
	double d2 = 0.0;
#pragma unroll
	for (int e = 0; e < 9; e++) d2 += (A.m[e] - B.m[e]) * (A.m[e] - B.m[e]);
	return d2 <= prec * prec * fmin(m3_sqnorm(A), m3_sqnorm(B));
}
__device__ __forceinline__ bool v3_approx(const double a[3], const double b[3], double prec) {
	const double d2 = (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
	const double a2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], b2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
	return d2 <= prec * prec * fmin(a2, b2);
}
__device__ __forceinline__ bool m3_valid_rotation(const M3& R) {  // isValidRotation, OTG_6dof_cartesian.cpp:19-27
	M3 G = m3_tmul(R, R);
	G.m[0] -= 1.0;
	G.m[4] -= 1.0;
	G.m[8] -= 1.0;
	if (sqrt(m3_sqnorm(G)) > 1e-3) return false;
	const double det = R.m[0] * (R.m[4] * R.m[8] - R.m[5] * R.m[7]) - R.m[1] * (R.m[3] * R.m[8] - R.m[5] * R.m[6]) + R.m[2] * (R.m[3] * R.m[7] - R.m[4] * R.m[6]);
	return !(dabs(det - 1.0) > 1e-3);
}
// Eigen::AngleAxisd(angle, axis).toRotationMatrix() for v = angle * axis, identity below 1e-3 (getNextOrientation, :228-239)
__device__ __forceinline__ M3 m3_from_rotvec(const double v[3]) {
	M3 R;
	const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
	if (n < 1e-3) {
#pragma unroll
		for (int e = 0; e < 9; e++) R.m[e] = (e % 4 == 0) ? 1.0 : 0.0;
		return R;
	}
	const double ax[3] = {v[0] / n, v[1] / n, v[2] / n};
	double sn, c;
	sincos(n, &sn, &c);
	const double sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
	const double ca[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
	double tmp = ca[0] * ax[1];
	R.m[1] = tmp - sa[2];
	R.m[3] = tmp + sa[2];
	tmp = ca[0] * ax[2];
	R.m[2] = tmp + sa[1];
	R.m[6] = tmp - sa[1];
	tmp = ca[1] * ax[2];
	R.m[5] = tmp - sa[0];
	R.m[7] = tmp + sa[0];
	R.m[0] = ca[0] * ax[0] + c;
	R.m[4] = ca[1] * ax[1] + c;
	R.m[8] = ca[2] * ax[2] + c;
	return R;
}
// angle * axis of Eigen::AngleAxisd(R): rotation matrix -> quaternion (trace / largest-diagonal branches) -> angle-axis
__device__ __forceinline__ void rotvec_from_m3(const M3& R, double out[3]) {
	double qw, qv[3];
	double t = R.m[0] + R.m[4] + R.m[8];
	if (t > 0.0) {
		t = sqrt(t + 1.0);
		qw = 0.5 * t;
		t = 0.5 / t;
		qv[0] = (R.m[7] - R.m[5]) * t;
		qv[1] = (R.m[2] - R.m[6]) * t;
		qv[2] = (R.m[3] - R.m[1]) * t;
	} else {
		int i = 0;
		if (R.m[4] > R.m[0]) i = 1;
		if (R.m[8] > R.m[4 * i]) i = 2;
		const int j = (i + 1) % 3, k = (j + 1) % 3;
		t = sqrt(R.m[4 * i] - R.m[4 * j] - R.m[4 * k] + 1.0);
		double q[3];
		q[i] = 0.5 * t;
		t = 0.5 / t;
		qw = (R.m[3 * k + j] - R.m[3 * j + k]) * t;
		q[j] = (R.m[3 * j + i] + R.m[3 * i + j]) * t;
		q[k] = (R.m[3 * k + i] + R.m[3 * i + k]) * t;
		qv[0] = q[0];
		qv[1] = q[1];
		qv[2] = q[2];
	}
	double n = sqrt(qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2]);
	if (n != 0.0) {
		const double angle = 2.0 * atan2(n, dabs(qw));
		if (qw < 0.0) n = -n;
		out[0] = angle * (qv[0] / n);
		out[1] = angle * (qv[1] / n);
		out[2] = angle * (qv[2] / n);
	} else {
		out[0] = out[1] = out[2] = 0.0;  // angle 0 (axis (1,0,0))
	}
}

struct CartFrames {
	M3 ref, gob;    // _reference_frame, _goal_orientation_in_base_frame
	double gab[3];  // _goal_angular_velocity_in_base_frame
};

// setGoalOrientationAndAngularVelocity, OTG_6dof_cartesian.cpp:155-191 (group-uniform; each lane keeps its own component)
template <int GS>
__device__ __forceinline__ void cart_set_goal_orientation(const Grp<GS>& g, bool act, Lane& L, CartFrames& F, int& flags, const M3& Rg, const double wg[3]) {
	if (!m3_valid_rotation(Rg)) {  // the reference throws std::invalid_argument (:158-162): flagged, goal ignored
		flags |= OTG_FLAG_INVALID_GOAL;
		return;
	}
	if ((flags & OTG_FLAG_GOAL_ORI_VALID) && m3_approx(F.gob, Rg, 1e-3) && v3_approx(F.gab, wg, 1e-3)) return;
	flags &= ~OTG_FLAG_GOAL_REACHED;
	const int dof = g.dof;
	const double op[3] = {g.bcast(L.out_p, 3), g.bcast(L.out_p, 4), g.bcast(L.out_p, 5)};
	const double ov[3] = {g.bcast(L.out_v, 3), g.bcast(L.out_v, 4), g.bcast(L.out_v, 5)};
	const double oa[3] = {g.bcast(L.out_a, 3), g.bcast(L.out_a, 4), g.bcast(L.out_a, 5)};
	const M3 new_ref = m3_mul(F.ref, m3_from_rotvec(op));  // getNextOrientation()
	const M3 Rntp = m3_tmul(new_ref, F.ref);               // R_new_to_previous_reference
	F.ref = new_ref;
	F.gob = Rg;
	F.gab[0] = wg[0];
	F.gab[1] = wg[1];
	F.gab[2] = wg[2];
	flags |= OTG_FLAG_GOAL_ORI_VALID;
	double rv[3], ra[3];
	m3_vec(Rntp, ov, rv);
	m3_vec(Rntp, oa, ra);
	if (dof >= 3 && dof < 6) {
		L.out_p = 0.0;
		L.out_v = rv[dof - 3];
		L.out_a = ra[dof - 3];
	}
	if (act) {  // _output.pass_to_input(_input): all six dofs
		L.in_p = L.out_p;
		L.in_v = L.out_v;
		L.in_a = L.out_a;
	}
	double aa[3], tw[3];
	rotvec_from_m3(m3_tmul(F.ref, F.gob), aa);
	m3_tvec(F.ref, wg, tw);
	if (dof >= 3 && dof < 6) {
		L.tg_p = aa[dof - 3];
		L.tg_v = tw[dof - 3];
	}
}
// setGoalPositionAndLinearVelocity, :138-148
template <int GS>
__device__ __forceinline__ void cart_set_goal_position(const Grp<GS>& g, Lane& L, int& flags, const double gp[3], const double gv[3]) {
	const double tp[3] = {g.bcast(L.tg_p, 0), g.bcast(L.tg_p, 1), g.bcast(L.tg_p, 2)};
	const double tv[3] = {g.bcast(L.tg_v, 0), g.bcast(L.tg_v, 1), g.bcast(L.tg_v, 2)};
	if (v3_approx(gp, tp, 1e-3) && v3_approx(gv, tv, 1e-3)) return;
	flags &= ~OTG_FLAG_GOAL_REACHED;
	if (g.dof < 3) {
		L.tg_p = gp[g.dof];
		L.tg_v = gv[g.dof];
	}
}

}  // namespace

// mode 0: one control cycle (MotionForceTask.cpp:394-406).  mode 1 | (mask << 4): reInitialize (mask 3), reInitializeLinear
// (mask 1), reInitializeAngular (mask 2) at the current pose of the control frame (OTG_6dof_cartesian.cpp:46-84).
template <bool JERK = false>
__device__ __forceinline__ void otg_cartesian_body(const OtgDev& O, const int B, const int ld, const int mode, const int block) {
	constexpr int GS = 8;
	const Grp<GS> g;
	const int gtid = block * blockDim.x + threadIdx.x;
	const int inst = gtid / GS;
	const int dof = g.dof;
	const bool inst_ok = inst < B;
	const bool act = inst_ok && dof < 6;
	const size_t stride = (size_t)O.lanes;
	const size_t li = (size_t)gtid;
	double* S = O.state;
	const size_t ib = inst_ok ? (size_t)inst : 0;

	Lane L;
	lane_load<JERK>(L, S, stride, li, act);
	int flags = inst_ok ? O.flags[inst] : 0;
	CartFrames F;
#pragma unroll
	for (int e = 0; e < 9; e++) {
		F.ref.m[e] = O.frame[(size_t)e * ld + ib];
		F.gob.m[e] = O.frame[(size_t)(9 + e) * ld + ib];
	}
#pragma unroll
	for (int e = 0; e < 3; e++) F.gab[e] = O.frame[(size_t)(18 + e) * ld + ib];
	const double zero3[3] = {0.0, 0.0, 0.0};
	bool new_traj = false;
	int result = 1;
	bool run = false;

	if ((mode & 15) == 1) {
		const int mask = mode >> 4;
		double pw[3];
		M3 Rc;
		fk_control_frame(*O.model, *O.task, O.q, ld, (int)ib, pw, Rc.m);
		if (!(flags & OTG_FLAG_STATE_VALID)) F.ref = Rc;  // constructor: _reference_frame = initial_orientation (:41), _output zero (:39)
		if (mask & 1) cart_set_goal_position<GS>(g, L, flags, pw, zero3);
		if (mask & 2) cart_set_goal_orientation<GS>(g, act, L, F, flags, Rc, zero3);
		const bool mine = (dof < 3) ? (mask & 1) : (mask & 2);
		if (act && mine) {
			L.in_p = L.tg_p;  // :50-56
			L.in_v = 0.0;
			L.in_a = 0.0;
			L.out_p = L.tg_p;
			L.out_v = 0.0;
			L.out_a = 0.0;
		}
		flags = (flags & ~(OTG_FLAG_ERROR | OTG_FLAG_FINISHED_MOVING | OTG_FLAG_INVALID_GOAL)) | OTG_FLAG_STATE_VALID;
	} else {
		double gp[3], gv[3], gw[3];
		M3 Rg;
#pragma unroll
		for (int e = 0; e < 3; e++) {
			gp[e] = O.goal[(size_t)e * ld + ib];
			gv[e] = O.goal[(size_t)(12 + e) * ld + ib];
			gw[e] = O.goal[(size_t)(15 + e) * ld + ib];
		}
#pragma unroll
		for (int e = 0; e < 9; e++) Rg.m[e] = O.goal[(size_t)(3 + e) * ld + ib];
		const double vMax = act ? O.limits[dof] : 1.0, aMax = act ? O.limits[SAIP_MAXN + dof] : 1.0, jMax = (JERK && act) ? O.limits[2 * SAIP_MAXN + dof] : 1.0;
		const bool new_limits = inst_ok && (O.epoch != O.seen_epoch[inst]);
		if (new_limits && !JERK) L.in_a = 0.0;  // disableJerkLimits, OTG_6dof_cartesian.h:86-89
		flags &= ~OTG_FLAG_INVALID_GOAL;
		cart_set_goal_position<GS>(g, L, flags, gp, gv);                     // MotionForceTask.cpp:395-396
		cart_set_goal_orientation<GS>(g, act, L, F, flags, Rg, gw);          // :397-398
		run = inst_ok && !(flags & OTG_FLAG_GOAL_REACHED);                   // OTG_6dof_cartesian.cpp:194-196
		if (g.any(run)) {
			double time = O.time[ib], duration = O.duration[ib];
			if (JERK) result = ruckig_update3<GS>(g, act, run, 6, vMax, aMax, jMax, new_limits, O.dt, L, flags, time, duration, &new_traj);
			else result = ruckig_update<GS>(g, act, run, 6, vMax, aMax, new_limits, O.dt, L, flags, time, duration, &new_traj);
			if (run) {  // :201-225
				if (result == 1) {
					const double vn2 = g.sum(act ? L.out_v * L.out_v : 0.0);
					if (sqrt(vn2) < 1e-3) {
						flags |= OTG_FLAG_GOAL_REACHED;
					} else {  // :206-211: same goal pose, zero goal velocities
						flags |= OTG_FLAG_FINISHED_MOVING;
						const double tp[3] = {g.bcast(L.tg_p, 0), g.bcast(L.tg_p, 1), g.bcast(L.tg_p, 2)};
						cart_set_goal_position<GS>(g, L, flags, tp, zero3);
						const M3 gob = F.gob;
						cart_set_goal_orientation<GS>(g, act, L, F, flags, gob, zero3);
					}
				} else if (result == 0) {
					L.in_p = L.out_p;
					L.in_v = L.out_v;
					L.in_a = L.out_a;
				} else {
					L.in_v = 0.0;
					L.in_a = 0.0;
					flags |= OTG_FLAG_ERROR;
				}
				if (act && dof == 0) {
					O.time[inst] = time;
					O.duration[inst] = duration;
				}
			}
		}
		if (act && dof == 0 && new_limits && (run ? result == 0 || result == 1 : true)) O.seen_epoch[inst] = O.epoch;
	}

	// ------------------------------------------------------------ stores: lane state, frames, desired state (goal layout)
	if (act) lane_store<JERK>(L, S, stride, li, new_traj);
	if (inst_ok) {
		const double op[3] = {g.bcast(L.out_p, 3), g.bcast(L.out_p, 4), g.bcast(L.out_p, 5)};
		const double ov[3] = {g.bcast(L.out_v, 3), g.bcast(L.out_v, 4), g.bcast(L.out_v, 5)};
		const double oa[3] = {g.bcast(L.out_a, 3), g.bcast(L.out_a, 4), g.bcast(L.out_a, 5)};
		const M3 Rd = m3_mul(F.ref, m3_from_rotvec(op));  // getNextOrientation
		double wd[3], ad[3];
		m3_vec(F.ref, ov, wd);  // getNextAngularVelocity / Acceleration, OTG_6dof_cartesian.h:258-264
		m3_vec(F.ref, oa, ad);
		double* D = O.desired;
		if (dof < 3) {
			D[(size_t)dof * ld + inst] = L.out_p;
			D[(size_t)(12 + dof) * ld + inst] = L.out_v;
			D[(size_t)(18 + dof) * ld + inst] = L.out_a;
		} else if (dof < 6) {
			D[(size_t)(15 + dof - 3) * ld + inst] = wd[dof - 3];
			D[(size_t)(21 + dof - 3) * ld + inst] = ad[dof - 3];
		}
		D[(size_t)(3 + dof) * ld + inst] = Rd.m[dof];  // eight lanes, entries 0..7
		if (dof == 0) D[(size_t)11 * ld + inst] = Rd.m[8];
		if (dof < 6)  // goal force / moment and sensed force / moment pass through
			for (int e = 24 + dof; e < O.goal_comps; e += 6) D[(size_t)e * ld + inst] = O.goal[(size_t)e * ld + inst];
		if (dof == 0) {
#pragma unroll
			for (int e = 0; e < 9; e++) {
				O.frame[(size_t)e * ld + inst] = F.ref.m[e];
				O.frame[(size_t)(9 + e) * ld + inst] = F.gob.m[e];
			}
#pragma unroll
			for (int e = 0; e < 3; e++) O.frame[(size_t)(18 + e) * ld + inst] = F.gab[e];
			O.flags[inst] = flags;
			O.result[inst] = run ? result : 1;
		}
	}
}

template <int GS>
__global__ void __launch_bounds__(256) saip_otg_joints(const OtgDev O, int B, int ld, int mode) {
	otg_joints_body<GS>(O, B, ld, mode, blockIdx.x);
}
__global__ void __launch_bounds__(256) saip_otg_cartesian(const OtgDev O, int B, int ld, int mode) {
	otg_cartesian_body(O, B, ld, mode, blockIdx.x);
}
// jerk-limited mode: kernels of their own, so that the third-order solver's registers and private memory stay out of the default mode's
template <int GS>
__global__ void __launch_bounds__(256) saip_otg3_joints(const OtgDev O, int B, int ld, int mode) {
	otg_joints_body<GS, true>(O, B, ld, mode, blockIdx.x);
}
__global__ void __launch_bounds__(256) saip_otg3_cartesian(const OtgDev O, int B, int ld, int mode) {
	otg_cartesian_body<true>(O, B, ld, mode, blockIdx.x);
}
// One launch for the two OTGs of a { MotionForceTask, JointTask } stack in their cycle mode: they are independent and each is a few
// latency-bound wavefronts, so the second launch only added its latency.  Workgroups [0, grid_c) run the Cartesian OTG, the rest the
// joint OTG (8-lane groups).
__global__ void __launch_bounds__(256) saip_otg_pair(const OtgDev Oc, const OtgDev Oj, int B, int ld, int grid_c) {
	if ((int)blockIdx.x < grid_c) otg_cartesian_body(Oc, B, ld, 0, blockIdx.x);
	else otg_joints_body<8>(Oj, B, ld, 0, blockIdx.x - grid_c);
}

hipError_t launch_otg_pair(const OtgDev& Oc, const OtgDev& Oj, int B, int ld, hipStream_t stream) {
	const int block = 256;
	const int grid_c = (int)(((long long)Oc.lanes + block - 1) / block), grid_j = (int)(((long long)Oj.lanes + block - 1) / block);
	hipLaunchKernelGGL(saip_otg_pair, dim3(grid_c + grid_j), dim3(block), 0, stream, Oc, Oj, B, ld, grid_c);
	return hipGetLastError();
}

// Closed-loop rollouts (saip_batch_rollout_async): the integration step of period k and the trajectory generation of period k + 1 do not
// depend on each other (the OTGs evolve their own state, they never read the robot's), and each is a handful of latency-bound
// wavefronts -- one launch instead of two.  Workgroups [0, grid_i) integrate (four wavefronts = four groups of eight instances each),
// then the Cartesian OTG, then the joint OTG.
// (two wavefronts per SIMD: at 260 registers the launch needed two rounds of wavefronts; capped at 256 it fits one -- 21.7 -> 20.3 us per period)
__global__ void __launch_bounds__(256, 2) saip_integrate_otg_pair(const SimParams S, const OtgDev Oc, const OtgDev Oj, int B, int ld, int grid_i, int grid_c) {
	__shared__ OctDynInst smem[32];
	const int blk = (int)blockIdx.x;
	if (blk < grid_i) {
		const int wave = (int)(threadIdx.x >> 6);
		integrate_oct_body(S, smem + 8 * wave, blk * 4 + wave, (int)(threadIdx.x & 63));
	} else if (blk < grid_i + grid_c) {
		otg_cartesian_body(Oc, B, ld, 0, blk - grid_i);
	} else {
		otg_joints_body<8>(Oj, B, ld, 0, blk - grid_i - grid_c);
	}
}

hipError_t launch_integrate_otg_pair(const SimParams& S, const OtgDev& Oc, const OtgDev& Oj, int B, int ld, hipStream_t stream) {
	const int block = 256;
	const int grid_i = (S.B + 31) / 32;
	const int grid_c = (int)(((long long)Oc.lanes + block - 1) / block), grid_j = (int)(((long long)Oj.lanes + block - 1) / block);
	hipLaunchKernelGGL(saip_integrate_otg_pair, dim3(grid_i + grid_c + grid_j), dim3(block), 0, stream, S, Oc, Oj, B, ld, grid_i, grid_c);
	return hipGetLastError();
}

hipError_t launch_otg_cartesian(const OtgDev& O, int B, int ld, int mode, hipStream_t stream) {
	const int block = 256;
	const int grid = (int)(((long long)O.lanes + block - 1) / block);
	if (O.jerk) hipLaunchKernelGGL(saip_otg3_cartesian, dim3(grid), dim3(block), 0, stream, O, B, ld, mode);
	else hipLaunchKernelGGL(saip_otg_cartesian, dim3(grid), dim3(block), 0, stream, O, B, ld, mode);
	return hipGetLastError();
}

int otg_state_fields() { return F_COUNT; }

hipError_t launch_otg_joints(const OtgDev& O, int B, int ld, int mode, hipStream_t stream) {
	const int gs = O.gs;
	const long long lanes = (long long)O.lanes;
	const int block = 256;
	const int grid = (int)((lanes + block - 1) / block);
	if (O.jerk) {
		if (gs == 8) hipLaunchKernelGGL((saip_otg3_joints<8>), dim3(grid), dim3(block), 0, stream, O, B, ld, mode);
		else hipLaunchKernelGGL((saip_otg3_joints<32>), dim3(grid), dim3(block), 0, stream, O, B, ld, mode);
	} else if (gs == 8) hipLaunchKernelGGL((saip_otg_joints<8>), dim3(grid), dim3(block), 0, stream, O, B, ld, mode);
	else hipLaunchKernelGGL((saip_otg_joints<32>), dim3(grid), dim3(block), 0, stream, O, B, ld, mode);
	return hipGetLastError();
}

}  // namespace saip

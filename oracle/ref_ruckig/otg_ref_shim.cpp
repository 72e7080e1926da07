// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product (sai-primitives_amd/).
//
// Reference build of the online-trajectory-generation row.  The arithmetic here is the REAL vendored
// Ruckig 0.10.1 of the reference, compiled from /root/reference/ruckig/{include,src} where those files lie (recipe:
// oracle/Makefile, target _ref/libsaip_otg_ref.so); nothing from the reference is copied into this repository.
//
// What this file adds is a driver with a C ABI, because the reference's own wrappers
//   /root/reference/src/helper_modules/OTG_joints.cpp        (needs Eigen -> cannot be compiled here)
//   /root/reference/src/helper_modules/OTG_6dof_cartesian.cpp (needs Eigen -> cannot be compiled here)
// are thin state machines around ruckig::Ruckig::update().  Their logic is restated below line by line on std::vector
// (ruckig::StandardVector), each step citing the wrapper line it follows.
#include <ruckig/ruckig.hpp>

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

using namespace ruckig;

namespace {

// Eigen's DenseBase::isApprox(other, prec): ||a-b||^2 <= prec^2 * min(||a||^2, ||b||^2)   [Eigen 3.4 fuzzy comparison; RECALLED]
bool is_approx(const double* a, const double* b, int n, double prec) {
	double d2 = 0, a2 = 0, b2 = 0;
	for (int i = 0; i < n; i++) {
		d2 += (a[i] - b[i]) * (a[i] - b[i]);
		a2 += a[i] * a[i];
		b2 += b[i] * b[i];
	}
	return d2 <= prec * prec * std::min(a2, b2);
}

struct JointsOtg {  // OTG_joints.h:172-189
	int dim;
	bool goal_reached = false;  // OTG_joints.h:175
	int result_value = Result::Finished;
	int finished_with_velocity = 0;  // counts the branch of OTG_joints.cpp:131-133 (the reference throws there, see below)
	Ruckig<DynamicDOFs> otg;
	InputParameter<DynamicDOFs> input;
	OutputParameter<DynamicDOFs> output;

	JointsOtg(int n, double dt, const double* q0) : dim(n), otg((size_t)n, dt), input((size_t)n), output((size_t)n) {
		input.synchronization = Synchronization::Phase;  // OTG_joints.cpp:23
		reinitialize(q0);                                // :25
	}
	void reinitialize(const double* q0) {  // OTG_joints.cpp:28-42
		std::vector<double> zero(dim, 0.0);
		set_goal(q0, zero.data());
		for (int i = 0; i < dim; i++) {
			output.new_position[i] = q0[i];
			output.new_velocity[i] = 0.0;
			output.new_acceleration[i] = 0.0;
		}
		output.pass_to_input(input);
	}
	void set_goal(const double* gp, const double* gv) {  // OTG_joints.cpp:98-114
		if (is_approx(gp, input.target_position.data(), dim, 1e-12) && is_approx(gv, input.target_velocity.data(), dim, 1e-12)) return;
		goal_reached = false;
		for (int i = 0; i < dim; i++) {
			input.target_position[i] = gp[i];
			input.target_velocity[i] = gv[i];
		}
	}
	void update() {  // OTG_joints.cpp:116-150
		if (goal_reached) return;
		OutputParameter<DynamicDOFs> previous_output = output;
		result_value = otg.update(input, output);
		if (result_value == Result::Finished) {
			double n2 = 0;
			for (int i = 0; i < dim; i++) n2 += output.new_velocity[i] * output.new_velocity[i];
			if (std::sqrt(n2) < 1e-3) {
				goal_reached = true;
			} else {
				// OTG_joints.cpp:132 calls setGoalPosition(_goal_position_eigen) with a member that is never assigned (size 0), which
				// throws std::invalid_argument (:100-105).  The evident intent (comment at :126-127, and the 6-dof wrapper,
				// OTG_6dof_cartesian.cpp:209-210) is "same target position, zero target velocity": do that and count the event.
				finished_with_velocity++;
				std::vector<double> zero(dim, 0.0), tp(input.target_position.begin(), input.target_position.end());
				// force the change through (isApprox of the position is true, of the velocity false unless it already is zero)
				set_goal(tp.data(), zero.data());
			}
			return;
		}
		if (result_value == Result::Working) {
			output.pass_to_input(input);
			return;
		}
		output = previous_output;  // :143-149 (the warning print is dropped)
		for (int i = 0; i < dim; i++) {
			input.current_velocity[i] = 0.0;
			input.current_acceleration[i] = 0.0;
		}
	}
};

}  // namespace

extern "C" {

void* otgref_joints_create(int dim, double dt, const double* q0, const double* vmax, const double* amax) {
	auto* o = new JointsOtg(dim, dt, q0);
	for (int i = 0; i < dim; i++) {  // enableInternalOtgAccelerationLimited: JointTask.cpp:377-379, OTG_joints.cpp:44-74, 90-93
		o->input.max_velocity[i] = vmax[i];
		o->input.max_acceleration[i] = amax[i];
		o->input.max_jerk[i] = std::numeric_limits<double>::infinity();
		o->input.current_acceleration[i] = 0.0;
	}
	return o;
}
// JointTask::enableInternalOtgAccelerationLimited on an already enabled, acceleration-limited OTG (JointTask.cpp:374-380):
// no re-initialisation, new limits, and disableJerkLimits() zeroes the input's current acceleration (OTG_joints.cpp:90-93).
void otgref_joints_set_limits(void* h, const double* vmax, const double* amax) {
	auto* o = (JointsOtg*)h;
	for (int i = 0; i < o->dim; i++) {
		o->input.max_velocity[i] = vmax[i];
		o->input.max_acceleration[i] = amax[i];
		o->input.max_jerk[i] = std::numeric_limits<double>::infinity();
		o->input.current_acceleration[i] = 0.0;
	}
}
// JointTask::enableInternalOtgJerkLimited (JointTask.cpp:383-410): setMaxVelocity / setMaxAcceleration / setMaxJerk (OTG_joints.cpp:44-86).
// The re-initialisation at the current position that precedes it when the OTG was off or acceleration-limited (:400-402) is the
// caller's otgref_joints_reinitialize.
void otgref_joints_set_jerk_limits(void* h, const double* vmax, const double* amax, const double* jmax) {
	auto* o = (JointsOtg*)h;
	for (int i = 0; i < o->dim; i++) {
		o->input.max_velocity[i] = vmax[i];
		o->input.max_acceleration[i] = amax[i];
		o->input.max_jerk[i] = jmax[i];
	}
}
void otgref_joints_destroy(void* h) { delete (JointsOtg*)h; }
void otgref_joints_reinitialize(void* h, const double* q0) { ((JointsOtg*)h)->reinitialize(q0); }

// One control cycle of JointTask.cpp:313-319: setGoalPositionAndVelocity, update, read the next state.
// Returns the ruckig result code of the cycle (or 1 = Finished when the goal had already been reached).
int otgref_joints_step(void* h, const double* goal_pos, const double* goal_vel, double* pos, double* vel, double* acc, int* goal_reached) {
	auto* o = (JointsOtg*)h;
	o->set_goal(goal_pos, goal_vel);
	o->update();
	for (int i = 0; i < o->dim; i++) {
		pos[i] = o->output.new_position[i];
		vel[i] = o->output.new_velocity[i];
		acc[i] = o->output.new_acceleration[i];
	}
	*goal_reached = o->goal_reached ? 1 : 0;
	return o->result_value;
}
int otgref_joints_finished_with_velocity(void* h) { return ((JointsOtg*)h)->finished_with_velocity; }

// Raw state-to-state trajectory (acceleration-limited, position interface): duration, per-dof segment times / accelerations of the
// chosen profile, brake pre-trajectory, and samples at the given times.  sync: 0 Time, 1 TimeIfNecessary, 2 Phase, 3 None.
int otgref_calculate(int dim, const double* p0, const double* v0, const double* pf, const double* vf, const double* vmax, const double* amax,
					 int sync, double* duration, double* t_out /*[dim][7]*/, double* a_out /*[dim][7]*/, double* brake_out /*[dim][2]: t, a*/,
					 int nsamples, const double* times, double* samples /*[nsamples][3][dim]*/) {
	Ruckig<DynamicDOFs> otg((size_t)dim);
	InputParameter<DynamicDOFs> in((size_t)dim);
	Trajectory<DynamicDOFs> traj((size_t)dim);
	in.synchronization = (Synchronization)sync;
	for (int i = 0; i < dim; i++) {
		in.current_position[i] = p0[i];
		in.current_velocity[i] = v0[i];
		in.current_acceleration[i] = 0.0;
		in.target_position[i] = pf[i];
		in.target_velocity[i] = vf[i];
		in.target_acceleration[i] = 0.0;
		in.max_velocity[i] = vmax[i];
		in.max_acceleration[i] = amax[i];
		in.max_jerk[i] = std::numeric_limits<double>::infinity();
	}
	Result r = otg.calculate(in, traj);
	if (r != Result::Working) return (int)r;
	*duration = traj.get_duration();
	auto profiles = traj.get_profiles();
	for (int i = 0; i < dim; i++) {
		const Profile& p = profiles[0][i];
		for (int k = 0; k < 7; k++) {
			t_out[7 * i + k] = p.t[k];
			a_out[7 * i + k] = p.a[k];
		}
		brake_out[2 * i] = p.brake.duration;
		brake_out[2 * i + 1] = p.brake.a[0];
	}
	std::vector<double> np(dim), nv(dim), na(dim);
	for (int s = 0; s < nsamples; s++) {
		traj.at_time(times[s], np, nv, na);
		for (int i = 0; i < dim; i++) {
			samples[(3 * s + 0) * dim + i] = np[i];
			samples[(3 * s + 1) * dim + i] = nv[i];
			samples[(3 * s + 2) * dim + i] = na[i];
		}
	}
	return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------
// OTG_6dof_cartesian (src/helper_modules/OTG_6dof_cartesian.cpp) restated on ruckig::Ruckig<6> with plain 3x3 helpers.  The
// Eigen calls of the wrapper (isApprox, AngleAxisd(Matrix3d), AngleAxisd::toRotationMatrix, determinant, norm) are restated from
// Eigen 3.4's documented algorithms [RECALLED: Eigen is not in this image]; the trajectory arithmetic is the real Ruckig.
namespace {
struct M3 {
	double m[9];
};
M3 mul(const M3& A, const M3& B) {
	M3 C;
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++) C.m[3 * r + c] = A.m[3 * r] * B.m[c] + A.m[3 * r + 1] * B.m[3 + c] + A.m[3 * r + 2] * B.m[6 + c];
	return C;
}
M3 tmul(const M3& A, const M3& B) {
	M3 C;
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++) C.m[3 * r + c] = A.m[r] * B.m[c] + A.m[3 + r] * B.m[3 + c] + A.m[6 + r] * B.m[6 + c];
	return C;
}
void mvec(const M3& A, const double* v, double* o) {
	for (int r = 0; r < 3; r++) o[r] = A.m[3 * r] * v[0] + A.m[3 * r + 1] * v[1] + A.m[3 * r + 2] * v[2];
}
void tvec(const M3& A, const double* v, double* o) {
	for (int r = 0; r < 3; r++) o[r] = A.m[r] * v[0] + A.m[3 + r] * v[1] + A.m[6 + r] * v[2];
}
bool valid_rotation(const M3& R) {  // OTG_6dof_cartesian.cpp:19-27
	M3 G = tmul(R, R);
	G.m[0] -= 1;
	G.m[4] -= 1;
	G.m[8] -= 1;
	double n2 = 0;
	for (double x : G.m) n2 += x * x;
	if (std::sqrt(n2) > 1e-3) return false;
	const double det = R.m[0] * (R.m[4] * R.m[8] - R.m[5] * R.m[7]) - R.m[1] * (R.m[3] * R.m[8] - R.m[5] * R.m[6]) + R.m[2] * (R.m[3] * R.m[7] - R.m[4] * R.m[6]);
	return !(std::abs(det - 1) > 1e-3);
}
M3 from_rotvec(const double* v) {  // getNextOrientation, :228-239
	M3 R;
	const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
	if (n < 1e-3) {
		for (int e = 0; e < 9; e++) R.m[e] = (e % 4 == 0) ? 1.0 : 0.0;
		return R;
	}
	const double ax[3] = {v[0] / n, v[1] / n, v[2] / n};
	const double sn = std::sin(n), c = std::cos(n);
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
void rotvec(const M3& R, double* out) {  // AngleAxisd(Matrix3d): matrix -> quaternion -> angle * axis
	double qw, qv[3];
	double t = R.m[0] + R.m[4] + R.m[8];
	if (t > 0.0) {
		t = std::sqrt(t + 1.0);
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
		t = std::sqrt(R.m[4 * i] - R.m[4 * j] - R.m[4 * k] + 1.0);
		qv[i] = 0.5 * t;
		t = 0.5 / t;
		qw = (R.m[3 * k + j] - R.m[3 * j + k]) * t;
		qv[j] = (R.m[3 * j + i] + R.m[3 * i + j]) * t;
		qv[k] = (R.m[3 * k + i] + R.m[3 * i + k]) * t;
	}
	double n = std::sqrt(qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2]);
	if (n != 0.0) {
		const double angle = 2.0 * std::atan2(n, std::abs(qw));
		if (qw < 0.0) n = -n;
		for (int e = 0; e < 3; e++) out[e] = angle * (qv[e] / n);
	} else {
		out[0] = out[1] = out[2] = 0.0;
	}
}

struct CartesianOtg {  // OTG_6dof_cartesian.h:271-291
	bool goal_reached = false;
	int result_value = Result::Finished;
	int finished_with_velocity = 0, invalid_goal = 0;
	bool goal_ori_assigned = false;  // the reference compares against an uninitialised member on the very first call (:164)
	M3 ref, gob;
	double gab[3] = {0, 0, 0};
	Ruckig<6> otg;
	InputParameter<6> input;
	OutputParameter<6> output;

	CartesianOtg(const double* p0, const double* R0, double dt) : otg(dt) {  // :30-44
		input.synchronization = Synchronization::Phase;
		for (int i = 0; i < 6; i++) {
			output.new_position[i] = 0.0;
			input.current_position[i] = input.target_position[i] = 0.0;
		}
		std::memcpy(ref.m, R0, sizeof(ref.m));
		reinitialize(p0, R0, 3);
	}
	void set_goal_position(const double* gp, const double* gv) {  // :138-148
		if (is_approx(gp, input.target_position.data(), 3, 1e-3) && is_approx(gv, input.target_velocity.data(), 3, 1e-3)) return;
		goal_reached = false;
		for (int i = 0; i < 3; i++) {
			input.target_position[i] = gp[i];
			input.target_velocity[i] = gv[i];
		}
	}
	M3 next_orientation() const {  // :228-239
		return mul(ref, from_rotvec(&output.new_position[3]));
	}
	void set_goal_orientation(const M3& Rg, const double* wg) {  // :150-191
		if (!valid_rotation(Rg)) {
			invalid_goal++;  // the reference throws std::invalid_argument
			return;
		}
		if (goal_ori_assigned && is_approx(gob.m, Rg.m, 9, 1e-3) && is_approx(gab, wg, 3, 1e-3)) return;
		goal_reached = false;
		const M3 new_ref = next_orientation();
		const M3 Rntp = tmul(new_ref, ref);
		ref = new_ref;
		gob = Rg;
		std::memcpy(gab, wg, sizeof(gab));
		goal_ori_assigned = true;
		double rv[3], ra[3];
		mvec(Rntp, &output.new_velocity[3], rv);
		mvec(Rntp, &output.new_acceleration[3], ra);
		for (int i = 0; i < 3; i++) {
			output.new_position[3 + i] = 0.0;
			output.new_velocity[3 + i] = rv[i];
			output.new_acceleration[3 + i] = ra[i];
		}
		output.pass_to_input(input);
		double aa[3], tw[3];
		rotvec(tmul(ref, gob), aa);
		tvec(ref, wg, tw);
		for (int i = 0; i < 3; i++) {
			input.target_position[3 + i] = aa[i];
			input.target_velocity[3 + i] = tw[i];
		}
	}
	void reinitialize(const double* p, const double* R, int mask) {  // reInitialize / Linear / Angular, :46-84
		const double zero[3] = {0, 0, 0};
		M3 Rm;
		std::memcpy(Rm.m, R, sizeof(Rm.m));
		if (mask & 1) set_goal_position(p, zero);
		if (mask & 2) set_goal_orientation(Rm, zero);
		for (int i = 0; i < 6; i++) {
			if (!((i < 3) ? (mask & 1) : (mask & 2))) continue;
			input.current_position[i] = input.target_position[i];
			input.current_velocity[i] = 0.0;
			input.current_acceleration[i] = 0.0;
			output.new_position[i] = input.target_position[i];
			output.new_velocity[i] = 0.0;
			output.new_acceleration[i] = 0.0;
		}
	}
	void update() {  // :193-226
		if (goal_reached) return;
		OutputParameter<6> previous_output = output;
		result_value = otg.update(input, output);
		if (result_value == Result::Finished) {
			double n2 = 0;
			for (int i = 0; i < 6; i++) n2 += output.new_velocity[i] * output.new_velocity[i];
			if (std::sqrt(n2) < 1e-3) {
				goal_reached = true;
			} else {
				finished_with_velocity++;
				const double zero[3] = {0, 0, 0};
				const double tp[3] = {input.target_position[0], input.target_position[1], input.target_position[2]};
				set_goal_position(tp, zero);
				const M3 g = gob;
				set_goal_orientation(g, zero);
			}
			return;
		}
		if (result_value == Result::Working) {
			output.pass_to_input(input);
			return;
		}
		output = previous_output;
		for (int i = 0; i < 6; i++) {
			input.current_velocity[i] = 0.0;
			input.current_acceleration[i] = 0.0;
		}
	}
};
}  // namespace

extern "C" {
// limits: {max linear velocity, max linear acceleration, max angular velocity, max angular acceleration} (MotionForceTask.cpp:510-523)
void otgref_cart_set_limits(void* h, const double* lim) {
	auto* o = (CartesianOtg*)h;
	for (int i = 0; i < 6; i++) {
		o->input.max_velocity[i] = lim[i < 3 ? 0 : 2];
		o->input.max_acceleration[i] = lim[i < 3 ? 1 : 3];
		o->input.max_jerk[i] = std::numeric_limits<double>::infinity();
		o->input.current_acceleration[i] = 0.0;  // disableJerkLimits, OTG_6dof_cartesian.h:86-89
	}
}
// MotionForceTask::enableInternalOtgJerkLimited (MotionForceTask.cpp:525-545): lim = {max linear velocity, acceleration, jerk, max angular
// velocity, acceleration, jerk} (OTG_6dof_cartesian.cpp:86-136); the preceding re-initialisation (:530-532) is the caller's
void otgref_cart_set_jerk_limits(void* h, const double* lim) {
	auto* o = (CartesianOtg*)h;
	for (int i = 0; i < 6; i++) {
		o->input.max_velocity[i] = lim[i < 3 ? 0 : 3];
		o->input.max_acceleration[i] = lim[i < 3 ? 1 : 4];
		o->input.max_jerk[i] = lim[i < 3 ? 2 : 5];
	}
}
void* otgref_cart_create(double dt, const double* p0, const double* R0, const double* lim) {
	auto* o = new CartesianOtg(p0, R0, dt);
	otgref_cart_set_limits(o, lim);
	return o;
}
void otgref_cart_destroy(void* h) { delete (CartesianOtg*)h; }
void otgref_cart_reinitialize(void* h, const double* p, const double* R, int mask) { ((CartesianOtg*)h)->reinitialize(p, R, mask); }
// one control cycle of MotionForceTask.cpp:394-406; out = [pos3, R9 (row-major), linvel3, angvel3, linacc3, angacc3] (the goal layout)
int otgref_cart_step(void* h, const double* gp, const double* gR, const double* gv, const double* gw, double* out, int* goal_reached) {
	auto* o = (CartesianOtg*)h;
	M3 Rg;
	std::memcpy(Rg.m, gR, sizeof(Rg.m));
	o->set_goal_position(gp, gv);
	o->set_goal_orientation(Rg, gw);
	o->update();
	const M3 Rd = o->next_orientation();
	double wd[3], ad[3];
	mvec(o->ref, &o->output.new_velocity[3], wd);
	mvec(o->ref, &o->output.new_acceleration[3], ad);
	for (int i = 0; i < 3; i++) {
		out[i] = o->output.new_position[i];
		out[12 + i] = o->output.new_velocity[i];
		out[15 + i] = wd[i];
		out[18 + i] = o->output.new_acceleration[i];
		out[21 + i] = ad[i];
	}
	for (int e = 0; e < 9; e++) out[3 + e] = Rd.m[e];
	*goal_reached = o->goal_reached ? 1 : 0;
	return o->result_value;
}
int otgref_cart_finished_with_velocity(void* h) { return ((CartesianOtg*)h)->finished_with_velocity; }
}  // extern "C"

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

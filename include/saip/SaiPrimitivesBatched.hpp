// SaiPrimitivesBatched.hpp -- header-only C++17 facade over the C-ABI (include/saip.h) that keeps the reference's
// class and method names for the control-cycle path, in a batched flavour: one object = B robot instances.
//
//   reference (one robot, Eigen)                                   this facade (B robots, SoA std::vector<double>)
//   SaiModel::SaiModel robot(urdf)                                  SaiPrimitivesBatched::SaiModel robot(links, B, device)
//   robot->setQ(q); robot->setDq(dq); robot->updateModel();         robot->setQ(q); robot->setDq(dq); robot->updateModel();   q: [dof][B]
//   MotionForceTask(robot, link, compliant_frame)                   MotionForceTask(robot, link, pos_in_link)
//   JointTask(robot) / JointTask(robot, S)                          JointTask(robot) / JointTask(robot, S, rows)
//   RobotController(robot, tasks)                                   RobotController(robot, tasks)
//   controller->updateControllerTaskModels();                       controller->updateControllerTaskModels();
//   task->setGoalPosition(x) ...                                    task->setGoalPosition(x) ...                               x: [3][B]
//   tau = controller->computeControlTorques();                      tau = controller->computeControlTorques();                  tau: [dof][B]
//   task->updateTaskModel(N_prec); task->computeTorques()           the same (TemplateTask.h:43-60; examples/04-task_and_redundancy):      N_prec: [dof*dof][B]
//   N_prec = task->getTaskAndPreviousNullspace()                    a task driven by hand without a RobotController owns a private one-task batch
//
// Reference interface: /root/reference/src/RobotController.h:47-90, src/tasks/TemplateTask.h:26-124,
// src/tasks/MotionForceTask.h:96-110,211-300,423,670-736, src/tasks/JointTask.h:56-75,140-175,237-257,323,363.
// Errors: std::invalid_argument where the reference throws it; std::runtime_error for device / unsupported / order errors.
// Every array is struct-of-arrays, component-major: value of component c for instance b at [c * B + b].
#pragma once
#include <cmath>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../saip.h"

namespace SaiPrimitivesBatched {

enum TaskType { UNDEFINED = 0, JOINT_LIMIT_AVOIDANCE_TASK = 1, JOINT_TASK = 2, MOTION_FORCE_TASK = 3 };  // TemplateTask.h:19-24
enum DynamicDecouplingType { FULL_DYNAMIC_DECOUPLING = 0, BOUNDED_INERTIA_ESTIMATES = 1, IMPEDANCE = 2 };  // SaiPrimitivesCommonDefinitions.h:14-20
struct PIDGains {  // SaiPrimitivesCommonDefinitions.h:26-32
	double kp, kv, ki;
	PIDGains(double kp_, double kv_, double ki_) : kp(kp_), kv(kv_), ki(ki_) {}
};

inline void check(saip_status st) {
	if (st == SAIP_OK) return;
	const std::string msg = saip_last_error();
	if (st == SAIP_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
	throw std::runtime_error("[saip status " + std::to_string((int)st) + "] " + msg);
}

class RobotController;
class TemplateTask;

// a [dof*dof][ld] nullspace matrix resident on the GPU: what getTaskAndPreviousNullspaceDevice() returns and updateTaskModel accepts
struct DeviceNullspace {
	saip_batch* batch = nullptr;
	const double* ptr = nullptr;
};

// B instances of one robot: constants + host-side q/dq staging (the kinematics/dynamics run inside the GPU cycle)
class SaiModel {
public:
	SaiModel(const std::vector<saip_link_desc>& links, int batch_size, int device = 0) : _batch(batch_size), _device(device) {
		check(saip_model_create_serial_chain(links.data(), (int)links.size(), &_model));
		_q.assign((size_t)dof() * _batch, 0.0);
		_dq.assign((size_t)dof() * _batch, 0.0);
	}
	~SaiModel() { saip_model_destroy(_model); }
	SaiModel(const SaiModel&) = delete;
	SaiModel& operator=(const SaiModel&) = delete;

	int dof() const { return saip_model_dof(_model); }
	int batchSize() const { return _batch; }
	int device() const { return _device; }
	const std::vector<double>& q() const { return _q; }
	const std::vector<double>& dq() const { return _dq; }
	void setQ(const std::vector<double>& q) {
		if (q.size() != _q.size()) throw std::invalid_argument("setQ: expected dof*batch values");
		_q = q;
		_version++;
	}
	void setDq(const std::vector<double>& dq) {
		if (dq.size() != _dq.size()) throw std::invalid_argument("setDq: expected dof*batch values");
		_dq = dq;
		_version++;
	}
	// pushes the state to every batch that mirrors this robot (the RobotController's and the private batches of tasks driven by hand)
	void updateModel() {
		for (auto& a : _attached) push(a);
	}
	const saip_model* handle() const { return _model; }

private:
	friend class RobotController;
	friend class TemplateTask;
	struct Attached {
		saip_batch* batch;
		long pushed;
	};
	void attach(saip_batch* b) { _attached.push_back({b, -1}); }
	void detach(saip_batch* b) {
		for (size_t i = 0; i < _attached.size(); i++)
			if (_attached[i].batch == b) {
				_attached.erase(_attached.begin() + i);
				return;
			}
	}
	void push(Attached& a) {
		if (a.pushed != _version) {
			check(saip_batch_set_state_host(a.batch, _q.data(), _dq.data()));
			a.pushed = _version;
		}
	}
	void pushTo(saip_batch* b) {
		for (auto& a : _attached)
			if (a.batch == b) push(a);
	}
	saip_model* _model = nullptr;
	int _batch, _device;
	std::vector<double> _q, _dq;
	long _version = 0;
	std::vector<Attached> _attached;
};

// TemplateTask.h:26-124
class TemplateTask {
public:
	TemplateTask(std::shared_ptr<SaiModel>& robot, const std::string& task_name, TaskType type, double loop_timestep)
		: _robot(robot), _task_name(task_name), _task_type(type), _loop_timestep(loop_timestep) {}
	virtual ~TemplateTask() { dropPrivateBatch(); }
	TemplateTask(const TemplateTask&) = delete;
	TemplateTask& operator=(const TemplateTask&) = delete;
	const std::shared_ptr<SaiModel>& getConstRobotModel() const { return _robot; }
	const double& getLoopTimestep() const { return _loop_timestep; }
	const TaskType& getTaskType() const { return _task_type; }
	const std::string& getTaskName() const { return _task_name; }

	void setDynamicDecouplingType(DynamicDecouplingType t) { cfg([=](saip_batch* b, int id) { return saip_batch_set_dynamic_decoupling_type(b, id, (int)t); }); }
	void setBoundedInertiaEstimateThreshold(double thr) {
		_bie_threshold = thr;
		cfg([=](saip_batch* b, int id) { return saip_batch_set_bie_threshold(b, id, thr); });
	}
	double getBoundedInertiaEstimateThreshold() const { return _bie_threshold; }
	// ---- the reference's per-task interface, TemplateTask.h:43-60, driven by hand in examples/04-task_and_redundancy/04-task_and_redundancy.cpp:141-206
	// N_prec: [dof*dof][B] (row-major per instance) or dof*dof values shared by every instance
	void updateTaskModel(const std::vector<double>& N_prec) {
		need();
		const size_t n = _robot->dof(), B = _robot->batchSize();
		if (N_prec.size() == n * n) {
			std::vector<double> all(n * n * B);
			for (size_t e = 0; e < n * n; e++)
				for (size_t b = 0; b < B; b++) all[e * B + b] = N_prec[e];
			pushAndUpdate(all.data());
		} else if (N_prec.size() == n * n * B) {
			pushAndUpdate(N_prec.data());
		} else {
			throw std::invalid_argument("N_prec matrix size not consistent with robot dof in updateTaskModel\n");  // JointTask.cpp:224-229
		}
	}
	// N_prec stays on the GPU: the handle another task's getTaskAndPreviousNullspaceDevice() returned
	void updateTaskModel(const DeviceNullspace& N_prec) {
		need();
		_robot->pushTo(_batch);
		if (N_prec.batch && N_prec.batch != _batch) check(saip_batch_wait_for(_batch, N_prec.batch));
		check(saip_batch_task_update_model_device(_batch, _id, N_prec.ptr));
		_manual = true;
	}
	// [dof][B] torques of THIS task (no RobotController post-processing); per-instance status in status()
	std::vector<double> computeTorques() { return torques(nullptr); }
	// ... with the feed-forward compensation of the torques of the previous tasks (JointTask.cpp:285-292)
	std::vector<double> computeTorques(const std::vector<double>& tau_prec) {
		if (tau_prec.size() != (size_t)_robot->dof() * _robot->batchSize()) throw std::invalid_argument("tau_prec: expected [dof][B]");
		return torques(tau_prec.data());
	}
	const std::vector<uint8_t>& status() const { return _status; }
	DeviceNullspace getTaskAndPreviousNullspaceDevice() {
		if (!_manual) throw std::runtime_error("getTaskAndPreviousNullspaceDevice: call updateTaskModel first");
		return DeviceNullspace{_batch, saip_batch_task_device_nullspace(_batch, _id, 2)};
	}
	// TemplateTask::reInitializeTask of this task alone
	void reInitializeTask() {
		need();
		_robot->pushTo(_batch);
		check(saip_batch_reinitialize_task(_batch, _id));
	}
	// resetIntegrators (both tasks), resetIntegratorsLinear / Angular (MotionForceTask.cpp:988-1002)
	void resetIntegrators() {
		need();
		check(saip_batch_reset_integrators(_batch, _id, 3));
	}
	void disableInternalOtg() {
		_otg_enabled = false;
		cfg([](saip_batch* b, int id) { return saip_batch_set_internal_otg(b, id, 0); });
	}
	bool getInternalOtgEnabled() const { return _otg_enabled; }
	// [goal_components][B]: what the control law tracks -- the internal OTG's output when enabled, else the goal (getDesired*)
	std::vector<double> getDesiredState() {
		need();
		std::vector<double> d((size_t)saip_batch_goal_components(_batch, _id) * _robot->batchSize());
		check(saip_batch_get_desired_host(_batch, _id, d.data()));
		return d;
	}
	// OTG::isGoalReached per instance
	std::vector<bool> isInternalOtgGoalReached() {
		need();
		std::vector<int> fl(_robot->batchSize());
		check(saip_batch_get_otg_status_host(_batch, _id, fl.data(), nullptr));
		std::vector<bool> r(fl.size());
		for (size_t i = 0; i < fl.size(); i++) r[i] = (fl[i] & 1) != 0;
		return r;
	}
	void disableVelocitySaturation() { cfg([](saip_batch* b, int id) { return saip_batch_set_velocity_saturation(b, id, 0); }); }
	// (B x dof x dof as [dof*dof][B]) nullspace projector of this task for the current state, TemplateTask.h:71-77
	std::vector<double> getTaskNullspace() {
		need();
		const int n = _robot->dof();
		std::vector<double> N((size_t)n * n * _robot->batchSize());
		if (_manual) check(saip_batch_task_get_nullspaces_host(_batch, _id, N.data(), nullptr, nullptr));
		else check(saip_batch_get_task_nullspace_host(_batch, _id, N.data()));
		return N;
	}
	// N_prec this task was updated with = N_{t-1} ... N_0 of the tasks above it (identity for the first), TemplateTask.h:79-83; same layout
	std::vector<double> getPreviousTasksNullspace() {
		need();
		const int n = _robot->dof();
		const size_t B = _robot->batchSize();
		std::vector<double> Np((size_t)n * n * B, 0.0), Ns(Np.size()), T(Np.size());
		if (_manual) {
			check(saip_batch_task_get_nullspaces_host(_batch, _id, nullptr, Np.data(), nullptr));
			return Np;
		}
		for (int i = 0; i < n; i++)
			for (size_t b = 0; b < B; b++) Np[((size_t)i * n + i) * B + b] = 1.0;
		for (int s = 0; s < _id; s++) {
			check(saip_batch_get_task_nullspace_host(_batch, s, Ns.data()));
			matmulBatched(Ns, Np, T, n, B);
			Np.swap(T);
		}
		return Np;
	}
	// N N_prec, what the next task of the hierarchy is updated with, TemplateTask.h:85-89
	std::vector<double> getTaskAndPreviousNullspace() {
		const int n = _robot->dof();
		const size_t B = _robot->batchSize();
		if (_manual) {
			std::vector<double> Nt((size_t)n * n * B);
			check(saip_batch_task_get_nullspaces_host(_batch, _id, nullptr, nullptr, Nt.data()));
			return Nt;
		}
		std::vector<double> N = getTaskNullspace(), Np = getPreviousTasksNullspace(), T(N.size());
		matmulBatched(N, Np, T, n, B);
		return T;
	}

protected:
	friend class RobotController;
	static void matmulBatched(const std::vector<double>& A, const std::vector<double>& Bm, std::vector<double>& C, int n, size_t B) {  // [n*n][B] layout
		for (int i = 0; i < n; i++)
			for (int j = 0; j < n; j++)
				for (size_t b = 0; b < B; b++) {
					double s = 0.0;
					for (int l = 0; l < n; l++) s += A[((size_t)i * n + l) * B + b] * Bm[((size_t)l * n + j) * B + b];
					C[((size_t)i * n + j) * B + b] = s;
				}
	}
	template <typename F>
	void cfg(F f) {
		_log.emplace_back(f);  // replayed into whichever batch the task joins later
		if (_batch) check(f(_batch, _id));
	}
	// the batch this task is evaluated in: its RobotController's, or -- for a task driven by hand like in the reference's example 04 --
	// a private one-task batch created on first use
	void need() {
		if (_batch) return;
		saip_batch* b = nullptr;
		check(saip_batch_create(_robot->handle(), _robot->batchSize(), _robot->device(), &b));
		try {
			check(add(b, &_id));
			check(saip_batch_finalize(b));
			for (auto& f : _log) check(f(b, _id));
		} catch (...) {
			saip_batch_destroy(b);
			throw;
		}
		_batch = b;
		_private = true;
		_robot->attach(b);
	}
	void dropPrivateBatch() {
		if (_private && _batch) {
			_robot->detach(_batch);
			saip_batch_destroy(_batch);
		}
		_private = false;
		_batch = nullptr;
	}
	void pushState() { _robot->pushTo(_batch); }
	void pushAndUpdate(const double* N_prec) {
		_robot->pushTo(_batch);
		check(saip_batch_task_update_model(_batch, _id, N_prec));
		_manual = true;
	}
	std::vector<double> torques(const double* tau_prec) {
		need();
		if (!_manual) throw std::runtime_error("task [" + _task_name + "]: call updateTaskModel(N_prec) before computeTorques()");
		std::vector<double> tau((size_t)_robot->dof() * _robot->batchSize());
		_status.assign(_robot->batchSize(), 0);
		check(saip_batch_task_compute_torques(_batch, _id, tau_prec, tau.data(), _status.data()));
		return tau;
	}
	void setField(int first, int comps, const std::vector<double>& v, const char* what) {
		need();
		if (v.size() != (size_t)comps * _robot->batchSize()) throw std::invalid_argument(what);
		check(saip_batch_set_goal_field_host(_batch, _id, first, comps, v.data()));
	}
	virtual saip_status add(saip_batch* b, int* id) = 0;

	std::shared_ptr<SaiModel> _robot;
	std::string _task_name;
	TaskType _task_type;
	double _loop_timestep;
	bool _otg_enabled = true;  // reference default (JointTask.h:38, MotionForceTask.h:67)
	double _bie_threshold = 0.1;
	saip_batch* _batch = nullptr;
	int _id = -1;
	bool _private = false;  // _batch is this task's own one-task batch (no RobotController)
	bool _manual = false;   // model last updated through updateTaskModel(N_prec)
	std::vector<uint8_t> _status;
	std::vector<std::function<saip_status(saip_batch*, int)>> _log;
};

class MotionForceTask : public TemplateTask {
public:
	// full 6-dof task, MotionForceTask.h:96-101
	MotionForceTask(std::shared_ptr<SaiModel>& robot, const std::string& link_name, const double (&pos_in_link)[3],
					const std::string& task_name = "motion_force_task", double loop_timestep = 0.001)
		: TemplateTask(robot, task_name, MOTION_FORCE_TASK, loop_timestep), _link(link_name), _partial(false) {
		for (int i = 0; i < 3; i++) _pos[i] = pos_in_link[i];
	}
	// partial task, MotionForceTask.h:103-110: controlled directions as flat xyz triples
	MotionForceTask(std::shared_ptr<SaiModel>& robot, const std::string& link_name, const std::vector<double>& controlled_directions_translation,
					const std::vector<double>& controlled_directions_rotation, const double (&pos_in_link)[3],
					const std::string& task_name = "partial_motion_force_task", double loop_timestep = 0.001)
		: TemplateTask(robot, task_name, MOTION_FORCE_TASK, loop_timestep), _link(link_name), _partial(true),
		  _dt(controlled_directions_translation), _dr(controlled_directions_rotation) {
		for (int i = 0; i < 3; i++) _pos[i] = pos_in_link[i];
		if (_dt.empty() && _dr.empty())
			throw std::invalid_argument("controlled_directions_translation and controlled_directions_rotation cannot both be empty in MotionForceTask::MotionForceTask\n");
	}
	void setGoalPosition(const std::vector<double>& x) { setField(0, 3, x, "setGoalPosition: expected [3][B]"); }
	void setGoalOrientation(const std::vector<double>& R) { setField(3, 9, R, "setGoalOrientation: expected [9][B] (row-major R)"); }
	void setGoalLinearVelocity(const std::vector<double>& v) { setField(12, 3, v, "setGoalLinearVelocity: expected [3][B]"); }
	void setGoalAngularVelocity(const std::vector<double>& w) { setField(15, 3, w, "setGoalAngularVelocity: expected [3][B]"); }
	void setGoalLinearAcceleration(const std::vector<double>& a) { setField(18, 3, a, "setGoalLinearAcceleration: expected [3][B]"); }
	void setGoalAngularAcceleration(const std::vector<double>& a) { setField(21, 3, a, "setGoalAngularAcceleration: expected [3][B]"); }
	void setPosControlGains(double kp, double kv, double ki = 0) { cfg([=](saip_batch* b, int id) { return saip_batch_set_pos_control_gains(b, id, &kp, &kv, &ki, 1); }); }
	void setOriControlGains(double kp, double kv, double ki = 0) { cfg([=](saip_batch* b, int id) { return saip_batch_set_ori_control_gains(b, id, &kp, &kv, &ki, 1); }); }
	void enableSingularityHandling() { cfg([](saip_batch* b, int id) { return saip_batch_set_singularity_handling(b, id, 1); }); }    // MotionForceTask.h:715
	void disableSingularityHandling() { cfg([](saip_batch* b, int id) { return saip_batch_set_singularity_handling(b, id, 0); }); }  // :723
	// blended type-1 / type-2 strategies for instances inside the singularity bounds: on by default like in the reference; off = flagged instead
	void setSingularityStrategies(bool enabled) { cfg([=](saip_batch* b, int id) { return saip_batch_set_singularity_strategies(b, id, enabled ? 1 : 0); }); }
	void setSingularityHandlingGains(double kp_type_1, double kv_type_1, double kv_type_2) {  // MotionForceTask.h:749
		cfg([=](saip_batch* b, int id) { return saip_batch_set_singularity_gains(b, id, kp_type_1, kv_type_1, kv_type_2); });
	}
	void handleAllSingularitiesAsType1(bool flag) { cfg([=](saip_batch* b, int id) { return saip_batch_set_all_singularities_type1(b, id, flag ? 1 : 0); }); }  // :698
	void setType1Posture(const std::vector<double>& q_des) {  // :707, one posture for every instance
		cfg([q_des](saip_batch* b, int id) { return saip_batch_set_type1_posture(b, id, q_des.data(), 0); });
	}
	void setSingularityHandlingBounds(double s_min, double s_max) { cfg([=](saip_batch* b, int id) { return saip_batch_set_singularity_bounds(b, id, s_min, s_max); }); }
	// MotionForceTask.cpp:510-523 (defaults MotionForceTask.h:68-71); the jerk-limited variant throws (not on the device)
	void enableInternalOtgAccelerationLimited(double max_linear_velocity = 0.3, double max_linear_acceleration = 2.0,
											  double max_angular_velocity = M_PI / 3, double max_angular_acceleration = 2 * M_PI) {
		_otg_enabled = true;
		cfg([=](saip_batch* b, int id) {
			const double v[2] = {max_linear_velocity, max_angular_velocity}, a[2] = {max_linear_acceleration, max_angular_acceleration};
			return saip_batch_set_otg_acceleration_limited(b, id, v, a, 2);
		});
	}
	// MotionForceTask::enableInternalOtgJerkLimited, MotionForceTask.cpp:525-545 (argument order of MotionForceTask.h:416-421)
	void enableInternalOtgJerkLimited(double max_linear_velocity, double max_linear_acceleration, double max_linear_jerk, double max_angular_velocity,
									  double max_angular_acceleration, double max_angular_jerk) {
		_otg_enabled = true;
		cfg([=](saip_batch* b, int id) {
			const double v[2] = {max_linear_velocity, max_angular_velocity}, a[2] = {max_linear_acceleration, max_angular_acceleration},
						 j[2] = {max_linear_jerk, max_angular_jerk};
			return saip_batch_set_otg_jerk_limited(b, id, v, a, j, 2);
		});
	}
	// MotionForceTask::enableVelocitySaturation(linear, angular), MotionForceTask.cpp:771-792
	void enableVelocitySaturation(double linear_vel_sat = 0.3, double angular_vel_sat = M_PI / 3) {
		cfg([=](saip_batch* b, int id) {
			const double v[2] = {linear_vel_sat, angular_vel_sat};
			saip_status st = saip_batch_set_saturation_velocities(b, id, v, 2);
			return st ? st : saip_batch_set_velocity_saturation(b, id, 1);
		});
	}
	// hybrid motion/force control, MotionForceTask.h:560-600 (open-loop force control; closed loop is refused)
	void parametrizeForceMotionSpaces(int force_space_dimension, double ax = 0, double ay = 0, double az = 0) {
		cfg([=](saip_batch* b, int id) {
			const double a[3] = {ax, ay, az};
			return saip_batch_parametrize_force_motion_spaces(b, id, force_space_dimension, a, nullptr);
		});
	}
	void parametrizeMomentRotMotionSpaces(int moment_space_dimension, double ax = 0, double ay = 0, double az = 0) {
		cfg([=](saip_batch* b, int id) {
			const double a[3] = {ax, ay, az};
			return saip_batch_parametrize_moment_rot_motion_spaces(b, id, moment_space_dimension, a, nullptr);
		});
	}
	void setGoalForce(const std::vector<double>& f) { setField(24, 3, f, "setGoalForce: expected [3][B]"); }
	void setGoalMoment(const std::vector<double>& m) { setField(27, 3, m, "setGoalMoment: expected [3][B]"); }
	void setForceControlGains(double kp, double kv, double ki) { cfg([=](saip_batch* b, int id) { return saip_batch_set_force_control_gains(b, id, kp, kv, ki); }); }
	void setMomentControlGains(double kp, double kv, double ki) { cfg([=](saip_batch* b, int id) { return saip_batch_set_moment_control_gains(b, id, kp, kv, ki); }); }
	void setClosedLoopForceControl(bool e) { cfg([=](saip_batch* b, int id) { return saip_batch_set_closed_loop_force_control(b, id, e); }); }
	void setClosedLoopMomentControl(bool e) { cfg([=](saip_batch* b, int id) { return saip_batch_set_closed_loop_moment_control(b, id, e); }); }
	// control point [3][B] and compliant-frame orientation [9][B] (row-major per instance) in the world frame at the robot's current
	// state, MotionForceTask.h:121-138
	std::vector<double> getCurrentPosition() {
		need();
		std::vector<double> p((size_t)3 * _robot->batchSize());
		pushState();
		check(saip_batch_get_current_pose_host(_batch, _id, p.data(), nullptr));
		return p;
	}
	std::vector<double> getCurrentOrientation() {
		need();
		std::vector<double> r((size_t)9 * _robot->batchSize());
		pushState();
		check(saip_batch_get_current_pose_host(_batch, _id, nullptr, r.data()));
		return r;
	}
	void resetIntegratorsLinear() {
		need();
		check(saip_batch_reset_integrators(_batch, _id, 1));
	}
	void resetIntegratorsAngular() {
		need();
		check(saip_batch_reset_integrators(_batch, _id, 2));
	}
	void enablePassivity() { cfg([](saip_batch* b, int id) { return saip_batch_set_passivity(b, id, 1); }); }    // MotionForceTask.h:630-631
	void disablePassivity() { cfg([](saip_batch* b, int id) { return saip_batch_set_passivity(b, id, 0); }); }
	// MotionForceTask.cpp:805-828: sensed force / moment in the sensor frame, [3][B] each
	void updateSensedForceAndMoment(const std::vector<double>& force, const std::vector<double>& moment) {
		setField(30, 3, force, "updateSensedForceAndMoment: expected [3][B]");
		setField(33, 3, moment, "updateSensedForceAndMoment: expected [3][B]");
	}
	// setFeedforwardForceGain / MomentGain, setMaxForceControlFeedbackOutput / Moment (MotionForceTask.h:330-355) in one call
	void setForceControlParameters(double kff_force = 0.95, double kff_moment = 0.95, double max_force_feedback = 20.0, double max_moment_feedback = 10.0) {
		cfg([=](saip_batch* b, int id) { return saip_batch_set_force_control_parameters(b, id, kff_force, kff_moment, max_force_feedback, max_moment_feedback); });
	}
	void setPosControlGainsUnsafe(double kp, double kv, double ki = 0) { setPosControlGains(kp, kv, ki); }
	void setOriControlGainsUnsafe(double kp, double kv, double ki = 0) { setOriControlGains(kp, kv, ki); }

protected:
	saip_status add(saip_batch* b, int* id) override {
		return saip_batch_add_motion_force_task(b, _task_name.c_str(), _link.c_str(), _pos, nullptr, _dt.empty() ? nullptr : _dt.data(),
												_partial ? (int)_dt.size() / 3 : -1, _dr.empty() ? nullptr : _dr.data(),
												_partial ? (int)_dr.size() / 3 : -1, _loop_timestep, id);
	}
	std::string _link;
	bool _partial;
	double _pos[3];
	std::vector<double> _dt, _dr;
};

class JointTask : public TemplateTask {
public:
	JointTask(std::shared_ptr<SaiModel>& robot, const std::string& task_name = "joint_task", double loop_timestep = 0.001)
		: TemplateTask(robot, task_name, JOINT_TASK, loop_timestep), _rows(0) {}
	// joint_selection_matrix: rows x dof, row-major (JointTask.h:66-75)
	JointTask(std::shared_ptr<SaiModel>& robot, const std::vector<double>& joint_selection_matrix, int rows,
			  const std::string& task_name = "partial_joint_task", double loop_timestep = 0.001)
		: TemplateTask(robot, task_name, JOINT_TASK, loop_timestep), _S(joint_selection_matrix), _rows(rows) {
		if (rows <= 0 || _S.size() != (size_t)rows * robot->dof())
			throw std::invalid_argument("joint selection matrix size not consistent with robot dof in JointTask constructor\n");
	}
	int getTaskDof() const { return _rows > 0 ? _rows : _robot->dof(); }
	bool isFullJointTask() const { return getTaskDof() == _robot->dof(); }
	void setGoalPosition(const std::vector<double>& q) { setField(0, getTaskDof(), q, "goal position vector size not consistent with task dof in JointTask::setGoalPosition\n"); }
	void setGoalVelocity(const std::vector<double>& dq) { setField(getTaskDof(), getTaskDof(), dq, "goal velocity vector size not consistent with task dof in JointTask::setGoalVelocity\n"); }
	void setGoalAcceleration(const std::vector<double>& ddq) { setField(2 * getTaskDof(), getTaskDof(), ddq, "goal acceleration vector size not consistent with task dof in JointTask::setGoalAcceleration\n"); }
	void setGains(double kp, double kv, double ki = 0) { cfg([=](saip_batch* b, int id) { return saip_batch_set_joint_gains(b, id, &kp, &kv, &ki, 1); }); }
	void setGainsUnsafe(double kp, double kv, double ki = 0) { setGains(kp, kv, ki); }
	// rows x dof, row-major (identity for the full task), JointTask.h getJointSelectionMatrix
	std::vector<double> getJointSelectionMatrix() const {
		if (_rows > 0) return _S;
		const int n = _robot->dof();
		std::vector<double> I((size_t)n * n, 0.0);
		for (int i = 0; i < n; i++) I[(size_t)i * n + i] = 1.0;
		return I;
	}
	// JointTask.cpp:358-381 (defaults JointTask.h:40-41); jerk-limited: JointTask.cpp:383-410
	void enableInternalOtgAccelerationLimited(double max_velocity = M_PI / 3.0, double max_acceleration = 2.0 * M_PI) {
		_otg_enabled = true;
		cfg([=](saip_batch* b, int id) { return saip_batch_set_otg_acceleration_limited(b, id, &max_velocity, &max_acceleration, 1); });
	}
	void enableInternalOtgJerkLimited(double max_velocity, double max_acceleration, double max_jerk) {
		_otg_enabled = true;
		cfg([=](saip_batch* b, int id) { return saip_batch_set_otg_jerk_limited(b, id, &max_velocity, &max_acceleration, &max_jerk, 1); });
	}
	void enableVelocitySaturation(double saturation_velocity = M_PI / 3.0) {  // JointTask.cpp:410-421
		cfg([=](saip_batch* b, int id) {
			saip_status st = saip_batch_set_saturation_velocities(b, id, &saturation_velocity, 1);
			return st ? st : saip_batch_set_velocity_saturation(b, id, 1);
		});
	}

protected:
	saip_status add(saip_batch* b, int* id) override {
		return saip_batch_add_joint_task(b, _task_name.c_str(), _rows > 0 ? _S.data() : nullptr, _rows, _loop_timestep, id);
	}
	std::vector<double> _S;
	int _rows;
};

// RobotController.h:47-90
class RobotController {
public:
	RobotController(std::shared_ptr<SaiModel>& robot, std::vector<std::shared_ptr<TemplateTask>>& tasks) : _robot(robot), _tasks(tasks) {
		if (tasks.empty()) throw std::invalid_argument("RobotController must have at least one task");
		for (auto& t : tasks)
			if (t->getConstRobotModel() != robot) throw std::invalid_argument("All tasks must have the same robot model in RobotController");
		for (auto& t : tasks)
			if (t->_batch && !t->_private) throw std::invalid_argument("task [" + t->getTaskName() + "] already belongs to a RobotController");
		check(saip_batch_create(robot->handle(), robot->batchSize(), robot->device(), &_batch));
		std::vector<int> ids(tasks.size(), -1);
		try {
			for (size_t i = 0; i < tasks.size(); i++) check(tasks[i]->add(_batch, &ids[i]));
			check(saip_batch_finalize(_batch));
		} catch (...) {
			saip_batch_destroy(_batch);
			_batch = nullptr;
			throw;
		}
		for (size_t i = 0; i < tasks.size(); i++) {
			auto& t = tasks[i];
			// a task that was driven by hand before moves here: configuration replayed, goal kept, integrators start afresh
			std::vector<double> goal;
			if (t->_private && robot->device() >= 0) {
				goal.resize((size_t)saip_batch_goal_components(t->_batch, t->_id) * robot->batchSize());
				check(saip_batch_get_goal_host(t->_batch, t->_id, goal.data()));
			}
			t->dropPrivateBatch();
			t->_batch = _batch;
			t->_id = ids[i];
			t->_manual = false;
			for (auto& f : t->_log) check(f(_batch, t->_id));
			if (!goal.empty()) check(saip_batch_set_goal_host(_batch, t->_id, goal.data()));
			_task_names.push_back(t->getTaskName());
		}
		robot->attach(_batch);
	}
	~RobotController() {
		for (auto& t : _tasks)
			if (t->_batch == _batch) {
				t->_batch = nullptr;
				t->_manual = false;
			}
		_robot->detach(_batch);
		saip_batch_destroy(_batch);
	}
	RobotController(const RobotController&) = delete;
	RobotController& operator=(const RobotController&) = delete;

	void updateControllerTaskModels() {
		pushState();
		check(saip_batch_update_task_models(_batch));
		for (auto& t : _tasks) t->_manual = false;
	}
	// [dof][B] joint torques; instances that left the non-singular branch carry NaN and status()[b] == 1
	std::vector<double> computeControlTorques() {
		std::vector<double> tau((size_t)_robot->dof() * _robot->batchSize());
		_status.assign(_robot->batchSize(), 0);
		check(saip_batch_compute_control_torques(_batch, tau.data(), _status.data()));
		return tau;
	}
	const std::vector<uint8_t>& status() const { return _status; }
	void enableGravityCompensation(bool e) { check(saip_batch_enable_gravity_compensation(_batch, e)); }
	void enableJointLimitAvoidance(bool e) { check(saip_batch_enable_joint_limit_avoidance(_batch, e)); }
	void enableTorqueSaturation(bool e) { check(saip_batch_enable_torque_saturation(_batch, e)); }
	// torques of instances that end a cycle flagged (status 1): false (default) = the last valid torques are held, true = NaN
	void setFlaggedTorquePolicy(bool nan) { check(saip_batch_set_flagged_torque_policy(_batch, nan ? 1 : 0)); }
	void setFlaggedRecompute(bool on_list) { check(saip_batch_set_flagged_recompute(_batch, on_list ? 1 : 0)); }
	void reinitializeTasks() {
		pushState();
		check(saip_batch_reinitialize_tasks(_batch));
	}
	const std::vector<std::string>& getTaskNames() const { return _task_names; }
	std::shared_ptr<JointTask> getJointTaskByName(const std::string& name) { return byName<JointTask>(name, JOINT_TASK, "JointTask"); }
	std::shared_ptr<MotionForceTask> getMotionForceTaskByName(const std::string& name) { return byName<MotionForceTask>(name, MOTION_FORCE_TASK, "MotionForceTask"); }
	saip_batch* handle() { return _batch; }
	void pushState() { _robot->pushTo(_batch); }

private:
	template <typename T>
	std::shared_ptr<T> byName(const std::string& name, TaskType type, const char* what) {
		for (auto& t : _tasks)
			if (t->getTaskName() == name) {
				if (t->getTaskType() != type)
					throw std::invalid_argument("Task " + name + " is not a " + what + ", and cannot be casted as such in RobotController::GetTaskByName");
				return std::dynamic_pointer_cast<T>(t);
			}
		throw std::invalid_argument("Task " + name + " not found in RobotController::GetTaskByName");
	}
	std::shared_ptr<SaiModel> _robot;
	std::vector<std::shared_ptr<TemplateTask>> _tasks;
	std::vector<std::string> _task_names;
	std::vector<uint8_t> _status;
	saip_batch* _batch = nullptr;
};

}  // namespace SaiPrimitivesBatched

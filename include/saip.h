/*
 * saip.h -- C-ABI of the MI355X-native batched operational-space control engine.
 *
 * The reference (manips-sai-org/sai-primitives) has no FFI/plugin layer: its drop-in boundary for the
 * hot path is the public C++ class API
 *     TemplateTask        /root/reference/src/tasks/TemplateTask.h:43-116
 *     MotionForceTask     /root/reference/src/tasks/MotionForceTask.h:96-110, 211-247, 272-300, 423, 670-736
 *     JointTask           /root/reference/src/tasks/JointTask.h:56-75, 140-175, 237-257, 323, 363
 *     RobotController     /root/reference/src/RobotController.h:47-90
 * driven once per control cycle as  robot->updateModel(); updateControllerTaskModels(); <set goals>;
 * computeControlTorques()  (/root/reference/examples/05-using_robot_controller/05-using_robot_controller.cpp:143-196).
 * This header is what a binding for that path would bind: one entry point per reference call, in a batched
 * flavour (B independent robot instances evaluated by one GPU launch).  Plain pointers and sizes only.
 *
 * Conventions
 *   - FP64 everywhere.  All per-instance arrays are struct-of-arrays: field component c of instance b lives
 *     at  ptr[c * ld + b]  where ld = saip_batch_ld(batch) (>= B, padded for alignment) for DEVICE arrays and
 *     ld = B for HOST arrays passed to the *_host setters/getters.
 *   - batch-uniform configuration (gains, flags, task definitions) mirrors the reference setters 1:1.
 *   - errors: status codes; saip_last_error() returns the text the reference would have thrown as
 *     std::invalid_argument (thread-local).  Nothing is computed on the CPU: every compute entry point
 *     fails with SAIP_ERR_NO_DEVICE when no HIP device is usable.
 *
 * Limits (what the reference accepts and this engine does not: every one fails loudly with SAIP_ERR_INVALID_ARGUMENT / _UNSUPPORTED)
 *   - robots are SERIAL CHAINS (link i's parent is link i - 1; sai-model also reads trees) with at most SAIP_MAX_DOF = 32 movable
 *     joints, revolute or prismatic; fixed links are merged into their parents.
 *   - a hierarchy holds at most SAIP_MAX_TASKS = 8 tasks (MotionForceTask / JointTask; RobotController's joint-limit-avoidance task is
 *     the controller option of saip_batch_enable_joint_limit_avoidance, not a ninth task).
 *   - a motion-force task of rank 1 (a single controlled direction) is refused: SingularityHandler's own loop leaves such a task with
 *     stale state in the reference.
 *   - haptic / teleoperation tasks and everything around the control path (Redis, logging, UI) are out of scope.
 */
#ifndef SAIP_H_
#define SAIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAIP_MAX_DOF 32
#define SAIP_MAX_TASKS 8
#define SAIP_NAME_LEN 48

typedef struct saip_model saip_model;
typedef struct saip_batch saip_batch;

typedef enum {
	SAIP_OK = 0,
	SAIP_ERR_INVALID_ARGUMENT = 1, /* what the reference reports by throwing std::invalid_argument */
	SAIP_ERR_UNSUPPORTED = 2,      /* valid in the reference, not (yet) implemented by the engine: fails loudly */
	SAIP_ERR_NO_DEVICE = 3,
	SAIP_ERR_DEVICE = 4,           /* HIP runtime error */
	SAIP_ERR_ORDER = 5             /* call-order contract violated (e.g. compute before update) */
} saip_status;

typedef enum { SAIP_JOINT_FIXED = 0, SAIP_JOINT_REVOLUTE = 1, SAIP_JOINT_PRISMATIC = 2 } saip_joint_type;

/* SaiPrimitives::TaskType, TemplateTask.h:19-24 */
typedef enum { SAIP_TASK_UNDEFINED = 0, SAIP_TASK_JOINT_LIMIT_AVOIDANCE = 1, SAIP_TASK_JOINT = 2, SAIP_TASK_MOTION_FORCE = 3 } saip_task_type;

/* SaiPrimitives::DynamicDecouplingType, helper_modules/SaiPrimitivesCommonDefinitions.h:14-20 (same values) */
typedef enum { SAIP_FULL_DYNAMIC_DECOUPLING = 0, SAIP_BOUNDED_INERTIA_ESTIMATES = 1, SAIP_IMPEDANCE = 2 } saip_decoupling_type;

/* per-instance status written by a cycle (uint8).  SINGULAR excludes the others; TRUNCATED, POPC_OVERFLOW and BLENDED are
 * informational bits on valid torques and may combine. */
enum {
	SAIP_INSTANCE_OK = 0,
	SAIP_INSTANCE_SINGULAR = 1,       /* the engine could not evaluate the instance (blended strategies switched off and the instance outside the
	                                     non-singular branch; sigma_min/sigma_0 < 1e-6; an ambiguous rank gap): its torques are NOT written -- the
	                                     last valid ones are held -- or NaN, see saip_batch_set_flagged_torque_policy; integrators do not advance */
	SAIP_INSTANCE_TRUNCATED = 2,      /* a motion-force task was reduced to its non-singular subspace (handling disabled) or passed through (fully singular) */
	SAIP_INSTANCE_POPC_OVERFLOW = 4,  /* the passivity observer's energy window exceeded the 1024 samples kept on the device */
	SAIP_INSTANCE_BLENDED = 8         /* a motion-force task ran the blended type-1 / type-2 singularity strategies (valid torques) */
};

/* One URDF link + the joint that attaches it to its parent (what sai-model/sai-urdfreader read from a URDF).
 * Link i's parent is link i-1 (serial chain); link 0 hangs off the fixed base.  Fixed links are merged into
 * their parent for dynamics (as RBDL does) and stay addressable by name for kinematics. */
typedef struct saip_link_desc {
	char name[SAIP_NAME_LEN];
	int joint_type;        /* saip_joint_type */
	double origin_xyz[3];  /* <origin xyz> of the joint: child frame in parent frame */
	double origin_rpy[3];  /* <origin rpy>: fixed-axis roll, pitch, yaw */
	double axis[3];        /* <axis xyz>, in the child frame */
	double mass;
	double com[3];         /* <inertial><origin xyz> */
	double inertia[6];     /* ixx iyy izz ixy ixz iyz about the COM, link axes */
	double q_lower, q_upper, velocity_limit, effort_limit; /* <limit> */
} saip_link_desc;

/* ---------------------------------------------------------------- robot model (constants only; replaces the
 * SaiModel constructor + URDF parse; the per-cycle SaiModel::updateModel() is part of the GPU cycle) */
saip_status saip_model_create_serial_chain(const saip_link_desc* links, int n_links, saip_model** out);
void saip_model_destroy(saip_model* model);
int saip_model_dof(const saip_model* model);                               /* SaiModel::dof() */
int saip_model_link_index(const saip_model* model, const char* link_name); /* -1 when absent */
saip_status saip_model_joint_limits(const saip_model* model, double* q_lower, double* q_upper, double* velocity,
									double* effort);                       /* SaiModel::jointLimits() */

/* ---------------------------------------------------------------- batch = B robots + one RobotController each */
/* device: HIP device ordinal.  The engine owns one stream and all device buffers of the batch. */
saip_status saip_batch_create(const saip_model* model, int batch_size, int device, saip_batch** out);
void saip_batch_destroy(saip_batch* batch);
int saip_batch_size(const saip_batch* batch);
int saip_batch_ld(const saip_batch* batch); /* leading dimension (instances) of device SoA arrays */
/* before saip_batch_finalize: a larger leading dimension than the default (B rounded up to 32; must stay a multiple of 32).  The shards of a
 * sharded run that differ by one instance take the largest shard's, so that the final all-gather moves slabs of one shape. */
saip_status saip_batch_set_leading_dimension(saip_batch* batch, int ld);
int saip_batch_dof(const saip_batch* batch);

/* MotionForceTask constructors, MotionForceTask.h:96-110 / MotionForceTask.cpp:16-89.
 * n_trans < 0 && n_rot < 0: full 6-dof task; otherwise dirs_* hold n_* controlled direction vectors (3 doubles
 * each).  rot_in_link may be NULL (identity compliant frame rotation).  Task order = hierarchy order. */
saip_status saip_batch_add_motion_force_task(saip_batch* batch, const char* task_name, const char* link_name,
											 const double pos_in_link[3], const double rot_in_link[9],
											 const double* dirs_trans, int n_trans, const double* dirs_rot, int n_rot,
											 double loop_timestep, int* task_id);
/* JointTask constructors, JointTask.h:56-75 / JointTask.cpp:14-43.  rows == 0: full joint task (S = I). */
saip_status saip_batch_add_joint_task(saip_batch* batch, const char* task_name, const double* joint_selection /*rows x dof*/,
									  int rows, double loop_timestep, int* task_id);
/* RobotController constructor validation, RobotController.cpp:8-66 (>=1 task, equal loop timesteps, unique
 * names, nothing after a full joint task); allocates the device arena.  Must precede any state/goal/cycle call. */
saip_status saip_batch_finalize(saip_batch* batch);

int saip_batch_task_count(const saip_batch* batch);
int saip_batch_task_type(const saip_batch* batch, int task_id);   /* TemplateTask::getTaskType */
const char* saip_batch_task_name(const saip_batch* batch, int task_id); /* TemplateTask::getTaskName / getTaskNames */
int saip_batch_task_by_name(const saip_batch* batch, const char* name); /* get{Joint,MotionForce}TaskByName; -1 when absent */
int saip_batch_task_dof(const saip_batch* batch, int task_id);    /* JointTask::getTaskDof (rows of S); 6-dof projector rank for motion-force */
int saip_batch_goal_components(const saip_batch* batch, int task_id); /* 36 (x3,R9 row-major,v3,w3,a3,alpha3,goal force 3,goal moment 3,sensed force 3,sensed moment 3) or 3*m (q,dq,ddq) */
/* MotionForceTask partial task projection P (6x6 row-major, MotionForceTask.cpp:55-87) and the orthonormal basis of
 * range(P) the engine works in (6x6 row-major, first *rank columns valid; = _current_task_range, :162-168) */
saip_status saip_batch_get_task_projection(const saip_batch* batch, int task_id, double* P36, double* basis36, int* rank);

/* ---- batch-uniform task configuration (reference setters; defaults = the reference DefaultParameters) */
/* MotionForceTask::setPosControlGains / setOriControlGains (.h:272-300): n_gains = 1 (isotropic) or 3 */
saip_status saip_batch_set_pos_control_gains(saip_batch*, int task, const double* kp, const double* kv, const double* ki, int n_gains);
saip_status saip_batch_set_ori_control_gains(saip_batch*, int task, const double* kp, const double* kv, const double* ki, int n_gains);
/* JointTask::setGains (.h:237-257): n_gains = 1 or task dof */
saip_status saip_batch_set_joint_gains(saip_batch*, int task, const double* kp, const double* kv, const double* ki, int n_gains);
/* setDynamicDecouplingType (.h:670 / JointTask.h:363), setBoundedInertiaEstimateThreshold (.h:679 / JointTask.h:372) */
saip_status saip_batch_set_dynamic_decoupling_type(saip_batch*, int task, int type);
saip_status saip_batch_set_bie_threshold(saip_batch*, int task, double threshold);
/* MotionForceTask::enableSingularityHandling / disableSingularityHandling (.h:715-725 -> SingularityHandler.h:146-158).
 * enabled (reference default): instances that leave the fully non-singular branch run the reference's blended singularity
 * strategies (next entry).
 * disabled: the reference then uses only the non-singular part of the task, _N = _N_ns and tau = tau_ns
 * (SingularityHandler.cpp:146-148, 319-330): the task acts on the span of the leading left singular vectors of its projected
 * Jacobian, up to the first sigma_i / sigma_0 < s_max (:100-121); a fully singular task (sigma_0 < 1e-3) is passed through.
 * Implemented: such instances finish with valid torques and status 2 (flagged by the lane kernel, recomputed by the general
 * kernel launched behind it as a device-side slow path). */
saip_status saip_batch_set_singularity_handling(saip_batch*, int task, int enabled);
/* Blended singularity strategies of SingularityHandler (updateTaskModel :100-121, 146-228; classifySingularity :230-295;
 * computeTorques :310-367) for instances inside the bounds (s_min, s_max), with the handling strategy enabled.  ON by default, like in the
 * reference (which has no switch for them); switching them off is an engine extra: such instances are then flagged (status 1) and
 * nothing but the fully non-singular branch is ever evaluated.  On: the task is split into its non-singular and singular directions, the
 * singular ones are blended out by alpha = (sigma_s/sigma_0 - s_min)/(s_max - s_min) in favour of a joint strategy (type 1: hold
 * the entering posture, kp 50 / kv 14; type 2: constant torque 1e-2 x the torque limit along the open direction, damping kv 5),
 * chosen by the majority of the last 200 classifications; status bit 8.  Per-instance handler state (entering posture, type
 * window, type-2 direction) lives on the device and is reset by this call.  Conventions where the reference depends on the sign
 * its SVD happens to return are fixed as DESIGN.md states.  A fully singular task (sigma_0 < 1e-3) is passed through like in the reference
 * (zero torques, N = N_prec, :149-151, 316-317; status bit 2; its classification is skipped); sigma_min/sigma_0 < 1e-6 -> status 1. */
saip_status saip_batch_set_singularity_strategies(saip_batch*, int task, int enabled);
/* MotionForceTask::setSingularityHandlingGains (.h:749 -> SingularityHandler.h:119; defaults 50, 14, 5) */
saip_status saip_batch_set_singularity_gains(saip_batch*, int task, double kp_type_1, double kv_type_1, double kv_type_2);
/* MotionForceTask::handleAllSingularitiesAsType1 (.h:698 -> SingularityHandler.h:131) */
saip_status saip_batch_set_all_singularities_type1(saip_batch*, int task, int flag);
/* MotionForceTask::setType1Posture (.h:707 -> SingularityHandler.h:140): the posture the type-1 strategy holds, q_des[dof]
 * (per_instance = 0) or [B][dof] (1).  As in the reference it lasts until an instance next ENTERS a singular region (:232-235). */
saip_status saip_batch_set_type1_posture(saip_batch*, int task, const double* q_des, int per_instance);
/* MotionForceTask::setSingularityHandlingBounds (.h:736) */
saip_status saip_batch_set_singularity_bounds(saip_batch*, int task, double s_min, double s_max);
/* disableInternalOtg / enableInternalOtg* (MotionForceTask.h:423, JointTask.h:323).  The reference default is ENABLED.
 * The acceleration-limited OTG (the reference default mode; Ruckig second-order position interface with phase
 * synchronisation) runs on the device, one launch ahead of the cycle kernel, and the control law tracks its output:
 * joint tasks OTG_joints.cpp (JointTask.cpp:313-319), motion-force tasks OTG_6dof_cartesian.cpp (MotionForceTask.cpp:394-406). */
saip_status saip_batch_set_internal_otg(saip_batch*, int task, int enabled);
/* JointTask::enableInternalOtgAccelerationLimited(max_velocity, max_acceleration) (JointTask.cpp:358-381): count = 1 (scalar) or
 * task dof; values <= 0 -> SAIP_ERR_INVALID_ARGUMENT (OTG_joints.cpp:44-74).  Defaults pi/3 rad/s, 2 pi rad/s^2 (JointTask.h:40-41).
 * MotionForceTask::enableInternalOtgAccelerationLimited(max_lin_vel, max_lin_acc, max_ang_vel, max_ang_acc)
 * (MotionForceTask.cpp:510-523): count = 2, max_velocity = {linear, angular}, max_acceleration = {linear, angular}; defaults
 * 0.3 m/s, 2 m/s^2, pi/3 rad/s, 2 pi rad/s^2 (MotionForceTask.h:68-71).
 * Enabling a previously disabled OTG re-initialises it at the current task position / pose on the next cycle. */
saip_status saip_batch_set_otg_acceleration_limited(saip_batch*, int task, const double* max_velocity, const double* max_acceleration, int count);
/* JointTask::enableInternalOtgJerkLimited(max_velocity, max_acceleration, max_jerk) (JointTask.cpp:383-410; OTG_joints::setMaxJerk,
 * OTG_joints.cpp:73-86): count = 1 or the task dof.  MotionForceTask::enableInternalOtgJerkLimited (MotionForceTask.cpp:525-545): count = 2,
 * {linear, angular} each.  Third-order (jerk-limited) Ruckig profiles on the device (csrc/saip_otg3.h: the reference's vendored Ruckig
 * 0.10.1 position interface, ruckig/src/ruckig/position-third-step{1,2}.cpp); limits must be positive.  The OTG is re-initialised at the
 * current task position / pose on the next cycle when it was off or acceleration-limited (JointTask.cpp:400-402). */
saip_status saip_batch_set_otg_jerk_limited(saip_batch*, int task, const double* max_velocity, const double* max_acceleration, const double* max_jerk, int count);
/* ---- the step after the path: what the reference's examples do with the external simulator,
 * sim->setJointTorques(tau); sim->integrate() (examples/05-using_robot_controller/05-using_robot_controller.cpp:225-231).
 * Forward dynamics M(q) qdd + b(q,dq) + g(q) = tau - damping*dq on the resident state with the torques of the last cycle held,
 * then `substeps` semi-implicit Euler steps of size dt (dq += dt*qdd; q += dt*dq).  gravity = NULL uses the model's gravity
 * (0, 0, -9.81); pass {0,0,0} for the gravity-free worlds of the reference's examples.  Instances whose torque is NaN (flagged
 * as outside the non-singular branch) coast.  The state changed: saip_batch_update_task_models is due before the next
 * saip_batch_compute_control_torques, exactly like after robot->setQ(). */
saip_status saip_batch_integrate(saip_batch*, double dt, int substeps, const double gravity[3], double damping);
/* `steps` closed-loop control periods { internal OTGs -> control cycle -> integrate(sim_dt, substeps) } enqueued back to back on the
 * engine stream without host synchronisation (follow with saip_batch_synchronize); goals stay as they are on the device. */
saip_status saip_batch_rollout_async(saip_batch*, int steps, double sim_dt, int substeps, const double gravity[3], double damping);
/* overwrite the resident torque array ([dof][B]) that saip_batch_integrate applies -- sim->setJointTorques(name, control_torques +
 * ui_torques) in the examples (05-...cpp:226-228) when the applied torque is not just the last cycle's output */
saip_status saip_batch_set_torques_host(saip_batch*, const double* tau);
/* read the resident state back: q, dq [dof][B] (either may be NULL) */
saip_status saip_batch_get_state_host(saip_batch*, double* q, double* dq);
/* getDesiredPosition/Velocity/Acceleration (JointTask.h:185-200; MotionForceTask desired* likewise): [goal_components][B] in the
 * goal layout -- the internal OTG's output of the last cycle when it is enabled, otherwise the goal itself. */
saip_status saip_batch_get_desired_host(saip_batch*, int task, double* desired);
/* per-instance OTG state of a task after the last cycle: flags (bit 0 goal reached = OTG_joints::isGoalReached, bit 3 an
 * error occurred since the last re-initialisation, bit 4 a trajectory finished with non-zero velocity) and the ruckig::Result of
 * the last cycle (0 Working, 1 Finished, < 0 error codes of ruckig/result.hpp).  Motion-force tasks additionally: bit 6 = the
 * goal orientation of the last cycle was not a rotation matrix (the reference throws, OTG_6dof_cartesian.cpp:158-162; the goal
 * is ignored).  Either pointer may be NULL. */
saip_status saip_batch_get_otg_status_host(saip_batch*, int task, int* flags, int* result);
/* velocity saturation: MotionForceTask::enableVelocitySaturation(linear, angular) / disableVelocitySaturation
 * (MotionForceTask.cpp:771-792, law :416-430, :449-462); JointTask::enableVelocitySaturation(value | vector) (JointTask.cpp:410-436,
 * law :327-341).  n_values: motion-force task 2 = {linear, angular}; joint task 1 or task dof; 0 keeps the stored/default limits
 * (MotionForceTask.h:63-64: 0.3, pi/3; JointTask.h:44: pi/3). */
saip_status saip_batch_set_velocity_saturation(saip_batch*, int task, int enabled);
saip_status saip_batch_set_saturation_velocities(saip_batch*, int task, const double* values, int n_values);
/* MotionForceTask::parametrizeForceMotionSpaces / parametrizeMomentRotMotionSpaces (MotionForceTask.cpp:828-890): dimension 0..3,
 * axis used for dimensions 1 and 2.  *changed (optional) = the reference's return value; when the parametrisation changed and the
 * device holds a state, the linear (angular) goal is reset to the current pose and the integrators to zero like the reference does. */
saip_status saip_batch_parametrize_force_motion_spaces(saip_batch*, int task, int force_space_dimension, const double axis[3], int* changed);
saip_status saip_batch_parametrize_moment_rot_motion_spaces(saip_batch*, int task, int moment_space_dimension, const double axis[3], int* changed);
/* constructor argument is_force_motion_parametrization_in_compliant_frame (MotionForceTask.h:96-110): axes and goal force/moment
 * are expressed in the compliant frame instead of the world frame */
saip_status saip_batch_set_parametrization_in_compliant_frame(saip_batch*, int task, int enabled);
/* MotionForceTask::setForceControlGains / setMomentControlGains: open-loop force control uses kv only (MotionForceTask.cpp:350-354, :379-383) */
saip_status saip_batch_set_force_control_gains(saip_batch*, int task, double kp, double kv, double ki);
saip_status saip_batch_set_moment_control_gains(saip_batch*, int task, double kp, double kv, double ki);
/* setClosedLoopForceControl / setClosedLoopMomentControl (MotionForceTask.cpp:973-986; law :327-349, :357-378, :484-487): PI feedback
 * on the sensed force / moment inside the force / moment space, saturated at the maximum feedback output, goal force feed-forward
 * scaled by kff.  The sensed force and moment of the current cycle are rows 30..32 / 33..35 of the task's goal block, in the SENSOR
 * frame (updateSensedForceAndMoment, :805-828: the engine applies _T_control_to_sensor and the control frame's world rotation of the
 * current state).  A change of either flag resets the linear / angular integrators like the reference.
 * saip_batch_set_passivity = enablePassivity / disablePassivity (.h:630-631): the windowed passivity observer + passivity controller
 * of POPCExplicitForceControl.cpp:29-96 around the force loop (off by default like in the reference; disabling re-initialises it).
 * The reference's energy window is an unbounded std::queue; the device keeps 1024 samples per instance and sets status bit 2 (value
 * 4) on an instance whose window would have grown beyond that (more than ~0.77 s of uninterrupted activity at 1 kHz). */
saip_status saip_batch_set_closed_loop_force_control(saip_batch*, int task, int enabled);
saip_status saip_batch_set_closed_loop_moment_control(saip_batch*, int task, int enabled);
saip_status saip_batch_set_passivity(saip_batch*, int task, int enabled);
/* setFeedforwardForceGain / setFeedforwardMomentGain / setMaxForceControlFeedbackOutput / setMaxMomentControlFeedbackOutput
 * (MotionForceTask.h:330-355); defaults 0.95, 0.95, 20 N, 10 Nm (.h:56-59) */
saip_status saip_batch_set_force_control_parameters(saip_batch*, int task, double kff_force, double kff_moment, double max_force_feedback, double max_moment_feedback);
/* setForceSensorFrame (MotionForceTask.cpp:794-803) given as _T_control_to_sensor = compliant_frame^-1 * T_link_sensor:
 * rotation 3x3 row-major and translation; default identity (:94) */
saip_status saip_batch_set_control_to_sensor_transform(saip_batch*, int task, const double* R_row_major, const double* translation);
/* RobotController::enable{GravityCompensation,JointLimitAvoidance,TorqueSaturation}, RobotController.h:64-75 */
saip_status saip_batch_enable_gravity_compensation(saip_batch*, int enabled);
saip_status saip_batch_enable_joint_limit_avoidance(saip_batch*, int enabled); /* JLA wrap, RobotController.cpp:96-112 */
saip_status saip_batch_enable_torque_saturation(saip_batch*, int enabled);
/* torques of an instance that ends a cycle flagged (status 1): 0 (default) = not written, i.e. the last valid torques of that instance are
 * held (zero before the first valid cycle); 1 = NaN.  The reference has no such state: it never refuses an instance. */
saip_status saip_batch_set_flagged_torque_policy(saip_batch*, int nan);
/* where the eight-lane kernels recompute the instances they cannot finish themselves (a task outside SingularityHandler's non-singular branch,
 * SingularityHandler.cpp:100-121, 146-158, 310-367): 0 (default) = in the kernel, by the wavefront that met them, right behind its epilogue -- no
 * second launch; costs nothing when there are none and ~70 us per instance and wavefront when there are, one after the other; 1 = on a device-side
 * list that a second launch behind every cycle spreads over the chip (+4.5 us per cycle, always) -- faster when MANY instances of one group of
 * eight are singular at once (measured, config 3's stack, 4096 instances: 1/8 of the batch packed into whole groups: 96 against 271 us per cycle;
 * spread evenly: 89 against 86).  Same results either way. */
saip_status saip_batch_set_flagged_recompute(saip_batch*, int on_list);
/* integrator state policy: 0 = advance the integral terms only while the task's ki != 0 (default; elides the
 * state traffic), 1 = advance every cycle like the reference does (MotionForceTask.cpp:411-413,446; JointTask.cpp:323) */
saip_status saip_batch_set_integrator_tracking(saip_batch*, int always);

/* ---- per-instance inputs (host staging: H2D copies on the engine stream) */
saip_status saip_batch_set_state_host(saip_batch*, const double* q /*[dof][B]*/, const double* dq /*[dof][B]*/); /* setQ/setDq */
/* whole goal block of a task, [goal_components][B]: setGoalPosition/Orientation/LinearVelocity/AngularVelocity/
 * LinearAcceleration/AngularAcceleration (MotionForceTask.h:211-247) or setGoalPosition/Velocity/Acceleration (JointTask.h:140-175) */
saip_status saip_batch_set_goal_host(saip_batch*, int task, const double* goal);
/* one goal field: component offset/count inside the goal block (e.g. position: 0,3; orientation: 3,9) */
saip_status saip_batch_set_goal_field_host(saip_batch*, int task, int first_component, int n_components, const double* values);
saip_status saip_batch_get_goal_host(saip_batch*, int task, double* goal);
/* TemplateTask::reInitializeTask / RobotController::reinitializeTasks: goal := current pose of every instance,
 * velocities/accelerations := 0, integrators := 0 (MotionForceTask.cpp:204-245, JointTask.cpp:91-107). GPU kernel. */
saip_status saip_batch_reinitialize_tasks(saip_batch*);
/* MotionForceTask::getCurrentPosition / getCurrentOrientation (MotionForceTask.h:121-138) at the state last set: pos [3][B], rot [9][B]
 * (row-major per instance); either pointer may be NULL */
saip_status saip_batch_get_current_pose_host(saip_batch*, int task, double* pos, double* rot);
/* TemplateTask::reInitializeTask of ONE task (MotionForceTask.cpp:204-245, JointTask.cpp:95-106) */
saip_status saip_batch_reinitialize_task(saip_batch*, int task);
/* MotionForceTask::resetIntegrators / resetIntegratorsLinear / resetIntegratorsAngular (MotionForceTask.cpp:988-1002; the linear part
 * also clears the force integrator, the angular part the moment integrator), JointTask::resetIntegrators.  parts: 1 linear (or the
 * joint task's), 2 angular, 3 both. */
saip_status saip_batch_reset_integrators(saip_batch*, int task, int parts);

/* ---- zero-copy access for resident pipelines (simulator / rollout on the same GPU) */
double* saip_batch_device_q(saip_batch*);
double* saip_batch_device_dq(saip_batch*);
double* saip_batch_device_goal(saip_batch*, int task);
double* saip_batch_device_tau(saip_batch*);          /* [dof][ld] */
uint8_t* saip_batch_device_status(saip_batch*);      /* [ld] */
/* write torques to a caller-owned device buffer (e.g. a torch tensor used for the RCCL gather); NULL restores the internal one */
saip_status saip_batch_bind_tau_device(saip_batch*, double* tau_dev /*[dof][ld]*/);
void* saip_batch_stream(saip_batch*);                /* hipStream_t */

/* ---- the control cycle */
/* robot->updateModel() + RobotController::updateControllerTaskModels() (RobotController.cpp:68-77) */
saip_status saip_batch_update_task_models(saip_batch*);
/* RobotController::computeControlTorques() (RobotController.cpp:79-118); requires update_task_models since the last
 * state change.  tau_host may be NULL (results stay on the device); status_host may be NULL. */
saip_status saip_batch_compute_control_torques(saip_batch*, double* tau_host /*[dof][B]*/, uint8_t* status_host /*[B]*/);
/* both of the above as ONE asynchronous launch on the engine stream; no host synchronisation */
saip_status saip_batch_step_async(saip_batch*);
saip_status saip_batch_synchronize(saip_batch*);
saip_status saip_batch_get_torques_host(saip_batch*, double* tau_host, uint8_t* status_host);
/* per-task diagnostics of the hierarchy, for parity tests of the task models: TemplateTask::getTaskNullspace()
 * ([dof*dof][B] row-major per instance).  Re-evaluates the task models with the general kernel (no control law; torques, status,
 * integrators, OTG and handler state of the last cycle stay untouched). */
saip_status saip_batch_get_task_nullspace_host(saip_batch*, int task, double* N /*[dof*dof][B]*/);

/* ---- the reference's per-task plug-in interface, TemplateTask.h:43-60 (pure virtuals of every task):
 *        virtual void updateTaskModel(const Eigen::MatrixXd& N_prec);
 *        virtual Eigen::VectorXd computeTorques();
 *        virtual Eigen::VectorXd computeTorques(const Eigen::VectorXd& tau_prec);
 *        getTaskNullspace() / getPreviousTasksNullspace() / getTaskAndPreviousNullspace()   (:71-89)
 * A caller that builds its own hierarchy drives the tasks by hand (examples/04-task_and_redundancy/04-task_and_redundancy.cpp:141-206):
 *        N_prec = I;  mf->updateTaskModel(N_prec);  N_prec = mf->getTaskAndPreviousNullspace();  jt->updateTaskModel(N_prec);
 *        tau = mf->computeTorques() + jt->computeTorques();
 * Nullspace matrices are [dof*dof][B] (host) / [dof*dof][ld] (device), row-major per instance.  The task keeps its own copy of N_prec
 * (MotionForceTask.cpp:259-267 / JointTask.cpp:230).  One launch of the general kernel per call, restricted to the task:
 * update_model evaluates the model for the current state (nullspaces out; SingularityHandler's classification state advances here like in
 * SingularityHandler::updateTaskModel, .cpp:227), compute_torques evaluates the control law (integrators, internal OTG, the type-2
 * direction memory advance here like in the reference's computeTorques) and returns THIS task's torques, without any of
 * RobotController's post-processing.  tau_prec = NULL is computeTorques(); otherwise the joint task subtracts its disturbance
 * compensation (JointTask.cpp:285-292) and the motion-force task adds nothing (its _Lambda is never written, MotionForceTask.cpp:140,273).
 * compute_torques after a state change without a new update_model is SAIP_ERR_ORDER (the reference would mix a fresh Jacobian with
 * stale Lambda / N_prec, MotionForceTask.cpp:280-283).  status: per-instance codes as for the whole cycle; where it is 1 the task's
 * torques follow saip_batch_set_flagged_torque_policy like the controller's (default: left as they were -- zero before the first valid
 * call --, 1: NaN): check the status, not isnan. */
saip_status saip_batch_task_update_model(saip_batch*, int task, const double* N_prec_host /* NULL = identity */);
saip_status saip_batch_task_update_model_device(saip_batch*, int task, const double* N_prec_dev /* NULL = identity; asynchronous */);
saip_status saip_batch_task_compute_torques(saip_batch*, int task, const double* tau_prec_host /*[dof][B] or NULL*/, double* tau_host /*[dof][B]*/, uint8_t* status_host /*[B] or NULL*/);
/* asynchronous flavour on the engine stream: tau_prec_dev [dof][ld] or NULL, tau_dev [dof][ld] or NULL (= the task's own buffer, saip_batch_task_device_torques) */
saip_status saip_batch_task_compute_torques_device(saip_batch*, int task, const double* tau_prec_dev, double* tau_dev);
/* getTaskNullspace (N), getPreviousTasksNullspace (N_prec), getTaskAndPreviousNullspace (N N_prec) of the last update_model; any pointer may be NULL */
saip_status saip_batch_task_get_nullspaces_host(saip_batch*, int task, double* N, double* N_prec, double* N_total);
/* the same on the device: which = 0 N, 1 N_prec (NULL when the identity was used), 2 N N_prec; NULL before the first update_model */
const double* saip_batch_task_device_nullspace(saip_batch*, int task, int which);
double* saip_batch_task_device_torques(saip_batch*, int task);
/* robot->setQ/setDq from arrays already on the device ([dof][ld], e.g. the state of another batch of the same robots): asynchronous D2D copy */
saip_status saip_batch_set_state_device(saip_batch*, const double* q_dev, const double* dq_dev);
/* device-side ordering between two batches' streams (no host synchronisation): everything enqueued so far on `producer` happens before
 * what is enqueued on `waiter` from now on.  Needed when one batch consumes device arrays another batch has just written (tasks driven
 * by hand that live in batches of their own). */
saip_status saip_batch_wait_for(saip_batch* waiter, saip_batch* producer);

/* kernel selection: 0 = auto, 1 = general workgroup-per-instance kernel, 2 = lane-per-instance register kernel (dof <= 8),
 * 3 = eight-lanes-per-instance kernel (7-dof chain, full MotionForceTask + full JointTask; the auto choice for batches <= 24576 of that stack, and at every batch size for its partial-task /
 *     reduced-task / joint-first / 6- and 8-dof instantiations),
 * 4 = wavefront-per-instance kernel on the FP64 matrix cores (chains of 9..32 dof; the auto choice there) */
saip_status saip_batch_set_kernel(saip_batch*, int which);
const char* saip_batch_kernel_name(saip_batch*);
/* HIP-event timing of `steps` back-to-back cycles on the engine stream (after `warmup` untimed ones):
 * total elapsed ms over the timed region.  Used by bench.py for the live roofline figure. */
saip_status saip_batch_time_steps(saip_batch*, int steps, int warmup, double* elapsed_ms);
/* the same without a wait inside: _begin records the first event, enqueues `steps` cycles, records the second event and returns; the caller waits
 * for the device in whatever way it waits anyway (bench.py: the torch.cuda.synchronize() its timing contract prescribes) and then asks _end for the
 * event time.  One wait instead of two: every runtime wait is its own marker round trip (4 - 15 us), which a 20-step timed region notices. */
saip_status saip_batch_time_steps_begin(saip_batch*, int steps);
saip_status saip_batch_time_steps_end(saip_batch*, double* elapsed_ms);

/* ---- multi-GPU (the reference is single-robot / single-thread, /root/reference/src/RobotController.cpp:68-118 has nothing to
 * shard): instances shard embarrassingly across the GPUs of a node, one batch per device, NO data-path collective; the only exchange is the
 * final gather of the [dof][ld] torque slabs, one ncclAllGather over xGMI.  C++ on RCCL, loaded on first use -- no PyTorch.
 * (a) one process per GPU: rank 0 draws a unique id, the 128 bytes reach the other ranks by the launcher's rendezvous, every rank creates its
 *     communicator; saip_batch_all_gather_torques enqueues the all-gather on the batch's stream: gathered_dev = [world][dof][ld]. */
#define SAIP_COMM_ID_BYTES 128
typedef struct saip_comm saip_comm;
/* can this process join a communicator on `device` (librccl loads, the device can be selected)?  Ask on EVERY rank and reduce the answers
 * before any rank calls saip_comm_create: a rank that fails locally would leave the others blocked inside ncclCommInitRank. */
saip_status saip_comm_probe(int device);
saip_status saip_comm_unique_id(void* id128);
saip_status saip_comm_create(int device, int world, int rank, const void* id128, saip_comm** out);
void saip_comm_destroy(saip_comm* comm);
int saip_comm_world(const saip_comm* comm); /* the number of ranks RCCL counts in the communicator (ncclCommCount) */
int saip_comm_rank(const saip_comm* comm);
/* every rank's batch must have the same dof and leading dimension (saip_batch_set_leading_dimension for uneven shards) */
saip_status saip_batch_all_gather_torques(saip_batch* batch, saip_comm* comm, double* gathered_dev);
/* HIP-event timing of `steps` cycles WITH the torque gather in the timed region: gather_mode 0 = one all-gather behind the last cycle (north_star:
 * "RCCL ... only for the final torque gather"), 1 = one behind EVERY cycle (what a consumer that takes the torques each control period sees,
 * /root/reference/examples/05-using_robot_controller/05-using_robot_controller.cpp:193-196, 225-231).  elapsed_ms = events around the whole region,
 * gather_ms = the sum over the gathers of the event time of each (so gather_ms / number of gathers is the cost of one).  comm == NULL (one
 * rank): no collective is issued and gather_ms = 0. */
saip_status saip_batch_time_steps_gather(saip_batch* batch, saip_comm* comm, double* gathered_dev, int steps, int gather_mode, double* elapsed_ms, double* gather_ms);
/* (b) one process, n devices: one batch + stream + communicator per device.  saip_multi_create makes the (unfinalized) batches -- add the same
 *     tasks to every saip_multi_batch(m, i) -- saip_multi_finalize finalizes them, runs ncclCommInitAll and allocates the gather buffers;
 *     saip_multi_step_async = saip_batch_step_async on every device, saip_multi_all_gather_torques = one grouped ncclAllGather, after which
 *     every device holds [n_devices][dof][ld] at saip_multi_gathered_device(m, i). */
typedef struct saip_multi saip_multi;
saip_status saip_multi_create(const saip_model* model, int batch_per_device, const int* devices, int n_devices, saip_multi** out);
int saip_multi_size(const saip_multi* m);
saip_batch* saip_multi_batch(saip_multi* m, int i);
saip_status saip_multi_finalize(saip_multi* m);
saip_status saip_multi_step_async(saip_multi* m);
saip_status saip_multi_all_gather_torques(saip_multi* m);
saip_status saip_multi_synchronize(saip_multi* m);
/* the same timing for the one-process form: every device runs `steps` cycles (+ the grouped all-gather, gather_mode as above); elapsed_ms / gather_ms =
 * the maximum over the devices of each device's event times */
saip_status saip_multi_time_steps(saip_multi* m, int steps, int gather_mode, double* elapsed_ms, double* gather_ms);
double* saip_multi_gathered_device(saip_multi* m, int i);
saip_status saip_multi_get_gathered_host(saip_multi* m, int i, double* out /*[n_devices][dof][B]*/);
void saip_multi_destroy(saip_multi* m);

const char* saip_last_error(void);
const char* saip_version(void);
int saip_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* SAIP_H_ */

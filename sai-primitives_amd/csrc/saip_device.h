// Device-visible constant blocks shared by the host engine and the HIP kernels.
// Layout notes: everything here is batch-uniform (wave-uniform), read through scalar loads.
#pragma once
#include <stdint.h>

#define SAIP_MAXN 32
#define SAIP_MAXT 8

namespace saip {

// everything the forward-kinematics loop needs about ONE joint/body, packed into four consecutive 64-byte lines so that the
// rolled loop of the lane kernel issues one burst of scalar loads per iteration (and a prewarm pass can touch all of it up front)
struct alignas(256) JointRec {
	double R0[9], p0[3], axis[3], com[3], inertia[6], mass;
	int jtype, axis_is_z, iso_inertia, pad_;
	double pad2_[5];
};
static_assert(sizeof(JointRec) == 256, "JointRec must be exactly four cache lines");

// one movable body of the serial chain (fixed links already merged on the host, engine.cpp: merge_fixed_links)
struct ModelDev {
	int n;
	int all_axis_z;              // every joint axis == (0,0,1) (the usual URDF convention): the eight-lane kernels build R0 Rz(q) from two columns
	int jtype[SAIP_MAXN];        // 1 revolute, 2 prismatic
	int axis_is_z[SAIP_MAXN];    // joint axis == (0,0,1) exactly: the joint rotation touches two columns only
	int iso_inertia[SAIP_MAXN];  // body inertia tensor = i * identity: invariant under rotation
	double R0[SAIP_MAXN][9];     // joint origin rotation (parent body frame -> joint frame at q=0), row-major
	double p0[SAIP_MAXN][3];     // joint origin translation in the parent body frame
	double axis[SAIP_MAXN][3];   // unit joint axis in the body frame
	double mass[SAIP_MAXN];
	double com[SAIP_MAXN][3];    // body frame
	double inertia[SAIP_MAXN][6];// about the COM, body axes: xx yy zz xy xz yz
	double effort[SAIP_MAXN];
	double q_lower[SAIP_MAXN], q_upper[SAIP_MAXN], vel_limit[SAIP_MAXN];  // jointLimits(), used by the joint-limit-avoidance wrap
	double gravity[3];
	double pad_align_[5];
	JointRec jrec[SAIP_MAXN];    // packed copy of the per-joint constants above (lane kernel FK loop)
};

enum { TASK_JOINT = 2, TASK_MOTION_FORCE = 3 };

// SingularityHandler state of one instance of one motion-force task with the blended strategies on (SingularityHandler.h:226-243):
// entering posture, type-2 torque direction (bit i set = -1), type counters and their 200-cycle window.  Zero-initialised.
enum { SH_BUFFER = 200 };
struct ShState {
	double q_prior[SAIP_MAXN];
	unsigned dir_neg;
	int ntypes, c1, c2, head, size;
	int last_cycle;    // cycle number (CycleParams::task_cycle) of the last cycle this instance had singular directions: the counters and the
	                   // window above are void unless that was the previous cycle (SingularityHandler.cpp:238-244 clears them whenever a cycle
	                   // finds none) -- instances outside the singular region never have to touch their state
	unsigned char hist[SH_BUFFER];
};

enum { DEC_FULL = 0, DEC_BIE = 1, DEC_IMPEDANCE = 2 };

struct TaskDev {
	int type;
	int body;          // motion force: movable body the control frame is attached to
	int k;             // motion force: task rank = pos_range + ori_range
	int m;             // joint: rows of S
	int decoupling;
	int s_identity;    // joint: S == I
	int has_ki;        // any integral gain != 0
	int goal_comps;
	int bm_identity;   // motion force: Bm == I6 (full task)
	int sing_handling; // motion force: SingularityHandler::_enforce_handling_strategy (1 = reference default)
	int sing_strategies; // motion force: blended type-1 / type-2 strategies on the device (general kernel) instead of flagging
	int sh_force_type1;  // handleAllSingularitiesAsType1 (SingularityHandler.h:131)
	double dt, bie_threshold, s_min, s_max, s_abs_tol;
	double sh_kp1, sh_kv1, sh_kv2;  // setSingularityHandlingGains (SingularityHandler.h:119; defaults 50, 14, 5)
	double cert_kroot; // k^(-1/8): lower/upper bound ratio of the trace-power bound on lambda_max (singular-value certificate)
	double pos[3];     // control point in the body frame
	double rot[9];     // control frame rotation in the body frame
	double Bm[36];     // 6 x 6 row-major, first k columns = orthonormal basis of range(P)
	double Ppos[9], Pori[9];   // diagonal blocks of the partial task projection P
	double sig_p[9], sig_o[9]; // sigmaPosition / sigmaOrientation (force & moment space dimension 0)
	double kp_pos[3], kv_pos[3], ki_pos[3], kp_ori[3], kv_ori[3], ki_ori[3];
	double kp[SAIP_MAXN], kv[SAIP_MAXN], ki[SAIP_MAXN];
	// velocity saturation, force / moment spaces (open-loop force control): MotionForceTask.cpp:306-383, 416-430, 449-462, 892-971; JointTask.cpp:327-341
	int general_law;   // 0: defaults (no velocity saturation, force and moment space dimension 0) -> short control-law path
	int vel_sat, force_dim, moment_dim, param_in_compliant_frame;
	int law_identity;  // motion force: Bm, Ppos, Pori, sigmaPosition, sigmaOrientation are all EXACTLY the identity and the short control law applies (the
	                   // full 6-dof task without force / moment space: the reference default) -- set at upload; the wavefront kernel skips the products
	int pad1_[2];
	double lin_sat, ang_sat, kv_force, kv_moment;
	double force_axis[3], moment_axis[3];      // unit axes
	// closed-loop force / moment control (MotionForceTask.cpp:327-349, 357-378, 484-487), passivity controller off (POPC default)
	int cl_force, cl_moment;
	int popc_enabled, popc_cap;                // POPCExplicitForceControl around the force loop; capacity of the energy window ring
	double kp_force, ki_force, kp_moment, ki_moment, kff_force, kff_moment, max_force_fb, max_moment_fb;
	double Rcs[9], tcs[3];                     // _T_control_to_sensor (MotionForceTask.cpp:94, 802)
	double kvinv_pos[3], kvinv_ori[3];         // pseudo-inverse of the diagonal kv gains (computePseudoInverse)
	double kvinv[SAIP_MAXN], sat[SAIP_MAXN];   // joint task
	double S[SAIP_MAXN * SAIP_MAXN]; // m x n row-major
	double* goal;      // [goal_comps][ld]
	double* integ;     // [12 or m][ld]
	double* integ_new; // same shape: what this cycle's control law advanced the integrators to; committed to `integ` at the end of the
	                   // cycle unless the instance ends flagged (status 1) -- a flagged instance advances nothing, whichever kernel ran
	double* diag_N;    // optional [n*n][ld] task nullspace dump (diagnostic variant), else nullptr
	ShState* sh;       // [B] handler states when sing_strategies, else nullptr
	double* popc;      // [7 + popc_cap][ld] passivity observer state: PO, E_correction, Rc, sum |vcl|^2, counter, ring head, ring size, ring
};

struct CycleParams {
	int B, ld, n, nt;
	// rollouts without internal OTGs: the eight-lane kernel integrates the state under the torques it has just written (0 = off)
	int sim_substeps;
	double sim_dt, sim_damping, sim_gravity[3];
	int gravity_comp, torque_sat, integ_always, diag;
	int jla;           // RobotController::enableJointLimitAvoidance
	int only_flagged;  // general kernel as the slow path behind the lane kernel: workgroups whose instance has status 0 exit at once
	const double* q;   // [n][ld]
	const double* dq;  // [n][ld]
	double* tau;       // [n][ld]
	uint8_t* status;   // [ld]
	const ModelDev* model;
	const TaskDev* tasks;
	// lane-per-instance kernel only
	int mf_task[2];               // task ids of the (at most two) motion-force tasks, -1 when absent
	int rank_bound[SAIP_MAXT];    // batch-uniform upper bound on the rank of each task in the hierarchy
	int lane_task_is_joint[SAIP_MAXT];
	int lane_prefetch_ok;         // stack is {motion-force, full joint task}: the prefetch layout of the latency variant applies
	int lane_active;              // instances per wavefront in the lane kernel (64 = every lane)
	int lane_general;             // some task needs the general control laws (velocity saturation / force or moment space)
	int oct_general_joint;        // eight-lane kernel, joint task: 0 rank <= 1 behind a 6-dof task, 1 Gram path (<= 4 rows), 2 full task behind a partial one
	int reinit_task, reinit_mask; // reinit kernel: task (-1 = all) and parts (1 linear/joint, 2 angular, 4 goal force+moment)
	int any_bie;
	int oct_partial_mf;           // eight-lane kernel: the motion-force task is partial (k < 6)
	int oct_truncate;             // eight-lane kernel: singularity handling disabled -> near-singular instances are reduced in the kernel
	int wave_general_joint;       // wavefront kernel: some joint task needs a range basis (anything but one full joint task closing the hierarchy)
	double bie_thr;
	// per-task entry points of the general kernel (TemplateTask::updateTaskModel(N_prec) / computeTorques() / computeTorques(tau_prec),
	// TemplateTask.h:43-60): evaluate ONE task with a caller-supplied N_prec instead of walking the hierarchy
	int single_task;              // -1: the whole hierarchy (RobotController); >= 0: this task only, no RobotController post-processing
	int phase;                    // 0: model + torques fused; 1: model only (updateTaskModel: nullspaces out, handler classification state
	                              // advanced, no control law); 2: torques only (computeTorques: control law, integrators, no classification)
	const double* ext_nprec;      // [n*n][ld] row-major per instance, nullptr = identity
	const double* ext_tau_prec;   // [n][ld] torques of the previous tasks (disturbance compensation), nullptr = none
	double* out_N;                // [n*n][ld] task nullspace N (getTaskNullspace), optional
	double* out_Ntot;             // [n*n][ld] N N_prec (getTaskAndPreviousNullspace), optional
	int task_cycle[SAIP_MAXT];    // per task: how many times its model has been updated (SingularityHandler state: see ShState::last_cycle)
	int flag_nan;                 // torques of an instance that ends flagged (status 1): 0 = left as they were (the last valid torque is held), 1 = NaN
	// instances the lane / eight-lane kernel could not finish (status 1) are appended here; the general kernel launched behind as the
	// device-side slow path (a fixed grid striding over the list) recomputes them.  Two count / list pairs used alternately: the lane /
	// eight-lane kernel of a cycle appends to `flag_count` / `flag_list` and zeroes `flag_count_next` (whose readers -- the slow path of the
	// previous cycle -- are done by stream order), so an empty list costs the slow path one load per workgroup and no atomic at all
	int* flag_count;
	int* flag_list;
	int* flag_count_next;
	// eight-lane kernels: the wavefront that could not finish an instance recomputes it itself, right behind its epilogue, with the general
	// kernel's body on its own (by then dead) LDS block -- no list, no second launch (round 4; 0: flagged instances go to the list / stay flagged)
	int slow_tail;
	// eight-lane kernels: TaskDev::goal / ::integ of the stack's two tasks once more, as kernel arguments.  Read from the task block they are
	// a pointer chase at the very start of every wavefront -- kernel arguments, then the pointer, then the data: one memory round trip more
	// in front of the goals than in front of anything else the wavefront asks for.
	const double* goal2[2];
	double* integ2[2];
};

// forward dynamics + semi-implicit Euler step (saip_dynamics.hip)
struct SimParams {
	int B, ld, n, substeps;
	double dt, damping;    // viscous joint damping coefficient (0 = none)
	double gravity[3];
	const ModelDev* model;
	double* q;             // [n][ld] in/out
	double* dq;            // [n][ld] in/out
	const double* tau;     // [n][ld] joint torques held over the call (NaN = coast)
	double* ddq;           // optional [n][ld] joint accelerations of the last substep
};

// Internal OTG of one joint task (saip_otg.hip).  Passed to the kernel by value.
enum { OTG_FLAG_GOAL_REACHED = 1, OTG_FLAG_RUCKIG_INIT = 2, OTG_FLAG_STATE_VALID = 4, OTG_FLAG_ERROR = 8, OTG_FLAG_FINISHED_MOVING = 16,
	   OTG_FLAG_GOAL_ORI_VALID = 32, OTG_FLAG_INVALID_GOAL = 64 };
struct OtgDev {
	int m, gs;             // task dof; lanes per robot (8 or 32)
	int epoch, jerk;       // epoch: bumped when the limits change (forces a new trajectory like InputParameter::operator!= does); jerk: 1 =
	                       // jerk-limited mode (third-order profiles, OTG_joints::setMaxJerk), 0 = acceleration-limited (the reference default)
	long long lanes;       // B * gs = stride of one state field
	double dt;             // control period (Ruckig delta_time)
	const double* limits;  // [3][SAIP_MAXN] device copy of the limits: max velocity, max acceleration, max jerk per task dof
	double* state;         // [fields][lanes], lane-major
	double* time;          // [ld] OutputParameter::time
	double* duration;      // [ld] Trajectory::duration
	int* flags;            // [ld] OTG_FLAG_*
	int* seen_epoch;       // [ld]
	int* result;           // [ld] ruckig::Result of the last cycle
	const double* goal;    // [3m][ld] user goal (position, velocity, acceleration rows)
	double* desired;       // [3m][ld] desired state fed to the control law
	const double* q;       // [n][ld] (re-initialisation only)
	const TaskDev* task;   // selection matrix S / control frame (re-initialisation only)
	int n, goal_comps;
	// Cartesian OTG of a motion-force task (OTG_6dof_cartesian): per-instance frames [21][ld] = _reference_frame (9),
	// _goal_orientation_in_base_frame (9), _goal_angular_velocity_in_base_frame (3)
	double* frame;
	const ModelDev* model;
};

}  // namespace saip

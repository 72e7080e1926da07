// Host side of the engine behind include/saip.h: robot constants, task stacks, device arena, launches.
// C++ (the reference is a C++ library), HIP runtime only -- no PyTorch, no Eigen.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/saip.h"
#include "saip_device.h"

namespace saip {
hipError_t launch_cycle_wg(const CycleParams& P, hipStream_t stream);
hipError_t launch_cycle_wg_list(const CycleParams& P, hipStream_t stream);
hipError_t launch_reinit(const CycleParams& P, hipStream_t stream);
hipError_t launch_cycle_lane(const CycleParams& P, hipStream_t stream, bool* supported);
hipError_t launch_cycle_oct(const CycleParams& P, hipStream_t stream);
hipError_t launch_cycle_wave(const CycleParams& P, hipStream_t stream);
hipError_t launch_cycle_octjf(const CycleParams& P, hipStream_t stream);
hipError_t launch_pose(const CycleParams& P, int task, double* out, hipStream_t stream);
hipError_t launch_otg_joints(const OtgDev& O, int B, int ld, int mode, hipStream_t stream);
hipError_t launch_otg_cartesian(const OtgDev& O, int B, int ld, int mode, hipStream_t stream);
hipError_t launch_otg_pair(const OtgDev& Oc, const OtgDev& Oj, int B, int ld, hipStream_t stream);
hipError_t launch_integrate_otg_pair(const SimParams& S, const OtgDev& Oc, const OtgDev& Oj, int B, int ld, hipStream_t stream);
int otg_state_fields();
hipError_t launch_integrate(const SimParams& S, hipStream_t stream);
}  // namespace saip

using saip::CycleParams;
using saip::ModelDev;
using saip::OtgDev;
using saip::SimParams;
using saip::TaskDev;

static thread_local std::string g_err;
static saip_status fail(saip_status st, const char* fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_err = buf;
	return st;
}
namespace saip {
saip_status fail_external(saip_status st, const char* fmt, ...) {  // for the other translation units of the library (saip_comm.cpp)
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_err = buf;
	return st;
}
}  // namespace saip
#define HIP_TRY(expr)                                                                                        \
	do {                                                                                                     \
		hipError_t e_ = (expr);                                                                              \
		if (e_ != hipSuccess) return fail(SAIP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
	} while (0)

// ------------------------------------------------------------------ tiny 3x3 helpers (row-major)
static void m3_mul(const double* A, const double* B, double* C) {
	double T[9];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
	memcpy(C, T, sizeof(T));
}
static void m3_vec(const double* A, const double* v, double* o) {
	double t[3];
	for (int i = 0; i < 3; i++) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
	memcpy(o, t, sizeof(t));
}
static void m3_T(const double* A, double* B) {
	double T[9];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * j + i];
	memcpy(B, T, sizeof(T));
}
static void m3_eye(double* A) {
	memset(A, 0, 9 * sizeof(double));
	A[0] = A[4] = A[8] = 1.0;
}
static void rpy_to_R(const double* rpy, double* R) {  // URDF fixed-axis rpy: R = Rz(yaw) Ry(pitch) Rx(roll)
	double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
	double Rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr}, Ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp}, Rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1}, T[9];
	m3_mul(Ry, Rx, T);
	m3_mul(Rz, T, R);
}
// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi); eigenvalues descending, eigenvectors in columns of V
static void sym3_eig(const double* A_in, double* lam, double* V) {
	double A[9];
	memcpy(A, A_in, sizeof(A));
	m3_eye(V);
	for (int sweep = 0; sweep < 50; sweep++) {
		double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
		if (off < 1e-300) break;
		for (int p = 0; p < 2; p++)
			for (int q = p + 1; q < 3; q++) {
				double apq = A[3 * p + q];
				if (fabs(apq) < 1e-300) continue;
				double theta = (A[3 * q + q] - A[3 * p + p]) / (2 * apq);
				double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
				double c = 1 / sqrt(t * t + 1), s = t * c;
				for (int i = 0; i < 3; i++) {
					double a = A[3 * i + p], b = A[3 * i + q];
					A[3 * i + p] = c * a - s * b;
					A[3 * i + q] = s * a + c * b;
				}
				for (int j = 0; j < 3; j++) {
					double a = A[3 * p + j], b = A[3 * q + j];
					A[3 * p + j] = c * a - s * b;
					A[3 * q + j] = s * a + c * b;
				}
				for (int i = 0; i < 3; i++) {
					double a = V[3 * i + p], b = V[3 * i + q];
					V[3 * i + p] = c * a - s * b;
					V[3 * i + q] = s * a + c * b;
				}
			}
	}
	int idx[3] = {0, 1, 2};
	for (int a = 0; a < 3; a++)
		for (int b = a + 1; b < 3; b++)
			if (A[4 * idx[b]] > A[4 * idx[a]]) std::swap(idx[a], idx[b]);
	double Vs[9];
	for (int j = 0; j < 3; j++) {
		lam[j] = A[4 * idx[j]];
		for (int i = 0; i < 3; i++) Vs[3 * i + j] = V[3 * i + idx[j]];
	}
	memcpy(V, Vs, sizeof(Vs));
}
// SaiModel::matrixRangeBasis for a 3 x cnt matrix whose columns are `dirs` (cnt vectors of 3): orthonormal basis of
// the column space with the reference's tolerance semantics (sigma_i/sigma_0 >= 1e-3; identity when rank 3).
// Returns the rank (0 = empty range); basis (3 x rank) row-major with leading dimension 3.
static int range_basis_3(const double* dirs, int cnt, double* basis) {
	double G[9] = {0};
	for (int c = 0; c < cnt; c++)
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) G[3 * i + j] += dirs[3 * c + i] * dirs[3 * c + j];
	const double tol = 1e-3;
	memset(basis, 0, 9 * sizeof(double));
	if (cnt <= 0 || sqrt(G[0] + G[4] + G[8]) < tol) return 0;
	double lam[3], V[9];
	sym3_eig(G, lam, V);
	double s0 = sqrt(fmax(lam[0], 0.0));
	if (s0 < tol) return 0;
	int maxr = cnt < 3 ? cnt : 3, rank = maxr;
	for (int i = maxr - 1; i > 0; i--) {
		if (sqrt(fmax(lam[i], 0.0)) / s0 < tol) rank--;
		else break;
	}
	if (rank == 3) {
		m3_eye(basis);
		return 3;
	}
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < rank; j++) basis[3 * i + j] = V[3 * i + j];
	return rank;
}

// ------------------------------------------------------------------ model
struct LinkInfo {
	std::string name;
	int body;        // movable body this link is rigidly attached to (-1: attached to the fixed base)
	double R[9], p[3];  // link frame in the body frame
};
struct saip_model {
	int n = 0;
	std::vector<LinkInfo> links;
	ModelDev dev;
	double q_lower[SAIP_MAXN], q_upper[SAIP_MAXN], vel[SAIP_MAXN], effort[SAIP_MAXN];
};

// combine rigid-body inertials expressed in one frame
struct Inertial {
	double m = 0, c[3] = {0, 0, 0}, I[9] = {0};  // I about the COM
};
static void inertial_add(Inertial& a, double m2, const double* c2, const double* I2) {
	double m = a.m + m2;
	if (m <= 0) return;
	double c[3];
	for (int i = 0; i < 3; i++) c[i] = (a.m * a.c[i] + m2 * c2[i]) / m;
	double I[9] = {0};
	auto shift = [&](double mm, const double* cc, const double* II) {
		double d[3] = {cc[0] - c[0], cc[1] - c[1], cc[2] - c[2]}, dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) I[3 * i + j] += II[3 * i + j] + mm * ((i == j ? dd : 0.0) - d[i] * d[j]);
	};
	shift(a.m, a.c, a.I);
	shift(m2, c2, I2);
	a.m = m;
	memcpy(a.c, c, sizeof(c));
	memcpy(a.I, I, sizeof(I));
}

extern "C" saip_status saip_model_create_serial_chain(const saip_link_desc* links, int n_links, saip_model** out) {
	if (!links || !out || n_links <= 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_model_create_serial_chain: null or empty link list");
	auto* M = new saip_model();
	memset(&M->dev, 0, sizeof(ModelDev));
	// pending fixed transform between the last movable body frame (or the base) and the current link
	double Rp[9], pp[3] = {0, 0, 0};
	m3_eye(Rp);
	int body = -1;
	std::vector<Inertial> inertials;
	for (int l = 0; l < n_links; l++) {
		const saip_link_desc& L = links[l];
		double R0[9], Rl[9], pl[3], t[3];
		rpy_to_R(L.origin_rpy, R0);
		m3_vec(Rp, L.origin_xyz, t);
		for (int i = 0; i < 3; i++) pl[i] = pp[i] + t[i];
		m3_mul(Rp, R0, Rl);  // link frame (at q = 0) in the frame of the last movable body
		double Il[9] = {L.inertia[0], L.inertia[3], L.inertia[4], L.inertia[3], L.inertia[1], L.inertia[5], L.inertia[4], L.inertia[5], L.inertia[2]};
		if (L.joint_type == SAIP_JOINT_FIXED) {
			if (body >= 0) {  // merge the inertial into the parent movable body (links welded to the base carry no dynamics)
				double c2[3], T[9], I2[9], RlT[9];
				m3_vec(Rl, L.com, c2);
				for (int i = 0; i < 3; i++) c2[i] += pl[i];
				m3_mul(Rl, Il, T);
				m3_T(Rl, RlT);
				m3_mul(T, RlT, I2);
				inertial_add(inertials[body], L.mass, c2, I2);
			}
			memcpy(Rp, Rl, sizeof(Rl));
			memcpy(pp, pl, sizeof(pl));
		} else if (L.joint_type == SAIP_JOINT_REVOLUTE || L.joint_type == SAIP_JOINT_PRISMATIC) {
			if (M->n >= SAIP_MAXN) {
				delete M;
				return fail(SAIP_ERR_UNSUPPORTED, "robot has more than %d degrees of freedom", SAIP_MAXN);
			}
			double an = sqrt(L.axis[0] * L.axis[0] + L.axis[1] * L.axis[1] + L.axis[2] * L.axis[2]);
			if (an < 1e-12) {
				delete M;
				return fail(SAIP_ERR_INVALID_ARGUMENT, "joint of link %s has a zero axis", L.name);
			}
			int j = M->n++;
			body = j;
			M->dev.jtype[j] = L.joint_type;
			memcpy(M->dev.R0[j], Rl, sizeof(Rl));
			memcpy(M->dev.p0[j], pl, sizeof(pl));
			for (int i = 0; i < 3; i++) M->dev.axis[j][i] = L.axis[i] / an;
			M->dev.axis_is_z[j] = (M->dev.axis[j][0] == 0.0 && M->dev.axis[j][1] == 0.0 && M->dev.axis[j][2] == 1.0) ? 1 : 0;
			Inertial in;
			inertial_add(in, L.mass, L.com, Il);
			if (L.mass <= 0) memcpy(in.c, L.com, sizeof(in.c));
			inertials.push_back(in);
			M->q_lower[j] = L.q_lower;
			M->q_upper[j] = L.q_upper;
			M->vel[j] = L.velocity_limit;
			M->effort[j] = L.effort_limit;
			M->dev.effort[j] = L.effort_limit;
			M->dev.q_lower[j] = L.q_lower;
			M->dev.q_upper[j] = L.q_upper;
			M->dev.vel_limit[j] = L.velocity_limit;
			m3_eye(Rp);
			pp[0] = pp[1] = pp[2] = 0;
		} else {
			delete M;
			return fail(SAIP_ERR_INVALID_ARGUMENT, "link %s: unknown joint type %d", L.name, L.joint_type);
		}
		LinkInfo li;
		li.name = std::string(L.name, strnlen(L.name, SAIP_NAME_LEN));
		li.body = body;
		memcpy(li.R, Rp, sizeof(Rp));
		memcpy(li.p, pp, sizeof(pp));
		M->links.push_back(li);
	}
	if (M->n == 0) {
		delete M;
		return fail(SAIP_ERR_INVALID_ARGUMENT, "robot has no movable joint");
	}
	M->dev.n = M->n;
	for (int j = 0; j < M->n; j++) {
		const Inertial& in = inertials[j];
		M->dev.mass[j] = in.m;
		memcpy(M->dev.com[j], in.c, sizeof(in.c));
		M->dev.inertia[j][0] = in.I[0];
		M->dev.inertia[j][1] = in.I[4];
		M->dev.inertia[j][2] = in.I[8];
		M->dev.inertia[j][3] = in.I[1];
		M->dev.inertia[j][4] = in.I[2];
		M->dev.inertia[j][5] = in.I[5];
		M->dev.iso_inertia[j] = (in.I[0] == in.I[4] && in.I[0] == in.I[8] && in.I[1] == 0.0 && in.I[2] == 0.0 && in.I[5] == 0.0) ? 1 : 0;
	}
	M->dev.gravity[0] = 0;
	M->dev.gravity[1] = 0;
	M->dev.gravity[2] = -9.81;
	for (int j = 0; j < M->n; j++) {  // packed per-joint records
		saip::JointRec& r = M->dev.jrec[j];
		memcpy(r.R0, M->dev.R0[j], sizeof(r.R0));
		memcpy(r.p0, M->dev.p0[j], sizeof(r.p0));
		memcpy(r.axis, M->dev.axis[j], sizeof(r.axis));
		memcpy(r.com, M->dev.com[j], sizeof(r.com));
		memcpy(r.inertia, M->dev.inertia[j], sizeof(r.inertia));
		r.mass = M->dev.mass[j];
		r.jtype = M->dev.jtype[j];
		r.axis_is_z = M->dev.axis_is_z[j];
		r.iso_inertia = M->dev.iso_inertia[j];
	}
	M->dev.all_axis_z = 1;
	for (int j = 0; j < M->n; j++)
		if (!M->dev.axis_is_z[j]) M->dev.all_axis_z = 0;
	*out = M;
	return SAIP_OK;
}
extern "C" void saip_model_destroy(saip_model* m) { delete m; }
extern "C" int saip_model_dof(const saip_model* m) { return m ? m->n : 0; }
extern "C" int saip_model_link_index(const saip_model* m, const char* name) {
	if (!m || !name) return -1;
	for (size_t i = 0; i < m->links.size(); i++)
		if (m->links[i].name == name) return (int)i;
	return -1;
}
extern "C" saip_status saip_model_joint_limits(const saip_model* m, double* lo, double* hi, double* vel, double* eff) {
	if (!m) return fail(SAIP_ERR_INVALID_ARGUMENT, "null model");
	for (int j = 0; j < m->n; j++) {
		if (lo) lo[j] = m->q_lower[j];
		if (hi) hi[j] = m->q_upper[j];
		if (vel) vel[j] = m->vel[j];
		if (eff) eff[j] = m->effort[j];
	}
	return SAIP_OK;
}

// ------------------------------------------------------------------ batch
struct TaskHost {
	std::string name;
	TaskDev dev;
	double P[36];
	bool otg_enabled = true;  // reference default (MotionForceTask.h:67, JointTask.h:38)
	// internal OTG of a joint task (saip_otg.hip): acceleration-limited, defaults JointTask.h:39-41
	bool otg_alloc = false, otg_inited = false, otg_limits_dirty = true;
	OtgDev otg;
	double otg_limits[3 * SAIP_MAXN];  // max velocity, max acceleration, max jerk per task dof (the jerk row only in jerk-limited mode)
	double* otg_limits_dev = nullptr;
	double* desired_dev = nullptr;
	bool vel_sat = false;
	bool full_joint = false;
	double* goal_dev = nullptr;
	double* integ_dev = nullptr;
	double* integ_new_dev = nullptr;
	double* diag_dev = nullptr;
	int integ_rows = 0;
	// per-task entry points (TemplateTask::updateTaskModel(N_prec) / computeTorques): the N_prec the task was last updated with, its
	// nullspaces N and N N_prec, its own torques and status; allocated on first use
	double *nprec_dev = nullptr, *ntask_dev = nullptr, *ntot_dev = nullptr, *ttau_dev = nullptr, *tprec_dev = nullptr;
	uint8_t* tstatus_dev = nullptr;
	bool nprec_identity = true;
	long model_epoch = -1;  // state epoch of the last updateTaskModel (-1: never)
	int sh_cycle = 0;       // how many times this task's model has been updated (CycleParams::task_cycle; ShState::last_cycle)
};
struct saip_batch {
	const saip_model* model = nullptr;
	int B = 0, ld = 0, device = -1;
	bool finalized = false, models_valid = false, config_dirty = true, state_pushed = false;
	bool gravity_comp = false, torque_sat = false, integ_always = false, jla = false;
	int kernel_choice = 0;
	std::string kernel_name = "none";
	std::vector<TaskHost> tasks;
	hipStream_t stream = nullptr;
	double *q = nullptr, *dq = nullptr, *tau = nullptr, *tau_bound = nullptr;
	uint8_t* status = nullptr;
	ModelDev* model_dev = nullptr;
	TaskDev* tasks_dev = nullptr;
	std::vector<void*> allocs;
	double* pose_dev = nullptr;              // [12][ld] scratch of saip_batch_get_current_pose_host
	hipGraph_t rollout_graph = nullptr;      // one captured closed-loop period (saip_batch_rollout_async)
	hipGraphExec_t rollout_exec = nullptr;
	long state_epoch = 0;                    // bumped whenever the resident state changes (per-task models become stale)
	double* diag_tau = nullptr;              // scratch torques / status of diagnostic launches (the last cycle's results stay intact)
	uint8_t* diag_status = nullptr;
	hipEvent_t sync_event = nullptr;         // saip_batch_wait_for
	hipEvent_t time_ev[2] = {nullptr, nullptr};  // saip_batch_time_steps (created once: event creation is not part of a timed region)
	bool flag_nan = false;                   // saip_batch_set_flagged_torque_policy
	bool flagged_on_list = false;            // saip_batch_set_flagged_recompute: eight-lane kernels hand flagged instances to the list launch instead of their slow tail
	int* flag_buf = nullptr;                 // [2 * (32 + ld)]: two { count, list } pairs of the device-side slow path, used alternately (CycleParams::flag_*)
	unsigned flag_parity = 0;                // advanced only by a launch that hands the pair to a kernel
	bool flag_clean[2] = {true, true};       // whether each pair's count is known to be zero (the buffer is zeroed at allocation)
	bool otg_prelaunched = false;            // rollouts: the paired OTG step of the coming cycle already ran, fused with the previous integrate
	int sim_want = 0;                        // rollouts: substeps the next cycle launch should integrate itself if it can (0 = no) ...
	double sim_dt = 0, sim_damping = 0, sim_gravity[3] = {0, 0, 0};
	bool sim_done = false;                   // ... and whether it did (eight-lane kernel, headline-type stack, no slow path behind)
};

// Waiting for the batch's stream.  hipStreamSynchronize parks the thread on an interrupt (4 - 5 us from the end of the last kernel to the
// return when the thread is the only waiter); polling hipStreamQuery returns within ~1 us.  Opt-in (SAIP_SPIN_WAIT_US = how long to poll before
// blocking; default 0 = block at once): a loop that waits once per cycle and then reads the torques from a bound device buffer gains the
// difference, but a caller that follows up with ANOTHER runtime wait (hipDeviceSynchronize, torch.cuda.synchronize) pays for that one's own
// marker round trip (~19 us instead of ~4 after a blocking wait) -- measured with bench.py at 20 steps: 10.2 us per step polling, 9.7 blocking.
static hipError_t wait_stream(hipStream_t stream) {
	static const long spin_us = [] {
		const char* e = getenv("SAIP_SPIN_WAIT_US");
		return e ? atol(e) : 0L;
	}();
	if (spin_us > 0) {
		const auto t0 = std::chrono::steady_clock::now();
		for (int it = 0;; it++) {
			const hipError_t e = hipStreamQuery(stream);
			if (e != hipErrorNotReady) return e;
			if ((it & 15) == 15 && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
		}
	}
	return hipStreamSynchronize(stream);
}

static bool has_device(const saip_batch* b) { return b->device >= 0; }
static saip_status need_ready(saip_batch* b, const char* fn);

extern "C" int saip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" saip_status saip_batch_create(const saip_model* model, int batch_size, int device, saip_batch** out) {
	if (!model || !out) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_create: null argument");
	if (batch_size <= 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_create: batch size must be positive");
	if (device >= 0) {
		int cnt = saip_device_count();
		if (cnt <= 0) return fail(SAIP_ERR_NO_DEVICE, "no HIP device available: the engine has no CPU path");
		if (device >= cnt) return fail(SAIP_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device, cnt);
	}
	auto* b = new saip_batch();
	b->model = model;
	b->B = batch_size;
	b->ld = (batch_size + 31) / 32 * 32;
	b->device = device;  // < 0: configuration-only batch (host logic can be exercised; every compute call fails)
	*out = b;
	return SAIP_OK;
}
// leading dimension of the device arrays (default: B rounded up to 32).  Shards of a sharded run that differ by an instance all take the
// ld of the largest one, so that the final all-gather moves slabs of one shape (sharding.shard_ld)
extern "C" saip_status saip_batch_set_leading_dimension(saip_batch* b, int ld) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "null batch");
	if (b->finalized) return fail(SAIP_ERR_ORDER, "the leading dimension is fixed by saip_batch_finalize");
	if (ld < b->B || ld % 32 != 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "leading dimension %d: must be a multiple of 32 and at least the batch size %d", ld, b->B);
	b->ld = ld;
	return SAIP_OK;
}
extern "C" void saip_batch_destroy(saip_batch* b) {
	if (!b) return;
	if (has_device(b)) {
		(void)hipSetDevice(b->device);
		if (b->stream) (void)hipStreamSynchronize(b->stream);  // nothing of this batch may still be in flight
		if (b->sync_event) (void)hipEventDestroy(b->sync_event);
		for (hipEvent_t e : b->time_ev)
			if (e) (void)hipEventDestroy(e);
		if (b->rollout_exec) (void)hipGraphExecDestroy(b->rollout_exec);
		if (b->rollout_graph) (void)hipGraphDestroy(b->rollout_graph);
		for (void* p : b->allocs) (void)hipFree(p);
		if (b->stream) (void)hipStreamDestroy(b->stream);
	}
	delete b;
}
extern "C" int saip_batch_size(const saip_batch* b) { return b ? b->B : 0; }
extern "C" int saip_batch_ld(const saip_batch* b) { return b ? b->ld : 0; }
extern "C" int saip_batch_dof(const saip_batch* b) { return (b && b->model) ? b->model->n : 0; }

static saip_status check_batch(const saip_batch* b, int task, const char* fn) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "%s: null batch", fn);
	if (task >= (int)b->tasks.size() || task < -1) return fail(SAIP_ERR_INVALID_ARGUMENT, "%s: task id %d out of range", fn, task);
	return SAIP_OK;
}
static void refresh_has_ki(TaskDev& d);
static void task_defaults(TaskDev& d, double dt) {
	memset(&d, 0, sizeof(TaskDev));
	d.dt = dt;
	d.decoupling = SAIP_BOUNDED_INERTIA_ESTIMATES;  // MotionForceTask.h:41-43, JointTask.h:35-37
	d.bie_threshold = 0.1;
}

extern "C" saip_status saip_batch_add_motion_force_task(saip_batch* b, const char* task_name, const char* link_name,
														const double pos_in_link[3], const double rot_in_link[9],
														const double* dirs_trans, int n_trans, const double* dirs_rot, int n_rot,
														double loop_timestep, int* task_id) {
	if (!b || !task_name || !link_name || !pos_in_link) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_add_motion_force_task: null argument");
	if (b->finalized) return fail(SAIP_ERR_ORDER, "tasks cannot be added after saip_batch_finalize");
	if ((int)b->tasks.size() >= SAIP_MAX_TASKS) return fail(SAIP_ERR_UNSUPPORTED, "more than %d tasks", SAIP_MAX_TASKS);
	int li = saip_model_link_index(b->model, link_name);
	if (li < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "link %s does not exist in the robot model", link_name);
	const LinkInfo& L = b->model->links[li];
	if (L.body < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "link %s is welded to the base: no controllable degree of freedom", link_name);
	TaskHost T;
	T.name = task_name;
	task_defaults(T.dev, loop_timestep);
	T.dev.type = saip::TASK_MOTION_FORCE;
	T.dev.body = L.body;
	double I3[9];
	m3_eye(I3);
	const double* Rin = rot_in_link ? rot_in_link : I3;
	double t[3];
	m3_vec(L.R, pos_in_link, t);
	for (int i = 0; i < 3; i++) T.dev.pos[i] = L.p[i] + t[i];
	m3_mul(L.R, Rin, T.dev.rot);
	double bt[9], br[9];
	int pr, orr;
	if (n_trans < 0 && n_rot < 0) {  // full task, MotionForceTask.cpp:28
		m3_eye(bt);
		m3_eye(br);
		pr = orr = 3;
	} else {
		if (n_trans <= 0 && n_rot <= 0)  // MotionForceTask.cpp:47-53
			return fail(SAIP_ERR_INVALID_ARGUMENT, "controlled_directions_translation and controlled_directions_rotation cannot both be empty in MotionForceTask::MotionForceTask");
		if ((n_trans > 0 && !dirs_trans) || (n_rot > 0 && !dirs_rot)) return fail(SAIP_ERR_INVALID_ARGUMENT, "null direction array");
		pr = range_basis_3(dirs_trans, n_trans > 0 ? n_trans : 0, bt);   // :55-87
		orr = range_basis_3(dirs_rot, n_rot > 0 ? n_rot : 0, br);
	}
	if (pr + orr == 0)  // :154-160
		return fail(SAIP_ERR_INVALID_ARGUMENT, "controlled_directions_translation and controlled_directions_rotation cannot both be empty in MotionForceTask::MotionForceTask");
	if (pr + orr == 1)
		return fail(SAIP_ERR_UNSUPPORTED, "rank-1 MotionForceTask: the reference's SingularityHandler never initialises its model for task_rank == 1 (SingularityHandler.cpp:100), refusing");
	memset(T.P, 0, sizeof(T.P));
	memset(T.dev.Bm, 0, sizeof(T.dev.Bm));
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			double sp = 0, so = 0;
			for (int c = 0; c < pr; c++) sp += bt[3 * i + c] * bt[3 * j + c];
			for (int c = 0; c < orr; c++) so += br[3 * i + c] * br[3 * j + c];
			T.P[6 * i + j] = sp;
			T.P[6 * (3 + i) + 3 + j] = so;
			T.dev.Ppos[3 * i + j] = sp;
			T.dev.Pori[3 * i + j] = so;
		}
	for (int i = 0; i < 3; i++) {
		for (int c = 0; c < pr; c++) T.dev.Bm[6 * i + c] = bt[3 * i + c];
		for (int c = 0; c < orr; c++) T.dev.Bm[6 * (3 + i) + pr + c] = br[3 * i + c];
	}
	// sigmaPosition = Ppos (I - sigmaForce) Ppos^T with force space dimension 0 (MotionForceTask.cpp:927-930), same for orientation
	double PT[9];
	m3_T(T.dev.Ppos, PT);
	m3_mul(T.dev.Ppos, PT, T.dev.sig_p);
	m3_T(T.dev.Pori, PT);
	m3_mul(T.dev.Pori, PT, T.dev.sig_o);
	T.dev.k = pr + orr;
	T.dev.goal_comps = 36;  // x3 R9 v3 w3 a3 alpha3 + goal force 3 + goal moment 3 + sensed force 3 + sensed moment 3 (sensor frame)
	T.dev.kv_force = 10.0;   // MotionForceTask.h:51,54
	T.dev.kv_moment = 10.0;
	T.dev.lin_sat = 0.3;     // :63-64
	T.dev.ang_sat = M_PI / 3;
	T.dev.force_axis[2] = T.dev.moment_axis[2] = 1.0;
	T.dev.bm_identity = (pr == 3 && orr == 3) ? 1 : 0;
	T.dev.cert_kroot = pow((double)T.dev.k, -1.0 / 8.0);
	for (int i = 0; i < 3; i++) {  // MotionForceTask.h:44-49
		T.dev.kp_pos[i] = 100.0; T.dev.kv_pos[i] = 20.0; T.dev.ki_pos[i] = 0.0;
		T.dev.kp_ori[i] = 200.0; T.dev.kv_ori[i] = 28.3; T.dev.ki_ori[i] = 0.0;
	}
	T.dev.s_min = 6e-3;   // MotionForceTask.cpp:197
	T.dev.s_max = 6e-2;
	T.dev.s_abs_tol = 1e-3;  // SingularityHandler.cpp:11
	T.dev.sing_handling = 1; // _enforce_handling_strategy = true, SingularityHandler.cpp:61
	T.dev.sing_strategies = 1;  // the reference always runs its blended type-1 / type-2 strategies while the handling is enforced (:100-121, 146-158, 310-367)
	T.dev.sh_kp1 = 50.0;     // KP_TYPE_1, KV_TYPE_1, KV_TYPE_2, SingularityHandler.cpp:17-19
	T.dev.sh_kv1 = 14.0;
	T.dev.sh_kv2 = 5.0;
	T.integ_rows = 12;  // position 3, orientation 3, force 3, moment 3
	T.dev.kp_force = T.dev.kp_moment = 0.7;  // MotionForceTask.h:50-59
	T.dev.ki_force = T.dev.ki_moment = 1.3;
	T.dev.kff_force = T.dev.kff_moment = 0.95;
	T.dev.max_force_fb = 20.0;
	T.dev.max_moment_fb = 10.0;
	T.dev.Rcs[0] = T.dev.Rcs[4] = T.dev.Rcs[8] = 1.0;  // _T_control_to_sensor = identity, MotionForceTask.cpp:94
	refresh_has_ki(T.dev);
	memset(&T.otg, 0, sizeof(T.otg));
	T.otg.m = 6;
	T.otg.gs = 8;
	for (int i = 0; i < SAIP_MAXN; i++) T.otg_limits[i] = T.otg_limits[SAIP_MAXN + i] = T.otg_limits[2 * SAIP_MAXN + i] = 1.0;
	for (int i = 0; i < 3; i++) {  // MotionForceTask.h:68-71
		T.otg_limits[i] = 0.3;
		T.otg_limits[SAIP_MAXN + i] = 2.0;
		T.otg_limits[3 + i] = M_PI / 3.0;
		T.otg_limits[SAIP_MAXN + 3 + i] = 2.0 * M_PI;
	}
	b->tasks.push_back(T);
	b->config_dirty = true;
	if (task_id) *task_id = (int)b->tasks.size() - 1;
	return SAIP_OK;
}

extern "C" saip_status saip_batch_add_joint_task(saip_batch* b, const char* task_name, const double* S, int rows, double loop_timestep, int* task_id) {
	if (!b || !task_name) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_add_joint_task: null argument");
	if (b->finalized) return fail(SAIP_ERR_ORDER, "tasks cannot be added after saip_batch_finalize");
	if ((int)b->tasks.size() >= SAIP_MAX_TASKS) return fail(SAIP_ERR_UNSUPPORTED, "more than %d tasks", SAIP_MAX_TASKS);
	const int n = b->model->n;
	TaskHost T;
	T.name = task_name;
	task_defaults(T.dev, loop_timestep);
	T.dev.type = saip::TASK_JOINT;
	if (rows <= 0 || !S) {  // JointTask.cpp:18-19
		T.dev.m = n;
		T.dev.s_identity = 1;
		for (int i = 0; i < n; i++) T.dev.S[i * n + i] = 1.0;
		T.full_joint = true;
	} else {
		if (rows > n)  // a rows x n matrix with rows > n cannot have full row rank (JointTask.cpp:34-39)
			return fail(SAIP_ERR_INVALID_ARGUMENT, "joint selection matrix is not full rank in JointTask constructor");
		// full row rank check (the reference uses FullPivLU, JointTask.cpp:34-39): Gaussian elimination with full pivoting
		std::vector<double> A(S, S + (size_t)rows * n);
		double amax = 0;
		for (double v : A) amax = fmax(amax, fabs(v));
		int rank = 0;
		std::vector<char> rused(rows, 0), cused(n, 0);
		for (int step = 0; step < rows; step++) {
			int pi = -1, pj = -1;
			double best = 0;
			for (int i = 0; i < rows; i++)
				if (!rused[i])
					for (int j = 0; j < n; j++)
						if (!cused[j] && fabs(A[(size_t)i * n + j]) > best) {
							best = fabs(A[(size_t)i * n + j]);
							pi = i;
							pj = j;
						}
			if (pi < 0 || best <= amax * 1e-12 * (rows > n ? rows : n)) break;
			rank++;
			rused[pi] = cused[pj] = 1;
			for (int i = 0; i < rows; i++)
				if (!rused[i]) {
					double f = A[(size_t)i * n + pj] / A[(size_t)pi * n + pj];
					for (int j = 0; j < n; j++) A[(size_t)i * n + j] -= f * A[(size_t)pi * n + j];
				}
		}
		if (rank != rows) return fail(SAIP_ERR_INVALID_ARGUMENT, "joint selection matrix is not full rank in JointTask constructor");
		T.dev.m = rows;
		memcpy(T.dev.S, S, sizeof(double) * rows * n);
		bool ident = (rows == n);
		for (int i = 0; i < rows && ident; i++)
			for (int j = 0; j < n; j++)
				if (S[i * n + j] != (i == j ? 1.0 : 0.0)) ident = false;
		T.dev.s_identity = ident;
		T.full_joint = (rows == n);  // JointTask::isFullJointTask(): task dof == robot dof
	}
	for (int i = 0; i < T.dev.m; i++) {  // JointTask.h:32-34
		T.dev.kp[i] = 50.0;
		T.dev.kv[i] = 14.0;
		T.dev.ki[i] = 0.0;
		T.dev.sat[i] = M_PI / 3.0;  // JointTask.h:44
	}
	T.dev.goal_comps = 3 * T.dev.m;
	T.integ_rows = T.dev.m;
	refresh_has_ki(T.dev);
	memset(&T.otg, 0, sizeof(T.otg));
	T.otg.m = T.dev.m;
	T.otg.gs = T.dev.m <= 8 ? 8 : 32;
	for (int i = 0; i < SAIP_MAXN; i++) {
		T.otg_limits[i] = M_PI / 3.0;              // DefaultParameters::otg_max_velocity, JointTask.h:40
		T.otg_limits[SAIP_MAXN + i] = 2.0 * M_PI;  // DefaultParameters::otg_max_acceleration, JointTask.h:41
		T.otg_limits[2 * SAIP_MAXN + i] = 10.0 * M_PI;  // DefaultParameters::otg_max_jerk, JointTask.h:42 (used in jerk-limited mode only)
	}
	b->tasks.push_back(T);
	b->config_dirty = true;
	if (task_id) *task_id = (int)b->tasks.size() - 1;
	return SAIP_OK;
}

template <typename Tp>
static saip_status dev_alloc(saip_batch* b, Tp** p, size_t count) {
	void* v = nullptr;
	HIP_TRY(hipMalloc(&v, count * sizeof(Tp)));
	HIP_TRY(hipMemset(v, 0, count * sizeof(Tp)));
	b->allocs.push_back(v);
	*p = (Tp*)v;
	return SAIP_OK;
}

extern "C" saip_status saip_batch_finalize(saip_batch* b) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "null batch");
	if (b->finalized) return SAIP_OK;
	// RobotController constructor checks, RobotController.cpp:8-66
	if (b->tasks.empty()) return fail(SAIP_ERR_INVALID_ARGUMENT, "RobotController must have at least one task");
	bool closed = false;
	for (size_t i = 0; i < b->tasks.size(); i++) {
		if (b->tasks[i].dev.dt != b->tasks[0].dev.dt) return fail(SAIP_ERR_INVALID_ARGUMENT, "All tasks must have the same loop timestep in RobotController");
		for (size_t j = 0; j < i; j++)
			if (b->tasks[j].name == b->tasks[i].name) return fail(SAIP_ERR_INVALID_ARGUMENT, "Tasks in RobotController must have unique names");
		if (closed)
			return fail(SAIP_ERR_INVALID_ARGUMENT, "task [%s] cannot be added to the controller because it is in the nullspace of a full joint task", b->tasks[i].name.c_str());
		if (b->tasks[i].dev.type == saip::TASK_JOINT && b->tasks[i].full_joint) closed = true;
	}
	if (has_device(b)) {
		HIP_TRY(hipSetDevice(b->device));
		HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
		const size_t n = b->model->n, ld = b->ld;
		saip_status st;
		if ((st = dev_alloc(b, &b->q, n * ld)) || (st = dev_alloc(b, &b->dq, n * ld)) || (st = dev_alloc(b, &b->tau, n * ld)) ||
			(st = dev_alloc(b, &b->status, ld)) || (st = dev_alloc(b, &b->model_dev, 1)) || (st = dev_alloc(b, &b->tasks_dev, b->tasks.size())) ||
			(st = dev_alloc(b, &b->flag_buf, 2 * (ld + 32))))
			return st;
		for (auto& T : b->tasks) {
			if ((st = dev_alloc(b, &T.goal_dev, (size_t)T.dev.goal_comps * ld)) || (st = dev_alloc(b, &T.integ_dev, (size_t)T.integ_rows * ld)) ||
				(st = dev_alloc(b, &T.integ_new_dev, (size_t)T.integ_rows * ld)))
				return st;
			T.dev.goal = T.goal_dev;
			T.dev.integ = T.integ_dev;
			T.dev.integ_new = T.integ_new_dev;
		}
		HIP_TRY(hipMemcpy(b->model_dev, &b->model->dev, sizeof(ModelDev), hipMemcpyHostToDevice));
	}
	b->finalized = true;
	b->config_dirty = true;
	return SAIP_OK;
}

extern "C" int saip_batch_task_count(const saip_batch* b) { return b ? (int)b->tasks.size() : 0; }
extern "C" int saip_batch_task_type(const saip_batch* b, int t) {
	if (!b || t < 0 || t >= (int)b->tasks.size()) return SAIP_TASK_UNDEFINED;
	return b->tasks[t].dev.type;
}
extern "C" const char* saip_batch_task_name(const saip_batch* b, int t) {
	if (!b || t < 0 || t >= (int)b->tasks.size()) return nullptr;
	return b->tasks[t].name.c_str();
}
extern "C" int saip_batch_task_by_name(const saip_batch* b, const char* name) {
	if (!b || !name) return -1;
	for (size_t i = 0; i < b->tasks.size(); i++)
		if (b->tasks[i].name == name) return (int)i;
	return -1;
}
extern "C" int saip_batch_task_dof(const saip_batch* b, int t) {
	if (!b || t < 0 || t >= (int)b->tasks.size()) return 0;
	return b->tasks[t].dev.type == saip::TASK_JOINT ? b->tasks[t].dev.m : b->tasks[t].dev.k;
}
extern "C" int saip_batch_goal_components(const saip_batch* b, int t) {
	if (!b || t < 0 || t >= (int)b->tasks.size()) return 0;
	return b->tasks[t].dev.goal_comps;
}
// projection of a motion-force task (parity tests): P 6x6 row-major, basis 6x6 row-major (first *rank columns valid)
extern "C" saip_status saip_batch_get_task_projection(const saip_batch* b, int t, double* P36, double* basis36, int* rank) {
	saip_status st = check_batch(b, t, "saip_batch_get_task_projection");
	if (st) return st;
	if (t < 0 || b->tasks[t].dev.type != saip::TASK_MOTION_FORCE) return fail(SAIP_ERR_INVALID_ARGUMENT, "task %d is not a MotionForceTask", t);
	if (P36) memcpy(P36, b->tasks[t].P, sizeof(double) * 36);
	if (basis36) memcpy(basis36, b->tasks[t].dev.Bm, sizeof(double) * 36);
	if (rank) *rank = b->tasks[t].dev.k;
	return SAIP_OK;
}

static saip_status set_gain3(double* kp, double* kv, double* ki, const double* p, const double* v, const double* i, int n_gains, const char* who) {
	if (!p || !v || !i) return fail(SAIP_ERR_INVALID_ARGUMENT, "%s: null gain pointer", who);
	if (n_gains != 1 && n_gains != 3) return fail(SAIP_ERR_INVALID_ARGUMENT, "kp, kv and ki should be of size 1 or 3 in %s", who);
	for (int c = 0; c < n_gains; c++)
		if (p[c] < 0 || v[c] < 0 || i[c] < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "all gains should be positive or zero in %s", who);
	for (int c = 0; c < 3; c++) {
		kp[c] = p[n_gains == 1 ? 0 : c];
		kv[c] = v[n_gains == 1 ? 0 : c];
		ki[c] = i[n_gains == 1 ? 0 : c];
	}
	return SAIP_OK;
}
static void refresh_has_ki(TaskDev& d) {
	// derived fields: pseudo-inverse of the diagonal kv gains (SaiModel::computePseudoInverse), control-law variant
	for (int c = 0; c < 3; c++) {
		d.kvinv_pos[c] = fabs(d.kv_pos[c]) > 1e-6 ? 1.0 / d.kv_pos[c] : 0.0;
		d.kvinv_ori[c] = fabs(d.kv_ori[c]) > 1e-6 ? 1.0 / d.kv_ori[c] : 0.0;
	}
	for (int c = 0; c < d.m; c++) d.kvinv[c] = fabs(d.kv[c]) > 1e-6 ? 1.0 / d.kv[c] : 0.0;
	d.general_law = (d.vel_sat || d.force_dim || d.moment_dim || d.cl_force || d.cl_moment) ? 1 : 0;
	d.has_ki = 0;
	if (d.type == saip::TASK_MOTION_FORCE) {
		for (int c = 0; c < 3; c++)
			if (d.ki_pos[c] != 0 || d.ki_ori[c] != 0) d.has_ki = 1;
	} else {
		for (int c = 0; c < d.m; c++)
			if (d.ki[c] != 0) d.has_ki = 1;
	}
}
static saip_status need_type(saip_batch* b, int t, int type, const char* fn) {
	saip_status st = check_batch(b, t, fn);
	if (st) return st;
	if (t < 0 || b->tasks[t].dev.type != type) return fail(SAIP_ERR_INVALID_ARGUMENT, "%s: task %d has the wrong type", fn, t);
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_pos_control_gains(saip_batch* b, int t, const double* kp, const double* kv, const double* ki, int ng) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_pos_control_gains");
	if (st) return st;
	TaskDev& d = b->tasks[t].dev;
	st = set_gain3(d.kp_pos, d.kv_pos, d.ki_pos, kp, kv, ki, ng, "MotionForceTask::setPosControlGains");
	refresh_has_ki(d);
	b->config_dirty = true;
	return st;
}
extern "C" saip_status saip_batch_set_ori_control_gains(saip_batch* b, int t, const double* kp, const double* kv, const double* ki, int ng) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_ori_control_gains");
	if (st) return st;
	TaskDev& d = b->tasks[t].dev;
	st = set_gain3(d.kp_ori, d.kv_ori, d.ki_ori, kp, kv, ki, ng, "MotionForceTask::setOriControlGains");
	refresh_has_ki(d);
	b->config_dirty = true;
	return st;
}
extern "C" saip_status saip_batch_set_joint_gains(saip_batch* b, int t, const double* kp, const double* kv, const double* ki, int ng) {
	saip_status st = need_type(b, t, saip::TASK_JOINT, "saip_batch_set_joint_gains");
	if (st) return st;
	TaskDev& d = b->tasks[t].dev;
	if (!kp || !kv || !ki) return fail(SAIP_ERR_INVALID_ARGUMENT, "JointTask::setGains: null gain pointer");
	if (ng != 1 && ng != d.m) return fail(SAIP_ERR_INVALID_ARGUMENT, "size of gain vectors inconsistent with number of task dofs in JointTask::setGains");
	for (int c = 0; c < ng; c++)
		if (kp[c] < 0 || kv[c] < 0 || ki[c] < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "gains must be positive or zero in JointTask::setGains");
	for (int c = 0; c < d.m; c++) {
		d.kp[c] = kp[ng == 1 ? 0 : c];
		d.kv[c] = kv[ng == 1 ? 0 : c];
		d.ki[c] = ki[ng == 1 ? 0 : c];
	}
	refresh_has_ki(d);
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_dynamic_decoupling_type(saip_batch* b, int t, int type) {
	saip_status st = check_batch(b, t, "saip_batch_set_dynamic_decoupling_type");
	if (st) return st;
	if (t < 0 || type < 0 || type > 2) return fail(SAIP_ERR_INVALID_ARGUMENT, "Dynamic decoupling type not recognized");
	b->tasks[t].dev.decoupling = type;
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_bie_threshold(saip_batch* b, int t, double thr) {
	saip_status st = check_batch(b, t, "saip_batch_set_bie_threshold");
	if (st) return st;
	if (t < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task");
	// JointTask clamps negative thresholds to 0 (JointTask.h:372-378); SingularityHandler stores them as-is (SingularityHandler.h:81-86)
	if (thr < 0 && b->tasks[t].dev.type == saip::TASK_JOINT) thr = 0;
	b->tasks[t].dev.bie_threshold = thr;
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_singularity_bounds(saip_batch* b, int t, double s_min, double s_max) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_singularity_bounds");
	if (st) return st;
	if (s_min < 0 || s_max < s_min) return fail(SAIP_ERR_INVALID_ARGUMENT, "singularity bounds must satisfy 0 <= s_min <= s_max");
	b->tasks[t].dev.s_min = s_min;
	b->tasks[t].dev.s_max = s_max;
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_singularity_handling(saip_batch* b, int t, int enabled) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_singularity_handling");
	if (st) return st;
	b->tasks[t].dev.sing_handling = enabled ? 1 : 0;
	b->config_dirty = true;
	return SAIP_OK;
}
// SingularityHandler state of every instance back to "never singular" (SingularityHandler.cpp:55-63); allocates on first use
static saip_status sh_reinit(saip_batch* b, TaskHost& T) {
	if (!T.dev.sh) {
		saip_status st = dev_alloc(b, &T.dev.sh, (size_t)b->ld);
		if (st) return st;
	}
	HIP_TRY(hipMemsetAsync(T.dev.sh, 0, (size_t)b->ld * sizeof(saip::ShState), b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_singularity_strategies(saip_batch* b, int t, int enabled) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_singularity_strategies");
	if (st) return st;
	TaskHost& T = b->tasks[t];
	T.dev.sing_strategies = enabled ? 1 : 0;
	b->config_dirty = true;
	if (enabled && b->finalized && has_device(b)) {
		if ((st = need_ready(b, "saip_batch_set_singularity_strategies"))) return st;
		return sh_reinit(b, T);
	}
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_singularity_gains(saip_batch* b, int t, double kp_type_1, double kv_type_1, double kv_type_2) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_singularity_gains");
	if (st) return st;
	b->tasks[t].dev.sh_kp1 = kp_type_1;
	b->tasks[t].dev.sh_kv1 = kv_type_1;
	b->tasks[t].dev.sh_kv2 = kv_type_2;
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_all_singularities_type1(saip_batch* b, int t, int flag) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_all_singularities_type1");
	if (st) return st;
	b->tasks[t].dev.sh_force_type1 = flag ? 1 : 0;
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_type1_posture(saip_batch* b, int t, const double* q_des, int per_instance) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_type1_posture");
	if (st) return st;
	if (!q_des) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_set_type1_posture: null posture");
	TaskHost& T = b->tasks[t];
	// _q_prior is overwritten with the current posture whenever an instance enters a singular region (SingularityHandler.cpp:232-235):
	// the call only matters for instances that are inside one, which needs the device state
	if (!T.dev.sing_strategies || !T.dev.sh || !b->finalized || !has_device(b)) return SAIP_OK;
	if ((st = need_ready(b, "saip_batch_set_type1_posture"))) return st;
	const int n = b->model->n;
	std::vector<double> host((size_t)b->B * n);
	for (int i = 0; i < b->B; i++)
		for (int j = 0; j < n; j++) host[(size_t)i * n + j] = per_instance ? q_des[(size_t)i * n + j] : q_des[j];
	HIP_TRY(hipMemcpy2DAsync(T.dev.sh, sizeof(saip::ShState), host.data(), (size_t)n * sizeof(double), (size_t)n * sizeof(double), (size_t)b->B,
							 hipMemcpyHostToDevice, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_internal_otg(saip_batch* b, int t, int enabled) {
	saip_status st = check_batch(b, t, "saip_batch_set_internal_otg");
	if (st) return st;
	if (t < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task");
	TaskHost& T = b->tasks[t];
	if (enabled && !T.otg_enabled) T.otg_inited = false;  // enableInternalOtg* re-initialises a disabled OTG (JointTask.cpp:374-376)
	T.otg_enabled = enabled != 0;
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_otg_acceleration_limited(saip_batch* b, int t, const double* max_velocity, const double* max_acceleration, int count) {
	saip_status st = check_batch(b, t, "saip_batch_set_otg_acceleration_limited");
	if (st) return st;
	if (t < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task");
	TaskHost& T = b->tasks[t];
	if (T.dev.type == saip::TASK_MOTION_FORCE) {
		// MotionForceTask::enableInternalOtgAccelerationLimited(max_lin_vel, max_lin_acc, max_ang_vel, max_ang_acc), MotionForceTask.cpp:510-523
		if (!max_velocity || !max_acceleration || count != 2)
			return fail(SAIP_ERR_INVALID_ARGUMENT, "motion-force task: pass {linear, angular} max velocities and max accelerations (count = 2)");
		for (int i = 0; i < 2; i++) {
			if (!(max_velocity[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max velocity set to 0 or negative value in some directions in OTG_6dof_cartesian::setMax%sVelocity", i ? "Angular" : "Linear");
			if (!(max_acceleration[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max acceleration set to 0 or negative value in some directions in OTG_6dof_cartesian::setMax%sAcceleration", i ? "Angular" : "Linear");
		}
		for (int i = 0; i < 6; i++) {
			T.otg_limits[i] = max_velocity[i / 3];
			T.otg_limits[SAIP_MAXN + i] = max_acceleration[i / 3];
		}
		T.otg_limits_dirty = true;
		T.otg.epoch++;
		if (!T.otg_enabled || T.otg.jerk) T.otg_inited = false;  // MotionForceTask.cpp:513-515: re-initialised when the OTG was off or jerk-limited
		T.otg.jerk = 0;
		T.otg_enabled = true;
		b->config_dirty = true;
		return SAIP_OK;
	}
	const int m = T.dev.m;
	if (!max_velocity || !max_acceleration || (count != 1 && count != m))
		return fail(SAIP_ERR_INVALID_ARGUMENT, "max velocity or max acceleration vector size not consistent with task dof in JointTask::enableInternalOtgAccelerationLimited");
	for (int i = 0; i < count; i++) {
		if (!(max_velocity[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max velocity cannot be 0 or negative in any directions in OTG_joints::setMaxVelocity");
		if (!(max_acceleration[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max acceleration cannot be 0 or negative in any directions in OTG_joints::setMaxAcceleration");
	}
	for (int i = 0; i < m; i++) {
		T.otg_limits[i] = max_velocity[count == 1 ? 0 : i];
		T.otg_limits[SAIP_MAXN + i] = max_acceleration[count == 1 ? 0 : i];
	}
	T.otg_limits_dirty = true;
	T.otg.epoch++;
	if (!T.otg_enabled || T.otg.jerk) T.otg_inited = false;  // JointTask.cpp:374-376
	T.otg.jerk = 0;
	T.otg_enabled = true;
	b->config_dirty = true;
	return SAIP_OK;
}
// JointTask::enableInternalOtgJerkLimited (JointTask.cpp:383-410; OTG_joints::setMaxVelocity / setMaxAcceleration / setMaxJerk, OTG_joints.cpp:44-86) and
// MotionForceTask::enableInternalOtgJerkLimited (MotionForceTask.cpp:525-545; OTG_6dof_cartesian::setMaxJerk, OTG_6dof_cartesian.cpp:126-136):
// third-order Ruckig profiles on the device (csrc/saip_otg3.h).  The OTG is re-initialised at the current task position when it was off or
// acceleration-limited (:400-402 / :530-532).
extern "C" saip_status saip_batch_set_otg_jerk_limited(saip_batch* b, int t, const double* max_velocity, const double* max_acceleration, const double* max_jerk, int count) {
	saip_status st = check_batch(b, t, "saip_batch_set_otg_jerk_limited");
	if (st) return st;
	if (t < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task");
	TaskHost& T = b->tasks[t];
	if (T.dev.type == saip::TASK_MOTION_FORCE) {
		if (!max_velocity || !max_acceleration || !max_jerk || count != 2)
			return fail(SAIP_ERR_INVALID_ARGUMENT, "motion-force task: pass {linear, angular} max velocities, max accelerations and max jerks (count = 2)");
		for (int i = 0; i < 2; i++) {
			if (!(max_velocity[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max velocity set to 0 or negative value in some directions in OTG_6dof_cartesian::setMax%sVelocity", i ? "Angular" : "Linear");
			if (!(max_acceleration[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max acceleration set to 0 or negative value in some directions in OTG_6dof_cartesian::setMax%sAcceleration", i ? "Angular" : "Linear");
			if (!(max_jerk[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max jerk set to 0 or negative value in some directions in OTG_6dof_cartesian::setMaxJerk");
		}
		for (int i = 0; i < 6; i++) {
			T.otg_limits[i] = max_velocity[i / 3];
			T.otg_limits[SAIP_MAXN + i] = max_acceleration[i / 3];
			T.otg_limits[2 * SAIP_MAXN + i] = max_jerk[i / 3];
		}
	} else {
		const int m = T.dev.m;
		if (!max_velocity || !max_acceleration || !max_jerk || (count != 1 && count != m))
			return fail(SAIP_ERR_INVALID_ARGUMENT, "max velocity, max acceleration or max jerk vector size not consistent with task dof in JointTask::enableInternalOtgJerkLimited");
		for (int i = 0; i < count; i++) {
			if (!(max_velocity[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max velocity cannot be 0 or negative in any directions in OTG_joints::setMaxVelocity");
			if (!(max_acceleration[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max acceleration cannot be 0 or negative in any directions in OTG_joints::setMaxAcceleration");
			if (!(max_jerk[i] > 0)) return fail(SAIP_ERR_INVALID_ARGUMENT, "max jerk cannot be 0 or negative in any directions in OTG_joints::setMaxJerk");
		}
		for (int i = 0; i < m; i++) {
			T.otg_limits[i] = max_velocity[count == 1 ? 0 : i];
			T.otg_limits[SAIP_MAXN + i] = max_acceleration[count == 1 ? 0 : i];
			T.otg_limits[2 * SAIP_MAXN + i] = max_jerk[count == 1 ? 0 : i];
		}
	}
	T.otg_limits_dirty = true;
	T.otg.epoch++;
	if (!T.otg_enabled || !T.otg.jerk) T.otg_inited = false;
	T.otg.jerk = 1;
	T.otg_enabled = true;
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_velocity_saturation(saip_batch* b, int t, int enabled) {
	saip_status st = check_batch(b, t, "saip_batch_set_velocity_saturation");
	if (st) return st;
	if (t < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task");
	b->tasks[t].dev.vel_sat = enabled ? 1 : 0;
	refresh_has_ki(b->tasks[t].dev);
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_saturation_velocities(saip_batch* b, int t, const double* v, int nv) {
	saip_status st = check_batch(b, t, "saip_batch_set_saturation_velocities");
	if (st) return st;
	if (t < 0 || (nv > 0 && !v)) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task or null values");
	TaskDev& d = b->tasks[t].dev;
	if (d.type == saip::TASK_MOTION_FORCE) {
		if (nv != 2) return fail(SAIP_ERR_INVALID_ARGUMENT, "MotionForceTask::enableVelocitySaturation takes a linear and an angular saturation velocity");
		if (v[0] <= 0 || v[1] <= 0)  // MotionForceTask.cpp:773-777
			return fail(SAIP_ERR_INVALID_ARGUMENT, "Velocity saturation values should be strictly positive or zero in MotionForceTask::enableVelocitySaturation");
		d.lin_sat = v[0];
		d.ang_sat = v[1];
	} else {
		if (nv != 1 && nv != d.m)  // JointTask.cpp:423-427
			return fail(SAIP_ERR_INVALID_ARGUMENT, "saturation velocity vector size not consistent with task dof in JointTask::enableVelocitySaturation");
		for (int i = 0; i < nv; i++)
			if (v[i] <= 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "saturation velocity must be positive in JointTask::enableVelocitySaturation");
		for (int i = 0; i < d.m; i++) d.sat[i] = v[nv == 1 ? 0 : i];
	}
	b->config_dirty = true;
	return SAIP_OK;
}
static saip_status launch_reinit_masked(saip_batch* b, int task, int mask);
static saip_status parametrize_space(saip_batch* b, int t, int dim, const double* axis, int* changed, bool moment, const char* who) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, who);
	if (st) return st;
	if (dim < 0 || dim > 3)  // MotionForceTask.cpp:831-835, 864-868
		return fail(SAIP_ERR_INVALID_ARGUMENT, "%s space dimension should be between 0 and 3 in %s", moment ? "Moment" : "Force", who);
	TaskDev& d = b->tasks[t].dev;
	int& cur_dim = moment ? d.moment_dim : d.force_dim;
	double* cur_axis = moment ? d.moment_axis : d.force_axis;
	bool reset = dim != cur_dim;
	if (dim == 1 || dim == 2) {
		if (!axis) return fail(SAIP_ERR_INVALID_ARGUMENT, "%s: null axis", who);
		double nrm = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
		if (nrm < 1e-2)  // :838-842, 871-875
			return fail(SAIP_ERR_INVALID_ARGUMENT, "%s axis should be a non singular vector in %s", moment ? "Moment or rot motion" : "Force or motion", who);
		double a[3] = {axis[0] / nrm, axis[1] / nrm, axis[2] / nrm};
		double diff = fabs(a[0] - cur_axis[0]) + fabs(a[1] - cur_axis[1]) + fabs(a[2] - cur_axis[2]);
		reset = reset || diff > 1e-12;
		memcpy(cur_axis, a, sizeof(a));
	}
	cur_dim = dim;
	refresh_has_ki(d);
	b->config_dirty = true;
	if (changed) *changed = reset ? 1 : 0;
	if (reset && b->finalized && b->device >= 0 && b->state_pushed)  // goal := current, integrators := 0 (:846-852, 880-886)
		return launch_reinit_masked(b, t, moment ? 2 : 1);
	return SAIP_OK;
}
extern "C" saip_status saip_batch_parametrize_force_motion_spaces(saip_batch* b, int t, int dim, const double* axis, int* changed) {
	return parametrize_space(b, t, dim, axis, changed, false, "MotionForceTask::parametrizeForceMotionSpaces");
}
extern "C" saip_status saip_batch_parametrize_moment_rot_motion_spaces(saip_batch* b, int t, int dim, const double* axis, int* changed) {
	return parametrize_space(b, t, dim, axis, changed, true, "MotionForceTask::parametrizeMomentRotMotionSpaces");
}
extern "C" saip_status saip_batch_set_parametrization_in_compliant_frame(saip_batch* b, int t, int enabled) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_parametrization_in_compliant_frame");
	if (st) return st;
	b->tasks[t].dev.param_in_compliant_frame = enabled ? 1 : 0;
	b->config_dirty = true;
	return SAIP_OK;
}
static saip_status set_fm_gains(saip_batch* b, int t, double kp, double kv, double ki, bool moment, const char* who) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, who);
	if (st) return st;
	if (kp < 0 || kv < 0 || ki < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "all gains should be positive or zero in %s", who);
	TaskDev& d = b->tasks[t].dev;
	(moment ? d.kv_moment : d.kv_force) = kv;
	(moment ? d.kp_moment : d.kp_force) = kp;  // kp, ki only act in closed-loop control
	(moment ? d.ki_moment : d.ki_force) = ki;
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_force_control_gains(saip_batch* b, int t, double kp, double kv, double ki) {
	return set_fm_gains(b, t, kp, kv, ki, false, "MotionForceTask::setForceControlGains");
}
extern "C" saip_status saip_batch_set_moment_control_gains(saip_batch* b, int t, double kp, double kv, double ki) {
	return set_fm_gains(b, t, kp, kv, ki, true, "MotionForceTask::setMomentControlGains");
}
// setClosedLoopForceControl / setClosedLoopMomentControl (MotionForceTask.cpp:973-986): a change resets the linear / angular integrators
static saip_status set_closed_loop(saip_batch* b, int t, int enabled, bool moment, const char* who) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, who);
	if (st) return st;
	TaskDev& d = b->tasks[t].dev;
	int& flag = moment ? d.cl_moment : d.cl_force;
	const bool changed = (flag != 0) != (enabled != 0);
	flag = enabled ? 1 : 0;
	refresh_has_ki(d);
	b->config_dirty = true;
	if (changed && b->finalized && has_device(b)) {
		// resetIntegratorsLinear / Angular: position + force (orientation + moment) integrators := 0
		const int rows[2] = {moment ? 3 : 0, moment ? 9 : 6};
		for (int r : rows) HIP_TRY(hipMemsetAsync(b->tasks[t].integ_dev + (size_t)r * b->ld, 0, (size_t)3 * b->ld * sizeof(double), b->stream));
		HIP_TRY(hipStreamSynchronize(b->stream));
	}
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_closed_loop_force_control(saip_batch* b, int t, int enabled) {
	return set_closed_loop(b, t, enabled, false, "saip_batch_set_closed_loop_force_control");
}
extern "C" saip_status saip_batch_set_closed_loop_moment_control(saip_batch* b, int t, int enabled) {
	return set_closed_loop(b, t, enabled, true, "saip_batch_set_closed_loop_moment_control");
}
// POPCExplicitForceControl::reInitialize (POPCExplicitForceControl.cpp:10-22) for every instance; allocates the state on first use
static saip_status popc_reinit(saip_batch* b, TaskHost& T) {
	const int cap = T.dev.popc_cap;
	const size_t rows = 7 + (size_t)cap, ld = b->ld;
	if (!T.dev.popc) {
		saip_status st = dev_alloc(b, &T.dev.popc, rows * ld);
		if (st) return st;
	}
	HIP_TRY(hipMemsetAsync(T.dev.popc, 0, rows * ld * sizeof(double), b->stream));
	std::vector<double> one(ld, 1.0), fifty(ld, 50.0);  // _Rc = 1, _PO_counter = _PO_max_counter
	HIP_TRY(hipMemcpyAsync(T.dev.popc + 2 * ld, one.data(), ld * sizeof(double), hipMemcpyHostToDevice, b->stream));
	HIP_TRY(hipMemcpyAsync(T.dev.popc + 4 * ld, fifty.data(), ld * sizeof(double), hipMemcpyHostToDevice, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return SAIP_OK;
}
// enablePassivity / disablePassivity (MotionForceTask.h:630-631 -> POPCExplicitForceControl::enable / disable, .cpp:24-29)
extern "C" saip_status saip_batch_set_passivity(saip_batch* b, int t, int enabled) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_passivity");
	if (st) return st;
	TaskHost& T = b->tasks[t];
	T.dev.popc_enabled = enabled ? 1 : 0;
	if (T.dev.popc_cap == 0) T.dev.popc_cap = 1024;
	b->config_dirty = true;
	if (b->finalized && has_device(b)) {
		if ((st = need_ready(b, "saip_batch_set_passivity"))) return st;
		if (!enabled || !T.dev.popc) return popc_reinit(b, T);  // disable() re-initialises; the first enable() creates the state
	}
	return SAIP_OK;
}
// setFeedforwardForceGain / MomentGain, setMaxForceControlFeedbackOutput / Moment (MotionForceTask.h:330-355)
extern "C" saip_status saip_batch_set_force_control_parameters(saip_batch* b, int t, double kff_force, double kff_moment, double max_force_feedback, double max_moment_feedback) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_force_control_parameters");
	if (st) return st;
	TaskDev& d = b->tasks[t].dev;
	d.kff_force = kff_force;
	d.kff_moment = kff_moment;
	d.max_force_fb = max_force_feedback;
	d.max_moment_fb = max_moment_feedback;
	b->config_dirty = true;
	return SAIP_OK;
}
// setForceSensorFrame (MotionForceTask.cpp:794-803), given directly as _T_control_to_sensor = compliant_frame^-1 * T_link_sensor
extern "C" saip_status saip_batch_set_control_to_sensor_transform(saip_batch* b, int t, const double* R_row_major, const double* translation) {
	saip_status st = need_type(b, t, saip::TASK_MOTION_FORCE, "saip_batch_set_control_to_sensor_transform");
	if (st) return st;
	if (!R_row_major || !translation) return fail(SAIP_ERR_INVALID_ARGUMENT, "null transform");
	TaskDev& d = b->tasks[t].dev;
	for (int i = 0; i < 9; i++) d.Rcs[i] = R_row_major[i];
	for (int i = 0; i < 3; i++) d.tcs[i] = translation[i];
	b->config_dirty = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_enable_gravity_compensation(saip_batch* b, int e) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "null batch");
	b->gravity_comp = e != 0;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_enable_joint_limit_avoidance(saip_batch* b, int e) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "null batch");
	b->jla = e != 0;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_enable_torque_saturation(saip_batch* b, int e) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "null batch");
	b->torque_sat = e != 0;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_flagged_torque_policy(saip_batch* b, int nan) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "null batch");
	b->flag_nan = nan != 0;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_flagged_recompute(saip_batch* b, int on_list) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "null batch");
	b->flagged_on_list = on_list != 0;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_integrator_tracking(saip_batch* b, int always) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "null batch");
	b->integ_always = always != 0;
	return SAIP_OK;
}

static saip_status need_ready(saip_batch* b, const char* fn) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "%s: null batch", fn);
	if (!b->finalized) return fail(SAIP_ERR_ORDER, "%s: call saip_batch_finalize first", fn);
	if (!has_device(b)) return fail(SAIP_ERR_NO_DEVICE, "%s: configuration-only batch (no HIP device): the engine has no CPU path", fn);
	if (hipSetDevice(b->device) != hipSuccess) return fail(SAIP_ERR_DEVICE, "hipSetDevice(%d) failed", b->device);
	return SAIP_OK;
}
// host [comps][B] <-> device [comps][ld]
static saip_status copy_h2d(saip_batch* b, double* dev, const double* host, int comps) {
	HIP_TRY(hipMemcpy2DAsync(dev, (size_t)b->ld * sizeof(double), host, (size_t)b->B * sizeof(double), (size_t)b->B * sizeof(double), comps,
							 hipMemcpyHostToDevice, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));  // the host buffer may be reused by the caller right away
	return SAIP_OK;
}
static saip_status copy_d2h(saip_batch* b, double* host, const double* dev, int comps) {
	HIP_TRY(hipMemcpy2DAsync(host, (size_t)b->B * sizeof(double), dev, (size_t)b->ld * sizeof(double), (size_t)b->B * sizeof(double), comps,
							 hipMemcpyDeviceToHost, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return SAIP_OK;
}

extern "C" saip_status saip_batch_set_state_host(saip_batch* b, const double* q, const double* dq) {
	saip_status st = need_ready(b, "saip_batch_set_state_host");
	if (st) return st;
	if (!q || !dq) return fail(SAIP_ERR_INVALID_ARGUMENT, "null state pointer");
	b->models_valid = false;
	b->state_epoch++;
	if ((st = copy_h2d(b, b->q, q, b->model->n))) return st;
	b->state_pushed = true;
	return copy_h2d(b, b->dq, dq, b->model->n);
}
extern "C" saip_status saip_batch_set_goal_host(saip_batch* b, int t, const double* goal) {
	saip_status st = need_ready(b, "saip_batch_set_goal_host");
	if (st) return st;
	if (t < 0 || t >= (int)b->tasks.size() || !goal) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task id or null goal");
	return copy_h2d(b, b->tasks[t].goal_dev, goal, b->tasks[t].dev.goal_comps);
}
extern "C" saip_status saip_batch_set_goal_field_host(saip_batch* b, int t, int first, int count, const double* values) {
	saip_status st = need_ready(b, "saip_batch_set_goal_field_host");
	if (st) return st;
	if (t < 0 || t >= (int)b->tasks.size() || !values) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task id or null values");
	if (first < 0 || count <= 0 || first + count > b->tasks[t].dev.goal_comps)  // e.g. JointTask.cpp:110-114
		return fail(SAIP_ERR_INVALID_ARGUMENT, "goal vector size not consistent with task dof");
	return copy_h2d(b, b->tasks[t].goal_dev + (size_t)first * b->ld, values, count);
}
extern "C" saip_status saip_batch_get_goal_host(saip_batch* b, int t, double* goal) {
	saip_status st = need_ready(b, "saip_batch_get_goal_host");
	if (st) return st;
	if (t < 0 || t >= (int)b->tasks.size() || !goal) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task id or null goal");
	return copy_d2h(b, goal, b->tasks[t].goal_dev, b->tasks[t].dev.goal_comps);
}

extern "C" double* saip_batch_device_q(saip_batch* b) { return b ? b->q : nullptr; }
extern "C" double* saip_batch_device_dq(saip_batch* b) { return b ? b->dq : nullptr; }
extern "C" double* saip_batch_device_goal(saip_batch* b, int t) { return (b && t >= 0 && t < (int)b->tasks.size()) ? b->tasks[t].goal_dev : nullptr; }
extern "C" double* saip_batch_device_tau(saip_batch* b) { return b ? (b->tau_bound ? b->tau_bound : b->tau) : nullptr; }
extern "C" uint8_t* saip_batch_device_status(saip_batch* b) { return b ? b->status : nullptr; }
extern "C" saip_status saip_batch_bind_tau_device(saip_batch* b, double* tau_dev) {
	if (!b) return fail(SAIP_ERR_INVALID_ARGUMENT, "null batch");
	b->tau_bound = tau_dev;
	return SAIP_OK;
}
extern "C" void* saip_batch_stream(saip_batch* b) { return b ? (void*)b->stream : nullptr; }

// lazily allocate the OTG state of a joint task: [fields][B*gs] lane-major doubles + per-instance scalars
static saip_status ensure_otg(saip_batch* b, TaskHost& T) {
	if (T.otg_alloc) return SAIP_OK;
	saip_status st;
	OtgDev& O = T.otg;
	O.lanes = (long long)b->B * O.gs;
	O.dt = T.dev.dt;
	O.n = b->model->n;
	if ((st = dev_alloc(b, &O.state, (size_t)saip::otg_state_fields() * (size_t)O.lanes)) || (st = dev_alloc(b, &O.time, (size_t)b->ld)) ||
		(st = dev_alloc(b, &O.duration, (size_t)b->ld)) || (st = dev_alloc(b, &O.flags, (size_t)b->ld)) || (st = dev_alloc(b, &O.seen_epoch, (size_t)b->ld)) ||
		(st = dev_alloc(b, &O.result, (size_t)b->ld)) || (st = dev_alloc(b, &T.desired_dev, (size_t)T.dev.goal_comps * b->ld)) ||
		(st = dev_alloc(b, &T.otg_limits_dev, (size_t)3 * SAIP_MAXN)))
		return st;
	if (T.dev.type == saip::TASK_MOTION_FORCE && (st = dev_alloc(b, &O.frame, (size_t)21 * b->ld))) return st;
	O.model = b->model_dev;
	O.goal_comps = T.dev.goal_comps;
	O.limits = T.otg_limits_dev;
	T.otg_limits_dirty = true;
	O.desired = T.desired_dev;
	O.goal = T.goal_dev;
	O.q = b->q;
	T.otg_alloc = true;
	T.otg_inited = false;
	return SAIP_OK;
}
// mode 1: OTG_joints::reInitialize(S q) for every instance; mode 0: one cycle of setGoal + update
static saip_status run_otg(saip_batch* b, int t, int mode, int mask = 3) {
	TaskHost& T = b->tasks[t];
	T.otg.goal_comps = T.dev.goal_comps;
	T.otg.task = b->tasks_dev + t;
	T.otg.dt = T.dev.dt;
	if (T.otg_limits_dirty) {
		HIP_TRY(hipMemcpyAsync(T.otg_limits_dev, T.otg_limits, sizeof(T.otg_limits), hipMemcpyHostToDevice, b->stream));
		HIP_TRY(hipStreamSynchronize(b->stream));  // limits change rarely; keeps the host array free to change again
		T.otg_limits_dirty = false;
	}
	hipError_t e = T.dev.type == saip::TASK_JOINT ? saip::launch_otg_joints(T.otg, b->B, b->ld, mode, b->stream)
												  : saip::launch_otg_cartesian(T.otg, b->B, b->ld, mode | (mask << 4), b->stream);
	if (e != hipSuccess) return fail(SAIP_ERR_DEVICE, "OTG kernel launch failed: %s", hipGetErrorString(e));
	if (mode == 1 && mask == 3) T.otg_inited = true;
	return SAIP_OK;
}

static saip_status make_params(saip_batch* b, CycleParams& P, bool diag) {
	for (auto& T : b->tasks) {
		if (T.otg_enabled) {
			saip_status st = ensure_otg(b, T);
			if (st) return st;
		}
		if (T.dev.type == saip::TASK_MOTION_FORCE && T.dev.sing_strategies && !T.dev.sh) {  // enabled before finalize
			saip_status st = sh_reinit(b, T);
			if (st) return st;
			b->config_dirty = true;
		}
		if (T.dev.type == saip::TASK_MOTION_FORCE && T.dev.popc_enabled && !T.dev.popc) {  // enabled before finalize
			saip_status st = popc_reinit(b, T);
			if (st) return st;
			b->config_dirty = true;
		}
	}
	if (b->config_dirty || diag) {
		std::vector<TaskDev> tmp;
		for (auto& T : b->tasks) {
			TaskDev d = T.dev;
			d.diag_N = diag ? T.diag_dev : nullptr;
			if (T.otg_enabled) d.goal = T.desired_dev;  // the law tracks the OTG output (JointTask.cpp:317-319, MotionForceTask.cpp:394-406)
			d.law_identity = 0;
			if (d.type == saip::TASK_MOTION_FORCE && d.k == 6 && !d.general_law) {
				bool id = true;
				for (int i = 0; i < 36 && id; i++) id = d.Bm[i] == ((i % 7 == 0) ? 1.0 : 0.0);
				for (int i = 0; i < 9 && id; i++) {
					const double e = (i % 4 == 0) ? 1.0 : 0.0;
					id = d.Ppos[i] == e && d.Pori[i] == e && d.sig_p[i] == e && d.sig_o[i] == e;
				}
				d.law_identity = id ? 1 : 0;
			}
			tmp.push_back(d);
		}
		HIP_TRY(hipMemcpyAsync(b->tasks_dev, tmp.data(), tmp.size() * sizeof(TaskDev), hipMemcpyHostToDevice, b->stream));
		HIP_TRY(hipStreamSynchronize(b->stream));  // tmp is a stack object
		b->config_dirty = diag;  // a diagnostic upload must be replaced before the next normal launch
	}
	P.B = b->B;
	P.ld = b->ld;
	P.n = b->model->n;
	P.nt = (int)b->tasks.size();
	P.sim_substeps = 0;
	P.sim_dt = P.sim_damping = 0.0;
	P.sim_gravity[0] = P.sim_gravity[1] = P.sim_gravity[2] = 0.0;
	P.gravity_comp = b->gravity_comp;
	P.torque_sat = b->torque_sat;
	P.integ_always = b->integ_always;
	P.jla = b->jla;
	P.only_flagged = 0;
	P.diag = diag;
	P.q = b->q;
	P.dq = b->dq;
	P.tau = b->tau_bound ? b->tau_bound : b->tau;
	P.status = b->status;
	if (diag) {
		// a diagnostic launch re-evaluates the task models only (phase 1: no control law, no integrator / OTG / handler / passivity state
		// is advanced) and writes its torques and status to scratch: the results of the last cycle stay what integrate() / getTorques() see
		saip_status st;
		if (!b->diag_tau && ((st = dev_alloc(b, &b->diag_tau, (size_t)b->model->n * b->ld)) || (st = dev_alloc(b, &b->diag_status, (size_t)b->ld)))) return st;
		P.tau = b->diag_tau;
		P.status = b->diag_status;
	}
	P.model = b->model_dev;
	P.tasks = b->tasks_dev;
	for (int t = 0; t < 2; t++) {
		P.goal2[t] = t < (int)b->tasks.size() ? (b->tasks[t].otg_enabled ? b->tasks[t].desired_dev : b->tasks[t].dev.goal) : nullptr;  // (as in the task block uploaded above)
		P.integ2[t] = t < (int)b->tasks.size() ? b->tasks[t].dev.integ : nullptr;
	}
	// lane kernel: motion-force task slots, batch-uniform rank bounds, the shared BIE threshold
	P.mf_task[0] = P.mf_task[1] = -1;
	P.any_bie = 0;
	P.bie_thr = 0.0;
	int nmf = 0, used = 0;
	for (int t = 0; t < SAIP_MAXT; t++) P.rank_bound[t] = P.lane_task_is_joint[t] = 0;
	for (size_t t = 0; t < b->tasks.size(); t++) {
		const TaskDev& d = b->tasks[t].dev;
		int own = d.type == saip::TASK_MOTION_FORCE ? d.k : d.m;
		int bound = own < P.n - used ? own : P.n - used;  // rank(J N_prec) <= n - sum of the ranks certified above
		P.rank_bound[t] = bound < 0 ? 0 : bound;
		P.lane_task_is_joint[t] = d.type == saip::TASK_JOINT;
		if (d.type == saip::TASK_MOTION_FORCE) {
			if (nmf < 2) P.mf_task[nmf] = (int)t;
			nmf++;
			used += d.k;  // exact whenever the instance stays in the non-singular branch (otherwise it is flagged anyway)
		}
		if (d.decoupling == SAIP_BOUNDED_INERTIA_ESTIMATES) {
			P.any_bie = 1;
			P.bie_thr = d.bie_threshold;
		}
	}
	P.reinit_task = -1;
	P.reinit_mask = 7;
	P.single_task = -1;
	P.phase = diag ? 1 : 0;
	P.ext_nprec = nullptr;
	P.ext_tau_prec = nullptr;
	P.out_N = nullptr;
	P.out_Ntot = nullptr;
	for (int t = 0; t < SAIP_MAXT; t++) P.task_cycle[t] = t < (int)b->tasks.size() ? b->tasks[t].sh_cycle : 0;
	P.flag_nan = b->flag_nan ? 1 : 0;
	P.flag_count = nullptr;  // set by launch_cycle when a device-side slow path follows the lane / eight-lane kernel
	P.flag_count_next = nullptr;
	P.flag_list = nullptr;
	P.slow_tail = 0;
	P.lane_general = 0;
	P.oct_general_joint = 0;
	P.oct_partial_mf = 0;
	P.oct_truncate = 0;
	for (auto& T : b->tasks)
		if (T.dev.general_law) P.lane_general = 1;
	P.wave_general_joint = 0;
	for (size_t t = 0; t < b->tasks.size(); t++) {
		const TaskDev& T = b->tasks[t].dev;
		if (T.type == saip::TASK_JOINT && !(T.s_identity && T.m == P.n && t + 1 == b->tasks.size())) P.wave_general_joint = 1;
	}
	P.lane_prefetch_ok = (b->tasks.size() == 2 && b->tasks[0].dev.type == saip::TASK_MOTION_FORCE && b->tasks[1].dev.type == saip::TASK_JOINT &&
						  b->tasks[1].dev.m == P.n && !b->tasks[0].dev.general_law) ? 1 : 0;
	{
		static const char* la = getenv("SAIP_LANE_ACTIVE");  // diagnostic knob, read once
		static const int a = la ? atoi(la) : 64;
		P.lane_active = (a >= 1 && a <= 64) ? a : 64;
	}
	return SAIP_OK;
}
// the lane-per-instance register kernel covers 6..8 dof, at most two motion-force tasks and one shared BIE threshold
static bool lane_eligible(const saip_batch* b) {
	const int n = b->model->n;
	if (n < 6 || n > 8) return false;
	int nmf = 0;
	double thr = -1.0;
	for (auto& T : b->tasks) {
		if (T.dev.type == saip::TASK_MOTION_FORCE) nmf++;
		// closed-loop force / moment control runs in the lane kernel's general-law instantiations; with the passivity controller around
		// the force loop (per-instance energy window in HBM) the stack stays with the general kernel
		if (T.dev.type == saip::TASK_MOTION_FORCE && (T.dev.cl_force || T.dev.cl_moment) && T.dev.popc_enabled) return false;
		if (T.dev.decoupling == SAIP_BOUNDED_INERTIA_ESTIMATES) {
			if (thr >= 0.0 && thr != T.dev.bie_threshold) return false;
			thr = T.dev.bie_threshold;
		}
	}
	return nmf <= 2;
}
// wavefront-per-instance kernel (saip_kernel_wave.hip): chains of 9..32 dof, any stack of motion-force tasks (rank >= 2) and joint tasks.
// A passivity controller around a force loop keeps its stack on the general kernel: its observer state in HBM is advanced by the control
// law itself, so an instance this kernel flags late in the hierarchy would advance it twice when the general kernel recomputes it.
static bool wave_eligible(const saip_batch* b) {
	const int n = b->model->n;
	if (n <= 8 || n > 32) return false;
	for (auto& T : b->tasks) {
		if (T.dev.type == saip::TASK_MOTION_FORCE && (T.dev.cl_force || T.dev.cl_moment) && T.dev.popc_enabled) return false;
		if (T.dev.type == saip::TASK_MOTION_FORCE && (T.dev.k < 2 || T.dev.k > 6)) return false;
	}
	return true;
}
// eight-lanes-per-instance kernel (saip_kernel_oct.hip): 7-dof chain, { MotionForceTask, JointTask }.  Either the headline stack
// (full 6-dof motion-force task + full joint task: the joint task has rank <= 1) or any motion-force task of rank >= 2 with a joint
// task of at most four rows (general range basis); default or general control laws, no closed-loop force control.
static bool oct_eligible(const saip_batch* b, CycleParams& P) {
	const int n = b->model->n;
	if (n < 6 || n > 8 || b->tasks.size() != 2) return false;
	const TaskDev& mf = b->tasks[0].dev;
	const TaskDev& jt = b->tasks[1].dev;
	if (mf.type != saip::TASK_MOTION_FORCE || jt.type != saip::TASK_JOINT) return false;
	if (mf.cl_force || mf.cl_moment || mf.k < 2) return false;
	const bool full_mf = mf.k == 6 && mf.bm_identity;
	const bool full_jt = jt.m == n && jt.s_identity;
	if (n != 7) {
		// 6- and 8-dof chains: the general instantiation only -- full joint task behind a motion-force task, Jp = N_1 of rank n - k (<= 5) by the
		// multi-pivot Gram-Schmidt path; singularity handling on (a reduced task would need the 7-dof bookkeeping), no joint limit avoidance
		if (!full_jt || n - mf.k > 5 || n - mf.k < 0 || !mf.sing_handling || P.jla) return false;
		if (!full_mf && mf.general_law) return false;
		if (full_mf)
			for (int i = 0; i < 9; i++) {
				const double id = (i % 4 == 0) ? 1.0 : 0.0;
				if (mf.Ppos[i] != id || mf.Pori[i] != id) return false;
				if (!mf.general_law && (mf.sig_p[i] != id || mf.sig_o[i] != id)) return false;
			}
		if (!lane_eligible(b)) return false;
		P.oct_general_joint = 2;
		P.oct_partial_mf = full_mf ? 0 : 1;
		return true;
	}
	if (!full_mf && mf.general_law) return false;  // the general laws are wired for the full task only
	if (full_mf) {
		for (int i = 0; i < 9; i++) {  // full task: the projections are identities; the short control law also relies on identity sigma matrices
			const double id = (i % 4 == 0) ? 1.0 : 0.0;
			if (mf.Ppos[i] != id || mf.Pori[i] != id) return false;
			if (!mf.general_law && (mf.sig_p[i] != id || mf.sig_o[i] != id)) return false;
		}
	}
	const bool full_behind_partial = full_jt && !full_mf && mf.k <= 5;  // Jp = N_1 has rank 7 - k: multi-pivot Gram-Schmidt path
	if (!(full_mf && full_jt) && !full_behind_partial && jt.m > 4) return false;
	if (!lane_eligible(b)) return false;
	P.oct_general_joint = (full_mf && full_jt) ? 0 : (full_behind_partial ? 2 : 1);
	if (full_mf && full_jt && !mf.sing_handling && !mf.general_law) {  // disableSingularityHandling(): reduced tasks need the multi-pivot joint-task path
		P.oct_general_joint = 2;
		P.oct_truncate = 1;
	}
	P.oct_partial_mf = full_mf ? 0 : 1;
	return true;
}
// eight-lanes-per-instance kernel for hierarchies that START with a joint task (saip_kernel_octjf.hip): 7- or 8-dof chain,
// { JointTask of <= 4 rows, each selecting one joint; full 6-dof MotionForceTask in its nullspace } -- the stack of examples/06.
// Default or general (open-loop) control laws, any decoupling type, gravity compensation, torque saturation; joint limit avoidance and
// closed-loop force control keep the stack on the lane kernel.
static bool octjf_eligible(const saip_batch* b, const CycleParams& P) {
	const int n = b->model->n;
	if ((n != 7 && n != 8) || b->tasks.size() != 2 || P.jla) return false;
	const TaskDev& jt = b->tasks[0].dev;
	const TaskDev& mf = b->tasks[1].dev;
	if (jt.type != saip::TASK_JOINT || mf.type != saip::TASK_MOTION_FORCE) return false;
	if (jt.m < 1 || jt.m > 4 || n - jt.m < 6) return false;  // (fewer than six joints left: the 6-dof task behind is always singular)
	unsigned seen = 0;
	for (int a = 0; a < jt.m; a++) {  // rows of S: distinct unit vectors (S then has full row rank: matrixRangeBasis returns the identity)
		int hit = -1;
		for (int l = 0; l < n; l++) {
			const double v = jt.S[a * n + l];
			if (v == 1.0 && hit < 0) hit = l;
			else if (v != 0.0) return false;
		}
		if (hit < 0 || (seen >> hit & 1u)) return false;
		seen |= 1u << hit;
	}
	if (mf.k != 6 || !mf.bm_identity || mf.cl_force || mf.cl_moment) return false;  // (goal rows 30..35, the sensed force and moment, are only read by the closed-loop laws)
	for (int i = 0; i < 9; i++) {
		const double id = (i % 4 == 0) ? 1.0 : 0.0;
		if (mf.Ppos[i] != id || mf.Pori[i] != id) return false;
	}
	return lane_eligible(b);
}
// both OTGs of a { MotionForceTask, JointTask } stack on, initialised and with clean limits: their cycle-mode steps share one launch
static bool otg_pair_ready(saip_batch* b) {
	if (!(b->tasks.size() == 2 && b->tasks[0].otg_enabled && b->tasks[1].otg_enabled && b->tasks[0].dev.type == saip::TASK_MOTION_FORCE &&
		  b->tasks[1].dev.type == saip::TASK_JOINT && b->tasks[1].otg.gs == 8 && !b->tasks[0].otg_limits_dirty && !b->tasks[1].otg_limits_dirty &&
		  !b->tasks[0].otg.jerk && !b->tasks[1].otg.jerk))
		return false;
	for (int t = 0; t < 2; t++) {
		TaskHost& T = b->tasks[t];
		if (!T.otg_inited) return false;
		T.otg.goal_comps = T.dev.goal_comps;
		T.otg.task = b->tasks_dev + t;
		T.otg.dt = T.dev.dt;
	}
	return true;
}

static saip_status launch_cycle(saip_batch* b, bool diag) {
	CycleParams P;
	if (!diag)
		for (auto& T : b->tasks) T.sh_cycle++;  // updateControllerTaskModels: every task's model is updated once per cycle
	saip_status st = make_params(b, P, diag);
	if (st) return st;
	for (size_t t = 0; t < b->tasks.size(); t++) {
		TaskHost& T = b->tasks[t];
		if (T.otg_enabled && !T.otg_inited && (st = run_otg(b, (int)t, 1))) return st;
	}
	// { MotionForceTask, JointTask } with both OTGs on: one launch for the two cycle-mode kernels
	bool paired = false;
	if (!diag && otg_pair_ready(b)) {
		if (b->otg_prelaunched) {  // a rollout ran this step together with the previous period's integration
			b->otg_prelaunched = false;
		} else {
			hipError_t e = saip::launch_otg_pair(b->tasks[0].otg, b->tasks[1].otg, b->B, b->ld, b->stream);
			if (e != hipSuccess) return fail(SAIP_ERR_DEVICE, "OTG kernel launch failed: %s", hipGetErrorString(e));
		}
		paired = true;
	}
	for (size_t t = 0; t < b->tasks.size() && !paired; t++) {
		TaskHost& T = b->tasks[t];
		if (!T.otg_enabled) continue;
		if (!diag && (st = run_otg(b, (int)t, 0))) return st;  // a diagnostic re-launch must not advance the trajectory
	}
	bool lane_ok = false;
	if (!diag && b->kernel_choice == 3 && !oct_eligible(b, P) && !octjf_eligible(b, P)) return fail(SAIP_ERR_UNSUPPORTED, "the eight-lanes-per-instance kernel does not cover this robot/task stack");
	if (!diag && b->kernel_choice != 1 && lane_eligible(b)) {
		// slow path: instances the lane / eight-lane kernel flags (outside the fully non-singular branch) are recomputed by the general kernel
		// when a task can handle them there (blended strategies -- the reference default -- or singularity handling disabled: the task is
		// reduced to its non-singular subspace).  Flagged instances are appended to a list on the device; the general kernel launched
		// behind strides over it with a small fixed grid, no host round trip, and leaves at once when the list is empty.
		bool slow = false;
		for (auto& T : b->tasks)
			if (T.dev.type == saip::TASK_MOTION_FORCE && (!T.dev.sing_handling || T.dev.sing_strategies)) slow = true;
		if (slow) {
			int* cur = b->flag_buf + (size_t)(b->flag_parity & 1u) * (b->ld + 32);
			int* nxt = b->flag_buf + (size_t)((b->flag_parity + 1u) & 1u) * (b->ld + 32);
			P.flag_count = cur;
			P.flag_list = cur + 32;
			P.flag_count_next = nxt;
		}
		// small batches of the headline stack: eight lanes per instance (the lane kernel would leave most of the chip idle)
		// Up to which batch: 1024 wavefronts (8192 instances) are resident at once, larger launches run in rounds.  Measured against the lane
		// kernel (tools/ab_cliff.sh, round 3): the lean instantiation (config 2's stack) stays ahead up to 24 576 instances (26.8 against
		// 31.1 us) and is level at 32 768; every other instantiation -- partial tasks, reduced tasks, joint task first, 6 / 8 dof -- is ahead
		// at every size (config 3: 109 against 195 us at 65 536, 384 against 627 at 262 144; config 6: 102 against 190 at 65 536), and
		// stacks whose instances leave the non-singular branch are not a contest (the lane kernel hands those to the general kernel).
		static const char* oct_env = getenv("SAIP_OCT_MAX_BATCH");
		const bool oct_ok = oct_eligible(b, P);
		const bool oct_lean = oct_ok && P.oct_general_joint == 0 && !(P.jla || P.lane_general || P.oct_partial_mf);
		const int oct_max = oct_env ? atoi(oct_env) : (oct_lean ? 24576 : INT_MAX);
		const bool oct = oct_ok && (b->kernel_choice == 3 || (b->kernel_choice == 0 && b->B <= oct_max));
		const bool octjf = !oct && octjf_eligible(b, P) && (b->kernel_choice == 3 || (b->kernel_choice == 0 && b->B <= oct_max));
		// the eight-lane kernel runs the blended singularity strategies of the headline stack itself (and passes a fully singular task
		// through): with the handling enforced nothing is left for a slow path, and what it still refuses the general kernel would too
		if (oct && P.oct_general_joint == 0 && b->tasks[0].dev.sing_handling) {
			slow = false;
			P.flag_count = P.flag_count_next = P.flag_list = nullptr;
		}
		// every other eight-lane stack: the wavefront that flags an instance recomputes it itself behind its epilogue (the general kernel's body on
		// its own LDS block) -- no list and no second launch behind the kernel (round 4; SAIP_NO_SLOW_TAIL=1 restores the list launch for A/B runs)
		static const bool tail_on = getenv("SAIP_NO_SLOW_TAIL") == nullptr;
		P.slow_tail = 0;
		if (slow && tail_on && !b->flagged_on_list && ((oct && P.oct_general_joint != 0) || octjf)) {
			slow = false;
			P.slow_tail = 1;
			P.flag_count = P.flag_count_next = P.flag_list = nullptr;
		}
		b->sim_done = false;
		if (b->sim_want > 0 && oct && !slow && P.oct_general_joint == 0) {  // nothing recomputes torques behind this launch: it can integrate as well
			P.sim_substeps = b->sim_want;
			P.sim_dt = b->sim_dt;
			P.sim_damping = b->sim_damping;
			for (int i = 0; i < 3; i++) P.sim_gravity[i] = b->sim_gravity[i];
			b->sim_done = true;
		}
		if (slow && !b->flag_clean[b->flag_parity & 1u]) {
			// the pair this launch appends to was not zeroed by the cycle before (a cycle without a slow path, another kernel choice, or a
			// failed launch came in between): zero its count here, or stale entries would be recomputed a second time
			HIP_TRY(hipMemsetAsync(P.flag_count, 0, sizeof(int), b->stream));
			b->flag_clean[b->flag_parity & 1u] = true;
		}
		hipError_t e = oct ? saip::launch_cycle_oct(P, b->stream) : (octjf ? saip::launch_cycle_octjf(P, b->stream) : saip::launch_cycle_lane(P, b->stream, &lane_ok));
		if (oct || octjf) lane_ok = true;
		if (slow && lane_ok && e == hipSuccess) {  // the kernel appends to the current pair and zeroes the other one
			b->flag_clean[b->flag_parity & 1u] = false;
			b->flag_clean[(b->flag_parity + 1u) & 1u] = true;
			b->flag_parity++;
		}
		if (lane_ok) {
			if (e != hipSuccess) return fail(SAIP_ERR_DEVICE, "lane kernel launch failed: %s", hipGetErrorString(e));
			b->kernel_name = oct ? "saip_cycle_oct" : (octjf ? "saip_cycle_octjf" : "saip_cycle_lane");
			if (slow) {
				hipError_t e2 = saip::launch_cycle_wg_list(P, b->stream);
				if (e2 != hipSuccess) return fail(SAIP_ERR_DEVICE, "slow-path kernel launch failed: %s", hipGetErrorString(e2));
			}
			return SAIP_OK;
		}
	}
	if (!diag && b->kernel_choice == 2) return fail(SAIP_ERR_UNSUPPORTED, "the lane-per-instance kernel does not cover this robot/task stack");
	if (!diag && b->kernel_choice == 4 && !wave_eligible(b)) return fail(SAIP_ERR_UNSUPPORTED, "the wavefront-per-instance kernel covers chains of 9 to 32 dof without a passivity controller");
	if (!diag && (b->kernel_choice == 0 || b->kernel_choice == 4) && wave_eligible(b)) {
		// chains of 9..32 dof: one wavefront per instance, matrices in MFMA operand form (saip_kernel_wave.hip).  What it cannot certify
		// (a task outside the non-singular branch, an ambiguous rank gap, ...) it leaves untouched on the device-side work list; the
		// general kernel behind recomputes those instances -- an empty list costs that launch one scalar load per workgroup
		int* cur = b->flag_buf + (size_t)(b->flag_parity & 1u) * (b->ld + 32);
		int* nxt = b->flag_buf + (size_t)((b->flag_parity + 1u) & 1u) * (b->ld + 32);
		P.flag_count = cur;
		P.flag_list = cur + 32;
		P.flag_count_next = nxt;
		if (!b->flag_clean[b->flag_parity & 1u]) {
			HIP_TRY(hipMemsetAsync(P.flag_count, 0, sizeof(int), b->stream));
			b->flag_clean[b->flag_parity & 1u] = true;
		}
		hipError_t e = saip::launch_cycle_wave(P, b->stream);
		if (e != hipSuccess) return fail(SAIP_ERR_DEVICE, "wavefront-per-instance kernel launch failed: %s", hipGetErrorString(e));
		b->flag_clean[b->flag_parity & 1u] = false;
		b->flag_clean[(b->flag_parity + 1u) & 1u] = true;
		b->flag_parity++;
		b->kernel_name = "saip_cycle_wave";
		hipError_t e2 = saip::launch_cycle_wg_list(P, b->stream);
		if (e2 != hipSuccess) return fail(SAIP_ERR_DEVICE, "slow-path kernel launch failed: %s", hipGetErrorString(e2));
		return SAIP_OK;
	}
	hipError_t e = saip::launch_cycle_wg(P, b->stream);
	if (e != hipSuccess) return fail(SAIP_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
	b->kernel_name = P.n <= 8 ? "saip_cycle_wg<8,64>" : "saip_cycle_wg<32,512>";
	return SAIP_OK;
}

static saip_status launch_reinit_masked(saip_batch* b, int task, int mask) {
	std::vector<bool> otg;
	for (auto& T : b->tasks) otg.push_back(T.otg_enabled), T.otg_enabled = false;  // reinit does not depend on the OTG flag
	CycleParams P;
	saip_status st = make_params(b, P, false);
	bool any_otg = false;
	for (size_t i = 0; i < b->tasks.size(); i++) {
		b->tasks[i].otg_enabled = otg[i];
		any_otg = any_otg || otg[i];
	}
	if (any_otg) b->config_dirty = true;  // the upload above pointed the tasks at the raw goal rows: the next cycle must re-point them at the OTG output
	if (st) return st;
	P.reinit_task = task;
	P.reinit_mask = mask;
	hipError_t e = saip::launch_reinit(P, b->stream);
	if (e != hipSuccess) return fail(SAIP_ERR_DEVICE, "reinit launch failed: %s", hipGetErrorString(e));
	for (size_t t = 0; t < b->tasks.size(); t++) {  // JointTask::reInitializeTask -> _otg->reInitialize (JointTask.cpp:106)
		TaskHost& T = b->tasks[t];
		if ((task >= 0 && (int)t != task) || !T.otg_enabled) continue;
		const int m3 = T.dev.type == saip::TASK_JOINT ? ((mask & 1) ? 3 : 0) : (mask & 3);  // linear / angular parts (reInitializeLinear / Angular)
		if (m3 == 0) continue;
		if ((st = ensure_otg(b, T))) return st;
		if (m3 != 3 && !T.otg_inited && (st = run_otg(b, (int)t, 1, 3))) return st;
		if ((st = run_otg(b, (int)t, 1, m3))) return st;
	}
	HIP_TRY(hipStreamSynchronize(b->stream));
	return SAIP_OK;
}
extern "C" saip_status saip_batch_reinitialize_tasks(saip_batch* b) {
	saip_status st = need_ready(b, "saip_batch_reinitialize_tasks");
	if (st) return st;
	return launch_reinit_masked(b, -1, 7);
}

// MotionForceTask::getCurrentPosition / getCurrentOrientation (MotionForceTask.h:121-138) for the state last pushed: pos [3][B], rot [9][B]
extern "C" saip_status saip_batch_get_current_pose_host(saip_batch* b, int task, double* pos, double* rot) {
	saip_status st = need_type(b, task, saip::TASK_MOTION_FORCE, "saip_batch_get_current_pose_host");
	if (st) return st;
	if ((st = need_ready(b, "saip_batch_get_current_pose_host"))) return st;
	if (!b->pose_dev && (st = dev_alloc(b, &b->pose_dev, (size_t)12 * b->ld))) return st;
	CycleParams P;
	if ((st = make_params(b, P, false))) return st;
	hipError_t e = saip::launch_pose(P, task, b->pose_dev, b->stream);
	if (e != hipSuccess) return fail(SAIP_ERR_DEVICE, "pose kernel launch failed: %s", hipGetErrorString(e));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (pos && (st = copy_d2h(b, pos, b->pose_dev, 3))) return st;
	if (rot && (st = copy_d2h(b, rot, b->pose_dev + 3 * (size_t)b->ld, 9))) return st;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_reinitialize_task(saip_batch* b, int task) {
	saip_status st = need_ready(b, "saip_batch_reinitialize_task");
	if (st) return st;
	if (task < 0 || task >= (int)b->tasks.size()) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_reinitialize_task: bad task");
	return launch_reinit_masked(b, task, 7);
}
// MotionForceTask::resetIntegrators / Linear / Angular (MotionForceTask.cpp:988-1002), JointTask::resetIntegrators: parts bit 0 = linear
// (or the joint task's), bit 1 = angular
extern "C" saip_status saip_batch_reset_integrators(saip_batch* b, int task, int parts) {
	saip_status st = need_ready(b, "saip_batch_reset_integrators");
	if (st) return st;
	if (task < 0 || task >= (int)b->tasks.size()) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_reset_integrators: bad task");
	TaskHost& T = b->tasks[task];
	if (!T.integ_dev) return SAIP_OK;
	const size_t row = (size_t)b->ld * sizeof(double);
	if (T.dev.type == saip::TASK_JOINT) {
		if (parts & 1) HIP_TRY(hipMemsetAsync(T.integ_dev, 0, row * T.dev.m, b->stream));
	} else {
		if (parts & 1) HIP_TRY(hipMemsetAsync(T.integ_dev, 0, row * 3, b->stream));                      // position, then force (rows 6..8)
		if (parts & 1) HIP_TRY(hipMemsetAsync(T.integ_dev + 6 * (size_t)b->ld, 0, row * 3, b->stream));
		if (parts & 2) HIP_TRY(hipMemsetAsync(T.integ_dev + 3 * (size_t)b->ld, 0, row * 3, b->stream));  // orientation, then moment (rows 9..11)
		if (parts & 2) HIP_TRY(hipMemsetAsync(T.integ_dev + 9 * (size_t)b->ld, 0, row * 3, b->stream));
	}
	return SAIP_OK;
}

extern "C" saip_status saip_batch_update_task_models(saip_batch* b) {
	saip_status st = need_ready(b, "saip_batch_update_task_models");
	if (st) return st;
	// The task models are a pure function of the state set by saip_batch_set_state_host; the fused cycle kernel
	// evaluates them together with the control law when computeControlTorques is called (goals may still change).
	b->models_valid = true;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_step_async(saip_batch* b) {
	saip_status st = need_ready(b, "saip_batch_step_async");
	if (st) return st;
	return launch_cycle(b, false);
}
extern "C" saip_status saip_batch_synchronize(saip_batch* b) {
	saip_status st = need_ready(b, "saip_batch_synchronize");
	if (st) return st;
	HIP_TRY(wait_stream(b->stream));  // (kernels only: the host-copy entry points below keep the blocking wait of their hipMemcpyAsync)
	return SAIP_OK;
}
extern "C" saip_status saip_batch_get_torques_host(saip_batch* b, double* tau_host, uint8_t* status_host) {
	saip_status st = need_ready(b, "saip_batch_get_torques_host");
	if (st) return st;
	if (tau_host && (st = copy_d2h(b, tau_host, b->tau_bound ? b->tau_bound : b->tau, b->model->n))) return st;
	if (status_host) {
		HIP_TRY(hipMemcpyAsync(status_host, b->status, b->B, hipMemcpyDeviceToHost, b->stream));
		HIP_TRY(hipStreamSynchronize(b->stream));
	}
	return SAIP_OK;
}
extern "C" saip_status saip_batch_compute_control_torques(saip_batch* b, double* tau_host, uint8_t* status_host) {
	saip_status st = need_ready(b, "saip_batch_compute_control_torques");
	if (st) return st;
	if (!b->models_valid)
		return fail(SAIP_ERR_ORDER, "computeControlTorques: the robot state changed since the last updateControllerTaskModels (stale task models are not supported)");
	if ((st = launch_cycle(b, false))) return st;
	HIP_TRY(hipStreamSynchronize(b->stream));
	return saip_batch_get_torques_host(b, tau_host, status_host);
}
extern "C" saip_status saip_batch_get_task_nullspace_host(saip_batch* b, int t, double* N) {
	saip_status st = need_ready(b, "saip_batch_get_task_nullspace_host");
	if (st) return st;
	if (t < 0 || t >= (int)b->tasks.size() || !N) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task id or null output");
	const int n = b->model->n;
	TaskHost& T = b->tasks[t];
	if (!T.diag_dev && (st = dev_alloc(b, &T.diag_dev, (size_t)n * n * b->ld))) return st;
	// the diagnostic pass re-evaluates the models only (make_params: phase 1, scratch outputs): no state of the batch is advanced
	st = launch_cycle(b, true);
	b->config_dirty = true;
	if (st) return st;
	HIP_TRY(hipStreamSynchronize(b->stream));
	return copy_d2h(b, N, T.diag_dev, n * n);
}
// ------------------------------------------------------------------ per-task entry points, TemplateTask.h:43-60
// The reference's plug-in interface: a caller that builds its own hierarchy drives each task by hand
// (examples/04-task_and_redundancy/04-task_and_redundancy.cpp:141-206): updateTaskModel(N_prec), N_prec = getTaskAndPreviousNullspace(),
// computeTorques() / computeTorques(tau_prec).  One launch of the general kernel per call, restricted to the task.
static saip_status ensure_task_buffers(saip_batch* b, TaskHost& T) {
	if (T.tstatus_dev) return SAIP_OK;
	const size_t n = b->model->n, ld = b->ld;
	saip_status st;
	if ((st = dev_alloc(b, &T.nprec_dev, n * n * ld)) || (st = dev_alloc(b, &T.ntask_dev, n * n * ld)) || (st = dev_alloc(b, &T.ntot_dev, n * n * ld)) ||
		(st = dev_alloc(b, &T.ttau_dev, n * ld)) || (st = dev_alloc(b, &T.tprec_dev, n * ld)) || (st = dev_alloc(b, &T.tstatus_dev, ld)))
		return st;
	return SAIP_OK;
}
static saip_status launch_task(saip_batch* b, int t, int phase, const double* tau_prec_dev, double* tau_out_dev) {
	CycleParams P;
	saip_status st = make_params(b, P, false);
	if (st) return st;
	TaskHost& T = b->tasks[t];
	if (phase == 1) P.task_cycle[t] = ++T.sh_cycle;
	if (phase == 2 && T.otg_enabled) {
		// computeTorques steps the task's internal OTG (JointTask.cpp:313-319, MotionForceTask.cpp:394-406); updateTaskModel does not
		if (!T.otg_inited && (st = run_otg(b, t, 1))) return st;
		if ((st = run_otg(b, t, 0))) return st;
	}
	P.single_task = t;
	P.phase = phase;
	P.rank_bound[t] = T.dev.type == saip::TASK_JOINT ? T.dev.m : T.dev.k;  // N_prec is the caller's: nothing is known about the ranks above
	P.ext_nprec = T.nprec_identity ? nullptr : T.nprec_dev;
	P.ext_tau_prec = phase == 2 ? tau_prec_dev : nullptr;
	P.out_N = phase == 1 ? T.ntask_dev : nullptr;
	P.out_Ntot = phase == 1 ? T.ntot_dev : nullptr;
	P.tau = tau_out_dev ? tau_out_dev : T.ttau_dev;
	P.status = T.tstatus_dev;
	hipError_t e = saip::launch_cycle_wg(P, b->stream);
	if (e != hipSuccess) return fail(SAIP_ERR_DEVICE, "per-task kernel launch failed: %s", hipGetErrorString(e));
	return SAIP_OK;
}
static saip_status task_ready(saip_batch* b, int t, const char* fn) {
	saip_status st = need_ready(b, fn);
	if (st) return st;
	if (t < 0 || t >= (int)b->tasks.size()) return fail(SAIP_ERR_INVALID_ARGUMENT, "%s: task id %d out of range", fn, t);
	if (!b->state_pushed) return fail(SAIP_ERR_ORDER, "%s: no robot state has been set (saip_batch_set_state_host)", fn);
	return ensure_task_buffers(b, b->tasks[t]);
}
static saip_status task_update_model(saip_batch* b, int t) {
	saip_status st = launch_task(b, t, 1, nullptr, nullptr);
	if (st) return st;
	b->tasks[t].model_epoch = b->state_epoch;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_task_update_model(saip_batch* b, int t, const double* N_prec_host) {
	saip_status st = task_ready(b, t, "saip_batch_task_update_model");
	if (st) return st;
	TaskHost& T = b->tasks[t];
	T.nprec_identity = N_prec_host == nullptr;
	if (N_prec_host && (st = copy_h2d(b, T.nprec_dev, N_prec_host, b->model->n * b->model->n))) return st;
	return task_update_model(b, t);
}
extern "C" saip_status saip_batch_task_update_model_device(saip_batch* b, int t, const double* N_prec_dev) {
	saip_status st = task_ready(b, t, "saip_batch_task_update_model_device");
	if (st) return st;
	TaskHost& T = b->tasks[t];
	T.nprec_identity = N_prec_dev == nullptr;
	if (N_prec_dev && N_prec_dev != T.nprec_dev)  // the task keeps its own copy like the reference (_N_prec = N_prec)
		HIP_TRY(hipMemcpyAsync(T.nprec_dev, N_prec_dev, (size_t)b->model->n * b->model->n * b->ld * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
	return task_update_model(b, t);
}
static saip_status task_model_fresh(saip_batch* b, int t, const char* fn) {
	if (b->tasks[t].model_epoch != b->state_epoch)
		return fail(SAIP_ERR_ORDER, "%s: task [%s] has no model for the current robot state: call updateTaskModel(N_prec) first (stale task models are not supported)",
					fn, b->tasks[t].name.c_str());
	return SAIP_OK;
}
extern "C" saip_status saip_batch_task_compute_torques_device(saip_batch* b, int t, const double* tau_prec_dev, double* tau_dev) {
	saip_status st = task_ready(b, t, "saip_batch_task_compute_torques_device");
	if (st || (st = task_model_fresh(b, t, "computeTorques"))) return st;
	return launch_task(b, t, 2, tau_prec_dev, tau_dev);
}
extern "C" saip_status saip_batch_task_compute_torques(saip_batch* b, int t, const double* tau_prec_host, double* tau_host, uint8_t* status_host) {
	saip_status st = task_ready(b, t, "saip_batch_task_compute_torques");
	if (st || (st = task_model_fresh(b, t, "computeTorques"))) return st;
	TaskHost& T = b->tasks[t];
	if (tau_prec_host && (st = copy_h2d(b, T.tprec_dev, tau_prec_host, b->model->n))) return st;
	if ((st = launch_task(b, t, 2, tau_prec_host ? T.tprec_dev : nullptr, nullptr))) return st;
	if (tau_host && (st = copy_d2h(b, tau_host, T.ttau_dev, b->model->n))) return st;
	if (status_host) HIP_TRY(hipMemcpyAsync(status_host, T.tstatus_dev, b->B, hipMemcpyDeviceToHost, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return SAIP_OK;
}
extern "C" const double* saip_batch_task_device_nullspace(saip_batch* b, int t, int which) {
	if (!b || t < 0 || t >= (int)b->tasks.size() || b->tasks[t].model_epoch < 0) return nullptr;
	const TaskHost& T = b->tasks[t];
	return which == 0 ? T.ntask_dev : (which == 1 ? (T.nprec_identity ? nullptr : T.nprec_dev) : (which == 2 ? T.ntot_dev : nullptr));
}
extern "C" double* saip_batch_task_device_torques(saip_batch* b, int t) {
	return (b && t >= 0 && t < (int)b->tasks.size()) ? b->tasks[t].ttau_dev : nullptr;
}
extern "C" saip_status saip_batch_task_get_nullspaces_host(saip_batch* b, int t, double* N, double* N_prec, double* N_total) {
	saip_status st = task_ready(b, t, "saip_batch_task_get_nullspaces_host");
	if (st) return st;
	TaskHost& T = b->tasks[t];
	if (T.model_epoch < 0) return fail(SAIP_ERR_ORDER, "task [%s]: updateTaskModel has not been called", T.name.c_str());
	const int n = b->model->n;
	if (N && (st = copy_d2h(b, N, T.ntask_dev, n * n))) return st;
	if (N_total && (st = copy_d2h(b, N_total, T.ntot_dev, n * n))) return st;
	if (N_prec) {
		if (T.nprec_identity) {
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++)
					for (int k = 0; k < b->B; k++) N_prec[((size_t)i * n + j) * b->B + k] = (i == j) ? 1.0 : 0.0;
		} else if ((st = copy_d2h(b, N_prec, T.nprec_dev, n * n))) {
			return st;
		}
	}
	return SAIP_OK;
}
// resident state shared between batches (tasks driven by hand live in batches of their own): device-to-device copy of q, dq [dof][ld]
extern "C" saip_status saip_batch_set_state_device(saip_batch* b, const double* q_dev, const double* dq_dev) {
	saip_status st = need_ready(b, "saip_batch_set_state_device");
	if (st) return st;
	if (!q_dev || !dq_dev) return fail(SAIP_ERR_INVALID_ARGUMENT, "null state pointer");
	const size_t bytes = (size_t)b->model->n * b->ld * sizeof(double);
	b->models_valid = false;
	b->state_epoch++;
	if (q_dev != b->q) HIP_TRY(hipMemcpyAsync(b->q, q_dev, bytes, hipMemcpyDeviceToDevice, b->stream));
	if (dq_dev != b->dq) HIP_TRY(hipMemcpyAsync(b->dq, dq_dev, bytes, hipMemcpyDeviceToDevice, b->stream));
	b->state_pushed = true;
	return SAIP_OK;
}
// everything enqueued so far on `producer`'s stream happens before what is enqueued on `waiter`'s stream from now on (device-side
// ordering, no host synchronisation): lets one batch consume device arrays another batch has just written
extern "C" saip_status saip_batch_wait_for(saip_batch* waiter, saip_batch* producer) {
	saip_status st = need_ready(waiter, "saip_batch_wait_for");
	if (st || (st = need_ready(producer, "saip_batch_wait_for"))) return st;
	if (waiter == producer) return SAIP_OK;
	if (waiter->device != producer->device) return fail(SAIP_ERR_UNSUPPORTED, "saip_batch_wait_for: batches live on different devices");
	if (!producer->sync_event) HIP_TRY(hipEventCreateWithFlags(&producer->sync_event, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(producer->sync_event, producer->stream));
	HIP_TRY(hipStreamWaitEvent(waiter->stream, producer->sync_event, 0));
	return SAIP_OK;
}

extern "C" saip_status saip_batch_set_kernel(saip_batch* b, int which) {
	if (!b || which < 0 || which > 4) return fail(SAIP_ERR_INVALID_ARGUMENT, "kernel selector must be 0, 1, 2, 3 or 4");
	b->kernel_choice = which;
	return SAIP_OK;
}
extern "C" const char* saip_batch_kernel_name(saip_batch* b) { return b ? b->kernel_name.c_str() : ""; }

// ---- the step after the path: forward dynamics + semi-implicit Euler on the resident state (saip_dynamics.hip)
static saip_status enqueue_integrate(saip_batch* b, double dt, int substeps, const double* gravity, double damping, bool with_next_otg = false) {
	SimParams S;
	S.B = b->B;
	S.ld = b->ld;
	S.n = b->model->n;
	S.substeps = substeps;
	S.dt = dt;
	S.damping = damping;
	for (int i = 0; i < 3; i++) S.gravity[i] = gravity ? gravity[i] : b->model->dev.gravity[i];
	S.model = b->model_dev;
	S.q = b->q;
	S.dq = b->dq;
	S.tau = b->tau_bound ? b->tau_bound : b->tau;
	S.ddq = nullptr;
	hipError_t e;
	static const bool no_fuse = getenv("SAIP_ROLLOUT_NO_FUSE") != nullptr;  // A/B measurements
	if (with_next_otg && !no_fuse && S.n == 7 && otg_pair_ready(b)) {
		// rollouts: this integration and the NEXT period's trajectory generation in one launch (they are independent)
		e = saip::launch_integrate_otg_pair(S, b->tasks[0].otg, b->tasks[1].otg, b->B, b->ld, b->stream);
		b->otg_prelaunched = true;
	} else {
		e = saip::launch_integrate(S, b->stream);
	}
	if (e != hipSuccess) return fail(SAIP_ERR_DEVICE, "integrate launch failed: %s", hipGetErrorString(e));
	b->models_valid = false;  // the state moved: like after robot->setQ(), updateControllerTaskModels() is due
	b->state_epoch++;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_integrate(saip_batch* b, double dt, int substeps, const double* gravity, double damping) {
	saip_status st = need_ready(b, "saip_batch_integrate");
	if (st) return st;
	if (!(dt > 0) || substeps < 1 || damping < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_integrate: dt > 0, substeps >= 1, damping >= 0 required");
	return enqueue_integrate(b, dt, substeps, gravity, damping);
}
// steps x { internal OTGs, control cycle, integrate } on the engine stream, no host synchronisation.  One period is 3-5 small
// launches.  Plain back-to-back stream launches are the default: they were measured FASTER than replaying a hipGraph of the period
// (68.6 vs 74.2 us per period at B = 4096, 65.7 vs 70.4 us at B = 256, tools/rollout_bench.py) -- the host enqueues far ahead of the
// device either way, and the graph adds inter-node latency.  SAIP_ROLLOUT_GRAPH=1 selects the graph path (first period eager: it
// uploads dirty task constants and initialises OTG state, which must not be baked into the graph; second period captured).
static void drop_rollout_graph(saip_batch* b) {
	if (b->rollout_exec) (void)hipGraphExecDestroy(b->rollout_exec);
	if (b->rollout_graph) (void)hipGraphDestroy(b->rollout_graph);
	b->rollout_exec = nullptr;
	b->rollout_graph = nullptr;
}
extern "C" saip_status saip_batch_rollout_async(saip_batch* b, int steps, double sim_dt, int substeps, const double* gravity, double damping) {
	saip_status st = need_ready(b, "saip_batch_rollout_async");
	if (st) return st;
	if (steps < 1 || !(sim_dt > 0) || substeps < 1 || damping < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_rollout_async: bad arguments");
	bool any_otg = false;
	for (auto& T : b->tasks) any_otg = any_otg || T.otg_enabled;
	static const bool no_fuse_sim = getenv("SAIP_ROLLOUT_NO_FUSE") != nullptr;  // A/B measurements
	auto period = [&](const bool more = false) -> saip_status {  // more: another period follows inside this call
		// no internal OTG in the stack: the cycle launch integrates the state itself when it can (eight-lane kernel, no slow path behind)
		b->sim_want = (!any_otg && !no_fuse_sim && b->model->n == 7) ? substeps : 0;
		b->sim_dt = sim_dt;
		b->sim_damping = damping;
		for (int i = 0; i < 3; i++) b->sim_gravity[i] = gravity ? gravity[i] : b->model->dev.gravity[i];
		b->sim_done = false;
		saip_status s2 = launch_cycle(b, false);
		b->sim_want = 0;
		if (s2) {
			b->otg_prelaunched = false;  // a failed period must not leave the next standalone cycle believing its OTG step has already run
			return s2;
		}
		if (b->sim_done) {  // the bookkeeping of enqueue_integrate
			b->sim_done = false;
			b->models_valid = false;
			b->state_epoch++;
			return SAIP_OK;
		}
		s2 = enqueue_integrate(b, sim_dt, substeps, gravity, damping, more);
		if (s2) b->otg_prelaunched = false;
		return s2;
	};
	static const bool use_graph = getenv("SAIP_ROLLOUT_GRAPH") != nullptr;
	bool cycle_dependent = false;  // the device-side slow path and the blended singularity strategies read per-cycle launch parameters
	for (auto& T : b->tasks)               // (list parity, CycleParams::task_cycle): not capturable
		if (T.dev.type == saip::TASK_MOTION_FORCE && (!T.dev.sing_handling || T.dev.sing_strategies)) cycle_dependent = true;
	int done = 0;
	if (use_graph && steps >= 4 && !cycle_dependent) {
		if ((st = period())) return st;  // eager: uploads, OTG initialisation
		done = 1;
		HIP_TRY(hipStreamSynchronize(b->stream));  // a graph still executing from an earlier call must not be destroyed under it
		drop_rollout_graph(b);
		HIP_TRY(hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal));
		st = period();
		hipError_t ec = hipStreamEndCapture(b->stream, &b->rollout_graph);
		if (st) return st;
		if (ec != hipSuccess) return fail(SAIP_ERR_DEVICE, "hipStreamEndCapture failed: %s", hipGetErrorString(ec));
		HIP_TRY(hipGraphInstantiate(&b->rollout_exec, b->rollout_graph, nullptr, nullptr, 0));
		for (; done < steps; done++) HIP_TRY(hipGraphLaunch(b->rollout_exec, b->stream));
		return SAIP_OK;
	}
	for (; done < steps; done++)
		if ((st = period(done + 1 < steps))) return st;
	return SAIP_OK;
}
extern "C" saip_status saip_batch_set_torques_host(saip_batch* b, const double* tau) {
	saip_status st = need_ready(b, "saip_batch_set_torques_host");
	if (st) return st;
	if (!tau) return fail(SAIP_ERR_INVALID_ARGUMENT, "null torque pointer");
	return copy_h2d(b, b->tau_bound ? b->tau_bound : b->tau, tau, b->model->n);
}
extern "C" saip_status saip_batch_get_state_host(saip_batch* b, double* q, double* dq) {
	saip_status st = need_ready(b, "saip_batch_get_state_host");
	if (st) return st;
	if (q && (st = copy_d2h(b, q, b->q, b->model->n))) return st;
	if (dq && (st = copy_d2h(b, dq, b->dq, b->model->n))) return st;
	return SAIP_OK;
}

// desired state of a task = what its control law tracks: the internal OTG's output when enabled, else the goal (JointTask.h:185-200)
extern "C" saip_status saip_batch_get_desired_host(saip_batch* b, int t, double* desired) {
	saip_status st = need_ready(b, "saip_batch_get_desired_host");
	if (st) return st;
	if ((st = check_batch(b, t, "saip_batch_get_desired_host"))) return st;
	if (t < 0 || !desired) return fail(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_get_desired_host: bad argument");
	TaskHost& T = b->tasks[t];
	const bool otg = T.otg_enabled && T.otg_alloc && T.otg_inited;
	return copy_d2h(b, desired, otg ? T.desired_dev : T.goal_dev, T.dev.goal_comps);
}
extern "C" saip_status saip_batch_get_otg_status_host(saip_batch* b, int t, int* flags, int* result) {
	saip_status st = check_batch(b, t, "saip_batch_get_otg_status_host");
	if (st) return st;
	if (t < 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad task");
	if ((st = need_ready(b, "saip_batch_get_otg_status_host"))) return st;
	TaskHost& T = b->tasks[t];
	if (!T.otg_alloc) return fail(SAIP_ERR_ORDER, "saip_batch_get_otg_status_host: the internal OTG of task [%s] has not run", T.name.c_str());
	if (flags) HIP_TRY(hipMemcpyAsync(flags, T.otg.flags, (size_t)b->B * sizeof(int), hipMemcpyDeviceToHost, b->stream));
	if (result) HIP_TRY(hipMemcpyAsync(result, T.otg.result, (size_t)b->B * sizeof(int), hipMemcpyDeviceToHost, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return SAIP_OK;
}

extern "C" saip_status saip_batch_time_steps(saip_batch* b, int steps, int warmup, double* elapsed_ms) {
	saip_status st = need_ready(b, "saip_batch_time_steps");
	if (st) return st;
	if (steps <= 0 || warmup < 0 || !elapsed_ms) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad timing arguments");
	for (int i = 0; i < warmup; i++)
		if ((st = launch_cycle(b, false))) return st;
	if (!b->time_ev[0]) {
		HIP_TRY(hipEventCreate(&b->time_ev[0]));
		HIP_TRY(hipEventCreate(&b->time_ev[1]));
	}
	hipEvent_t e0 = b->time_ev[0], e1 = b->time_ev[1];
	HIP_TRY(hipEventRecord(e0, b->stream));
	for (int i = 0; i < steps; i++)
		if ((st = launch_cycle(b, false))) return st;
	HIP_TRY(hipEventRecord(e1, b->stream));
	// one wait for the stream (not for the event and then, in the caller, for the stream or the device: each is its own ~15 us
	// marker round trip, tools/bench_overhead_probe.py), polling first (wait_stream)
	HIP_TRY(wait_stream(b->stream));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	*elapsed_ms = ms;
	return SAIP_OK;
}

extern "C" saip_status saip_batch_time_steps_begin(saip_batch* b, int steps) {
	saip_status st = need_ready(b, "saip_batch_time_steps_begin");
	if (st) return st;
	if (steps <= 0) return fail(SAIP_ERR_INVALID_ARGUMENT, "bad timing arguments");
	if (!b->time_ev[0]) {
		HIP_TRY(hipEventCreate(&b->time_ev[0]));
		HIP_TRY(hipEventCreate(&b->time_ev[1]));
	}
	HIP_TRY(hipEventRecord(b->time_ev[0], b->stream));
	for (int i = 0; i < steps; i++)
		if ((st = launch_cycle(b, false))) return st;
	HIP_TRY(hipEventRecord(b->time_ev[1], b->stream));
	return SAIP_OK;
}
extern "C" saip_status saip_batch_time_steps_end(saip_batch* b, double* elapsed_ms) {
	saip_status st = need_ready(b, "saip_batch_time_steps_end");
	if (st) return st;
	if (!elapsed_ms || !b->time_ev[0]) return fail(SAIP_ERR_ORDER, "saip_batch_time_steps_end: call saip_batch_time_steps_begin first");
	HIP_TRY(hipEventSynchronize(b->time_ev[1]));  // (returns at once when the caller has waited for the device, as it should have)
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, b->time_ev[0], b->time_ev[1]));
	*elapsed_ms = ms;
	return SAIP_OK;
}

extern "C" const char* saip_last_error(void) { return g_err.c_str(); }
extern "C" const char* saip_version(void) { return "saip 0.1 (gfx950)"; }

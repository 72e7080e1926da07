/*
 * saip_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked, imported or called by the product).
 *
 * Plain-C, single-robot, FP64 restatement of the reference's per-cycle control path, written to
 * follow the reference line by line (literal path: SVDs, explicit general inverses, duplicated
 * Jacobian/FK evaluation, per-cycle joint-limit-avoidance model update), so that it can serve as
 *   (1) the parity oracle the HIP engine is compared with in tests/ and __graft_entry__.smoke(), and
 *   (2) the "port" CPU baseline timed by bench.py's cpu_baseline leg.
 *
 * Reference lines followed (all under /root/reference/src):
 *   RobotController.cpp:68-118                 -> orc_instance_cycle()
 *   tasks/MotionForceTask.cpp:247-268          -> mf_update_task_model()
 *   tasks/MotionForceTask.cpp:270-509,892-971  -> mf_compute_torques()
 *   tasks/SingularityHandler.cpp:75-228        -> sh_update_task_model()   (non-singular branch; truncated branch with the handling disabled)
 *   tasks/SingularityHandler.cpp:100-121,146-228,230-295 -> sh_blend_update()  (blended strategies + classifySingularity, opt-in; three stated
 *                                                 conventions where the reference depends on SVD signs / an unassigned member)
 *   tasks/SingularityHandler.cpp:297-367       -> tail of mf_compute_torques()
 *   tasks/JointTask.cpp:218-283, 285-356       -> jt_update_task_model(), jt_compute_torques()
 *   tasks/JointLimitAvoidanceTask.cpp:16-36, 124-421 -> jla_update_limit_status(), jla_update_task_model(), jla_compute_torques()
 *   tasks/MotionForceTask.cpp:28,55-87,143-168 -> orc_add_motion_force_task() (partial task projector)
 * sai-model / RBDL / Eigen are NOT in /root/reference (un-vendored, unpinned @master dependency);
 * their semantics are restated from the algebraic definitions of SURVEY.md Appendix B:
 *   M(q) (here: sum over bodies of Jacobian-form kinetic energy, equal to CRBA), M^-1 (general inverse),
 *   JWorldFrame = [Jv;Jw], positionInWorld/rotationInWorld, operationalSpaceMatrices
 *   (Lambda=(J M^-1 J^T)^-1, Jbar=M^-1 J^T Lambda, N=I-Jbar J), matrixRangeBasis (SVD, tol 1e-3),
 *   orientationError(Rd,Rc) = -1/2 sum_i Rc[:,i] x Rd[:,i].
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path and cannot be
 * built here (Eigen3, sai-model, RBDL absent).  This oracle is pinned only by (i) the analytic invariants
 * of SURVEY.md Appendix E (tests/test_oracle_invariants.py) and (ii) agreement <=1e-10 with the independent
 * NumPy restatement in oracle/restatement.py (tests/golden/, tests/test_oracle.py).
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXN 32  /* max dof */
#define MAXL 40  /* max links incl. fixed */
#define MAXT 8   /* max tasks */

enum { JT_FIXED = 0, JT_REVOLUTE = 1, JT_PRISMATIC = 2 };
enum { TASK_MOTION_FORCE = 1, TASK_JOINT = 2 };
enum { FULL_DYNAMIC_DECOUPLING = 0, BOUNDED_INERTIA_ESTIMATES = 1, IMPEDANCE = 2 };
enum { ST_OK = 0, ST_SINGULAR = 1, ST_TRUNCATED = 2, ST_BLENDED = 8 };

typedef struct {
	int type;
	double dt;
	int decoupling;
	double bie_threshold;
	/* motion force */
	int link;
	double pos_in_link[3], rot_in_link[9];
	double P[36];          /* _partial_task_projection */
	int pos_range, ori_range;
	double kp_pos[3], kv_pos[3], ki_pos[3], kp_ori[3], kv_ori[3], ki_ori[3];
	double s_min, s_max, s_abs_tol;
	int sing_handling;     /* SingularityHandler::_enforce_handling_strategy (default true, SingularityHandler.cpp:61) */
	int sing_strategies, sh_force_type1;
	double sh_kp1, sh_kv1, sh_kv2; /* setSingularityHandlingGains, SingularityHandler.h:119 */   /* blended type-1 / type-2 strategies restated (sh_blend_*); otherwise such instances are flagged */
	double kv_force, kv_moment, lin_sat, ang_sat; /* MotionForceTask.h:50-66 */
	int vel_sat, force_dim, moment_dim, param_in_compliant_frame;
	/* closed-loop force / moment control, MotionForceTask.h:50-60 */
	int cl_force, cl_moment, passivity;
	double kp_force, ki_force, kp_moment, ki_moment, kff_force, kff_moment, max_force_fb, max_moment_fb;
	double Rcs[9], tcs[3]; /* _T_control_to_sensor */
	double force_axis[3], moment_axis[3];
	/* joint */
	int m;
	double S[MAXN * MAXN]; /* m x n row-major */
	double kp[MAXN], kv[MAXN], ki[MAXN];
	double sat[MAXN]; /* joint velocity saturation (uses vel_sat) */
} orc_task;

typedef struct {
	int nl, n;
	int joint_type[MAXL], dof_of_link[MAXL];
	double R0[MAXL][9], p0[MAXL][3], axis[MAXL][3];
	double mass[MAXL], com[MAXL][3], inertia[MAXL][9];
	double q_lo[MAXN], q_hi[MAXN], vel[MAXN], effort[MAXN];
	int nt;
	orc_task task[MAXT];
	int gravity_comp, jla_enabled, torque_sat;
	double gravity[3];
	/* per-instance integrator state (lazily sized) */
	int state_B;
	double* integ; /* [B][MAXT][MAXN] (motion force: 12) */
	void* popc;    /* [B][popc_tasks] passivity observer states (tasks with enablePassivity), lazily sized */
	int popc_tasks, popc_slot[MAXT];
	void* sh;      /* [B][MAXT] blended-strategy states (sh_state), allocated when a task has sing_strategies */
	int lean;      /* orc_set_lean: the closed forms of SURVEY App. A-4 / A-5 where a certificate allows them (see there), literal path otherwise */
} orc_ctx;

/* ------------------------------------------------------------------ small dense helpers (row-major) */
static void mm(const double* A, const double* B, double* C, int m, int k, int p) {
	for (int i = 0; i < m; i++)
		for (int j = 0; j < p; j++) {
			double s = 0;
			for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * p + j];
			C[i * p + j] = s;
		}
}
static void mmT(const double* A, const double* B, double* C, int m, int k, int p) { /* C = A * B^T, B is p x k */
	for (int i = 0; i < m; i++)
		for (int j = 0; j < p; j++) {
			double s = 0;
			for (int l = 0; l < k; l++) s += A[i * k + l] * B[j * k + l];
			C[i * p + j] = s;
		}
}
static void mTm(const double* A, const double* B, double* C, int k, int m, int p) { /* C = A^T * B, A is k x m, B k x p */
	for (int i = 0; i < m; i++)
		for (int j = 0; j < p; j++) {
			double s = 0;
			for (int l = 0; l < k; l++) s += A[l * m + i] * B[l * p + j];
			C[i * p + j] = s;
		}
}
static double fro(const double* A, int len) {
	double s = 0;
	for (int i = 0; i < len; i++) s += A[i] * A[i];
	return sqrt(s);
}
static void eye(double* A, int n) {
	memset(A, 0, sizeof(double) * n * n);
	for (int i = 0; i < n; i++) A[i * n + i] = 1;
}
/* general inverse, Gauss-Jordan with partial pivoting (Eigen's dynamic .inverse() is PartialPivLU) */
static int inverse(const double* A, double* Ainv, int n) {
	double W[MAXN * 2 * MAXN];
	int w = 2 * n;
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) {
			W[i * w + j] = A[i * n + j];
			W[i * w + n + j] = (i == j);
		}
	for (int c = 0; c < n; c++) {
		int piv = c;
		for (int r = c + 1; r < n; r++)
			if (fabs(W[r * w + c]) > fabs(W[piv * w + c])) piv = r;
		if (W[piv * w + c] == 0) return -1;
		if (piv != c)
			for (int j = 0; j < w; j++) {
				double t = W[c * w + j];
				W[c * w + j] = W[piv * w + j];
				W[piv * w + j] = t;
			}
		double d = 1.0 / W[c * w + c];
		for (int j = 0; j < w; j++) W[c * w + j] *= d;
		for (int r = 0; r < n; r++)
			if (r != c) {
				double f = W[r * w + c];
				if (f != 0)
					for (int j = 0; j < w; j++) W[r * w + j] -= f * W[c * w + j];
			}
	}
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) Ainv[i * n + j] = W[i * w + n + j];
	return 0;
}

/* thin SVD by one-sided (Hestenes) Jacobi.  A m x n -> U m x r, s r, V n x r, r=min(m,n), descending.
 * columns of U belonging to sigma==0 are set to zero (never used by the callers). */
static void svd_tall(const double* A, int m, int n, double* U, double* s, double* V) { /* m >= n */
	double W[MAXN * MAXN], VV[MAXN * MAXN];
	memcpy(W, A, sizeof(double) * m * n);
	eye(VV, n);
	for (int sweep = 0; sweep < 80; sweep++) {
		int rotated = 0;
		for (int p = 0; p < n - 1; p++)
			for (int q = p + 1; q < n; q++) {
				double al = 0, be = 0, ga = 0;
				for (int i = 0; i < m; i++) {
					al += W[i * n + p] * W[i * n + p];
					be += W[i * n + q] * W[i * n + q];
					ga += W[i * n + p] * W[i * n + q];
				}
				if (al == 0 || be == 0) continue;
				if (fabs(ga) <= 1e-16 * sqrt(al * be)) continue;
				rotated = 1;
				double zeta = (be - al) / (2 * ga);
				double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
				double c = 1 / sqrt(1 + t * t), sn = c * t;
				for (int i = 0; i < m; i++) {
					double a = W[i * n + p], b = W[i * n + q];
					W[i * n + p] = c * a - sn * b;
					W[i * n + q] = sn * a + c * b;
				}
				for (int i = 0; i < n; i++) {
					double a = VV[i * n + p], b = VV[i * n + q];
					VV[i * n + p] = c * a - sn * b;
					VV[i * n + q] = sn * a + c * b;
				}
			}
		if (!rotated) break;
	}
	double nrm[MAXN];
	int idx[MAXN];
	for (int j = 0; j < n; j++) {
		double a = 0;
		for (int i = 0; i < m; i++) a += W[i * n + j] * W[i * n + j];
		nrm[j] = sqrt(a);
		idx[j] = j;
	}
	for (int a = 0; a < n; a++) /* selection sort, descending */
		for (int b = a + 1; b < n; b++)
			if (nrm[idx[b]] > nrm[idx[a]]) {
				int t = idx[a];
				idx[a] = idx[b];
				idx[b] = t;
			}
	double smax = nrm[idx[0]];
	for (int j = 0; j < n; j++) {
		int c = idx[j];
		s[j] = nrm[c];
		int dead = !(nrm[c] > 1e-300 && nrm[c] > 1e-14 * smax);
		for (int i = 0; i < m; i++) U[i * n + j] = dead ? 0.0 : W[i * n + c] / nrm[c];
		for (int i = 0; i < n; i++) V[i * n + j] = VV[i * n + c];
	}
}
static void svd_thin(const double* A, int m, int n, double* U, double* s, double* V) {
	if (m >= n) {
		svd_tall(A, m, n, U, s, V);
	} else {
		double At[MAXN * MAXN];
		for (int i = 0; i < m; i++)
			for (int j = 0; j < n; j++) At[j * m + i] = A[i * n + j];
		/* A^T (n x m) = U' S V'^T  =>  A = V' S U'^T */
		svd_tall(At, n, m, V, s, U);
	}
}

/* SaiModel::matrixRangeBasis [RECALLED, SURVEY App. A-5/B]: returns number of columns (>=1); a single zero
 * column means "empty range". U is rows x cols_out row-major. */
static int matrix_range_basis(const double* A, int rows, int cols, double* U, double tol) {
	if (rows == 0 || cols == 0 || fro(A, rows * cols) < tol) {
		for (int i = 0; i < rows; i++) U[i] = 0;
		return 1;
	}
	int r = rows < cols ? rows : cols;
	double Uf[MAXN * MAXN], s[MAXN], V[MAXN * MAXN];
	svd_thin(A, rows, cols, Uf, s, V);
	if (s[0] < tol) {
		for (int i = 0; i < rows; i++) U[i] = 0;
		return 1;
	}
	int task_dof = r;
	for (int i = r - 1; i > 0; i--) {
		if (s[i] / s[0] < tol) task_dof--;
		else break;
	}
	if (task_dof == rows) {
		eye(U, rows);
		return rows;
	}
	for (int i = 0; i < rows; i++)
		for (int j = 0; j < task_dof; j++) U[i * task_dof + j] = Uf[i * r + j];
	return task_dof;
}

/* LEAN variant of matrixRangeBasis (SURVEY 8(d) "lean", App. A-5): column-pivoted Gram-Schmidt instead of the SVD.  Every use of U downstream is
 * of the form U(...)U^T, so any orthonormal basis of the same range gives the same torques; what has to agree with the literal function is the RANK
 * decision sigma_i / sigma_0 >= tol.  It is taken only when certified from both sides:
 *   dropped part:  sigma_{r+1} <= ||A - A_r||_F (residual after r columns) and sigma_0 >= largest column norm  => ratio < tol / 4
 *   kept part:     sigma_r >= sigma_min(R11) >= 1 / ||R11^-1||_F (the r chosen columns are Q R11) and sigma_0 <= ||A||_F  => ratio > 4 tol
 * Returns -1 when the certificate fails (the caller then runs the literal SVD form). */
static int matrix_range_basis_lean(const double* A, int rows, int cols, double* U, double tol) {
	if (rows == 0 || cols == 0) return -1;
	double W[MAXN * MAXN], Q[MAXN * MAXN], R[MAXN * MAXN];
	double nA = fro(A, rows * cols);
	if (nA < 4 * tol) return -1; /* near the "empty range" test: literal */
	memcpy(W, A, sizeof(double) * rows * cols);
	int r = 0, rmax = rows < cols ? rows : cols, piv[MAXN];
	double first = 0;
	for (; r < rmax; r++) {
		int best = -1;
		double bn = 0, res = 0;
		for (int j = 0; j < cols; j++) {
			double a = 0;
			for (int i = 0; i < rows; i++) a += W[i * cols + j] * W[i * cols + j];
			res += a;
			if (a > bn) { bn = a; best = j; }
		}
		bn = sqrt(bn);
		if (r == 0) first = bn;
		if (sqrt(res) < 0.25 * tol * first) break; /* everything left is certified below the tolerance */
		if (best < 0) return -1;
		piv[r] = best;
		for (int i = 0; i < rows; i++) Q[i * rmax + r] = W[i * cols + best] / bn;
		for (int j = 0; j < cols; j++) {
			double d = 0;
			for (int i = 0; i < rows; i++) d += Q[i * rmax + r] * W[i * cols + j];
			for (int i = 0; i < rows; i++) W[i * cols + j] -= d * Q[i * rmax + r];
		}
	}
	if (r == 0) return -1;
	if (r < rmax) { /* residual certificate held at the break; re-check it is not a fall-through */
		double res = fro(W, rows * cols);
		if (!(res < 0.25 * tol * first)) return -1;
	}
	/* R11 = Q^T A[:, piv] (r x r, upper triangular up to rounding); its inverse by back substitution */
	for (int a = 0; a < r; a++)
		for (int b = 0; b < r; b++) {
			double d = 0;
			for (int i = 0; i < rows; i++) d += Q[i * rmax + a] * A[i * cols + piv[b]];
			R[a * r + b] = b >= a ? d : 0;
		}
	double inv2 = 0;
	for (int b = 0; b < r; b++) { /* column b of R11^-1 */
		double x[MAXN];
		for (int a = r - 1; a >= 0; a--) {
			double d = a == b ? 1.0 : 0.0;
			for (int k2 = a + 1; k2 < r; k2++) d -= R[a * r + k2] * x[k2];
			if (R[a * r + a] == 0) return -1;
			x[a] = d / R[a * r + a];
			inv2 += x[a] * x[a];
		}
	}
	if (!(1.0 / sqrt(inv2) > 4 * tol * nA)) return -1;
	if (r == rows) {
		eye(U, rows);
		return rows;
	}
	for (int i = 0; i < rows; i++)
		for (int j = 0; j < r; j++) U[i * r + j] = Q[i * rmax + j];
	return r;
}

/* ------------------------------------------------------------------ robot model (sai-model restatement) */
typedef struct {
	double q[MAXN], dq[MAXN];
	double R[MAXL][9], o[MAXL][3]; /* world pose of every link frame */
	double M[MAXN * MAXN], Minv[MAXN * MAXN];
} robot_state;

static void rot_rpy(const double rpy[3], double R[9]) {
	double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
	/* Rz(y) Ry(p) Rx(r) */
	R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
	R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
	R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}
static void rot_axis(const double a[3], double q, double R[9]) {
	double c = cos(q), s = sin(q), v = 1 - c;
	R[0] = c + a[0] * a[0] * v;        R[1] = a[0] * a[1] * v - a[2] * s; R[2] = a[0] * a[2] * v + a[1] * s;
	R[3] = a[1] * a[0] * v + a[2] * s; R[4] = c + a[1] * a[1] * v;        R[5] = a[1] * a[2] * v - a[0] * s;
	R[6] = a[2] * a[0] * v - a[1] * s; R[7] = a[2] * a[1] * v + a[0] * s; R[8] = c + a[2] * a[2] * v;
}
static void mat3v(const double R[9], const double v[3], double out[3]) {
	for (int i = 0; i < 3; i++) out[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
static void cross3(const double a[3], const double b[3], double c[3]) {
	c[0] = a[1] * b[2] - a[2] * b[1];
	c[1] = a[2] * b[0] - a[0] * b[2];
	c[2] = a[0] * b[1] - a[1] * b[0];
}

static void update_kinematics(const orc_ctx* c, robot_state* rs) {
	double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, op[3] = {0, 0, 0};
	for (int l = 0; l < c->nl; l++) {
		double t[3], Rt[9];
		mat3v(Rp, c->p0[l], t);
		for (int i = 0; i < 3; i++) rs->o[l][i] = op[i] + t[i];
		mm(Rp, c->R0[l], Rt, 3, 3, 3);
		if (c->joint_type[l] == JT_REVOLUTE) {
			double Rq[9];
			rot_axis(c->axis[l], rs->q[c->dof_of_link[l]], Rq);
			mm(Rt, Rq, rs->R[l], 3, 3, 3);
		} else {
			memcpy(rs->R[l], Rt, sizeof(Rt));
			if (c->joint_type[l] == JT_PRISMATIC) {
				double z[3];
				mat3v(Rt, c->axis[l], z);
				for (int i = 0; i < 3; i++) rs->o[l][i] += z[i] * rs->q[c->dof_of_link[l]];
			}
		}
		memcpy(Rp, rs->R[l], sizeof(Rp));
		memcpy(op, rs->o[l], sizeof(op));
	}
}
/* JWorldFrame(link, pos_in_link): 6 x n, rows [Jv; Jw] */
static void jacobian_world(const orc_ctx* c, const robot_state* rs, int link, const double p_world[3], double* J) {
	int n = c->n;
	memset(J, 0, sizeof(double) * 6 * n);
	for (int l = 0; l <= link; l++) {
		int d = c->dof_of_link[l];
		if (d < 0) continue;
		double z[3];
		mat3v(rs->R[l], c->axis[l], z);
		if (c->joint_type[l] == JT_REVOLUTE) {
			double r[3] = {p_world[0] - rs->o[l][0], p_world[1] - rs->o[l][1], p_world[2] - rs->o[l][2]}, v[3];
			cross3(z, r, v);
			for (int i = 0; i < 3; i++) {
				J[i * n + d] = v[i];
				J[(3 + i) * n + d] = z[i];
			}
		} else {
			for (int i = 0; i < 3; i++) J[i * n + d] = z[i];
		}
	}
}
static void position_in_world(const robot_state* rs, int link, const double pos_in_link[3], double p[3]) {
	double t[3];
	mat3v(rs->R[link], pos_in_link, t);
	for (int i = 0; i < 3; i++) p[i] = rs->o[link][i] + t[i];
}
/* updateModel(): kinematics + M (body sum, equals CRBA) + M^-1 */
static void update_model(const orc_ctx* c, robot_state* rs) {
	int n = c->n;
	update_kinematics(c, rs);
	memset(rs->M, 0, sizeof(double) * n * n);
	for (int l = 0; l < c->nl; l++) {
		double pc[3], J[6 * MAXN], Iw[9], T[9], IJ[3 * MAXN];
		position_in_world(rs, l, c->com[l], pc);
		jacobian_world(c, rs, l, pc, J);
		mm(rs->R[l], c->inertia[l], T, 3, 3, 3);
		mmT(T, rs->R[l], Iw, 3, 3, 3);
		mm(Iw, J + 3 * n, IJ, 3, 3, n);
		for (int i = 0; i < n; i++)
			for (int j = 0; j < n; j++) {
				double s = 0;
				for (int k = 0; k < 3; k++) s += c->mass[l] * J[k * n + i] * J[k * n + j] + J[(3 + k) * n + i] * IJ[k * n + j];
				rs->M[i * n + j] += s;
			}
	}
	inverse(rs->M, rs->Minv, n);
}
static void joint_gravity_vector(const orc_ctx* c, const robot_state* rs, double* g) {
	int n = c->n;
	for (int i = 0; i < n; i++) g[i] = 0;
	for (int l = 0; l < c->nl; l++) {
		double pc[3], J[6 * MAXN];
		position_in_world(rs, l, c->com[l], pc);
		jacobian_world(c, rs, l, pc, J);
		for (int i = 0; i < n; i++)
			for (int k = 0; k < 3; k++) g[i] -= c->mass[l] * J[k * n + i] * c->gravity[k];
	}
}
/* operationalSpaceMatrices(J): Lambda k x k, Jbar n x k, N n x n */
static void op_space_matrices(const robot_state* rs, int n, const double* J, int k, double* Lambda, double* Jbar, double* N) {
	double T[MAXN * MAXN], Li[MAXN * MAXN], T2[MAXN * MAXN];
	mm(J, rs->Minv, T, k, n, n);
	mmT(T, J, Li, k, n, k);
	inverse(Li, Lambda, k);
	for (int i = 0; i < n; i++) /* T2 = Minv J^T (n x k) */
		for (int j = 0; j < k; j++) {
			double s = 0;
			for (int l = 0; l < n; l++) s += rs->Minv[i * n + l] * J[j * n + l];
			T2[i * k + j] = s;
		}
	mm(T2, Lambda, Jbar, n, k, k);
	mm(Jbar, J, N, n, k, n);
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) N[i * n + j] = (i == j) - N[i * n + j];
}
static void bie_inverse(const robot_state* rs, int n, double thr, double* MinvB) {
	double Mb[MAXN * MAXN];
	memcpy(Mb, rs->M, sizeof(double) * n * n);
	for (int i = 0; i < n; i++)
		if (Mb[i * n + i] < thr) Mb[i * n + i] = thr;
	inverse(Mb, MinvB, n);
}
static void orientation_error(const double Rd[9], const double Rc[9], double e[3]) {
	e[0] = e[1] = e[2] = 0;
	for (int col = 0; col < 3; col++) {
		double a[3] = {Rc[col], Rc[3 + col], Rc[6 + col]}, b[3] = {Rd[col], Rd[3 + col], Rd[6 + col]}, x[3];
		cross3(a, b, x);
		for (int i = 0; i < 3; i++) e[i] += -0.5 * x[i];
	}
}

/* POPCExplicitForceControl (src/helper_modules/POPCExplicitForceControl.cpp), per instance and task.  The reference's window is an
 * unbounded std::queue; POPC_CAP samples are kept here (ample for the test horizons; overflow aborts the step with status 4). */
#define POPC_CAP 2048
typedef struct {
	int inited;
	double PO, Ecorr, stored, Rc, vsum;
	int counter, head, size;
	double ring[POPC_CAP];
} popc_state;
static void popc_reinit(popc_state* s) { /* :10-22 */
	s->PO = s->Ecorr = s->stored = s->vsum = 0;
	s->counter = 50;
	s->Rc = 1.0;
	s->head = s->size = 0;
	s->inited = 1;
}
/* computePassivitySaturatedForce with _is_enabled, :35-95; returns Rc */
static double popc_step(popc_state* s, double dt, double kff, double kv, const double* fd, const double* fs, const double* vcl, const double* vr, int* overflow) {
	const int PO_WINDOW = 250, PO_MAX = 50;
	double Fcmd[3], vc2 = 0, p = 0;
	for (int i = 0; i < 3; i++) {
		Fcmd[i] = kff * fd[i] + s->Rc * vcl[i] - kv * vr[i];
		vc2 += vcl[i] * vcl[i];
		p += (fs[i] - fd[i]) * vcl[i] - Fcmd[i] * vr[i];
	}
	p *= dt;
	s->PO += p;
	if (s->size == POPC_CAP) { *overflow = 1; s->head = (s->head + 1) % POPC_CAP; s->size--; }
	s->ring[(s->head + s->size) % POPC_CAP] = p;
	s->size++;
	if (s->PO + s->stored + s->Ecorr > 0) {
		while (s->size > PO_WINDOW) {
			double front = s->ring[s->head];
			if (s->PO + s->Ecorr + s->stored > front) {
				if (front > 0) s->PO -= front;
				s->head = (s->head + 1) % POPC_CAP;
				s->size--;
			} else break;
		}
	}
	if (s->counter <= 0) {
		s->counter = PO_MAX;
		double old = s->Rc;
		if (s->PO + s->stored + s->Ecorr < 0) {
			s->Rc = 1 + (s->PO + s->stored + s->Ecorr) / (s->vsum * dt);
			if (s->Rc > 1) s->Rc = 1;
			if (s->Rc < 0) s->Rc = 0;
		} else {
			s->Rc = (1 + (0.1 * PO_MAX - 1) * s->Rc) / (double)(0.1 * PO_MAX);
		}
		s->Ecorr += (1 - old) * s->vsum * dt;
		s->vsum = 0;
	}
	s->counter--;
	s->vsum += vc2;
	return s->Rc;
}

/* SingularityHandler state that persists across cycles (SingularityHandler.h:197-230), per instance and motion-force task */
#define SH_BUFFER 200
#define SH_TYPE2_RATIO 1e-2
#define SH_TYPE2_ANGLE (5 * M_PI / 180)
typedef struct {
	int inited, ntypes, c1, c2, head, size;
	unsigned char hist[SH_BUFFER];
	double q_prior[MAXN], dir[MAXN];
} sh_state;

/* ------------------------------------------------------------------ per-task cached models */
typedef struct {
	double N_prec[MAXN * MAXN], N[MAXN * MAXN];
	/* motion force + singularity handler */
	double jacobian[6 * MAXN], projected_jacobian[6 * MAXN];
	int k;
	double task_range_ns[36], projected_jacobian_ns[6 * MAXN], Lambda_ns[36], Lambda_ns_modified[36];
	double sv[6];
	int singular;
	int truncated, skip;   /* handling disabled: task reduced to its non-singular part / passed through (fully singular) */
	/* blended strategies (handling enforced): singular part of the task and the joint-space strategy */
	int blend, nsg;        /* nsg = number of singular directions */
	double alpha, U_s[36], V_s[MAXN * 6], J_s[6 * MAXN], Lambda_s_mod[36], Jpp[6 * MAXN], Lambda_js_mod[36];
	sh_state* sh;
	/* joint */
	int r, empty;
	double jt_projected_jacobian[MAXN * MAXN], current_task_range[MAXN * MAXN], M_partial[MAXN * MAXN],
		M_partial_modified[MAXN * MAXN];
} task_model;


/* Blended singularity strategies (handling enforced): updateTaskModel :83-121, 146-228 and classifySingularity :230-295, with the
 * three choices documented at restatement.sh_blend (sign-invariant classification, intended type-2 torque ratio, invariance of the
 * torque formulas).  Returns 0 when the task is fully non-singular (the caller continues on the ordinary path). */
static void position_in_world(const robot_state* rs, int link, const double pos_in_link[3], double p[3]);
static void orientation_error(const double Rd[9], const double Rc[9], double e[3]);
static void lambda_modified(const orc_task* t, const robot_state* rs, int n, const double* J, int rows, const double* Lambda, double* out) {
	if (t->decoupling == FULL_DYNAMIC_DECOUPLING) memcpy(out, Lambda, sizeof(double) * rows * rows);
	else if (t->decoupling == IMPEDANCE) eye(out, rows);
	else {
		double MinvB[MAXN * MAXN], T[6 * MAXN], Li[36];
		bie_inverse(rs, n, t->bie_threshold, MinvB);
		mm(J, MinvB, T, rows, n, n);
		mmT(T, J, Li, rows, n, rows);
		inverse(Li, out, rows);
	}
}
static int sh_blend_update(const orc_ctx* c, const orc_task* t, const robot_state* rs, task_model* tm, const double* U, const double* s,
						   const double* V, int r, int k, int ns) {
	int n = c->n;
	sh_state* st = tm->sh;
	if (!st) return 0;
	if (!st->inited) { /* SingularityHandler.cpp:55-63 */
		memset(st, 0, sizeof(*st));
		for (int i = 0; i < n; i++) {
			st->q_prior[i] = 0.5 * (c->q_lo[i] + c->q_hi[i]);
			st->dir[i] = 1.0;
		}
		st->inited = 1;
	}
	if (st->ntypes == 0 || st->c2 > st->c1) memcpy(st->q_prior, rs->q, sizeof(double) * n); /* :232-235 */
	if (ns == k) { /* :238-244 */
		st->ntypes = st->c1 = st->c2 = st->head = st->size = 0;
		return 0;
	}
	const int nsg = k - ns;
	tm->blend = 1;
	tm->nsg = nsg;
	tm->k = ns;
	tm->alpha = ns == 0 ? 0.0 : fmin(fmax((s[ns] / s[0] - t->s_min) / (t->s_max - t->s_min), 0.0), 1.0);
	for (int i = 0; i < 6; i++)
		for (int j = 0; j < nsg; j++) tm->U_s[i * nsg + j] = U[i * r + ns + j];
	for (int i = 0; i < n; i++)
		for (int j = 0; j < nsg; j++) tm->V_s[i * nsg + j] = V[i * r + ns + j];
	mTm(tm->U_s, tm->projected_jacobian, tm->J_s, 6, nsg, n);
	if (ns == 0) { /* fully singular: _N = N_prec (:149-151), torques zero (:316-317) */
		memcpy(tm->N, tm->N_prec, sizeof(double) * n * n);
	} else {
		double Jbar[MAXN * 6], N_ns[MAXN * MAXN], N_js[MAXN * MAXN], Lam_s[36], Lam_js[36], T[MAXN * MAXN], T2[MAXN * MAXN], A[36];
		for (int i = 0; i < 6; i++)
			for (int j = 0; j < ns; j++) tm->task_range_ns[i * ns + j] = U[i * r + j];
		mTm(tm->task_range_ns, tm->projected_jacobian, tm->projected_jacobian_ns, 6, ns, n);
		op_space_matrices(rs, n, tm->projected_jacobian_ns, ns, tm->Lambda_ns, Jbar, N_ns);
		lambda_modified(t, rs, n, tm->projected_jacobian_ns, ns, tm->Lambda_ns, tm->Lambda_ns_modified);
		mm(tm->J_s, rs->Minv, T, nsg, n, n); /* Lambda_s = (J_s M^-1 J_s^T)^-1, :117 */
		mmT(T, tm->J_s, A, nsg, n, nsg);
		inverse(A, Lam_s, nsg);
		lambda_modified(t, rs, n, tm->J_s, nsg, Lam_s, tm->Lambda_s_mod);
		mm(N_ns, tm->N_prec, T, n, n, n); /* _posture_projected_jacobian = V_s^T N_ns N_prec, :153 */
		mTm(tm->V_s, T, tm->Jpp, n, nsg, n);
		op_space_matrices(rs, n, tm->Jpp, nsg, Lam_js, Jbar, N_js);
		lambda_modified(t, rs, n, tm->Jpp, nsg, Lam_js, tm->Lambda_js_mod);
		mm(N_js, N_ns, tm->N, n, n, n); /* :157 */
		(void)T2;
	}
	/* classifySingularity :246-294 (sign-invariant: the larger motion of q + 5 v and q - 5 v) */
	double x0[3], Rc0[9];
	position_in_world(rs, t->link, t->pos_in_link, x0);
	mm(rs->R[t->link], t->rot_in_link, Rc0, 3, 3, 3);
	int any1 = 0;
	for (int cidx = 0; cidx < nsg; cidx++) {
		double motion = 0;
		for (int sg = 0; sg < 2; sg++) {
			robot_state r2;
			memcpy(r2.q, rs->q, sizeof(double) * n);
			for (int i = 0; i < n; i++) r2.q[i] += (sg ? -5.0 : 5.0) * tm->V_s[i * nsg + cidx];
			update_kinematics(c, &r2);
			double x1[3], Rc1[9], dphi[3], d = 0;
			position_in_world(&r2, t->link, t->pos_in_link, x1);
			mm(r2.R[t->link], t->rot_in_link, Rc1, 3, 3, 3);
			orientation_error(Rc1, Rc0, dphi);
			for (int e = 0; e < 3; e++) d += (x1[e] - x0[e]) * tm->U_s[e * nsg + cidx] + dphi[e] * tm->U_s[(3 + e) * nsg + cidx];
			if (fabs(d) > motion) motion = fabs(d);
		}
		if (motion > 0.5) any1 = 1;
	}
	st->ntypes = nsg;
	unsigned char ty = any1 ? 1 : 2;
	if (st->size == SH_BUFFER) { /* pop the oldest (:285-292); pushing first and popping when size > buffer is the same thing */
		if (st->hist[st->head] == 1) st->c1--; else st->c2--;
		st->head = (st->head + 1) % SH_BUFFER;
		st->size--;
	}
	st->hist[(st->head + st->size) % SH_BUFFER] = ty;
	st->size++;
	if (ty == 1) st->c1++; else st->c2++;
	return 1;
}

/* SingularityHandler::updateTaskModel, SingularityHandler.cpp:75-228 (non-singular branch :123-141,146-148,175-193) */
static void sh_update_task_model(const orc_ctx* c, const orc_task* t, const robot_state* rs, task_model* tm) {
	int n = c->n, k = t->pos_range + t->ori_range;
	double U[36], s[6], V[MAXN * 6];
	tm->k = k;
	if (c->lean && k == 6 && n >= 6) {
		/* LEAN: for a full task U_ns is square orthogonal and cancels; the singular values only feed the branch predicates
		 * s_0 >= s_abs_tol and s_i / s_0 >= s_max.  With G = Jp Jp^T: lambda_max <= tr G, lambda_max >= tr G / 6, 1 / lambda_min <= tr G^-1,
		 * so tr G >= 6 tol^2 and tr G * tr G^-1 <= 1 / s_max^2 certify the fully non-singular branch; anything else runs the literal code. */
		double G[36], Gi[36], trG = 0, trGi = 0;
		mmT(tm->projected_jacobian, tm->projected_jacobian, G, 6, n, 6);
		if (inverse(G, Gi, 6) == 0) {
			for (int i = 0; i < 6; i++) { trG += G[i * 7]; trGi += Gi[i * 7]; }
			if (trG >= 6 * t->s_abs_tol * t->s_abs_tol && trGi > 0 && trG * trGi * t->s_max * t->s_max <= 1.0) {
				tm->singular = tm->truncated = tm->skip = tm->blend = 0;
				for (int i = 0; i < 6; i++) tm->sv[i] = NAN; /* not computed on this path */
				eye(tm->task_range_ns, 6);
				memcpy(tm->projected_jacobian_ns, tm->projected_jacobian, sizeof(double) * 6 * n);
				goto nonsingular;
			}
		}
	}
	svd_thin(tm->projected_jacobian, 6, n, U, s, V); /* :78-81 */
	int r = n < 6 ? n : 6;
	for (int i = 0; i < 6; i++) tm->sv[i] = i < r ? s[i] : 0;
	tm->singular = 0;
	tm->truncated = 0;
	tm->skip = 0;
	if (k < 2) { /* :100 loop never runs: rank-1 tasks leave stale state */
		tm->singular = 1;
		return;
	}
	tm->blend = 0;
	if (t->sing_handling && t->sing_strategies) { /* blended strategies restated: sh_blend_update decides */
		int ns = k;
		if (s[0] < t->s_abs_tol) ns = 0;
		else
			for (int i = 1; i < k; i++)
				if (s[i] / s[0] < t->s_max) { ns = i; break; }
		if (sh_blend_update(c, t, rs, tm, U, s, V, r, k, ns)) return;
	}
	if (s[0] < t->s_abs_tol) { /* :83 fully singular */
		if (t->sing_handling) {
			tm->singular = 1;
			return;
		}
		/* handling disabled: computeTorques returns zeros (:316-317); _N = _N_ns (:146-148) is a stale matrix in the reference,
		 * restated as the identity (the task is passed through) */
		tm->truncated = tm->skip = 1;
		eye(tm->N, n);
		return;
	}
	for (int i = 1; i < k; i++) {
		double inv_condition_number = s[i] / s[0];
		if (inv_condition_number < t->s_max) { /* :103 blending region */
			if (t->sing_handling) {
				tm->singular = 1; /* needs the blended type-1 / type-2 strategies: not restated */
				return;
			}
			/* disableSingularityHandling(): _N = _N_ns (:146-148), tau = tau_ns (:319-330) with _task_range_ns = U.leftCols(i) (:108) */
			tm->truncated = 1;
			tm->k = k = i;
			break;
		}
	}
	/* :123-141 fully non-singular: alpha = 1 (or the non-singular part, :107-115) */
	for (int i = 0; i < 6; i++)
		for (int j = 0; j < k; j++) tm->task_range_ns[i * k + j] = U[i * r + j]; /* _svd_U.leftCols(...) */
	mTm(tm->task_range_ns, tm->projected_jacobian, tm->projected_jacobian_ns, 6, k, n);
nonsingular:;
	double Jbar[MAXN * 6];
	op_space_matrices(rs, n, tm->projected_jacobian_ns, k, tm->Lambda_ns, Jbar, tm->N); /* _N = _N_ns :146-148 */
	switch (t->decoupling) {
		case FULL_DYNAMIC_DECOUPLING: memcpy(tm->Lambda_ns_modified, tm->Lambda_ns, sizeof(double) * k * k); break;
		case IMPEDANCE: eye(tm->Lambda_ns_modified, k); break;
		default: { /* BOUNDED_INERTIA_ESTIMATES :175-193 */
			double MinvB[MAXN * MAXN], T[6 * MAXN], Li[36];
			bie_inverse(rs, n, t->bie_threshold, MinvB);
			mm(tm->projected_jacobian_ns, MinvB, T, k, n, n);
			mmT(T, tm->projected_jacobian_ns, Li, k, n, k);
			inverse(Li, tm->Lambda_ns_modified, k);
		}
	}
}
/* MotionForceTask::updateTaskModel, MotionForceTask.cpp:247-268 */
static void mf_update_task_model(const orc_ctx* c, const orc_task* t, const robot_state* rs, const double* N_prec, task_model* tm) {
	int n = c->n;
	double p[3], Jw[6 * MAXN];
	memcpy(tm->N_prec, N_prec, sizeof(double) * n * n);
	position_in_world(rs, t->link, t->pos_in_link, p);
	jacobian_world(c, rs, t->link, p, Jw);
	mm(t->P, Jw, tm->jacobian, 6, 6, n);                       /* :261-263 */
	mm(tm->jacobian, tm->N_prec, tm->projected_jacobian, 6, n, n); /* :264 */
	sh_update_task_model(c, t, rs, tm);                        /* :266 */
}
/* sigmaForce / sigmaMoment, MotionForceTask.cpp:892-925, 932-966 */
static void sigma_space(const double Pb[9], int dim, const double axis[3], const double rot[9], double out[9]) {
	if (dim == 0) {
		memset(out, 0, 9 * sizeof(double));
		return;
	}
	if (dim == 3) {
		memcpy(out, Pb, 9 * sizeof(double));
		return;
	}
	double a[3], nrm = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]), an[3] = {axis[0] / nrm, axis[1] / nrm, axis[2] / nrm};
	mat3v(rot, an, a);
	double in[9], T[9];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) in[i * 3 + j] = dim == 1 ? a[i] * a[j] : (i == j) - a[i] * a[j];
	mm(Pb, in, T, 3, 3, 3);
	mmT(T, Pb, out, 3, 3, 3);
}
/* MotionForceTask::computeTorques(), :278-509, with sigma_force = sigma_moment = 0 (defaults, :897-899, :937-939),
 * OTG disabled (desired = goal, :385-391), velocity saturation off, open-loop force control. */
static void mf_compute_torques(const orc_ctx* c, const orc_task* t, const robot_state* rs, task_model* tm, const double* goal,
								double* integ, popc_state* popc, double* tau) {
	int n = c->n, k = tm->k;
	double p[3], Jw[6 * MAXN], Rc[9];
	position_in_world(rs, t->link, t->pos_in_link, p);         /* :286-287 */
	if (!c->lean) { /* LEAN keeps the Jacobians of updateTaskModel (same model state, same values) */
		jacobian_world(c, rs, t->link, p, Jw);
		mm(t->P, Jw, tm->jacobian, 6, 6, n);                       /* :280-282 (recomputed) */
		mm(tm->jacobian, tm->N_prec, tm->projected_jacobian, 6, n, n); /* :283 */
	}
	mm(rs->R[t->link], t->rot_in_link, Rc, 3, 3, 3);           /* :288-289 */
	const double *xd = goal, *Rd = goal + 3, *vd = goal + 12, *wd = goal + 15, *ad = goal + 18, *ald = goal + 21;
	double v[3] = {0, 0, 0}, w[3] = {0, 0, 0};
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < n; j++) {
			v[i] += tm->jacobian[i * n + j] * rs->dq[j];       /* :293-298 */
			w[i] += tm->jacobian[(3 + i) * n + j] * rs->dq[j];
		}
	double sig_p[9], sig_o[9], sig_f[9], sig_m[9], Pp[9], Po[9], rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			Pp[i * 3 + j] = t->P[i * 6 + j];
			Po[i * 3 + j] = t->P[(3 + i) * 6 + 3 + j];
		}
	if (t->param_in_compliant_frame) memcpy(rot, Rc, sizeof(rot)); /* :894-897, 934-937 */
	sigma_space(Pp, t->force_dim, t->force_axis, rot, sig_f);   /* sigmaForce :892-925 */
	sigma_space(Po, t->moment_dim, t->moment_axis, rot, sig_m); /* sigmaMoment :932-966 */
	{
		double ImS[9], T[9];
		for (int i = 0; i < 9; i++) ImS[i] = ((i % 4) == 0) - sig_f[i];
		mm(Pp, ImS, T, 3, 3, 3);
		mmT(T, Pp, sig_p, 3, 3, 3); /* sigmaPosition :927-930 */
		for (int i = 0; i < 9; i++) ImS[i] = ((i % 4) == 0) - sig_m[i];
		mm(Po, ImS, T, 3, 3, 3);
		mmT(T, Po, sig_o, 3, 3, 3); /* sigmaOrientation :968-971 */
	}
	double goal_force[3] = {0, 0, 0}, goal_moment[3] = {0, 0, 0}, tmp[3], e[3], F_um[6];
	if (t->force_dim || t->moment_dim) { /* getGoalForce / getGoalMoment :755-769 */
		mat3v(rot, goal + 24, goal_force);
		mat3v(rot, goal + 27, goal_moment);
	}
	double F_f[6];
	double f_sw[3] = {0, 0, 0}, m_sw[3] = {0, 0, 0};
	if (t->cl_force || t->cl_moment) { /* updateSensedForceAndMoment :805-828 */
		double f_c[3], m_c[3], x[3];
		mat3v(t->Rcs, goal + 30, f_c);
		mat3v(t->Rcs, goal + 33, m_c);
		cross3(t->tcs, f_c, x);
		for (int i = 0; i < 3; i++) m_c[i] += x[i];
		mat3v(Rc, f_c, f_sw);
		mat3v(Rc, m_c, m_sw);
	}
	if (t->cl_force) { /* :327-349; POPC disabled (POPCExplicitForceControl.cpp:31-33): vcl - kv_force * vr */
		double d[3], sd[3], fb[3], u[3], nrm = 0;
		for (int i = 0; i < 3; i++) d[i] = f_sw[i] - goal_force[i];
		mat3v(sig_f, d, sd);
		for (int i = 0; i < 3; i++) integ[6 + i] += sd[i] * t->dt;
		for (int i = 0; i < 3; i++) u[i] = -t->kp_force * d[i] - t->ki_force * integ[6 + i];
		mat3v(sig_f, u, fb);
		for (int i = 0; i < 3; i++) nrm += fb[i] * fb[i];
		nrm = sqrt(nrm);
		if (nrm > t->max_force_fb)
			for (int i = 0; i < 3; i++) fb[i] *= t->max_force_fb / nrm;
		double vcl[3], vr[3], rc = 1.0;
		mat3v(sig_f, fb, vcl);
		mat3v(sig_f, v, vr);
		if (t->passivity && popc) { /* enablePassivity(): :344-348 */
			double fd[3], fs[3];
			int ovf = 0;
			mat3v(sig_f, goal_force, fd);
			mat3v(sig_f, f_sw, fs);
			if (!popc->inited) popc_reinit(popc);
			rc = popc_step(popc, t->dt, t->kff_force, t->kv_force, fd, fs, vcl, vr, &ovf);
		}
		for (int i = 0; i < 3; i++) F_f[i] = rc * vcl[i] - t->kv_force * vr[i];
	} else { /* open loop :350-354 */
		for (int i = 0; i < 3; i++) tmp[i] = -t->kv_force * v[i];
		mat3v(sig_f, tmp, F_f);
	}
	if (t->cl_moment) { /* :357-378 */
		double d[3], sd[3], fb[3], u[3], nrm = 0;
		for (int i = 0; i < 3; i++) d[i] = m_sw[i] - goal_moment[i];
		mat3v(sig_m, d, sd);
		for (int i = 0; i < 3; i++) integ[9 + i] += sd[i] * t->dt;
		for (int i = 0; i < 3; i++) u[i] = -t->kp_moment * d[i] - t->ki_moment * integ[9 + i];
		mat3v(sig_m, u, fb);
		for (int i = 0; i < 3; i++) nrm += fb[i] * fb[i];
		nrm = sqrt(nrm);
		if (nrm > t->max_moment_fb)
			for (int i = 0; i < 3; i++) fb[i] *= t->max_moment_fb / nrm;
		for (int i = 0; i < 3; i++) tmp[i] = fb[i] - t->kv_moment * w[i];
		mat3v(sig_m, tmp, F_f + 3);
	} else { /* open loop :379-383 */
		for (int i = 0; i < 3; i++) tmp[i] = -t->kv_moment * w[i];
		mat3v(sig_m, tmp, F_f + 3);
	}
	for (int i = 0; i < 3; i++) e[i] = p[i] - xd[i];
	mat3v(sig_p, e, tmp);
	for (int i = 0; i < 3; i++) integ[i] += tmp[i] * t->dt;   /* :411-413 */
	if (t->vel_sat) { /* :416-430 */
		double se[3], vdes[3], nrm = 0;
		mat3v(sig_p, e, se);
		for (int i = 0; i < 3; i++) {
			double kvi = fabs(t->kv_pos[i]) > 1e-6 ? 1.0 / t->kv_pos[i] : 0.0; /* computePseudoInverse of the diagonal gain */
			vdes[i] = -t->kp_pos[i] * kvi * se[i] - t->ki_pos[i] * kvi * integ[i];
			nrm += vdes[i] * vdes[i];
		}
		nrm = sqrt(nrm);
		if (nrm > t->lin_sat)
			for (int i = 0; i < 3; i++) vdes[i] *= t->lin_sat / nrm;
		for (int i = 0; i < 3; i++) tmp[i] = ad[i] - t->kv_pos[i] * (v[i] - vdes[i]);
	} else {
		for (int i = 0; i < 3; i++) tmp[i] = ad[i] - t->kp_pos[i] * e[i] - t->kv_pos[i] * (v[i] - vd[i]) - t->ki_pos[i] * integ[i];
	}
	mat3v(sig_p, tmp, F_um);                                   /* :431-437 */
	double oe[3], dphi[3];
	orientation_error(Rd, Rc, oe);
	mat3v(sig_o, oe, dphi);                                    /* :441-443 */
	for (int i = 0; i < 3; i++) integ[3 + i] += dphi[i] * t->dt; /* :446 */
	if (t->vel_sat) { /* :449-462 */
		double wdes[3], nrm = 0;
		for (int i = 0; i < 3; i++) {
			double kvi = fabs(t->kv_ori[i]) > 1e-6 ? 1.0 / t->kv_ori[i] : 0.0;
			wdes[i] = -t->kp_ori[i] * kvi * dphi[i] - t->ki_ori[i] * kvi * integ[3 + i];
			nrm += wdes[i] * wdes[i];
		}
		nrm = sqrt(nrm);
		if (nrm > t->ang_sat)
			for (int i = 0; i < 3; i++) wdes[i] *= t->ang_sat / nrm;
		for (int i = 0; i < 3; i++) tmp[i] = ald[i] - t->kv_ori[i] * (w[i] - wdes[i]);
	} else {
		for (int i = 0; i < 3; i++) tmp[i] = ald[i] - t->kp_ori[i] * dphi[i] - t->kv_ori[i] * (w[i] - wd[i]) - t->ki_ori[i] * integ[3 + i];
	}
	mat3v(sig_o, tmp, F_um + 3);                               /* :463-468 */
	/* force_moment_contribution + feedforward :471-487; both parts are scaled by kff when closed-loop FORCE control is on (:484-487) */
	mat3v(sig_f, goal_force, tmp);
	for (int i = 0; i < 3; i++) F_f[i] += (t->cl_force ? t->kff_force : 1.0) * tmp[i];
	mat3v(sig_m, goal_moment, tmp);
	for (int i = 0; i < 3; i++) F_f[3 + i] += (t->cl_force ? t->kff_moment : 1.0) * tmp[i];
	/* SingularityHandler::computeTorques, :307-309 (and :316-330 with the handling strategy disabled) */
	if (tm->blend) { /* computeTorques in the singular branches, :310-367 (conventions: restatement.sh_blend_torques) */
		sh_state* st = tm->sh;
		const int ns = tm->k, nsg = tm->nsg;
		for (int i = 0; i < n; i++) tau[i] = 0;
		if (ns == 0) return; /* :316-317 */
		double a[6], b[6], f[6], tns[MAXN], ts[MAXN], tjs[MAXN], unit[MAXN], u[6], lu[6];
		mTm(tm->task_range_ns, F_um, a, 6, ns, 1);
		mTm(tm->task_range_ns, F_f, b, 6, ns, 1);
		if (t->decoupling == IMPEDANCE) { /* :311-313 */
			for (int i = 0; i < ns; i++) f[i] = a[i] + b[i];
			mTm(tm->projected_jacobian_ns, f, tau, ns, n, 1);
			return;
		}
		mm(tm->Lambda_ns_modified, a, f, ns, ns, 1);
		for (int i = 0; i < ns; i++) f[i] += b[i];
		mTm(tm->projected_jacobian_ns, f, tns, ns, n, 1);
		if (st->c1 > st->c2 || t->sh_force_type1) { /* type-1 strategy: hold the entering posture, :329-336 */
			for (int i = 0; i < n; i++) unit[i] = -t->sh_kp1 * (rs->q[i] - st->q_prior[i]) - t->sh_kv1 * rs->dq[i];
			mTm(tm->V_s, unit, u, n, nsg, 1);
			mm(tm->Lambda_js_mod, u, lu, nsg, nsg, 1);
			mTm(tm->Jpp, lu, tjs, nsg, n, 1);
		} else { /* type-2 strategy, :337-355 */
			for (int i = 0; i < n; i++)
				if (tm->V_s[i * nsg] != 0) {
					if (fabs(rs->q[i] - c->q_hi[i]) < SH_TYPE2_ANGLE) st->dir[i] = -1;
					else if (fabs(rs->q[i] - c->q_lo[i]) < SH_TYPE2_ANGLE) st->dir[i] = 1;
				}
			double fn = 0, fTd = 0;
			for (int i = 0; i < 6; i++) fn += (F_um[i] + F_f[i]) * (F_um[i] + F_f[i]);
			fn = sqrt(fn);
			for (int i = 0; i < 6; i++) fTd += (F_um[i] + F_f[i]) / fn * tm->U_s[i * nsg];
			for (int i = 0; i < n; i++) unit[i] = st->dir[i] * (fabs(fTd) * SH_TYPE2_RATIO * c->effort[i]);
			mTm(tm->V_s, unit, u, n, nsg, 1);
			mTm(tm->Jpp, u, tjs, nsg, n, 1);
			for (int i = 0; i < n; i++) unit[i] = -t->sh_kv2 * rs->dq[i];
			mTm(tm->V_s, unit, u, n, nsg, 1);
			mm(tm->Lambda_js_mod, u, lu, nsg, nsg, 1);
			mTm(tm->Jpp, lu, unit, nsg, n, 1);
			for (int i = 0; i < n; i++) tjs[i] += unit[i];
		}
		mTm(tm->U_s, F_um, a, 6, nsg, 1); /* :358-359 */
		mTm(tm->U_s, F_f, b, 6, nsg, 1);
		mm(tm->Lambda_s_mod, a, f, nsg, nsg, 1);
		for (int i = 0; i < nsg; i++) f[i] += b[i];
		mTm(tm->J_s, f, ts, nsg, n, 1);
		for (int i = 0; i < n; i++) {
			double v = ts[i];
			if (v != v) v = 0;
			else if (v > c->effort[i]) v = c->effort[i];
			else if (v < -c->effort[i]) v = -c->effort[i];
			tau[i] = tns[i] + tm->alpha * v + (1 - tm->alpha) * tjs[i];
		}
		return;
	}
	if (tm->skip) { /* fully singular task passed through: the integrators above still advanced, the torque is zero */
		for (int i = 0; i < n; i++) tau[i] = 0;
		return;
	}
	double a[6], b[6], f[6];
	mTm(tm->task_range_ns, F_um, a, 6, k, 1);
	mTm(tm->task_range_ns, F_f, b, 6, k, 1);
	mm(tm->Lambda_ns_modified, a, f, k, k, 1);
	for (int i = 0; i < k; i++) f[i] += b[i];
	mTm(tm->projected_jacobian_ns, f, tau, k, n, 1);
	/* computeTorques(tau_prec) :270-276: _Lambda is all zeros (never written) => no compensation (App. C-1) */
}
/* JointTask::updateTaskModel, JointTask.cpp:218-283 */
static void jt_update_task_model(const orc_ctx* c, const orc_task* t, const robot_state* rs, const double* N_prec, task_model* tm) {
	int n = c->n, m = t->m;
	memcpy(tm->N_prec, N_prec, sizeof(double) * n * n);
	mm(t->S, tm->N_prec, tm->jt_projected_jacobian, m, n, n); /* :231 */
	tm->r = c->lean ? matrix_range_basis_lean(tm->jt_projected_jacobian, m, n, tm->current_task_range, 1e-3) : -1;
	if (tm->r < 0) tm->r = matrix_range_basis(tm->jt_projected_jacobian, m, n, tm->current_task_range, 1e-3); /* :233 */
	tm->empty = 0;
	if (fro(tm->current_task_range, m * tm->r) == 0) {         /* :234-239 */
		eye(tm->N, n);
		tm->empty = 1;
		return;
	}
	int r = tm->r;
	double Jh[MAXN * MAXN], Jbar[MAXN * MAXN];
	mTm(tm->current_task_range, tm->jt_projected_jacobian, Jh, m, r, n);
	op_space_matrices(rs, n, Jh, r, tm->M_partial, Jbar, tm->N); /* :241-245 */
	switch (t->decoupling) {
		case FULL_DYNAMIC_DECOUPLING: memcpy(tm->M_partial_modified, tm->M_partial, sizeof(double) * r * r); break;
		case IMPEDANCE: eye(tm->M_partial_modified, r); break;
		default: { /* :253-267 */
			double MinvB[MAXN * MAXN], T[MAXN * MAXN], Li[MAXN * MAXN];
			bie_inverse(rs, n, t->bie_threshold, MinvB);
			mm(Jh, MinvB, T, r, n, n);
			mmT(T, Jh, Li, r, n, r);
			inverse(Li, tm->M_partial_modified, r);
		}
	}
}
/* JointTask::computeTorques(tau_prec), :285-356 */
static void jt_compute_torques(const orc_ctx* c, const orc_task* t, const robot_state* rs, task_model* tm, const double* goal,
								double* integ, const double* tau_prec, double* tau) {
	int n = c->n, m = t->m, r = tm->r;
	for (int i = 0; i < n; i++) tau[i] = 0;
	mm(t->S, tm->N_prec, tm->jt_projected_jacobian, m, n, n);  /* :296 */
	if (tm->empty) return;                                      /* :302-306 */
	double cur[MAXN], vel[MAXN], f[MAXN];
	mm(t->S, rs->q, cur, m, n, 1);
	mm(t->S, rs->dq, vel, m, n, 1);
	const double *qd = goal, *dqd = goal + m, *ddqd = goal + 2 * m;
	for (int i = 0; i < m; i++) {
		integ[i] += (cur[i] - qd[i]) * t->dt;                   /* :323-324 */
		if (t->vel_sat) { /* :327-341 */
			double kvi = fabs(t->kv[i]) > 1e-6 ? 1.0 / t->kv[i] : 0.0;
			double vdes = -t->kp[i] * kvi * (cur[i] - qd[i]) - t->ki[i] * kvi * integ[i];
			if (vdes > t->sat[i]) vdes = t->sat[i];
			else if (vdes < -t->sat[i]) vdes = -t->sat[i];
			f[i] = -t->kv[i] * (vel[i] - vdes);
		} else {
			f[i] = -t->kp[i] * (cur[i] - qd[i]) - t->kv[i] * (vel[i] - dqd[i]) - t->ki[i] * integ[i]; /* :342-345 */
		}
	}
	double a[MAXN], b[MAXN], fr[MAXN], g[MAXN], Ua[MAXN];
	mTm(tm->current_task_range, ddqd, a, m, r, 1);
	mTm(tm->current_task_range, f, b, m, r, 1);
	mm(tm->M_partial, a, fr, r, r, 1);
	mm(tm->M_partial_modified, b, g, r, r, 1);
	for (int i = 0; i < r; i++) fr[i] += g[i];                  /* :348-351 */
	mm(tm->current_task_range, fr, Ua, m, r, 1);
	mTm(tm->jt_projected_jacobian, Ua, tau, m, n, 1);           /* :354-355 */
	/* disturbance compensation :287-291 */
	double mt[MAXN], smt[MAXN], d[MAXN];
	mm(rs->Minv, tau_prec, mt, n, n, 1);
	mm(t->S, mt, smt, m, n, 1);
	mTm(tm->current_task_range, smt, a, m, r, 1);
	mm(tm->M_partial, a, b, r, r, 1);
	mm(tm->current_task_range, b, Ua, m, r, 1);
	mTm(tm->jt_projected_jacobian, Ua, d, m, n, 1);
	for (int i = 0; i < n; i++) tau[i] -= d[i];
}
/* JointLimitAvoidanceTask, JointLimitAvoidanceTask.cpp.  Defaults of JointLimitAvoidanceTask.h:26-35 (RobotController owns the task
 * privately, so they cannot be changed through the controller). */
enum { LS_OFF = 0, LS_POS_Z1, LS_POS_Z2, LS_VEL_Z1, LS_VEL_Z2 };
typedef struct {
	int status[MAXN], direction[MAXN], active[MAXN], na; /* direction: +1 POSITIVE, -1 NEGATIVE */
	double limit_value[MAXN], torque_limit_value[MAXN];
	double N_prec[MAXN * MAXN], S[MAXN * MAXN], U[MAXN * MAXN];
	int r, empty;
} jla_model;
static const double JLA_KV = 20.0, JLA_PZ1 = 9 * M_PI / 180.0, JLA_PZ2 = 6 * M_PI / 180.0, JLA_VZ1 = 0.5, JLA_VZ2 = 0.3, JLA_RP = 1.0,
					JLA_RV = 0.05;

/* updateLimitStatus :171-239 */
static void jla_update_limit_status(const orc_ctx* c, const robot_state* rs, jla_model* jm) {
	jm->na = 0;
	for (int i = 0; i < c->n; i++) {
		jm->status[i] = LS_OFF;
		jm->direction[i] = +1;
		jm->limit_value[i] = 0;
		jm->torque_limit_value[i] = 0;
		double q = rs->q[i], dq = rs->dq[i];
		int pos_valid = c->q_hi[i] - c->q_lo[i] > 2 * JLA_PZ1, vel_valid = c->vel[i] > 2 * JLA_VZ1; /* verifyValidityPerJoint :95-117 */
		if (pos_valid) {
			if (q > c->q_hi[i] - JLA_PZ1) { jm->direction[i] = +1; jm->limit_value[i] = c->q_hi[i]; jm->torque_limit_value[i] = c->effort[i]; jm->status[i] = LS_POS_Z1; }
			if (q > c->q_hi[i] - JLA_PZ2) jm->status[i] = LS_POS_Z2;
		}
		if (pos_valid) {
			if (q < c->q_lo[i] + JLA_PZ1) { jm->direction[i] = -1; jm->limit_value[i] = c->q_lo[i]; jm->torque_limit_value[i] = c->effort[i]; jm->status[i] = LS_POS_Z1; }
			if (q < c->q_lo[i] + JLA_PZ2) jm->status[i] = LS_POS_Z2;
		}
		if (vel_valid && (jm->status[i] == LS_OFF || jm->direction[i] == -1)) {
			if (dq > c->vel[i] - JLA_VZ1) { jm->direction[i] = +1; jm->limit_value[i] = c->vel[i]; jm->torque_limit_value[i] = c->effort[i]; jm->status[i] = LS_VEL_Z1; }
			if (dq > c->vel[i] - JLA_VZ2) jm->status[i] = LS_VEL_Z2;
		}
		if (vel_valid && (jm->status[i] == LS_OFF || jm->direction[i] == +1)) {
			if (dq < -c->vel[i] + JLA_VZ1) { jm->direction[i] = -1; jm->limit_value[i] = -c->vel[i]; jm->torque_limit_value[i] = c->effort[i]; jm->status[i] = LS_VEL_Z1; }
			if (dq < -c->vel[i] + JLA_VZ2) jm->status[i] = LS_VEL_Z2;
		}
		if (jm->status[i] != LS_OFF) jm->active[jm->na++] = i;
	}
}
/* updateTaskModel :124-169 (runs every cycle, RobotController.cpp:71, also when the controller has JLA disabled) */
static void jla_update_task_model(const orc_ctx* c, const robot_state* rs, const double* N_prec, double* N_constraints, jla_model* jm) {
	int n = c->n;
	jla_update_limit_status(c, rs, jm);
	int na = jm->na;
	memcpy(jm->N_prec, N_prec, sizeof(double) * n * n);
	double Jp[MAXN * MAXN];
	memset(jm->S, 0, sizeof(double) * (na ? na : 1) * n);
	for (int r = 0; r < na; r++) jm->S[r * n + jm->active[r]] = 1;     /* computeJointSelectionMatrix :241-252 */
	mm(jm->S, N_prec, Jp, na, n, n);                                  /* :145 */
	int r0 = matrix_range_basis(jm->S, na, n, jm->U, 1e-3);           /* :147-154 (N_unconstrained is not used by the controller) */
	(void)r0;
	jm->r = matrix_range_basis(Jp, na, n, jm->U, 1e-3);               /* :156 */
	jm->empty = fro(jm->U, na * jm->r) == 0;
	if (jm->empty) {                                                  /* :157-162 */
		eye(N_constraints, n);
		return;
	}
	double Jh[MAXN * MAXN], Lam[MAXN * MAXN], Jbar[MAXN * MAXN], N[MAXN * MAXN];
	mTm(jm->U, Jp, Jh, na, jm->r, n);
	op_space_matrices(rs, n, Jh, jm->r, Lam, Jbar, N);                /* :164-168 */
	mm(N, N_prec, N_constraints, n, n, n);
}
/* computeBlendingCoefficient :16-36 */
static double jla_blend(double z, double z1, double z2, int direction) {
	if (direction < 0) {
		if (z >= z1) return 0;
		if (z <= z2) return 1;
		return (z1 - z) / (z1 - z2);
	}
	if (z <= z1) return 0;
	if (z >= z2) return 1;
	return (z - z1) / (z2 - z1);
}
/* computeTorques(tau_tasks) :258-421 */
static void jla_compute_torques(const orc_ctx* c, const robot_state* rs, const jla_model* jm, const double* tau_tasks, double* out) {
	int n = c->n, na = jm->na;
	for (int i = 0; i < n; i++) out[i] = 0;
	if (na == 0 || jm->empty) return;                                 /* :261-263, :272-276 */
	double lat[MAXN], Jp[MAXN * MAXN];
	mm(jm->S, jm->N_prec, Jp, na, n, n);                              /* :266 */
	int cn = 0;
	for (int i = 0; i < n; i++) {
		double alpha, z1t = 0, z2t = 0, q = rs->q[i], dq = rs->dq[i], lv = jm->limit_value[i], tl = jm->torque_limit_value[i], t = tau_tasks[i];
		int d = jm->direction[i];
		if (jm->status[i] == LS_OFF) continue;
		if (d > 0) {
			switch (jm->status[i]) {
				case LS_POS_Z1:
					alpha = jla_blend(q, lv - JLA_PZ1, lv - JLA_PZ2, d);
					z1t = t - JLA_KV * dq;
					lat[cn] = (1 - alpha) * t + alpha * z1t;
					break;
				case LS_POS_Z2:
					alpha = jla_blend(q, lv - JLA_PZ2, lv, d);
					z1t = t - JLA_KV * dq;
					z2t = -tl * JLA_RP - JLA_KV * dq;
					lat[cn] = (1 - alpha) * z1t + alpha * z2t;
					break;
				case LS_VEL_Z1:
					alpha = jla_blend(dq, lv - JLA_VZ1, lv - JLA_VZ2, d);
					z1t = -JLA_KV * dq;
					lat[cn] = (1 - alpha) * t + alpha * z1t;
					break;
				default: /* VEL_Z2 */
					alpha = jla_blend(dq, lv - JLA_VZ2, lv, d);
					z1t = -JLA_KV * dq;
					z1t = fmax(fmin(z1t, tl * JLA_RV), -tl * JLA_RV);
					z2t = -alpha * tl * JLA_RV;
					lat[cn] = (1 - alpha) * z1t + alpha * z2t;
			}
		} else {
			switch (jm->status[i]) {
				case LS_POS_Z1:
					alpha = jla_blend(q, lv + JLA_PZ1, lv + JLA_PZ2, d);
					z1t = t - JLA_KV * dq;
					z1t = fmax(fmin(z1t, tl * JLA_RV), -tl * JLA_RV);
					lat[cn] = alpha * t + (1 - alpha) * z1t;          /* as written, :349-361 */
					break;
				case LS_POS_Z2:
					alpha = jla_blend(q, lv + JLA_PZ2, lv, d);
					z1t = t - JLA_KV * dq;
					z2t = tl * JLA_RP - JLA_KV * dq;
					lat[cn] = (1 - alpha) * z1t + alpha * z2t;
					break;
				case LS_VEL_Z1:
					alpha = jla_blend(dq, lv + JLA_VZ1, lv + JLA_VZ2, d);
					z1t = -JLA_KV * dq;
					z1t = fmax(fmin(z1t, tl * JLA_RV), -tl * JLA_RV);
					lat[cn] = (1 - alpha) * t + alpha * z1t;
					break;
				default:
					alpha = jla_blend(dq, lv + JLA_VZ2, lv, d);
					z1t = -JLA_KV * dq;
					z1t = fmax(fmin(z1t, tl * JLA_RV), -tl * JLA_RV);
					z2t = tl * JLA_RV;
					lat[cn] = (1 - alpha) * z1t + alpha * z2t;
			}
		}
		cn++;
	}
	double Ul[MAXN];
	mm(jm->U, lat, Ul, na, jm->r, 1);                                 /* :418-420: Jp^T U lat */
	mTm(Jp, Ul, out, na, n, 1);
}

/* ------------------------------------------------------------------ RobotController cycle for one instance */
typedef struct {
	double* N_out;      /* optional [nt][n*n] task nullspaces (N of each task) */
	double* Lambda_out; /* optional [nt][MAXN*MAXN] Lambda (k x k or r x r, row-major, leading dim = its size) */
	double* sv_out;     /* optional [nt][6] singular values of projected jacobian (motion force tasks) */
	double* M_out;      /* optional n*n */
	double* task_tau;   /* optional [nt][n] */
} orc_details;

static int orc_instance_cycle(const orc_ctx* c, const double* q, const double* dq, const double* const* goal, double* integ,
							  popc_state* popc, sh_state* sh, double* tau, orc_details* det) {
	int n = c->n;
	robot_state rs;
	task_model tm[MAXT];
	memcpy(rs.q, q, sizeof(double) * n);
	memcpy(rs.dq, dq, sizeof(double) * n);
	update_model(c, &rs); /* robot->updateModel() */
	if (det && det->M_out) memcpy(det->M_out, rs.M, sizeof(double) * n * n);

	/* updateControllerTaskModels, RobotController.cpp:68-77 */
	double N_prec[MAXN * MAXN], N_constraints[MAXN * MAXN], T[MAXN * MAXN];
	eye(N_prec, n);
	jla_model jm;
	if (c->lean && !c->jla_enabled) eye(N_constraints, n); /* LEAN: the result is unused when JLA is disabled (:96) */
	else jla_update_task_model(c, &rs, N_prec, N_constraints, &jm); /* :71-72, runs even when JLA is disabled */
	int status = ST_OK, info = ST_OK;
	for (int t = 0; t < c->nt; t++) {
		const orc_task* tk = &c->task[t];
		if (tk->type == TASK_MOTION_FORCE) {
			tm[t].sh = sh ? sh + t : NULL;
			mf_update_task_model(c, tk, &rs, N_prec, &tm[t]);
			if (tm[t].blend) info |= tm[t].k == 0 ? ST_TRUNCATED : ST_BLENDED; /* fully singular: passed through (:149-151, 316-317) */
			if (tm[t].singular) status = ST_SINGULAR;
			if (tm[t].truncated) info |= ST_TRUNCATED;
			if (det && det->sv_out) memcpy(det->sv_out + 6 * t, tm[t].sv, sizeof(double) * 6);
			if (status != ST_OK) break;
			if (det && det->Lambda_out) memcpy(det->Lambda_out + t * MAXN * MAXN, tm[t].Lambda_ns, sizeof(double) * tm[t].k * tm[t].k);
		} else {
			jt_update_task_model(c, tk, &rs, N_prec, &tm[t]);
			if (det && det->Lambda_out && !tm[t].empty)
				memcpy(det->Lambda_out + t * MAXN * MAXN, tm[t].M_partial, sizeof(double) * tm[t].r * tm[t].r);
		}
		if (det && det->N_out) memcpy(det->N_out + t * n * n, tm[t].N, sizeof(double) * n * n);
		mm(tm[t].N, tm[t].N_prec, T, n, n, n); /* getTaskAndPreviousNullspace */
		memcpy(N_prec, T, sizeof(double) * n * n);
	}
	if (status != ST_OK) {
		for (int i = 0; i < n; i++) tau[i] = NAN;
		return status;
	}
	/* computeControlTorques, RobotController.cpp:79-118 */
	double tt[MAXN];
	for (int i = 0; i < n; i++) tau[i] = 0;
	for (int t = 0; t < c->nt; t++) {
		const orc_task* tk = &c->task[t];
		if (tk->type == TASK_MOTION_FORCE)
			mf_compute_torques(c, tk, &rs, &tm[t], goal[t], integ + t * MAXN, (popc && c->popc_slot[t] >= 0) ? popc + c->popc_slot[t] : NULL, tt);
		else jt_compute_torques(c, tk, &rs, &tm[t], goal[t], integ + t * MAXN, tau, tt);
		if (det && det->task_tau) memcpy(det->task_tau + t * n, tt, sizeof(double) * n);
		for (int i = 0; i < n; i++) tau[i] += tt[i];
	}
	if (c->torque_sat) /* :86-94 */
		for (int i = 0; i < n; i++) {
			if (tau[i] > c->effort[i]) tau[i] = c->effort[i];
			else if (tau[i] < -c->effort[i]) tau[i] = -c->effort[i];
		}
	if (c->jla_enabled) { /* :96-112 */
		double tj[MAXN], nt[MAXN];
		jla_compute_torques(c, &rs, &jm, tau, tj);
		mTm(N_constraints, tau, nt, n, n, 1);
		for (int i = 0; i < n; i++) tau[i] = tj[i] + nt[i];
		if (c->torque_sat)
			for (int i = 0; i < n; i++) {
				if (tau[i] > c->effort[i]) tau[i] = c->effort[i];
				else if (tau[i] < -c->effort[i]) tau[i] = -c->effort[i];
			}
	}
	if (c->gravity_comp) { /* :114-116 */
		double g[MAXN];
		joint_gravity_vector(c, &rs, g);
		for (int i = 0; i < n; i++) tau[i] += g[i];
	}
	return info;
}

/* ------------------------------------------------------------------ exported API (ctypes) */
orc_ctx* orc_create(int n_links, const int* joint_type, const double* xyz, const double* rpy, const double* axis,
					const double* mass, const double* com, const double* inertia6, const double* q_lo, const double* q_hi,
					const double* vel, const double* effort) {
	if (n_links > MAXL) return NULL;
	orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
	c->nl = n_links;
	int n = 0;
	for (int l = 0; l < n_links; l++) {
		c->joint_type[l] = joint_type[l];
		rot_rpy(rpy + 3 * l, c->R0[l]);
		double an = sqrt(axis[3 * l] * axis[3 * l] + axis[3 * l + 1] * axis[3 * l + 1] + axis[3 * l + 2] * axis[3 * l + 2]);
		for (int i = 0; i < 3; i++) {
			c->p0[l][i] = xyz[3 * l + i];
			c->axis[l][i] = an > 0 ? axis[3 * l + i] / an : 0;
			c->com[l][i] = com[3 * l + i];
		}
		c->mass[l] = mass[l];
		const double* I = inertia6 + 6 * l; /* ixx iyy izz ixy ixz iyz */
		double Im[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]};
		memcpy(c->inertia[l], Im, sizeof(Im));
		if (joint_type[l] == JT_FIXED) {
			c->dof_of_link[l] = -1;
		} else {
			if (n >= MAXN) { free(c); return NULL; }
			c->dof_of_link[l] = n;
			c->q_lo[n] = q_lo[l]; c->q_hi[n] = q_hi[l]; c->vel[n] = vel[l]; c->effort[n] = effort[l];
			n++;
		}
	}
	c->n = n;
	c->gravity[2] = -9.81;
	return c;
}
void orc_destroy(orc_ctx* c) {
	if (c) { free(c->integ); free(c->popc); free(c->sh); free(c); }
}
int orc_dof(const orc_ctx* c) { return c->n; }

static void task_defaults(orc_task* t) {
	t->dt = 0.001;
	t->decoupling = BOUNDED_INERTIA_ESTIMATES; /* MotionForceTask.h:41-43, JointTask.h:35-37 */
	t->bie_threshold = 0.1;
}
/* MotionForceTask ctor: nt<0 && nr<0 => full task (P = I, MotionForceTask.cpp:28); otherwise :55-87 */
int orc_add_motion_force_task(orc_ctx* c, int link, const double* pos_in_link, const double* rot_in_link, const double* dirs_t,
							  int nt, const double* dirs_r, int nr) {
	if (c->nt >= MAXT || link < 0 || link >= c->nl) return -1;
	orc_task* t = &c->task[c->nt];
	memset(t, 0, sizeof(*t));
	task_defaults(t);
	t->type = TASK_MOTION_FORCE;
	t->link = link;
	memcpy(t->pos_in_link, pos_in_link, sizeof(double) * 3);
	memcpy(t->rot_in_link, rot_in_link, sizeof(double) * 9);
	if (nt < 0 && nr < 0) {
		eye(t->P, 6);
	} else {
		if (nt <= 0 && nr <= 0) return -1; /* :47-53 */
		if (nt > 8 || nr > 8) return -1;
		double bt[9] = {0}, br[9] = {0}, A[24];
		int ct = 1, cr = 1;
		if (nt > 0) { /* 3 x nt matrix whose columns are the directions */
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < nt; j++) A[i * nt + j] = dirs_t[3 * j + i];
			ct = matrix_range_basis(A, 3, nt, bt, 1e-3);
		}
		if (nr > 0) {
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < nr; j++) A[i * nr + j] = dirs_r[3 * j + i];
			cr = matrix_range_basis(A, 3, nr, br, 1e-3);
		}
		double Pt[9], Pr[9];
		mmT(bt, bt, Pt, 3, ct, 3);
		mmT(br, br, Pr, 3, cr, 3);
		memset(t->P, 0, sizeof(t->P));
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) {
				t->P[i * 6 + j] = Pt[i * 3 + j];
				t->P[(3 + i) * 6 + 3 + j] = Pr[i * 3 + j];
			}
	}
	/* initialSetup :143-152 */
	double Pp[9], Po[9], U[9];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			Pp[i * 3 + j] = t->P[i * 6 + j];
			Po[i * 3 + j] = t->P[(3 + i) * 6 + 3 + j];
		}
	int cp = matrix_range_basis(Pp, 3, 3, U, 1e-3);
	t->pos_range = fro(U, 3 * cp) == 0 ? 0 : cp;
	int co = matrix_range_basis(Po, 3, 3, U, 1e-3);
	t->ori_range = fro(U, 3 * co) == 0 ? 0 : co;
	if (t->pos_range + t->ori_range == 0) return -1;
	for (int i = 0; i < 3; i++) { /* MotionForceTask.h:44-49 */
		t->kp_pos[i] = 100.0; t->kv_pos[i] = 20.0; t->ki_pos[i] = 0.0;
		t->kp_ori[i] = 200.0; t->kv_ori[i] = 28.3; t->ki_ori[i] = 0.0;
	}
	t->s_min = 6e-3; t->s_max = 6e-2; /* MotionForceTask.cpp:197 */
	t->s_abs_tol = 1e-3;              /* SingularityHandler.cpp:11 */
	t->sing_handling = 1;             /* :61 */
	t->sh_kp1 = 50.0; t->sh_kv1 = 14.0; t->sh_kv2 = 5.0; /* KP_TYPE_1, KV_TYPE_1, KV_TYPE_2, :17-19 */
	t->kv_force = 10.0; t->kv_moment = 10.0; /* MotionForceTask.h:51,54 */
	t->lin_sat = 0.3; t->ang_sat = M_PI / 3;  /* :63-64 */
	t->force_axis[2] = 1.0; t->moment_axis[2] = 1.0;
	return c->nt++;
}
/* JointTask ctor: rows<=0 => full (S = I, JointTask.cpp:18-19) */
int orc_add_joint_task(orc_ctx* c, const double* S, int rows) {
	if (c->nt >= MAXT) return -1;
	orc_task* t = &c->task[c->nt];
	memset(t, 0, sizeof(*t));
	task_defaults(t);
	t->type = TASK_JOINT;
	if (rows <= 0) {
		t->m = c->n;
		eye(t->S, c->n);
	} else {
		if (rows > c->n) return -1;
		t->m = rows;
		memcpy(t->S, S, sizeof(double) * rows * c->n);
	}
	for (int i = 0; i < t->m; i++) { t->kp[i] = 50.0; t->kv[i] = 14.0; t->ki[i] = 0.0; t->sat[i] = M_PI / 3.0; } /* JointTask.h:32-34, 44 */
	return c->nt++;
}
void orc_set_mf_gains(orc_ctx* c, int task, const double* kp_pos, const double* kv_pos, const double* ki_pos, const double* kp_ori,
					  const double* kv_ori, const double* ki_ori) {
	orc_task* t = &c->task[task];
	for (int i = 0; i < 3; i++) {
		t->kp_pos[i] = kp_pos[i]; t->kv_pos[i] = kv_pos[i]; t->ki_pos[i] = ki_pos[i];
		t->kp_ori[i] = kp_ori[i]; t->kv_ori[i] = kv_ori[i]; t->ki_ori[i] = ki_ori[i];
	}
}
void orc_set_joint_gains(orc_ctx* c, int task, const double* kp, const double* kv, const double* ki) {
	orc_task* t = &c->task[task];
	for (int i = 0; i < t->m; i++) { t->kp[i] = kp[i]; t->kv[i] = kv[i]; t->ki[i] = ki[i]; }
}
void orc_set_decoupling(orc_ctx* c, int task, int type, double bie_threshold) {
	c->task[task].decoupling = type;
	c->task[task].bie_threshold = bie_threshold;
}
void orc_set_singularity_bounds(orc_ctx* c, int task, double s_min, double s_max) {
	c->task[task].s_min = s_min;
	c->task[task].s_max = s_max;
}
void orc_set_singularity_handling(orc_ctx* c, int task, int enabled) { c->task[task].sing_handling = enabled != 0; }
void orc_set_singularity_strategies(orc_ctx* c, int task, int enabled) {
	c->task[task].sing_strategies = enabled != 0;
	c->state_B = 0; /* state is re-sized on the next step */
}
void orc_set_singularity_gains(orc_ctx* c, int task, double kp1, double kv1, double kv2) {
	c->task[task].sh_kp1 = kp1; c->task[task].sh_kv1 = kv1; c->task[task].sh_kv2 = kv2;
}
void orc_set_all_singularities_type1(orc_ctx* c, int task, int flag) { c->task[task].sh_force_type1 = flag != 0; }
/* setType1Posture, SingularityHandler.h:140: q_des [B][n] or [n]; before the first cycle the state does not exist yet and the posture would be
 * overwritten on entering a singular region anyway (:232-235) */
void orc_set_type1_posture(orc_ctx* c, int task, const double* q_des, int per_instance) {
	if (!c->sh) return;
	for (int b = 0; b < c->state_B; b++) {
		sh_state* st = (sh_state*)c->sh + (size_t)b * MAXT + task;
		for (int i = 0; i < c->n; i++) st->q_prior[i] = per_instance ? q_des[(size_t)b * c->n + i] : q_des[i];
	}
}
void orc_set_timestep(orc_ctx* c, int task, double dt) { c->task[task].dt = dt; }
int orc_set_options(orc_ctx* c, int gravity_comp, int jla, int torque_sat) {
	c->jla_enabled = jla;
	c->gravity_comp = gravity_comp;
	c->torque_sat = torque_sat;
	return 0;
}
int orc_goal_size(const orc_ctx* c, int task) {
	const orc_task* t = &c->task[task];
	if (t->type == TASK_MOTION_FORCE) return (t->cl_force || t->cl_moment) ? 36 : ((t->force_dim || t->moment_dim) ? 30 : 24);
	return 3 * t->m;
}
void orc_set_mf_force_spaces(orc_ctx* c, int task, int force_dim, const double* force_axis, int moment_dim, const double* moment_axis,
							 int param_in_compliant_frame, double kv_force, double kv_moment) {
	orc_task* t = &c->task[task];
	t->force_dim = force_dim; t->moment_dim = moment_dim; t->param_in_compliant_frame = param_in_compliant_frame;
	t->kv_force = kv_force; t->kv_moment = kv_moment;
	for (int i = 0; i < 3; i++) { t->force_axis[i] = force_axis[i]; t->moment_axis[i] = moment_axis[i]; }
}
/* setClosedLoopForceControl / MomentControl, setForceControlGains, setFeedforward*Gain, setMax*ControlFeedbackOutput, setForceSensorFrame */
void orc_set_passivity(orc_ctx* c, int task, int enabled) { c->task[task].passivity = enabled != 0; c->state_B = 0; }
void orc_set_closed_loop_force_control(orc_ctx* c, int task, int cl_force, int cl_moment, const double* gains8, const double* Rcs, const double* tcs) {
	orc_task* t = &c->task[task];
	t->cl_force = cl_force; t->cl_moment = cl_moment;
	t->kp_force = gains8[0]; t->ki_force = gains8[1]; t->kp_moment = gains8[2]; t->ki_moment = gains8[3];
	t->kff_force = gains8[4]; t->kff_moment = gains8[5]; t->max_force_fb = gains8[6]; t->max_moment_fb = gains8[7];
	for (int i = 0; i < 9; i++) t->Rcs[i] = Rcs[i];
	for (int i = 0; i < 3; i++) t->tcs[i] = tcs[i];
}
void orc_set_velocity_saturation(orc_ctx* c, int task, int enabled, double lin_sat, double ang_sat, const double* joint_sat) {
	orc_task* t = &c->task[task];
	t->vel_sat = enabled; t->lin_sat = lin_sat; t->ang_sat = ang_sat;
	if (joint_sat) for (int i = 0; i < t->m; i++) t->sat[i] = joint_sat[i];
}
/* 1 = lean variant (closed forms of SURVEY App. A-4 / A-5 behind certificates, duplicate Jacobians and the unused JLA model skipped); 0 = literal */
void orc_set_lean(orc_ctx* c, int lean) { c->lean = lean != 0; }
void orc_reset_state(orc_ctx* c) {
	free(c->integ);
	c->integ = NULL;
	free(c->popc);
	c->popc = NULL;
	free(c->sh);
	c->sh = NULL;
	c->state_B = 0;
}

typedef struct {
	const orc_ctx* c;
	int B, b0, b1;
	const double *q, *dq;
	const double* const* goals;
	double* tau;
	unsigned char* status;
	double* integ;
} worker_arg;

static void* worker(void* p) {
	worker_arg* a = (worker_arg*)p;
	const orc_ctx* c = a->c;
	int n = c->n, B = a->B;
	double q[MAXN], dq[MAXN], tau[MAXN], gbuf[MAXT][3 * MAXN > 24 ? 3 * MAXN : 24];
	const double* gp[MAXT];
	for (int b = a->b0; b < a->b1; b++) {
		for (int i = 0; i < n; i++) { q[i] = a->q[(size_t)i * B + b]; dq[i] = a->dq[(size_t)i * B + b]; }
		for (int t = 0; t < c->nt; t++) {
			int gs = orc_goal_size(c, t);
			for (int i = 0; i < gs; i++) gbuf[t][i] = a->goals[t][(size_t)i * B + b];
			gp[t] = gbuf[t];
		}
		int st = orc_instance_cycle(c, q, dq, gp, a->integ + (size_t)b * MAXT * MAXN,
									c->popc ? (popc_state*)c->popc + (size_t)b * c->popc_tasks : NULL,
									c->sh ? (sh_state*)c->sh + (size_t)b * MAXT : NULL, tau, NULL);
		for (int i = 0; i < n; i++) a->tau[(size_t)i * B + b] = tau[i];
		if (a->status) a->status[b] = (unsigned char)st;
	}
	return NULL;
}

/* One control cycle for B instances. Layout: q,dq [n][B]; goals[t] [goal_size][B]; tau [n][B]; status [B]. */
int orc_step_batch(orc_ctx* c, int B, const double* q, const double* dq, const double* const* goals, double* tau,
				   unsigned char* status, int nthreads) {
	if (c->state_B != B) {
		free(c->integ);
		c->integ = (double*)calloc((size_t)B * MAXT * MAXN, sizeof(double));
		free(c->popc);
		c->popc = NULL;
		c->popc_tasks = 0;
		for (int t = 0; t < MAXT; t++) c->popc_slot[t] = -1;
		for (int t = 0; t < c->nt; t++)
			if (c->task[t].type == TASK_MOTION_FORCE && c->task[t].passivity) c->popc_slot[t] = c->popc_tasks++;
		if (c->popc_tasks) c->popc = calloc((size_t)B * c->popc_tasks, sizeof(popc_state));  /* inited = 0: re-initialised on first use */
		free(c->sh);
		c->sh = NULL;
		for (int t = 0; t < c->nt; t++)
			if (c->task[t].type == TASK_MOTION_FORCE && c->task[t].sing_strategies && !c->sh) c->sh = calloc((size_t)B * MAXT, sizeof(sh_state));
		c->state_B = B;
	}
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	if (nthreads > B) nthreads = B;
	pthread_t th[256];
	worker_arg wa[256];
	for (int i = 0; i < nthreads; i++) {
		wa[i] = (worker_arg){c, B, (int)((long long)B * i / nthreads), (int)((long long)B * (i + 1) / nthreads), q, dq, goals, tau, status, c->integ};
		if (nthreads == 1) worker(&wa[i]);
		else pthread_create(&th[i], NULL, worker, &wa[i]);
	}
	if (nthreads > 1)
		for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
	return 0;
}

/* Single instance with intermediate quantities (for invariant tests). Arrays are instance-contiguous. Integrators start at 0. */
int orc_step_details(const orc_ctx* c, const double* q, const double* dq, const double* const* goals, double* tau, double* N_out,
					 double* Lambda_out, double* sv_out, double* M_out, double* task_tau) {
	double integ[MAXT * MAXN];
	memset(integ, 0, sizeof(integ));
	orc_details d = {N_out, Lambda_out, sv_out, M_out, task_tau};
	return orc_instance_cycle(c, q, dq, goals, integ, NULL, NULL, tau, &d);
}
/* The reference's per-task interface for ONE instance (TemplateTask.h:43-60, driven by hand in examples/04-task_and_redundancy/
 * 04-task_and_redundancy.cpp:141-206): robot->updateModel(); task->updateTaskModel(N_prec); tau = task->computeTorques[(tau_prec)].
 * N_prec n*n row-major (NULL = identity), tau_prec n (NULL = computeTorques()).  N_out = getTaskNullspace(), Ntot_out =
 * getTaskAndPreviousNullspace().  Integrators start at 0, a fresh SingularityHandler (first cycle), no passivity state.  Returns the instance status. */
int orc_task_step(const orc_ctx* c, int task, const double* q, const double* dq, const double* N_prec_in, const double* tau_prec,
				  const double* goal, double* tau, double* N_out, double* Ntot_out) {
	int n = c->n;
	if (task < 0 || task >= c->nt) return -1;
	robot_state rs;
	task_model tm;
	double N_prec[MAXN * MAXN], integ[MAXN], zero[MAXN];
	memset(integ, 0, sizeof(integ));
	memset(zero, 0, sizeof(zero));
	memcpy(rs.q, q, sizeof(double) * n);
	memcpy(rs.dq, dq, sizeof(double) * n);
	update_model(c, &rs);
	if (N_prec_in) memcpy(N_prec, N_prec_in, sizeof(double) * n * n);
	else eye(N_prec, n);
	const orc_task* tk = &c->task[task];
	int status = ST_OK;
	sh_state sh; /* a fresh SingularityHandler: the state of the first cycle */
	memset(&sh, 0, sizeof(sh));
	if (tk->type == TASK_MOTION_FORCE) {
		tm.sh = &sh;
		mf_update_task_model(c, tk, &rs, N_prec, &tm);
		if (tm.singular) status = ST_SINGULAR;
		else if (tm.truncated) status = ST_TRUNCATED;
		else if (tm.blend) status = tm.k == 0 ? ST_TRUNCATED : ST_BLENDED;
	} else {
		jt_update_task_model(c, tk, &rs, N_prec, &tm);
	}
	if (status == ST_SINGULAR) {
		for (int i = 0; i < n; i++) tau[i] = NAN;
		return status;
	}
	if (N_out) memcpy(N_out, tm.N, sizeof(double) * n * n);
	if (Ntot_out) mm(tm.N, tm.N_prec, Ntot_out, n, n, n);
	if (tk->type == TASK_MOTION_FORCE) mf_compute_torques(c, tk, &rs, &tm, goal, integ, NULL, tau);
	else jt_compute_torques(c, tk, &rs, &tm, goal, integ, tau_prec ? tau_prec : zero, tau);
	return status;
}
int orc_maxn(void) { return MAXN; }

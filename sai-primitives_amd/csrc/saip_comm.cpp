// Multi-GPU side of the engine, C++ on RCCL (no PyTorch): instances shard embarrassingly across the GPUs of a node -- one batch per
// device, no data-path collective -- and the only exchange is the final gather of the [dof][ld] torque slabs over xGMI (SURVEY.md 8(e),
// north_star: "RCCL over xGMI only for the final torque gather").  Two ways in:
//   * one process per GPU (how the driver launches bench.py): saip_comm_unique_id on rank 0, the 128 bytes travel by whatever rendezvous
//     the launcher has, saip_comm_create on every rank, saip_batch_all_gather_torques enqueues ncclAllGather on the batch's stream;
//   * one process, n devices: saip_multi_* owns a batch, a stream and a communicator per device (ncclCommInitAll) and groups the n
//     all-gathers between ncclGroupStart / ncclGroupEnd.
// librccl is loaded on first use (dlopen): single-GPU users of libsaip.so never pay for it.  Built on the public C-ABI only.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/saip.h"

namespace saip {
saip_status fail_external(saip_status st, const char* fmt, ...);  // saip_engine.cpp: sets saip_last_error()
}
using saip::fail_external;

namespace {
struct Rccl {
	void* lib = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclCommCount) CommCount = nullptr;
	char why[256] = "librccl.so was never looked for";  // dlerror() of the failed load (dlerror itself returns NULL on later calls)
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl& rccl_state() {
	static Rccl r;
	return r;
}
Rccl* rccl() {
	Rccl& r = rccl_state();
	static bool tried = false;
	if (!tried) {
		tried = true;
		for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (r.lib) break;
			const char* e = dlerror();
			snprintf(r.why, sizeof(r.why), "%s", e ? e : "dlopen failed");
		}
		if (r.lib) {
#define SAIP_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.lib, sym)
			SAIP_SYM(GetUniqueId, "ncclGetUniqueId");
			SAIP_SYM(CommInitRank, "ncclCommInitRank");
			SAIP_SYM(CommInitAll, "ncclCommInitAll");
			SAIP_SYM(CommDestroy, "ncclCommDestroy");
			SAIP_SYM(CommCount, "ncclCommCount");
			SAIP_SYM(AllGather, "ncclAllGather");
			SAIP_SYM(GroupStart, "ncclGroupStart");
			SAIP_SYM(GroupEnd, "ncclGroupEnd");
			SAIP_SYM(GetErrorString, "ncclGetErrorString");
#undef SAIP_SYM
			if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.CommCount || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
				r.lib = nullptr;
				snprintf(r.why, sizeof(r.why), "librccl.so lacks one of the entry points the engine uses");
			}
		}
	}
	return r.lib ? &r : nullptr;
}
}  // namespace

#define RCCL_TRY(R, expr)                                                                                               \
	do {                                                                                                                \
		ncclResult_t r_ = (expr);                                                                                       \
		if (r_ != ncclSuccess) return fail_external(SAIP_ERR_DEVICE, "%s failed: %s", #expr, (R)->GetErrorString(r_)); \
	} while (0)
#define HIP_TRY2(expr)                                                                                                   \
	do {                                                                                                                 \
		hipError_t e_ = (expr);                                                                                          \
		if (e_ != hipSuccess) return fail_external(SAIP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));     \
	} while (0)

struct saip_comm {
	ncclComm_t comm = nullptr;
	int device = -1, world = 0, rank = 0;
};

// can this process take part in a communicator on `device`?  (librccl loads, the device can be selected.)  Every rank asks this BEFORE any
// rank enters ncclCommInitRank, and the launcher reduces the answers: a rank that failed locally would otherwise leave the others blocked
// inside the collective initialisation
extern "C" saip_status saip_comm_probe(int device) {
	Rccl* R = rccl();
	if (!R) return fail_external(SAIP_ERR_NO_DEVICE, "librccl.so could not be loaded: %s", rccl_state().why);
	HIP_TRY2(hipSetDevice(device));
	return SAIP_OK;
}
extern "C" saip_status saip_comm_unique_id(void* id128) {
	Rccl* R = rccl();
	if (!R) return fail_external(SAIP_ERR_NO_DEVICE, "librccl.so could not be loaded: %s", rccl_state().why);
	if (!id128) return fail_external(SAIP_ERR_INVALID_ARGUMENT, "saip_comm_unique_id: null buffer");
	static_assert(sizeof(ncclUniqueId) == SAIP_COMM_ID_BYTES, "unique id size");
	RCCL_TRY(R, R->GetUniqueId((ncclUniqueId*)id128));
	return SAIP_OK;
}
extern "C" saip_status saip_comm_create(int device, int world, int rank, const void* id128, saip_comm** out) {
	Rccl* R = rccl();
	if (!R) return fail_external(SAIP_ERR_NO_DEVICE, "librccl.so could not be loaded: %s", rccl_state().why);
	if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail_external(SAIP_ERR_INVALID_ARGUMENT, "saip_comm_create: bad arguments");
	HIP_TRY2(hipSetDevice(device));
	ncclUniqueId id;
	__builtin_memcpy(&id, id128, sizeof(id));
	saip_comm* c = new saip_comm();
	c->device = device;
	c->world = world;
	c->rank = rank;
	ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
	if (r != ncclSuccess) {
		delete c;
		return fail_external(SAIP_ERR_DEVICE, "ncclCommInitRank failed: %s", R->GetErrorString(r));
	}
	*out = c;
	return SAIP_OK;
}
extern "C" void saip_comm_destroy(saip_comm* c) {
	if (!c) return;
	if (Rccl* R = rccl())
		if (c->comm) (void)R->CommDestroy(c->comm);
	delete c;
}
// number of ranks RCCL itself counts in the communicator (ncclCommCount), not what the caller asked for
extern "C" int saip_comm_world(const saip_comm* c) {
	Rccl* R = rccl();
	if (!c || !R || !c->comm) return 0;
	int n = 0;
	return R->CommCount(c->comm, &n) == ncclSuccess ? n : 0;
}
extern "C" int saip_comm_rank(const saip_comm* c) { return c ? c->rank : -1; }

// the final torque gather: [dof][ld] of this rank -> gathered_dev [world][dof][ld] on every rank, asynchronously on the batch's stream
extern "C" saip_status saip_batch_all_gather_torques(saip_batch* b, saip_comm* c, double* gathered_dev) {
	Rccl* R = rccl();
	if (!R) return fail_external(SAIP_ERR_NO_DEVICE, "librccl.so could not be loaded: %s", rccl_state().why);
	if (!b || !c || !gathered_dev) return fail_external(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_all_gather_torques: null argument");
	double* tau = saip_batch_device_tau(b);
	hipStream_t stream = (hipStream_t)saip_batch_stream(b);
	if (!tau || !stream) return fail_external(SAIP_ERR_ORDER, "saip_batch_all_gather_torques: the batch is not finalized on a device");
	const size_t count = (size_t)saip_batch_dof(b) * (size_t)saip_batch_ld(b);
	HIP_TRY2(hipSetDevice(c->device));
	RCCL_TRY(R, R->AllGather(tau, gathered_dev, count, ncclDouble, c->comm, stream));
	return SAIP_OK;
}

// `steps` cycles with the gather inside the timed region (see saip.h): events around the region and around every gather, all created before it
namespace {
struct EventPool {
	std::vector<hipEvent_t> ev;
	~EventPool() {
		for (hipEvent_t e : ev) (void)hipEventDestroy(e);
	}
	bool grow(size_t n) {
		while (ev.size() < n) {
			hipEvent_t e = nullptr;
			if (hipEventCreate(&e) != hipSuccess) return false;
			ev.push_back(e);
		}
		return true;
	}
};
}  // namespace
extern "C" saip_status saip_batch_time_steps_gather(saip_batch* b, saip_comm* c, double* gathered_dev, int steps, int gather_mode, double* elapsed_ms, double* gather_ms) {
	if (!b || steps <= 0 || !elapsed_ms || !gather_ms || (gather_mode != 0 && gather_mode != 1) || (c && !gathered_dev))
		return fail_external(SAIP_ERR_INVALID_ARGUMENT, "saip_batch_time_steps_gather: bad arguments");
	hipStream_t stream = (hipStream_t)saip_batch_stream(b);
	if (!stream) return fail_external(SAIP_ERR_ORDER, "saip_batch_time_steps_gather: the batch is not finalized on a device");
	const int ngather = c ? (gather_mode == 1 ? steps : 1) : 0;
	EventPool pool;
	if (!pool.grow(2 + 2 * (size_t)ngather)) return fail_external(SAIP_ERR_DEVICE, "hipEventCreate failed");
	HIP_TRY2(hipEventRecord(pool.ev[0], stream));
	int g = 0;
	for (int i = 0; i < steps; i++) {
		saip_status st = saip_batch_step_async(b);
		if (st) return st;
		if (c && (gather_mode == 1 || i == steps - 1)) {
			HIP_TRY2(hipEventRecord(pool.ev[2 + 2 * g], stream));
			if ((st = saip_batch_all_gather_torques(b, c, gathered_dev))) return st;
			HIP_TRY2(hipEventRecord(pool.ev[3 + 2 * g], stream));
			g++;
		}
	}
	HIP_TRY2(hipEventRecord(pool.ev[1], stream));
	HIP_TRY2(hipStreamSynchronize(stream));
	float ms = 0;
	HIP_TRY2(hipEventElapsedTime(&ms, pool.ev[0], pool.ev[1]));
	*elapsed_ms = ms;
	double gsum = 0.0;
	for (int i = 0; i < g; i++) {
		HIP_TRY2(hipEventElapsedTime(&ms, pool.ev[2 + 2 * i], pool.ev[3 + 2 * i]));
		gsum += ms;
	}
	*gather_ms = gsum;
	return SAIP_OK;
}

// ------------------------------------------------------------------ one process, n devices
struct saip_multi {
	std::vector<int> devices;
	std::vector<saip_batch*> batches;
	std::vector<ncclComm_t> comms;
	std::vector<double*> gathered;  // per device: [n][dof][ld]
	bool finalized = false;
};
extern "C" saip_status saip_multi_create(const saip_model* model, int batch_per_device, const int* devices, int n_devices, saip_multi** out) {
	if (!model || !devices || !out || n_devices < 1) return fail_external(SAIP_ERR_INVALID_ARGUMENT, "saip_multi_create: bad arguments");
	saip_multi* m = new saip_multi();
	for (int i = 0; i < n_devices; i++) {
		saip_batch* b = nullptr;
		saip_status st = saip_batch_create(model, batch_per_device, devices[i], &b);
		if (st) {
			for (saip_batch* x : m->batches) saip_batch_destroy(x);
			delete m;
			return st;
		}
		m->devices.push_back(devices[i]);
		m->batches.push_back(b);
	}
	*out = m;
	return SAIP_OK;
}
extern "C" int saip_multi_size(const saip_multi* m) { return m ? (int)m->batches.size() : 0; }
extern "C" saip_batch* saip_multi_batch(saip_multi* m, int i) { return (m && i >= 0 && i < (int)m->batches.size()) ? m->batches[i] : nullptr; }
extern "C" saip_status saip_multi_finalize(saip_multi* m) {
	if (!m) return fail_external(SAIP_ERR_INVALID_ARGUMENT, "null multi");
	if (m->finalized) return SAIP_OK;
	Rccl* R = rccl();
	if (!R) return fail_external(SAIP_ERR_NO_DEVICE, "librccl.so could not be loaded: %s", rccl_state().why);
	for (saip_batch* b : m->batches) {
		saip_status st = saip_batch_finalize(b);
		if (st) return st;
	}
	const int n = (int)m->batches.size();
	m->comms.assign(n, nullptr);
	RCCL_TRY(R, R->CommInitAll(m->comms.data(), n, m->devices.data()));
	const size_t slab = (size_t)saip_batch_dof(m->batches[0]) * (size_t)saip_batch_ld(m->batches[0]);
	m->gathered.assign(n, nullptr);
	for (int i = 0; i < n; i++) {
		HIP_TRY2(hipSetDevice(m->devices[i]));
		HIP_TRY2(hipMalloc((void**)&m->gathered[i], slab * n * sizeof(double)));
	}
	m->finalized = true;
	return SAIP_OK;
}
extern "C" saip_status saip_multi_step_async(saip_multi* m) {
	if (!m || !m->finalized) return fail_external(SAIP_ERR_ORDER, "saip_multi_step_async: call saip_multi_finalize first");
	for (saip_batch* b : m->batches) {
		saip_status st = saip_batch_step_async(b);
		if (st) return st;
	}
	return SAIP_OK;
}
extern "C" saip_status saip_multi_all_gather_torques(saip_multi* m) {
	if (!m || !m->finalized) return fail_external(SAIP_ERR_ORDER, "saip_multi_all_gather_torques: call saip_multi_finalize first");
	Rccl* R = rccl();
	const size_t count = (size_t)saip_batch_dof(m->batches[0]) * (size_t)saip_batch_ld(m->batches[0]);
	RCCL_TRY(R, R->GroupStart());
	for (size_t i = 0; i < m->batches.size(); i++) {
		ncclResult_t r = R->AllGather(saip_batch_device_tau(m->batches[i]), m->gathered[i], count, ncclDouble, m->comms[i], (hipStream_t)saip_batch_stream(m->batches[i]));
		if (r != ncclSuccess) {
			(void)R->GroupEnd();
			return fail_external(SAIP_ERR_DEVICE, "ncclAllGather failed: %s", R->GetErrorString(r));
		}
	}
	RCCL_TRY(R, R->GroupEnd());
	return SAIP_OK;
}
extern "C" saip_status saip_multi_synchronize(saip_multi* m) {
	if (!m) return fail_external(SAIP_ERR_INVALID_ARGUMENT, "null multi");
	for (saip_batch* b : m->batches) {
		saip_status st = saip_batch_synchronize(b);
		if (st) return st;
	}
	return SAIP_OK;
}
extern "C" saip_status saip_multi_time_steps(saip_multi* m, int steps, int gather_mode, double* elapsed_ms, double* gather_ms) {
	if (!m || !m->finalized) return fail_external(SAIP_ERR_ORDER, "saip_multi_time_steps: call saip_multi_finalize first");
	if (steps <= 0 || !elapsed_ms || !gather_ms || (gather_mode != 0 && gather_mode != 1)) return fail_external(SAIP_ERR_INVALID_ARGUMENT, "saip_multi_time_steps: bad arguments");
	const size_t nd = m->batches.size();
	const bool coll = nd > 1;  // one device: there is nothing to gather, no collective is issued
	const int ngather = coll ? (gather_mode == 1 ? steps : 1) : 0;
	const size_t per = 2 + 2 * (size_t)ngather;
	std::vector<EventPool> pools(nd);
	for (size_t d = 0; d < nd; d++) {
		HIP_TRY2(hipSetDevice(m->devices[d]));
		if (!pools[d].grow(per)) return fail_external(SAIP_ERR_DEVICE, "hipEventCreate failed");
	}
	auto record = [&](size_t slot) -> hipError_t {
		for (size_t d = 0; d < nd; d++) {
			hipError_t e = hipSetDevice(m->devices[d]);
			if (e == hipSuccess) e = hipEventRecord(pools[d].ev[slot], (hipStream_t)saip_batch_stream(m->batches[d]));
			if (e != hipSuccess) return e;
		}
		return hipSuccess;
	};
	HIP_TRY2(record(0));
	int g = 0;
	for (int i = 0; i < steps; i++) {
		saip_status st = saip_multi_step_async(m);
		if (st) return st;
		if (coll && (gather_mode == 1 || i == steps - 1)) {
			HIP_TRY2(record(2 + 2 * (size_t)g));
			if ((st = saip_multi_all_gather_torques(m))) return st;
			HIP_TRY2(record(3 + 2 * (size_t)g));
			g++;
		}
	}
	HIP_TRY2(record(1));
	saip_status st = saip_multi_synchronize(m);
	if (st) return st;
	double emax = 0.0, gmax = 0.0;
	for (size_t d = 0; d < nd; d++) {
		HIP_TRY2(hipSetDevice(m->devices[d]));
		float ms = 0;
		HIP_TRY2(hipEventElapsedTime(&ms, pools[d].ev[0], pools[d].ev[1]));
		emax = ms > emax ? ms : emax;
		double gsum = 0.0;
		for (int i = 0; i < g; i++) {
			HIP_TRY2(hipEventElapsedTime(&ms, pools[d].ev[2 + 2 * (size_t)i], pools[d].ev[3 + 2 * (size_t)i]));
			gsum += ms;
		}
		gmax = gsum > gmax ? gsum : gmax;
	}
	*elapsed_ms = emax;
	*gather_ms = gmax;
	return SAIP_OK;
}
extern "C" double* saip_multi_gathered_device(saip_multi* m, int i) { return (m && m->finalized && i >= 0 && i < (int)m->gathered.size()) ? m->gathered[i] : nullptr; }
// the gathered slabs as device i holds them: out [n_devices][dof][B] (host); synchronises
extern "C" saip_status saip_multi_get_gathered_host(saip_multi* m, int i, double* out) {
	if (!m || !m->finalized || i < 0 || i >= (int)m->batches.size() || !out) return fail_external(SAIP_ERR_INVALID_ARGUMENT, "saip_multi_get_gathered_host: bad arguments");
	saip_status st = saip_multi_synchronize(m);
	if (st) return st;
	saip_batch* b = m->batches[i];
	const size_t n = m->batches.size(), dof = saip_batch_dof(b), ld = saip_batch_ld(b), B = saip_batch_size(b);
	HIP_TRY2(hipSetDevice(m->devices[i]));
	HIP_TRY2(hipMemcpy2D(out, B * sizeof(double), m->gathered[i], ld * sizeof(double), B * sizeof(double), n * dof, hipMemcpyDeviceToHost));
	return SAIP_OK;
}
extern "C" void saip_multi_destroy(saip_multi* m) {
	if (!m) return;
	Rccl* R = rccl();
	for (size_t i = 0; i < m->batches.size(); i++) {
		(void)hipSetDevice(m->devices[i]);
		if (m->finalized) (void)saip_batch_synchronize(m->batches[i]);
		if (i < m->gathered.size() && m->gathered[i]) (void)hipFree(m->gathered[i]);
		if (R && i < m->comms.size() && m->comms[i]) (void)R->CommDestroy(m->comms[i]);
		saip_batch_destroy(m->batches[i]);
	}
	delete m;
}

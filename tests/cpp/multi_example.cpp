// A C++ caller sharding a batch over the GPUs of a node through the C-ABI alone (no Python, no PyTorch): one batch per device, no data-path
// collective, one RCCL all-gather of the torque slabs at the end.
//   multi_example <robot.txt> <B per device> <n devices> <in.bin> <out.bin>
//     in  = q, dq, goal0[24], goal1[21] as [c][n*B] doubles (device d owns instances [d*B, (d+1)*B))
//     out = the gathered slabs as device 0 and as the last device hold them: 2 x [n][7][B]
// On a one-GPU box "n devices = 2" asks for GPU 0 twice; RCCL may refuse duplicates, the program then says so and runs with n = 1.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/saip.h"

static std::vector<saip_link_desc> read_robot(const char* path) {
	std::ifstream f(path);
	int n;
	f >> n;
	std::vector<saip_link_desc> links(n);
	for (auto& l : links) {
		std::string name;
		memset(&l, 0, sizeof(l));
		f >> name >> l.joint_type;
		strncpy(l.name, name.c_str(), SAIP_NAME_LEN - 1);
		for (double& v : l.origin_xyz) f >> v;
		for (double& v : l.origin_rpy) f >> v;
		for (double& v : l.axis) f >> v;
		f >> l.mass;
		for (double& v : l.com) f >> v;
		for (double& v : l.inertia) f >> v;
		f >> l.q_lower >> l.q_upper >> l.velocity_limit >> l.effort_limit;
	}
	return links;
}
#define OK(expr)                                                                     \
	do {                                                                             \
		if ((expr) != SAIP_OK) {                                                     \
			std::cerr << #expr << " failed: " << saip_last_error() << std::endl;     \
			return 1;                                                                \
		}                                                                            \
	} while (0)

static int configure(saip_batch* b) {
	const double pos[3] = {0.0, 0.0, 0.07};
	int mf = -1, jt = -1;
	OK(saip_batch_add_motion_force_task(b, "motion_force_task", "end-effector", pos, nullptr, nullptr, -1, nullptr, -1, 0.001, &mf));
	OK(saip_batch_add_joint_task(b, "joint_task", nullptr, 0, 0.001, &jt));
	OK(saip_batch_set_internal_otg(b, mf, 0));
	OK(saip_batch_set_internal_otg(b, jt, 0));
	return 0;
}

int main(int argc, char** argv) {
	if (argc != 6) return 2;
	auto links = read_robot(argv[1]);
	const int B = atoi(argv[2]);
	int n = atoi(argv[3]);
	saip_model* model = nullptr;
	OK(saip_model_create_serial_chain(links.data(), (int)links.size(), &model));
	const int dof = saip_model_dof(model);
	saip_multi* m = nullptr;
	for (;;) {
		std::vector<int> devices(n, 0);
		for (int i = 0; i < n && i < saip_device_count(); i++) devices[i] = i;  // distinct GPUs where the box has them
		OK(saip_multi_create(model, B, devices.data(), n, &m));
		bool bad = false;
		for (int i = 0; i < n; i++)
			if (configure(saip_multi_batch(m, i))) bad = true;
		if (bad) return 1;
		if (saip_multi_finalize(m) == SAIP_OK) break;
		std::cout << "MULTI_REFUSED n=" << n << ": " << saip_last_error() << std::endl;
		saip_multi_destroy(m);
		m = nullptr;
		if (n == 1) return 1;
		n = 1;
	}
	std::cout << "MULTI_DEVICES " << n << std::endl;
	const int nreq = atoi(argv[3]);
	const size_t total = (size_t)nreq * B;  // the input file always holds nreq * B instances
	std::vector<double> in((size_t)(2 * dof + 24 + 3 * dof) * total);
	std::ifstream f(argv[4], std::ios::binary);
	f.read((char*)in.data(), in.size() * sizeof(double));
	if (!f) return 3;
	// component c of instances [d*B, (d+1)*B)
	auto shard = [&](size_t first, size_t comps, int d) {
		std::vector<double> v(comps * B);
		for (size_t c = 0; c < comps; c++) memcpy(&v[c * B], &in[(first + c) * total + (size_t)d * B], B * sizeof(double));
		return v;
	};
	for (int d = 0; d < n; d++) {
		saip_batch* b = saip_multi_batch(m, d);
		auto q = shard(0, dof, d), dq = shard(dof, dof, d), g0 = shard(2 * dof, 24, d), g1 = shard(2 * dof + 24, 3 * dof, d);
		g0.resize((size_t)saip_batch_goal_components(b, 0) * B, 0.0);  // no goal / sensed force
		OK(saip_batch_set_state_host(b, q.data(), dq.data()));
		OK(saip_batch_set_goal_host(b, 0, g0.data()));
		OK(saip_batch_set_goal_host(b, 1, g1.data()));
		OK(saip_batch_update_task_models(b));
	}
	OK(saip_multi_step_async(m));            // every device computes its shard
	OK(saip_multi_all_gather_torques(m));    // the one collective: grouped ncclAllGather of the [dof][ld] slabs
	OK(saip_multi_synchronize(m));
	std::vector<double> first((size_t)n * dof * B), last((size_t)n * dof * B);
	OK(saip_multi_get_gathered_host(m, 0, first.data()));
	OK(saip_multi_get_gathered_host(m, n - 1, last.data()));
	std::ofstream o(argv[5], std::ios::binary);
	o.write((const char*)first.data(), first.size() * sizeof(double));
	o.write((const char*)last.data(), last.size() * sizeof(double));
	// round 4: the timed forms -- K cycles between HIP events on every device's stream with the grouped all-gather behind the last cycle (mode 0)
	// or behind every cycle (mode 1); with one device no collective is issued and the gather time is 0.  The torques must come out the same.
	for (int mode = 0; mode < 2; mode++) {
		double ms = -1.0, gms = -1.0;
		OK(saip_multi_time_steps(m, 10, mode, &ms, &gms));
		if (!(ms > 0.0) || !(gms >= 0.0) || (n == 1 && gms != 0.0)) {
			std::cerr << "saip_multi_time_steps: implausible times " << ms << " / " << gms << std::endl;
			return 1;
		}
		std::cout << "MULTI_TIMED mode " << mode << " ms " << ms << " gather_ms " << gms << std::endl;
	}
	if (n > 1) OK(saip_multi_all_gather_torques(m));
	std::vector<double> again((size_t)n * dof * B);
	if (n > 1) {
		OK(saip_multi_get_gathered_host(m, 0, again.data()));
		if (memcmp(again.data(), first.data(), again.size() * sizeof(double)) != 0) {
			std::cerr << "torques changed between the plain and the timed steps" << std::endl;
			return 1;
		}
	}
	{   // the one-process-per-GPU form without a communicator (one rank): begin / end and the gather-timed entry point
		saip_batch* b0 = saip_multi_batch(m, 0);
		double ms = -1.0, gms = -1.0;
		OK(saip_batch_time_steps_begin(b0, 10));
		OK(saip_batch_synchronize(b0));
		OK(saip_batch_time_steps_end(b0, &ms));
		if (!(ms > 0.0)) return 1;
		OK(saip_batch_time_steps_gather(b0, nullptr, nullptr, 10, 1, &ms, &gms));
		if (!(ms > 0.0) || gms != 0.0) return 1;
		if (saip_batch_time_steps_gather(b0, nullptr, nullptr, 0, 0, &ms, &gms) == SAIP_OK) return 1;  // bad arguments are refused
	}
	saip_multi_destroy(m);
	saip_model_destroy(model);
	std::cout << "MULTI_OK" << std::endl;
	return 0;
}

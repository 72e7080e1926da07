// Drives the C++ facade the way /root/reference/examples/05-using_robot_controller/05-using_robot_controller.cpp:103-196
// drives the reference: MotionForceTask + JointTask in a RobotController, one control cycle.
//   facade_example <robot.txt> cfgonly            host-logic checks without a GPU (device -1)
//   facade_example <robot.txt> run <B> <in.bin> <out.bin>   one cycle on GPU 0: in = q,dq,goal0[24],goal1[21] as [c][B] doubles
//   facade_example <robot.txt> ex04 <B> <in.bin> <out.bin>  the same tasks WITHOUT a controller, hierarchy built by hand like
//       /root/reference/examples/04-task_and_redundancy/04-task_and_redundancy.cpp:141-206; out = tau_mf, tau_joint, tau_joint with
//       computeTorques(tau_prec), tau_joint through the device-resident N_prec, then a RobotController on the same tasks: 5 x [dof][B]
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../include/saip/SaiPrimitivesBatched.hpp"

using namespace SaiPrimitivesBatched;

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
	if (!f) throw std::runtime_error("bad robot file");
	return links;
}

template <typename F>
static bool throws_invalid(F f) {
	try {
		f();
	} catch (const std::invalid_argument&) {
		return true;
	} catch (...) {
	}
	return false;
}

int main(int argc, char** argv) {
	if (argc < 3) return 2;
	auto links = read_robot(argv[1]);
	const double pos_in_link[3] = {0.0, 0.0, 0.07};
	if (std::string(argv[2]) == "cfgonly") {
		auto robot = std::make_shared<SaiModel>(links, 4, -1);
		int ok = 1;
		std::vector<std::shared_ptr<TemplateTask>> none;
		ok &= throws_invalid([&] { RobotController c(robot, none); });
		auto mf = std::make_shared<MotionForceTask>(robot, "end-effector", pos_in_link);
		auto jt = std::make_shared<JointTask>(robot);
		auto jt2 = std::make_shared<JointTask>(robot, "joint_task");  // duplicate name
		std::vector<std::shared_ptr<TemplateTask>> dup = {jt, jt2};
		ok &= throws_invalid([&] { RobotController c(robot, dup); });
		std::vector<std::shared_ptr<TemplateTask>> late = {jt, mf};  // motion-force task below a full joint task
		ok &= throws_invalid([&] { RobotController c(robot, late); });
		ok &= throws_invalid([&] { JointTask bad(robot, std::vector<double>(6, 1.0), 1); });
		std::vector<std::shared_ptr<TemplateTask>> tasks = {mf, jt};
		RobotController controller(robot, tasks);
		ok &= controller.getTaskNames().size() == 2 && controller.getJointTaskByName("joint_task")->isFullJointTask();
		ok &= throws_invalid([&] { controller.getJointTaskByName("motion_force_task"); });
		bool loud = false;
		try {
			controller.updateControllerTaskModels();  // no device: must fail, never compute on the CPU
		} catch (const std::runtime_error& e) {
			loud = std::string(e.what()).find("no CPU path") != std::string::npos;
		}
		ok &= loud;
		std::cout << (ok ? "FACADE_CFG_OK" : "FACADE_CFG_FAIL") << std::endl;
		return ok ? 0 : 1;
	}
	if (std::string(argv[2]) == "run" && argc == 6) {
		const int B = atoi(argv[3]);
		auto robot = std::make_shared<SaiModel>(links, B, 0);
		const int n = robot->dof();
		std::vector<double> in((size_t)(2 * n + 24 + 3 * n) * B);
		std::ifstream f(argv[4], std::ios::binary);
		f.read((char*)in.data(), in.size() * sizeof(double));
		if (!f) return 3;
		auto slice = [&](size_t first, size_t comps) { return std::vector<double>(in.begin() + first * B, in.begin() + (first + comps) * B); };
		// update robot model, create the tasks and the controller (05-using_robot_controller.cpp:105-131)
		auto motion_force_task = std::make_shared<MotionForceTask>(robot, "end-effector", pos_in_link);
		motion_force_task->disableInternalOtg();
		auto joint_task = std::make_shared<JointTask>(robot);
		joint_task->disableInternalOtg();
		std::vector<std::shared_ptr<TemplateTask>> task_list = {motion_force_task, joint_task};
		RobotController robot_controller(robot, task_list);
		// one control cycle (:143-196)
		robot->setQ(slice(0, n));
		robot->setDq(slice(n, n));
		robot->updateModel();
		robot_controller.updateControllerTaskModels();
		size_t g = 2 * n;
		motion_force_task->setGoalPosition(slice(g, 3));
		motion_force_task->setGoalOrientation(slice(g + 3, 9));
		motion_force_task->setGoalLinearVelocity(slice(g + 12, 3));
		motion_force_task->setGoalAngularVelocity(slice(g + 15, 3));
		motion_force_task->setGoalLinearAcceleration(slice(g + 18, 3));
		motion_force_task->setGoalAngularAcceleration(slice(g + 21, 3));
		g += 24;
		joint_task->setGoalPosition(slice(g, n));
		joint_task->setGoalVelocity(slice(g + n, n));
		joint_task->setGoalAcceleration(slice(g + 2 * n, n));
		std::vector<double> control_torques = robot_controller.computeControlTorques();
		std::ofstream o(argv[5], std::ios::binary);
		o.write((const char*)control_torques.data(), control_torques.size() * sizeof(double));
		std::cout << "FACADE_RUN_OK" << std::endl;
		return 0;
	}
	if (std::string(argv[2]) == "ex04" && argc == 6) {
		const int B = atoi(argv[3]);
		auto robot = std::make_shared<SaiModel>(links, B, 0);
		const int dof = robot->dof();
		std::vector<double> in((size_t)(2 * dof + 24 + 3 * dof) * B);
		std::ifstream f(argv[4], std::ios::binary);
		f.read((char*)in.data(), in.size() * sizeof(double));
		if (!f) return 3;
		auto slice = [&](size_t first, size_t comps) { return std::vector<double>(in.begin() + first * B, in.begin() + (first + comps) * B); };
		// tasks, no controller (04-task_and_redundancy.cpp:97-128)
		auto motion_force_task = std::make_shared<MotionForceTask>(robot, "end-effector", pos_in_link);
		motion_force_task->disableInternalOtg();
		auto joint_task = std::make_shared<JointTask>(robot);
		joint_task->disableInternalOtg();
		// one pass of the control loop (:136-206)
		robot->setQ(slice(0, dof));
		robot->setDq(slice(dof, dof));
		robot->updateModel();
		std::vector<double> N_prec((size_t)dof * dof, 0.0);  // MatrixXd::Identity(dof, dof)
		for (int i = 0; i < dof; i++) N_prec[(size_t)i * dof + i] = 1.0;
		motion_force_task->updateTaskModel(N_prec);
		N_prec = motion_force_task->getTaskAndPreviousNullspace();
		joint_task->updateTaskModel(N_prec);
		size_t g = 2 * dof;
		motion_force_task->setGoalPosition(slice(g, 3));
		motion_force_task->setGoalOrientation(slice(g + 3, 9));
		motion_force_task->setGoalLinearVelocity(slice(g + 12, 3));
		motion_force_task->setGoalAngularVelocity(slice(g + 15, 3));
		motion_force_task->setGoalLinearAcceleration(slice(g + 18, 3));
		motion_force_task->setGoalAngularAcceleration(slice(g + 21, 3));
		g += 24;
		joint_task->setGoalPosition(slice(g, dof));
		joint_task->setGoalVelocity(slice(g + dof, dof));
		joint_task->setGoalAcceleration(slice(g + 2 * dof, dof));
		std::vector<double> motion_force_task_torques = motion_force_task->computeTorques();
		std::vector<double> joint_task_torques = joint_task->computeTorques();
		std::vector<double> joint_task_torques_comp = joint_task->computeTorques(motion_force_task_torques);  // RobotController.cpp:82-84
		// the same with N_prec kept on the GPU
		joint_task->updateTaskModel(motion_force_task->getTaskAndPreviousNullspaceDevice());
		std::vector<double> joint_task_torques_dev = joint_task->computeTorques(motion_force_task_torques);
		// and the tasks handed to a RobotController afterwards (configuration and goals travel with them)
		std::vector<std::shared_ptr<TemplateTask>> task_list = {motion_force_task, joint_task};
		RobotController robot_controller(robot, task_list);
		robot_controller.updateControllerTaskModels();
		std::vector<double> control_torques = robot_controller.computeControlTorques();
		std::ofstream o(argv[5], std::ios::binary);
		for (auto* v : {&motion_force_task_torques, &joint_task_torques, &joint_task_torques_comp, &joint_task_torques_dev, &control_torques})
			o.write((const char*)v->data(), v->size() * sizeof(double));
		std::cout << "FACADE_EX04_OK" << std::endl;
		return 0;
	}
	return 2;
}

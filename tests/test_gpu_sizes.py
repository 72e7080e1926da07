"""Ragged batch sizes through the kernels added in round 3 (a single instance, fewer instances than a wavefront holds, one more than a
multiple of eight, one more than the automatic choice's batch limits): config 5 on the wavefront-per-instance kernel, config 6 on the
joint-first eight-lane kernel in both forms (one more instance than the chip holds at once: the launch then runs in rounds), config 15
(8-dof chain, motion-force task first) on the eight-lane kernel at every size; configs 3 / 12 / 14 (partial, reduced and blended tasks) one
chip-load and a bit, where the automatic choice used to hand over to the lane kernel.  Torques and status bits against the oracle."""
import numpy as np
import pytest

import workloads as W

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,B,kernel", [(5, 1, "saip_cycle_wave"), (5, 3, "saip_cycle_wave"), (5, 65, "saip_cycle_wave"),
                                          (6, 1, "saip_cycle_octjf"), (6, 7, "saip_cycle_octjf"), (6, 9, "saip_cycle_octjf"), (6, 4097, "saip_cycle_octjf"), (6, 8193, "saip_cycle_octjf"),
                                          (15, 1, "saip_cycle_oct"), (15, 5, "saip_cycle_oct"), (15, 8193, "saip_cycle_oct"),
                                          (3, 8193, "saip_cycle_oct"), (12, 8201, "saip_cycle_oct"), (14, 8201, "saip_cycle_oct")])
def test_ragged_batches(cfg, B, kernel):
    from oracle import Oracle
    from sai_primitives_amd.controller import controller_from_specs
    d = W.make_inputs(cfg, B)
    robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=0)
    ctrl.setFlaggedTorquePolicy(True)
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    ctrl.updateControllerTaskModels()
    ctrl.setGoals(d["goals"])
    tau = ctrl.computeControlTorques()
    assert ctrl.kernelName() == kernel
    ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=8)
    ok = st != 1
    assert np.array_equal(ctrl.status & 9, st & 9)
    assert W.torque_error(tau[ok], ref[ok]) < 1e-6

"""Batched mirror of the reference's task/controller interface on top of the C-ABI.

Same class and method names as the reference (camelCase kept on purpose so that call sequences read like
/root/reference/examples/05-using_robot_controller/05-using_robot_controller.cpp:103-196), with every
per-robot Eigen vector replaced by a (B, size) NumPy array.  Differences from the reference, all loud:
  * one SaiModel object = B robot instances of the same robot (state arrays are (B, dof));
  * the internal OTG is ENABLED by default like in the reference: the acceleration-limited mode (the reference default) of both
    the joint OTG (OTG_joints) and the Cartesian OTG (OTG_6dof_cartesian) runs on the device; the jerk-limited mode raises
    SaipUnsupported;
  * a task can be driven by hand exactly like in the reference (TemplateTask.h:43-60: updateTaskModel(N_prec), computeTorques(),
    computeTorques(tau_prec), get*Nullspace; examples/04-task_and_redundancy): inside a RobotController it uses the controller's batch,
    on its own it owns a private one-task batch on the same device.
"""
from __future__ import annotations

import ctypes as C
import weakref
import enum
import json
import os
from dataclasses import dataclass

import numpy as np

from . import capi

_ROBOT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "robots")
_JT = {"fixed": 0, "revolute": 1, "prismatic": 2}


class TaskType(enum.IntEnum):  # TemplateTask.h:19-24
    UNDEFINED = 0
    JOINT_LIMIT_AVOIDANCE_TASK = 1
    JOINT_TASK = 2
    MOTION_FORCE_TASK = 3


class DynamicDecouplingType(enum.IntEnum):  # SaiPrimitivesCommonDefinitions.h:14-20
    FULL_DYNAMIC_DECOUPLING = 0
    BOUNDED_INERTIA_ESTIMATES = 1
    IMPEDANCE = 2


@dataclass
class PIDGains:  # SaiPrimitivesCommonDefinitions.h:26-32
    kp: float
    kv: float
    ki: float


def load_robot_description(name_or_path: str) -> dict:
    path = name_or_path if os.path.exists(name_or_path) else os.path.join(_ROBOT_DIR, name_or_path + ".json")
    with open(path) as f:
        return json.load(f)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _check_otg_limits(who, **limits):
    """what the reference's OTG wrappers check before they touch any member (OTG_joints.cpp:41-86, OTG_6dof_cartesian.cpp:59-117): every
    limit strictly positive.  Checked here so that a task outside any batch rejects the call as well, before anything is logged."""
    for what, v in limits.items():
        if not np.all(np.asarray(v) > 0):
            raise ValueError(f"max {what} cannot be 0 or negative in any directions in {who}::setMax{what.capitalize()}")


def _soa(a, B, comps, what):
    """(B, comps) user array -> contiguous [comps][B] host staging array"""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1 and comps == a.shape[0]:
        a = np.broadcast_to(a, (B, comps))
    if a.shape != (B, comps):
        raise ValueError(f"{what}: expected shape ({B}, {comps}), got {a.shape}")
    return np.ascontiguousarray(a.T)


class SaiModel:
    """B instances of one robot (batched stand-in for SaiModel::SaiModel: constants + q/dq state)."""

    def __init__(self, description, batch_size: int, device: int = 0):
        desc = load_robot_description(description) if isinstance(description, str) else description
        self.description = desc
        links = desc["links"]
        arr = (capi.LinkDesc * len(links))()
        for d, l in zip(arr, links):
            d.name = l["name"].encode()
            d.joint_type = _JT[l["joint_type"]]
            d.origin_xyz[:] = l["origin_xyz"]
            d.origin_rpy[:] = l["origin_rpy"]
            d.axis[:] = l["axis"]
            d.mass = l["mass"]
            d.com[:] = l["com"]
            d.inertia[:] = l["inertia"]
            d.q_lower, d.q_upper = l["q_lower"], l["q_upper"]
            d.velocity_limit, d.effort_limit = l["velocity_limit"], l["effort_limit"]
        L = capi.lib()
        h = C.c_void_p()
        capi.check(L.saip_model_create_serial_chain(arr, len(links), C.byref(h)))
        self._h = h
        self._n = L.saip_model_dof(h)
        self.batch_size = int(batch_size)
        self.device = int(device)
        self._q = np.zeros((self.batch_size, self._n))
        self._dq = np.zeros((self.batch_size, self._n))
        self._controller = None
        # every batch that mirrors this robot's state: the RobotController and the private batches of tasks driven by hand.  Weak
        # references: a controller the user dropped must be collectable (its __del__ frees the device batch) and must not keep receiving
        # the robot's state at every updateModel()
        self._controllers = weakref.WeakSet()
        self._state_version = 0

    def __del__(self):
        if getattr(self, "_h", None):
            capi.lib().saip_model_destroy(self._h)
            self._h = None

    def dof(self) -> int:
        return self._n

    def q(self):
        return self._q

    def dq(self):
        return self._dq

    def setQ(self, q):
        q = np.asarray(q, float)
        if q.shape != (self.batch_size, self._n):
            raise ValueError(f"setQ: expected shape ({self.batch_size}, {self._n})")
        self._q = q.copy()
        self._state_version += 1

    def setDq(self, dq):
        dq = np.asarray(dq, float)
        if dq.shape != (self.batch_size, self._n):
            raise ValueError(f"setDq: expected shape ({self.batch_size}, {self._n})")
        self._dq = dq.copy()
        self._state_version += 1

    def updateModel(self):
        """pushes q/dq to the device; kinematics and dynamics are evaluated inside the cycle kernel"""
        for c in list(self._controllers):
            c._push_state()

    def jointLimits(self):
        n = self._n
        out = [np.zeros(n) for _ in range(4)]
        capi.check(capi.lib().saip_model_joint_limits(self._h, *[_dptr(a) for a in out]))
        return dict(position_lower=out[0], position_upper=out[1], velocity=out[2], effort=out[3])

    def linkIndex(self, name: str) -> int:
        return capi.lib().saip_model_link_index(self._h, name.encode())


class _Task:
    """common part of TemplateTask (TemplateTask.h:26-124)"""
    _type = TaskType.UNDEFINED

    def __init__(self, robot: SaiModel, task_name: str, loop_timestep: float):
        self._robot, self._name, self._dt = robot, task_name, float(loop_timestep)
        self._otg_enabled = True  # reference default (JointTask.h:38, MotionForceTask.h:67)
        self._ctrl = None
        self._id = -1
        self._log = []  # every configuration call, replayed into the batch the task joins (its RobotController's or its private one)
        self._keepalive = {}  # (setter, field) -> the arrays its logged arguments point into: replaced, hence released, with the log entry
        self._manual = False  # model last updated through updateTaskModel(N_prec) rather than updateControllerTaskModels()

    # -- TemplateTask accessors
    def getConstRobotModel(self):
        return self._robot

    def getLoopTimestep(self):
        return self._dt

    def getTaskType(self):
        return self._type

    def getTaskName(self):
        return self._name

    def _log_call(self, fn_name, args, key=None, keep=None):
        """the configuration log replayed into the batch the task joins holds the LAST call per setter (per field for setters that address
        several): a caller that sets gains every cycle neither grows it nor replays duplicates.  `keep`: the arrays the logged pointer arguments
        point into; they live exactly as long as the entry."""
        k = (fn_name, key)
        self._log = [e for e in self._log if e[2] != k]
        self._log.append((fn_name, args, k))
        if keep is not None:
            self._keepalive[k] = keep
        else:
            self._keepalive.pop(k, None)

    def _cfg(self, fn_name, *args, keep=None):
        """one configuration call: into the engine first when the task sits in a batch -- a call the engine rejects raises and leaves the log (and
        whatever the caller commits behind this call) as it was, like the reference leaves its members untouched when a setter throws
        (JointTask.cpp:400-409) -- then into the replay log"""
        if self._ctrl is not None:
            self._ctrl._call(fn_name, self._id, *args)
        self._log_call(fn_name, args, keep=keep)

    def reInitializeTask(self):
        """TemplateTask::reInitializeTask of this task alone: goal := current pose, integrators := 0, OTG re-initialised"""
        self._need_ctrl()._push_state()
        self._ctrl._call("saip_batch_reinitialize_task", self._id)

    # -- the reference's per-task interface, TemplateTask.h:43-60 (driven by hand in examples/04-task_and_redundancy/04-task_and_redundancy.cpp:141-206)
    def updateTaskModel(self, N_prec):
        """TemplateTask::updateTaskModel(N_prec): N_prec = nullspace of the higher-priority tasks, (dof, dof) for every instance,
        (B, dof, dof), or the DeviceNullspace another task's getTaskAndPreviousNullspace(device=True) returned (stays on the GPU)."""
        ctrl = self._need_ctrl()
        ctrl._push_state()
        n, B = self._robot.dof(), ctrl.batch_size
        L = capi.lib()
        if isinstance(N_prec, DeviceNullspace):
            if N_prec.n != n or N_prec.batch_size != B:
                raise ValueError("N_prec matrix size not consistent with robot dof in updateTaskModel")
            if N_prec.ctrl is not ctrl:
                capi.check(L.saip_batch_wait_for(ctrl._h, N_prec.ctrl._h))
            capi.check(L.saip_batch_task_update_model_device(ctrl._h, self._id, N_prec.ptr))
        else:
            a = np.asarray(N_prec, float)
            if a.ndim < 2 or a.shape[-1] != a.shape[-2]:  # JointTask.cpp:219-223
                raise ValueError("N_prec matrix not square in updateTaskModel")
            if a.shape[-1] != n or a.ndim > 3 or (a.ndim == 3 and a.shape[0] != B):  # :224-229
                raise ValueError("N_prec matrix size not consistent with robot dof in updateTaskModel")
            a = np.ascontiguousarray(np.broadcast_to(a, (B, n, n)).reshape(B, n * n).T)
            capi.check(L.saip_batch_task_update_model(ctrl._h, self._id, _dptr(a)))
        self._manual = True

    def computeTorques(self, tau_prec=None):
        """TemplateTask::computeTorques() / computeTorques(tau_prec): (B, dof) torques of THIS task; tau_prec (B, dof) = torques of the
        previous tasks (the joint task feed-forward compensates them, JointTask.cpp:285-292).  Per-instance status in self.status."""
        ctrl = self._need_ctrl()
        if not self._manual:
            raise capi.SaipError(f"task [{self._name}]: call updateTaskModel(N_prec) before computeTorques()")
        n, B = self._robot.dof(), ctrl.batch_size
        tau, st = np.empty((n, B)), np.zeros(B, np.uint8)
        tp = None if tau_prec is None else _soa(tau_prec, B, n, "tau_prec")
        capi.check(capi.lib().saip_batch_task_compute_torques(ctrl._h, self._id, None if tp is None else _dptr(tp), _dptr(tau),
                                                               st.ctypes.data_as(C.POINTER(C.c_ubyte))))
        self.status = st
        return tau.T.copy()

    def _manual_nullspace(self, which, device):
        ctrl = self._need_ctrl()
        n, B = self._robot.dof(), ctrl.batch_size
        if device:
            ptr = capi.lib().saip_batch_task_device_nullspace(ctrl._h, self._id, which)
            return DeviceNullspace(ctrl, self, which, ptr, n, B)
        out = np.empty((n * n, B))
        args = [None, None, None]
        args[which] = _dptr(out)
        capi.check(capi.lib().saip_batch_task_get_nullspaces_host(ctrl._h, self._id, *args))
        return out.T.reshape(B, n, n).copy()

    def _gains(self, fn, kp, kv, ki):
        self._gain_cache = getattr(self, "_gain_cache", {})
        self._gain_cache[fn] = tuple(np.atleast_1d(np.asarray(x, float)).copy() for x in (kp, kv, ki))
        kp, kv, ki = (np.atleast_1d(np.asarray(x, float)) for x in (kp, kv, ki))
        size = max(kp.shape[0], kv.shape[0], ki.shape[0])
        for x in (kp, kv, ki):
            if x.ndim != 1 or x.shape[0] not in (1, size):
                raise ValueError("kp, kv and ki must be scalars or vectors of the same size")
        k = [np.ascontiguousarray(np.broadcast_to(x, (size,))) for x in (kp, kv, ki)]
        self._cfg(fn, _dptr(k[0]), _dptr(k[1]), _dptr(k[2]), size, keep=k)

    def setDynamicDecouplingType(self, t):
        self._cfg("saip_batch_set_dynamic_decoupling_type", int(t))

    def setBoundedInertiaEstimateThreshold(self, thr: float):
        self._bie_threshold = max(float(thr), 0.0) if self._type == TaskType.JOINT_TASK else float(thr)  # JointTask.h:372-378 clamps, SingularityHandler.h:81-86 does not
        self._cfg("saip_batch_set_bie_threshold", float(thr))

    def disableInternalOtg(self):
        self._otg_enabled = False
        self._cfg("saip_batch_set_internal_otg", 0)

    def getInternalOtgEnabled(self):
        return self._otg_enabled

    def _desired_block(self):
        """(B, goal_components): what the control law tracks -- the internal OTG's output when enabled, else the goal"""
        ctrl = self._need_ctrl()
        gs = capi.lib().saip_batch_goal_components(ctrl._h, self._id)
        out = np.empty((gs, ctrl._robot.batch_size))
        ctrl._call("saip_batch_get_desired_host", self._id, _dptr(out))
        return out.T

    def getInternalOtgStatus(self):
        """(goal_reached (B,) bool = OTG::isGoalReached, flags (B,) int, result (B,) int = ruckig::Result of the last cycle)"""
        ctrl = self._need_ctrl()
        fl, res = np.zeros(ctrl._robot.batch_size, np.int32), np.zeros(ctrl._robot.batch_size, np.int32)
        ip = C.POINTER(C.c_int)
        ctrl._call("saip_batch_get_otg_status_host", self._id, fl.ctypes.data_as(ip), res.ctypes.data_as(ip))
        return (fl & 1).astype(bool), fl, res

    def enableVelocitySaturation(self, *values):
        """MotionForceTask: (linear_vel_sat, angular_vel_sat); JointTask: (value) or (vector of task dof); () keeps the defaults"""
        self._vel_sat = True
        if values:
            v = np.ascontiguousarray(np.concatenate([np.atleast_1d(np.asarray(x, float)) for x in values]))
            self._sat_vel = tuple(v.tolist())
            self._cfg("saip_batch_set_saturation_velocities", _dptr(v), int(v.shape[0]), keep=v)
        self._cfg("saip_batch_set_velocity_saturation", 1)

    def disableVelocitySaturation(self):
        self._vel_sat = False
        self._cfg("saip_batch_set_velocity_saturation", 0)

    def _cached_gains(self, fn, default):
        g = getattr(self, "_gain_cache", {}).get(fn)
        return [PIDGains(*d) for d in default] if g is None else [PIDGains(*v) for v in zip(*(np.broadcast_to(x, (max(len(y) for y in g),)) for x in g))]

    def _need_ctrl(self):
        """the batch this task is evaluated in: its RobotController's, or -- for a task driven by hand like in the reference's example 04 --
        a private one-task batch created on first use"""
        if self._ctrl is None:
            RobotController(self._robot, [self], _private=True)
        return self._ctrl

    def _set_field(self, first, comps, value, what):
        ctrl = self._need_ctrl()
        a = _soa(value, ctrl._robot.batch_size, comps, what)
        capi.check(capi.lib().saip_batch_set_goal_field_host(ctrl._h, self._id, first, comps, _dptr(a)))

    def _get_goal(self):
        ctrl = self._need_ctrl()
        gs = capi.lib().saip_batch_goal_components(ctrl._h, self._id)
        out = np.empty((gs, ctrl._robot.batch_size))
        capi.check(capi.lib().saip_batch_get_goal_host(ctrl._h, self._id, _dptr(out)))
        return out.T.copy()

    def getTaskNullspace(self, device=False):
        """(B, dof, dof) nullspace projector N of this task for the current state (TemplateTask.h:71-77)"""
        if self._manual:
            return self._manual_nullspace(0, device)
        ctrl = self._need_ctrl()
        n = self._robot.dof()
        out = np.empty((n * n, ctrl._robot.batch_size))
        capi.check(capi.lib().saip_batch_get_task_nullspace_host(ctrl._h, self._id, _dptr(out)))
        return out.T.reshape(ctrl._robot.batch_size, n, n).copy()

    def getPreviousTasksNullspace(self):
        """(B, dof, dof) N_prec this task was updated with: the product N_{t-1} ... N_0 of the tasks above it (TemplateTask.h:79-83,
        RobotController.cpp:68-77); the identity for the first task"""
        if self._manual:
            return self._manual_nullspace(1, False)
        ctrl = self._need_ctrl()
        n, B = self._robot.dof(), ctrl._robot.batch_size
        Np = np.broadcast_to(np.eye(n), (B, n, n)).copy()
        for s in range(self._id):
            out = np.empty((n * n, B))
            capi.check(capi.lib().saip_batch_get_task_nullspace_host(ctrl._h, s, _dptr(out)))
            Np = out.T.reshape(B, n, n) @ Np
        return Np

    def getTaskAndPreviousNullspace(self, device=False):
        """(B, dof, dof) N N_prec, what the next task in the hierarchy is updated with (TemplateTask.h:85-89); device=True (after
        updateTaskModel) returns a DeviceNullspace handle instead: pass it to the next task's updateTaskModel, nothing crosses PCIe"""
        if self._manual:
            return self._manual_nullspace(2, device)
        return self.getTaskNullspace() @ self.getPreviousTasksNullspace()


class DeviceNullspace:
    """a (B, dof, dof) nullspace matrix resident on the GPU ([dof*dof][ld] row-major per instance): what task.getTaskAndPreviousNullspace(
    device=True) returns and task.updateTaskModel accepts.  Valid until the producing task's next updateTaskModel."""

    def __init__(self, ctrl, task, which, ptr, n, batch_size):
        self.ctrl, self.task, self.which, self.ptr, self.n, self.batch_size = ctrl, task, which, ptr, n, batch_size

    def numpy(self):
        return self.task._manual_nullspace(self.which, False)


class MotionForceTask(_Task):
    """MotionForceTask.h:96-110.  controlled_directions_* = None -> full 6-dof task."""
    _type = TaskType.MOTION_FORCE_TASK

    def __init__(self, robot, link_name, compliant_frame_pos=(0.0, 0.0, 0.0), compliant_frame_rot=None,
                 controlled_directions_translation=None, controlled_directions_rotation=None,
                 task_name="motion_force_task", is_force_motion_parametrization_in_compliant_frame=False, loop_timestep=0.001):
        super().__init__(robot, task_name, loop_timestep)
        self._compliant_param = bool(is_force_motion_parametrization_in_compliant_frame)
        if is_force_motion_parametrization_in_compliant_frame:
            self._cfg("saip_batch_set_parametrization_in_compliant_frame", 1)
        self.link_name = link_name
        self.pos = np.asarray(compliant_frame_pos, float).reshape(3).copy()
        self.rot = None if compliant_frame_rot is None else np.ascontiguousarray(np.asarray(compliant_frame_rot, float).reshape(9))
        self.partial = controlled_directions_translation is not None or controlled_directions_rotation is not None
        none = np.zeros((0, 3))
        self.dt_ = np.ascontiguousarray(np.asarray(controlled_directions_translation if controlled_directions_translation is not None else none, float).reshape(-1, 3))
        self.dr_ = np.ascontiguousarray(np.asarray(controlled_directions_rotation if controlled_directions_rotation is not None else none, float).reshape(-1, 3))
        if self.partial and len(self.dt_) == 0 and len(self.dr_) == 0:  # MotionForceTask.cpp:47-53
            raise ValueError("controlled_directions_translation and controlled_directions_rotation cannot both be empty "
                             "in MotionForceTask::MotionForceTask")

    def _add(self, L, h):
        tid = C.c_int(-1)
        capi.check(L.saip_batch_add_motion_force_task(
            h, self._name.encode(), self.link_name.encode(), _dptr(self.pos), None if self.rot is None else _dptr(self.rot),
            _dptr(self.dt_) if len(self.dt_) else None, len(self.dt_) if self.partial else -1,
            _dptr(self.dr_) if len(self.dr_) else None, len(self.dr_) if self.partial else -1, self._dt, C.byref(tid)))
        return tid.value

    # goals, MotionForceTask.h:211-247
    def setGoalPosition(self, x):
        self._set_field(0, 3, x, "setGoalPosition")

    def setGoalOrientation(self, R):
        R = np.asarray(R, float)
        B = self._need_ctrl().batch_size
        if R.shape == (3, 3):
            R = np.broadcast_to(R, (B, 3, 3))
        self._set_field(3, 9, R.reshape(B, 9), "setGoalOrientation")

    def setGoalLinearVelocity(self, v):
        self._set_field(12, 3, v, "setGoalLinearVelocity")

    def setGoalAngularVelocity(self, w):
        self._set_field(15, 3, w, "setGoalAngularVelocity")

    def setGoalLinearAcceleration(self, a):
        self._set_field(18, 3, a, "setGoalLinearAcceleration")

    def setGoalAngularAcceleration(self, a):
        self._set_field(21, 3, a, "setGoalAngularAcceleration")

    def setGoalForce(self, f):  # MotionForceTask.h setGoalForce / setGoalMoment
        self._set_field(24, 3, f, "setGoalForce")

    def setGoalMoment(self, m):
        self._set_field(27, 3, m, "setGoalMoment")

    def _space(self, fn, dim, axis):
        a = np.ascontiguousarray(np.asarray(axis if axis is not None else (0.0, 0.0, 0.0), float).reshape(3))
        if self._ctrl is None:
            self._log_call(fn, (int(dim), _dptr(a), None), keep=a)
            return None
        ch = C.c_int(0)
        self._ctrl._call(fn, self._id, int(dim), _dptr(a), C.byref(ch))
        self._log_call(fn, (int(dim), _dptr(a), None), keep=a)
        return bool(ch.value)

    def parametrizeForceMotionSpaces(self, force_space_dimension, force_or_motion_single_axis=None):  # MotionForceTask.h:560-580
        self._force_dim = int(force_space_dimension)
        if force_or_motion_single_axis is not None and self._force_dim in (1, 2):
            self._force_axis = np.asarray(force_or_motion_single_axis, float).reshape(3) / np.linalg.norm(force_or_motion_single_axis)
        return self._space("saip_batch_parametrize_force_motion_spaces", force_space_dimension, force_or_motion_single_axis)

    def parametrizeMomentRotMotionSpaces(self, moment_space_dimension, moment_or_rot_motion_single_axis=None):
        self._moment_dim = int(moment_space_dimension)
        if moment_or_rot_motion_single_axis is not None and self._moment_dim in (1, 2):
            self._moment_axis = np.asarray(moment_or_rot_motion_single_axis, float).reshape(3) / np.linalg.norm(moment_or_rot_motion_single_axis)
        return self._space("saip_batch_parametrize_moment_rot_motion_spaces", moment_space_dimension, moment_or_rot_motion_single_axis)

    def setForceControlGains(self, kp, kv, ki=0.0):
        self._force_gains = (float(kp), float(kv), float(ki))
        self._cfg("saip_batch_set_force_control_gains", float(kp), float(kv), float(ki))

    def setMomentControlGains(self, kp, kv, ki=0.0):
        self._moment_gains = (float(kp), float(kv), float(ki))
        self._cfg("saip_batch_set_moment_control_gains", float(kp), float(kv), float(ki))

    def setClosedLoopForceControl(self, enabled: bool):
        self._cfg("saip_batch_set_closed_loop_force_control", int(enabled))

    def setClosedLoopMomentControl(self, enabled: bool):
        self._cfg("saip_batch_set_closed_loop_moment_control", int(enabled))

    def enablePassivity(self):  # MotionForceTask.h:630: POPC passivity observer / controller around the closed-loop force control
        self._cfg("saip_batch_set_passivity", 1)

    def disablePassivity(self):
        self._cfg("saip_batch_set_passivity", 0)

    def updateSensedForceAndMoment(self, sensed_force_sensor_frame, sensed_moment_sensor_frame):  # MotionForceTask.cpp:805-828
        self._set_field(30, 3, sensed_force_sensor_frame, "updateSensedForceAndMoment (force)")
        self._set_field(33, 3, sensed_moment_sensor_frame, "updateSensedForceAndMoment (moment)")

    def setForceControlParameters(self, kff_force=0.95, kff_moment=0.95, max_force_feedback=20.0, max_moment_feedback=10.0):
        """setFeedforwardForceGain / MomentGain, setMaxForceControlFeedbackOutput / Moment (MotionForceTask.h:330-355)"""
        self._force_params = dict(kff_force=float(kff_force), kff_moment=float(kff_moment), max_force_feedback=float(max_force_feedback),
                                  max_moment_feedback=float(max_moment_feedback))
        self._cfg("saip_batch_set_force_control_parameters", float(kff_force), float(kff_moment), float(max_force_feedback), float(max_moment_feedback))

    def setControlToSensorTransform(self, R, p):
        """_T_control_to_sensor of setForceSensorFrame (MotionForceTask.cpp:802): rotation (3,3) and translation (3,)"""
        R, p = np.ascontiguousarray(np.asarray(R, float).reshape(9)), np.ascontiguousarray(np.asarray(p, float).reshape(3))
        self._cfg("saip_batch_set_control_to_sensor_transform", _dptr(R), _dptr(p), keep=(R, p))

    # remaining MotionForceTask.h surface: aliases, host-side getters of what was configured, integrator resets
    def setPosControlGainsUnsafe(self, kp, kv, ki=0.0):  # MotionForceTask.h: same as the checked setter (the engine validates itself)
        self.setPosControlGains(kp, kv, ki)

    def setOriControlGainsUnsafe(self, kp, kv, ki=0.0):
        self.setOriControlGains(kp, kv, ki)

    def getPosControlGains(self):
        """list of PIDGains (one entry for isotropic gains), defaults MotionForceTask.h:40-45"""
        return self._cached_gains("saip_batch_set_pos_control_gains", [(100.0, 20.0, 0.0)])

    def getOriControlGains(self):
        return self._cached_gains("saip_batch_set_ori_control_gains", [(200.0, 28.3, 0.0)])

    def getForceControlGains(self):
        return PIDGains(*getattr(self, "_force_gains", (0.7, 10.0, 1.3)))  # MotionForceTask.h:50-52

    def getMomentControlGains(self):
        return PIDGains(*getattr(self, "_moment_gains", (0.7, 10.0, 1.3)))

    def getForceSpaceDimension(self):
        return getattr(self, "_force_dim", 0)

    def getMomentSpaceDimension(self):
        return getattr(self, "_moment_dim", 0)

    def getVelocitySaturationEnabled(self):
        return getattr(self, "_vel_sat", False)

    def getLinearSaturationVelocity(self):
        return getattr(self, "_sat_vel", (0.3, np.pi / 3))[0]  # MotionForceTask.h:64-65

    def getAngularSaturationVelocity(self):
        return getattr(self, "_sat_vel", (0.3, np.pi / 3))[1]

    def getBoundedInertiaEstimateThreshold(self):
        return getattr(self, "_bie_threshold", 0.1)

    def _fcp(self, **kw):
        cur = dict(kff_force=0.95, kff_moment=0.95, max_force_feedback=20.0, max_moment_feedback=10.0)
        cur.update(getattr(self, "_force_params", {}))
        cur.update(kw)
        self.setForceControlParameters(**cur)

    def setFeedforwardForceGain(self, k):  # MotionForceTask.h:330-355
        self._fcp(kff_force=float(k))

    def setFeedforwardmomentGain(self, k):
        self._fcp(kff_moment=float(k))

    def setMaxForceControlFeedbackOutput(self, v):
        self._fcp(max_force_feedback=float(v))

    def setMaxMomentControlFeedbackOutput(self, v):
        self._fcp(max_moment_feedback=float(v))

    def getFeedforwardForceGain(self):
        return getattr(self, "_force_params", {}).get("kff_force", 0.95)

    def getFeedforwardmomentGain(self):
        return getattr(self, "_force_params", {}).get("kff_moment", 0.95)

    def getMaxForceControlFeedbackOutput(self):
        return getattr(self, "_force_params", {}).get("max_force_feedback", 20.0)

    def getMaxMomentControlFeedbackOutput(self):
        return getattr(self, "_force_params", {}).get("max_moment_feedback", 10.0)

    def resetIntegrators(self):  # MotionForceTask.cpp:988-1002
        self._need_ctrl()._call("saip_batch_reset_integrators", self._id, 3)

    def _current_pose(self):
        ctrl = self._need_ctrl()
        ctrl._push_state()
        B = ctrl._robot.batch_size
        pos, rot = np.empty((3, B)), np.empty((9, B))
        capi.check(capi.lib().saip_batch_get_current_pose_host(ctrl._h, self._id, _dptr(pos), _dptr(rot)))
        return pos.T.copy(), rot.T.reshape(B, 3, 3).copy()

    def getCurrentPosition(self):
        """(B, 3) control point in the world frame at the robot's current state, MotionForceTask.h:121"""
        return self._current_pose()[0]

    def getCurrentOrientation(self):
        """(B, 3, 3) compliant-frame orientation in the world frame, MotionForceTask.h:136"""
        return self._current_pose()[1]

    def _sigma(self, dim, axis, sel, R):
        """sigmaPosition / sigmaOrientation, MotionForceTask.cpp:892-971: sel (I - sigma_force) sel^T with the force (moment) space given
        by `dim` and `axis`, in the compliant frame when the task is parametrised there"""
        B = R.shape[0]
        I = np.broadcast_to(np.eye(3), (B, 3, 3))
        if dim == 0:
            core = np.zeros((B, 3, 3))
        elif dim == 3:
            core = I
        else:
            a = (R @ axis) if self._compliant_param else np.broadcast_to(axis, (B, 3))
            aa = np.einsum("bi,bj->bij", a, a)
            core = aa if dim == 1 else I - aa
        sf = sel @ core @ sel.T if dim in (1, 2) else (sel if dim == 3 else core)
        return sel @ (I - sf) @ sel.T

    def goalPositionReached(self, tolerance, verbose=False):
        """(B,) bool: sqrt(e^T sigmaPosition e) < tolerance with e = goal - current position, MotionForceTask.cpp:548-563"""
        pos, R = self._current_pose()
        e = self.getGoalPosition() - pos
        S = self._sigma(getattr(self, "_force_dim", 0), getattr(self, "_force_axis", np.array([0.0, 0.0, 1.0])), self.getTaskProjection()[0][:3, :3], R)
        err = np.sqrt(np.maximum(np.einsum("bi,bij,bj->b", e, S, e), 0.0))
        if verbose:
            print("position error in MotionForceTask :", err, "\nTolerance :", tolerance)
        return err < tolerance

    def goalOrientationReached(self, tolerance, verbose=False):
        """(B,) bool: sqrt(dphi^T sigmaOrientation dphi) < tolerance, dphi = orientationError(goal, current), MotionForceTask.cpp:565-579"""
        _, R = self._current_pose()
        Rd = self.getGoalOrientation()
        dphi = -0.5 * sum(np.cross(R[:, :, c], Rd[:, :, c]) for c in range(3))
        S = self._sigma(getattr(self, "_moment_dim", 0), getattr(self, "_moment_axis", np.array([0.0, 0.0, 1.0])), self.getTaskProjection()[0][3:, 3:], R)
        err = np.sqrt(np.maximum(np.einsum("bi,bij,bj->b", dphi, S, dphi), 0.0))
        if verbose:
            print("orientation error in MotionForceTask :", err, "\nTolerance :", tolerance)
        return err < tolerance

    def resetIntegratorsLinear(self):
        self._need_ctrl()._call("saip_batch_reset_integrators", self._id, 1)

    def resetIntegratorsAngular(self):
        self._need_ctrl()._call("saip_batch_reset_integrators", self._id, 2)

    def getGoalPosition(self):
        return self._get_goal()[:, 0:3]

    def getGoalOrientation(self):
        return self._get_goal()[:, 3:12].reshape(-1, 3, 3)

    # internal Cartesian OTG (OTG_6dof_cartesian), acceleration-limited mode on the device.  MotionForceTask.h:387-423
    def enableInternalOtgAccelerationLimited(self, max_linear_velocity=0.3, max_linear_acceleration=2.0,
                                             max_angular_velocity=np.pi / 3, max_angular_acceleration=2 * np.pi):
        v = np.ascontiguousarray([float(max_linear_velocity), float(max_angular_velocity)])
        a = np.ascontiguousarray([float(max_linear_acceleration), float(max_angular_acceleration)])
        _check_otg_limits("OTG_6dof_cartesian", velocity=v, acceleration=a)
        self._cfg("saip_batch_set_otg_acceleration_limited", _dptr(v), _dptr(a), 2, keep=(v, a))
        self._otg_enabled = True

    def enableInternalOtgJerkLimited(self, max_linear_velocity=0.3, max_linear_acceleration=2.0, max_linear_jerk=10.0,
                                     max_angular_velocity=np.pi / 3, max_angular_acceleration=2 * np.pi, max_angular_jerk=10 * np.pi):
        """MotionForceTask.h:416-421 / MotionForceTask.cpp:525-545 (defaults: MotionForceTask.h:68-73)"""
        v = np.ascontiguousarray([float(max_linear_velocity), float(max_angular_velocity)])
        a = np.ascontiguousarray([float(max_linear_acceleration), float(max_angular_acceleration)])
        j = np.ascontiguousarray([float(max_linear_jerk), float(max_angular_jerk)])
        _check_otg_limits("OTG_6dof_cartesian", velocity=v, acceleration=a, jerk=j)
        self._cfg("saip_batch_set_otg_jerk_limited", _dptr(v), _dptr(a), _dptr(j), 2, keep=(v, a, j))
        self._otg_enabled = True

    # desired state = OTG output when enabled, else the goal (MotionForceTask.h getDesired*)
    def getDesiredPosition(self):
        return self._desired_block()[:, 0:3]

    def getDesiredOrientation(self):
        return self._desired_block()[:, 3:12].reshape(-1, 3, 3)

    def getDesiredLinearVelocity(self):
        return self._desired_block()[:, 12:15]

    def getDesiredAngularVelocity(self):
        return self._desired_block()[:, 15:18]

    def getDesiredLinearAcceleration(self):
        return self._desired_block()[:, 18:21]

    def getDesiredAngularAcceleration(self):
        return self._desired_block()[:, 21:24]

    # gains, MotionForceTask.h:272-300
    def setPosControlGains(self, kp, kv, ki=0.0):
        self._gains("saip_batch_set_pos_control_gains", kp, kv, ki)

    def setOriControlGains(self, kp, kv, ki=0.0):
        self._gains("saip_batch_set_ori_control_gains", kp, kv, ki)

    def enableSingularityHandling(self):  # MotionForceTask.h:715-725
        self._cfg("saip_batch_set_singularity_handling", 1)

    def disableSingularityHandling(self):
        """near-singular instances then use the non-singular part of the task only (status 2) instead of being flagged (status 1)"""
        self._cfg("saip_batch_set_singularity_handling", 0)

    def setSingularityStrategies(self, enabled=True):
        """blended type-1 / type-2 strategies of SingularityHandler for instances inside the singularity bounds (status bit 8): ON by default
        like in the reference; off (an engine extra): such instances are flagged (status 1) instead"""
        self._cfg("saip_batch_set_singularity_strategies", int(bool(enabled)))

    def setSingularityHandlingGains(self, kp_type_1, kv_type_1, kv_type_2):  # MotionForceTask.h:749
        self._cfg("saip_batch_set_singularity_gains", float(kp_type_1), float(kv_type_1), float(kv_type_2))

    def handleAllSingularitiesAsType1(self, flag):  # MotionForceTask.h:698
        self._cfg("saip_batch_set_all_singularities_type1", int(bool(flag)))

    def setType1Posture(self, q_des):  # MotionForceTask.h:707; (dof,) for every instance or (B, dof)
        q = np.ascontiguousarray(q_des, dtype=np.float64)
        self._cfg("saip_batch_set_type1_posture", q.ctypes.data_as(C.POINTER(C.c_double)), int(q.ndim == 2))

    def setSingularityHandlingBounds(self, s_min, s_max):  # MotionForceTask.h:736
        self._cfg("saip_batch_set_singularity_bounds", float(s_min), float(s_max))

    def getTaskProjection(self):
        ctrl = self._need_ctrl()
        P, Bm, k = np.zeros(36), np.zeros(36), C.c_int(0)
        capi.check(capi.lib().saip_batch_get_task_projection(ctrl._h, self._id, _dptr(P), _dptr(Bm), C.byref(k)))
        return P.reshape(6, 6), Bm.reshape(6, 6)[:, :k.value].copy()


class JointTask(_Task):
    """JointTask.h:56-75.  joint_selection_matrix = None -> full joint task."""
    _type = TaskType.JOINT_TASK

    def __init__(self, robot, joint_selection_matrix=None, task_name="joint_task", loop_timestep=0.001):
        super().__init__(robot, task_name, loop_timestep)
        self.S = None
        if joint_selection_matrix is not None:
            S = np.ascontiguousarray(np.asarray(joint_selection_matrix, float))
            if S.ndim != 2 or S.shape[1] != robot.dof():  # JointTask.cpp:28-32
                raise ValueError("joint selection matrix size not consistent with robot dof in JointTask constructor")
            self.S = S

    def _add(self, L, h):
        tid = C.c_int(-1)
        if self.S is None:
            capi.check(L.saip_batch_add_joint_task(h, self._name.encode(), None, 0, self._dt, C.byref(tid)))
        else:
            capi.check(L.saip_batch_add_joint_task(h, self._name.encode(), _dptr(self.S), self.S.shape[0], self._dt, C.byref(tid)))
        return tid.value

    def getTaskDof(self):
        return self._robot.dof() if self.S is None else self.S.shape[0]

    def isFullJointTask(self):
        return self.getTaskDof() == self._robot.dof()

    # goals, JointTask.h:140-175
    def setGoalPosition(self, q):
        self._set_field(0, self.getTaskDof(), q, "goal position vector size not consistent with task dof in JointTask::setGoalPosition")

    def setGoalVelocity(self, dq):
        m = self.getTaskDof()
        self._set_field(m, m, dq, "goal velocity vector size not consistent with task dof in JointTask::setGoalVelocity")

    def setGoalAcceleration(self, ddq):
        m = self.getTaskDof()
        self._set_field(2 * m, m, ddq, "goal acceleration vector size not consistent with task dof in JointTask::setGoalAcceleration")

    def getGoalPosition(self):
        return self._get_goal()[:, :self.getTaskDof()]

    def setGains(self, kp, kv, ki=0.0):  # JointTask.h:237-257
        self._gains("saip_batch_set_joint_gains", kp, kv, ki)

    def setGainsUnsafe(self, kp, kv, ki=0.0):  # same as the checked setter (the engine validates itself)
        self.setGains(kp, kv, ki)

    def getGains(self):
        """list of PIDGains (one entry for isotropic gains), defaults JointTask.h:31-33"""
        return self._cached_gains("saip_batch_set_joint_gains", [(50.0, 14.0, 0.0)])

    def getJointSelectionMatrix(self):
        return np.eye(self._robot.dof()) if self.S is None else self.S.reshape(self.getTaskDof(), self._robot.dof()).copy()

    def getCurrentPosition(self):
        """S q of the robot's current state, (B, task dof)"""
        return self._robot._q @ self.getJointSelectionMatrix().T

    def getCurrentVelocity(self):
        return self._robot._dq @ self.getJointSelectionMatrix().T

    def getGoalVelocity(self):
        m = self.getTaskDof()
        return self._get_goal()[:, m:2 * m]

    def getGoalAcceleration(self):
        m = self.getTaskDof()
        return self._get_goal()[:, 2 * m:3 * m]

    def getVelocitySaturationEnabled(self):
        return getattr(self, "_vel_sat", False)

    def goalPositionReached(self, tol):
        """(B,) bool: sqrt(e^T U U^T e) < tol with e = current - goal and U = matrixRangeBasis(S N_prec) (JointTask.cpp:437-446; tolerance
        1e-3 as JointTask.cpp:233); N_prec comes from the device, the small SVDs run on the host"""
        e = self.getCurrentPosition() - self.getGoalPosition()
        Jp = self.getJointSelectionMatrix()[None] @ self.getPreviousTasksNullspace()
        out = np.zeros(e.shape[0], bool)
        for b in range(e.shape[0]):
            U, sv, _ = np.linalg.svd(Jp[b], full_matrices=False)
            keep = (sv >= 1e-3 * sv[0]) if sv[0] >= 1e-3 and np.linalg.norm(Jp[b]) >= 1e-3 else np.zeros(len(sv), bool)
            c = U[:, keep].T @ e[b]
            out[b] = np.sqrt(c @ c) < tol
        return out

    def getVelocitySaturationMaxVelocity(self):
        v = np.asarray(getattr(self, "_sat_vel", (np.pi / 3,)), float)  # JointTask.h:44
        return np.broadcast_to(v, (self.getTaskDof(),)).copy() if v.shape[0] in (1, self.getTaskDof()) else v

    def getBoundedInertiaEstimateThreshold(self):
        return getattr(self, "_bie_threshold", 0.1)

    def resetIntegrators(self):
        self._need_ctrl()._call("saip_batch_reset_integrators", self._id, 1)

    # internal OTG, JointTask.h:272-327.  Acceleration-limited mode (the reference default) runs on the device.
    def enableInternalOtgAccelerationLimited(self, max_velocity=np.pi / 3.0, max_acceleration=2.0 * np.pi):
        v, a = np.atleast_1d(np.asarray(max_velocity, float)), np.atleast_1d(np.asarray(max_acceleration, float))
        m = self.getTaskDof()
        if v.shape != a.shape or v.ndim != 1 or v.shape[0] not in (1, m):  # JointTask.cpp:367-373
            raise ValueError("max velocity or max acceleration vector size not consistent with task dof in JointTask::enableInternalOtgAccelerationLimited")
        v, a = np.ascontiguousarray(v), np.ascontiguousarray(a)
        _check_otg_limits("OTG_joints", velocity=v, acceleration=a)
        self._cfg("saip_batch_set_otg_acceleration_limited", _dptr(v), _dptr(a), int(v.shape[0]), keep=(v, a))
        self._otg_enabled = True

    def enableInternalOtgJerkLimited(self, max_velocity, max_acceleration, max_jerk):
        """JointTask.h:298-316 / JointTask.cpp:383-410: third-order (jerk-limited) Ruckig profiles; scalars or one value per task dof.  The OTG
        is re-initialised at the current task position when it was off or acceleration-limited (the engine does so at the next cycle)"""
        v, a, j = (np.ascontiguousarray(np.atleast_1d(np.asarray(x, float))) for x in (max_velocity, max_acceleration, max_jerk))
        if not (v.shape == a.shape == j.shape) or v.shape[0] not in (1, self.getTaskDof()):
            raise ValueError("max velocity, max acceleration or max jerk vector size not consistent with task dof in JointTask::enableInternalOtgJerkLimited")
        _check_otg_limits("OTG_joints", velocity=v, acceleration=a, jerk=j)
        self._cfg("saip_batch_set_otg_jerk_limited", _dptr(v), _dptr(a), _dptr(j), int(v.shape[0]), keep=(v, a, j))
        self._otg_enabled = True

    def getDesiredPosition(self):  # JointTask.h:185-200: the OTG output when enabled, else the goal
        return self._desired_block()[:, :self.getTaskDof()]

    def getDesiredVelocity(self):
        m = self.getTaskDof()
        return self._desired_block()[:, m:2 * m]

    def getDesiredAcceleration(self):
        m = self.getTaskDof()
        return self._desired_block()[:, 2 * m:]


class RobotController:
    """RobotController.h:47-90 for B robots at once."""

    def __init__(self, robot: SaiModel, tasks, _private=False, leading_dimension=None, _adopt=None):
        """leading_dimension (engine extra): the leading dimension of the device arrays when it must be larger than the batch rounded up
        to 32 -- the shards of a sharded run all take the largest shard's, so that the final all-gather moves slabs of one shape.
        _adopt: an unfinalized batch handle owned by somebody else (saip_multi_batch of sharding.MultiController): tasks are added here, the
        owner finalizes all its batches together and then calls _after_finalize(); the handle is not destroyed by this object."""
        L = capi.lib()
        if len(tasks) == 0:  # RobotController.cpp:11-14
            raise ValueError("RobotController must have at least one task")
        for t in tasks:
            if t.getConstRobotModel() is not robot:  # :28-31
                raise ValueError("All tasks must have the same robot model in RobotController")
        for t in tasks:
            if t._ctrl is not None and not t._ctrl._private:
                raise ValueError(f"task [{t.getTaskName()}] already belongs to a RobotController")
        self._owns = _adopt is None
        if self._owns:
            h = C.c_void_p()
            capi.check(L.saip_batch_create(robot._h, robot.batch_size, robot.device, C.byref(h)))
        else:
            h = C.c_void_p(_adopt)
        if leading_dimension is not None:
            try:
                capi.check(L.saip_batch_set_leading_dimension(h, int(leading_dimension)))
            except Exception:
                if self._owns:
                    L.saip_batch_destroy(h)
                raise
        self._h = h
        self._robot = robot
        self.batch_size = robot.batch_size
        self._tasks = list(tasks)
        self._private = bool(_private)  # the one-task batch of a task driven by hand (TemplateTask.h:43-60), not a user-visible controller
        try:
            for t in tasks:
                t._id = t._add(L, h)
            if self._owns:
                capi.check(L.saip_batch_finalize(h))  # remaining constructor checks, :32-58
        except Exception:
            if self._owns:
                L.saip_batch_destroy(h)
            self._h = None
            raise
        if self._owns:
            self._after_finalize()

    def _after_finalize(self):
        """the part of the constructor behind saip_batch_finalize: the tasks' logged configuration is replayed into the batch"""
        L, h, robot, tasks, _private = capi.lib(), self._h, self._robot, self._tasks, self._private
        for t in tasks:
            goal = None
            if t._ctrl is not None:  # the task was driven by hand before: it moves here (configuration replayed, goal kept; integrators start afresh)
                if robot.device >= 0:
                    old_id, t._id = t._id, 0
                    goal = t._get_goal()
                    t._id = old_id
                t._ctrl._release()
            t._ctrl = self
            t._manual = False
            for fn, args, _key in t._log:
                self._call(fn, t._id, *args)
            if goal is not None:
                a = np.ascontiguousarray(goal.T)
                capi.check(L.saip_batch_set_goal_host(h, t._id, _dptr(a)))
        if not _private:
            robot._controller = self
        robot._controllers.add(self)
        self._pushed_version = -1
        self._has_device = robot.device >= 0

    def __del__(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owns", True):
                capi.lib().saip_batch_destroy(self._h)
            self._h = None

    def _release(self):
        """a private one-task batch whose task joins a RobotController"""
        self._robot._controllers.discard(self)
        self.__del__()

    def _call(self, fn_name, *args):
        capi.check(getattr(capi.lib(), fn_name)(self._h, *args))

    def _push_state(self):
        r = self._robot
        if self._pushed_version != r._state_version:
            q, dq = np.ascontiguousarray(r._q.T), np.ascontiguousarray(r._dq.T)
            capi.check(capi.lib().saip_batch_set_state_host(self._h, _dptr(q), _dptr(dq)))
            self._pushed_version = r._state_version

    # -- reference API
    def updateControllerTaskModels(self):
        self._push_state()
        self._call("saip_batch_update_task_models")
        for t in self._tasks:
            t._manual = False

    def computeControlTorques(self):
        """returns (B, dof) joint torques; per-instance status in self.status (0 ok; bit 0 = refused: that row holds the last valid torques --
        zero before the first -- or NaN with setFlaggedTorquePolicy(True); check the status, not isnan)"""
        n, B = self._robot.dof(), self.batch_size
        tau = np.empty((n, B))
        st = np.zeros(B, np.uint8)
        capi.check(capi.lib().saip_batch_compute_control_torques(self._h, _dptr(tau), st.ctypes.data_as(C.POINTER(C.c_ubyte))))
        self.status = st
        return tau.T.copy()

    def enableGravityCompensation(self, e: bool):
        self._call("saip_batch_enable_gravity_compensation", int(e))

    def enableJointLimitAvoidance(self, e: bool):
        self._call("saip_batch_enable_joint_limit_avoidance", int(e))

    def enableTorqueSaturation(self, e: bool):
        self._call("saip_batch_enable_torque_saturation", int(e))

    def reinitializeTasks(self):
        self._push_state()
        self._call("saip_batch_reinitialize_tasks")

    def getTaskNames(self):
        return [t.getTaskName() for t in self._tasks]

    def _by_name(self, name, typ, what):
        for t in self._tasks:
            if t.getTaskName() == name:
                if t.getTaskType() != typ:  # RobotController.cpp:124-158
                    raise ValueError(f"Task {name} is not a {what}, and cannot be casted as such in RobotController::GetTaskByName")
                return t
        raise ValueError(f"Task {name} not found in RobotController::GetTaskByName")

    def getJointTaskByName(self, name):
        return self._by_name(name, TaskType.JOINT_TASK, "JointTask")

    def getMotionForceTaskByName(self, name):
        return self._by_name(name, TaskType.MOTION_FORCE_TASK, "MotionForceTask")

    # -- engine extras (resident pipelines, benchmarking)
    def setFlaggedTorquePolicy(self, nan: bool):
        """torques of instances that end a cycle flagged (status 1): False (default) = the last valid torques are held, True = NaN"""
        self._call("saip_batch_set_flagged_torque_policy", int(bool(nan)))

    def setFlaggedRecompute(self, on_list: bool):
        """where the eight-lane kernels recompute instances outside the non-singular branch: False (default) = in the kernel's slow tail, True = on the
        device-side list behind every cycle (saip_batch_set_flagged_recompute: faster when many instances of one group are singular at once)"""
        self._call("saip_batch_set_flagged_recompute", int(bool(on_list)))

    def setIntegratorTracking(self, always: bool):
        self._call("saip_batch_set_integrator_tracking", int(always))

    def setKernel(self, which: int):
        self._call("saip_batch_set_kernel", int(which))

    def kernelName(self) -> str:
        return capi.lib().saip_batch_kernel_name(self._h).decode()

    def setGoals(self, goals):
        """goals[t]: (B, goal_components) whole goal block per task, controller order"""
        for t, g in zip(self._tasks, goals):
            gs = capi.lib().saip_batch_goal_components(self._h, t._id)
            g = np.asarray(g, float)
            if g.ndim == 2 and g.shape[1] in (24, 30) and gs > g.shape[1]:  # no goal / sensed force and moment given: zeros
                g = np.concatenate([g, np.zeros((g.shape[0], gs - g.shape[1]))], axis=1)
            a = _soa(g, self.batch_size, gs, f"goal of task {t.getTaskName()}")
            capi.check(capi.lib().saip_batch_set_goal_host(self._h, t._id, _dptr(a)))

    def stepAsync(self):
        self._call("saip_batch_step_async")

    # -- the step after the path: resident forward dynamics (what the examples do with sim->integrate(), examples/05-...cpp:225-231)
    @staticmethod
    def _grav(gravity):
        if gravity is None:
            return None, None
        g = np.ascontiguousarray(np.asarray(gravity, float).reshape(3))
        return g, _dptr(g)

    def integrate(self, dt: float, substeps: int = 1, gravity=None, damping: float = 0.0):
        """semi-implicit Euler steps of the resident state under the torques of the last cycle; gravity None = model gravity"""
        self._push_state()
        g, gp = self._grav(gravity)
        self._call("saip_batch_integrate", float(dt), int(substeps), gp, float(damping))

    def rolloutAsync(self, steps: int, sim_dt: float, substeps: int = 1, gravity=None, damping: float = 0.0):
        """`steps` closed-loop periods {internal OTGs, control cycle, integrate} enqueued without host synchronisation"""
        self._push_state()
        g, gp = self._grav(gravity)
        self._call("saip_batch_rollout_async", int(steps), float(sim_dt), int(substeps), gp, float(damping))

    def setTorques(self, tau):
        """overwrite the resident torques the next integrate() applies: (B, dof)"""
        a = _soa(np.asarray(tau, float), self.batch_size, self._robot.dof(), "torques")
        self._call("saip_batch_set_torques_host", _dptr(a))

    def pullState(self):
        """read the resident state back into the SaiModel mirror (after integrate / rolloutAsync); returns (q, dq) (B, dof)"""
        r = self._robot
        n, B = r.dof(), self.batch_size
        q, dq = np.empty((n, B)), np.empty((n, B))
        self._call("saip_batch_get_state_host", _dptr(q), _dptr(dq))
        r._q, r._dq = q.T.copy(), dq.T.copy()
        r._state_version += 1
        self._pushed_version = r._state_version  # the device already holds this state
        return r._q, r._dq

    def synchronize(self):
        self._call("saip_batch_synchronize")

    def getTorques(self):
        n, B = self._robot.dof(), self.batch_size
        tau = np.empty((n, B))
        st = np.zeros(B, np.uint8)
        capi.check(capi.lib().saip_batch_get_torques_host(self._h, _dptr(tau), st.ctypes.data_as(C.POINTER(C.c_ubyte))))
        self.status = st
        return tau.T.copy()

    def timeSteps(self, steps: int, warmup: int = 0) -> float:
        ms = C.c_double(0)
        capi.check(capi.lib().saip_batch_time_steps(self._h, steps, warmup, C.byref(ms)))
        return ms.value

    def timeStepsBegin(self, steps: int):
        """enqueue `steps` cycles between two HIP events on the engine stream and return without waiting (saip_batch_time_steps_begin)"""
        capi.check(capi.lib().saip_batch_time_steps_begin(self._h, int(steps)))

    def timeStepsEnd(self) -> float:
        """event time (ms) of the cycles timeStepsBegin enqueued; call it behind the caller's own wait for the device"""
        ms = C.c_double(0)
        capi.check(capi.lib().saip_batch_time_steps_end(self._h, C.byref(ms)))
        return ms.value

    def timeStepsGather(self, steps: int, comm=None, gathered_ptr=None, every_step: bool = False):
        """(elapsed_ms, gather_ms) of `steps` cycles with the torque all-gather inside the timed region (saip_batch_time_steps_gather): one gather
        behind the last cycle, or one behind every cycle; comm None (one rank) issues no collective"""
        e, g = C.c_double(0), C.c_double(0)
        capi.check(capi.lib().saip_batch_time_steps_gather(self._h, comm, C.c_void_p(gathered_ptr) if gathered_ptr else None, int(steps),
                                                           1 if every_step else 0, C.byref(e), C.byref(g)))
        return e.value, g.value

    def devicePointers(self):
        L = capi.lib()
        return dict(q=L.saip_batch_device_q(self._h), dq=L.saip_batch_device_dq(self._h), tau=L.saip_batch_device_tau(self._h),
                    status=L.saip_batch_device_status(self._h), ld=L.saip_batch_ld(self._h), stream=L.saip_batch_stream(self._h))

    def bindTauDevice(self, ptr):
        self._call("saip_batch_bind_tau_device", C.c_void_p(ptr))


def controller_from_specs(description, tasks, batch_size, device=0, *, disable_otg=True, leading_dimension=None):
    """Build (robot, controller, task objects) from workloads.py-style task specs (used by tests and bench)."""
    robot = SaiModel(description, batch_size, device)
    objs = tasks_from_specs(robot, tasks, disable_otg=disable_otg)
    ctrl = RobotController(robot, objs, leading_dimension=leading_dimension)
    return robot, ctrl, objs


def tasks_from_specs(robot, tasks, *, disable_otg=True):
    """task objects (not yet in any RobotController) from workloads.py-style task specs"""
    objs = []
    for t in tasks:
        if t["type"] == "motion_force":
            o = MotionForceTask(robot, t["link"], t["pos_in_link"], t.get("rot_in_link"), t.get("dirs_trans"), t.get("dirs_rot"),
                                task_name=t["name"], loop_timestep=t.get("dt", 0.001),
                                is_force_motion_parametrization_in_compliant_frame=t.get("param_in_compliant_frame", False))
            o.setPosControlGains(t["kp_pos"], t["kv_pos"], t["ki_pos"])
            o.setOriControlGains(t["kp_ori"], t["kv_ori"], t["ki_ori"])
            o.setSingularityHandlingBounds(t["s_min"], t["s_max"])
            if not t.get("singularity_handling", True):
                o.disableSingularityHandling()
            o.setSingularityStrategies(bool(t.get("singularity_strategies", True)))
            if "sh_gains" in t:
                o.setSingularityHandlingGains(*t["sh_gains"])
            if t.get("sh_all_type1", False):
                o.handleAllSingularitiesAsType1(True)
            if t.get("cl_force") or t.get("cl_moment"):
                o.setForceControlParameters(t.get("kff_force", 0.95), t.get("kff_moment", 0.95), t.get("max_force_fb", 20.0), t.get("max_moment_fb", 10.0))
                o.setControlToSensorTransform(t.get("Rcs", np.eye(3)), t.get("tcs", np.zeros(3)))
                o.setClosedLoopForceControl(bool(t.get("cl_force")))
                o.setClosedLoopMomentControl(bool(t.get("cl_moment")))
                if t.get("passivity"):
                    o.enablePassivity()
            o.setForceControlGains(t.get("kp_force", 0.7), t.get("kv_force", 10.0), t.get("ki_force", 1.3))
            o.setMomentControlGains(t.get("kp_moment", 0.7), t.get("kv_moment", 10.0), t.get("ki_moment", 1.3))
            if t.get("force_dim", 0):
                o.parametrizeForceMotionSpaces(t["force_dim"], t.get("force_axis"))
            if t.get("moment_dim", 0):
                o.parametrizeMomentRotMotionSpaces(t["moment_dim"], t.get("moment_axis"))
            if t.get("vel_sat", False):
                o.enableVelocitySaturation(t["lin_sat"], t["ang_sat"])
        else:
            S = t["S"]
            if S is not None:
                S = np.asarray(S, float)
                if S.ndim == 1:
                    rows = [int(i) for i in S]
                    S = np.zeros((len(rows), robot.dof()))
                    S[np.arange(len(rows)), rows] = 1.0
            o = JointTask(robot, S, task_name=t["name"], loop_timestep=t.get("dt", 0.001))
            o.setGains(t["kp"], t["kv"], t["ki"])
            if t.get("vel_sat", False):
                o.enableVelocitySaturation(t["sat"])
        o.setDynamicDecouplingType(t["decoupling"])
        o.setBoundedInertiaEstimateThreshold(t["bie_threshold"])
        if disable_otg:
            o.disableInternalOtg()
        objs.append(o)
    return objs

"""ctypes binding of oracle/_ref/libsaip_otg_ref.so -- the REAL vendored Ruckig of the reference behind a C-ABI driver
(oracle/ref_ruckig/otg_ref_shim.cpp).  TEST INFRASTRUCTURE ONLY: importable from tests/ and tests/golden/make_otg_golden.py,
never from the product package.  The library exists only where it was built (this container, `make -C oracle ref`) or where the
prebuilt file travelled to; available() says which."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_DIR, "_ref", "libsaip_otg_ref.so")
REFERENCE_TREE = "/root/reference/ruckig/include"
_LIB = None


def build() -> str | None:
    """(re)build from the reference tree when it is present; otherwise keep whatever prebuilt file is there"""
    if os.path.isdir(REFERENCE_TREE):
        src = os.path.join(_DIR, "ref_ruckig", "otg_ref_shim.cpp")
        if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _DIR, "-B", "ref"], stdout=subprocess.DEVNULL)
    return SO if os.path.exists(SO) else None


def available() -> bool:
    return os.path.exists(SO)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(SO)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.otgref_joints_create.restype = C.c_void_p
        L.otgref_joints_create.argtypes = [C.c_int, C.c_double, dp, dp, dp]
        L.otgref_joints_destroy.argtypes = [C.c_void_p]
        L.otgref_joints_reinitialize.argtypes = [C.c_void_p, dp]
        L.otgref_joints_set_limits.argtypes = [C.c_void_p, dp, dp]
        L.otgref_joints_set_jerk_limits.argtypes = [C.c_void_p, dp, dp, dp]
        L.otgref_cart_set_jerk_limits.argtypes = [C.c_void_p, dp]
        L.otgref_joints_step.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, ip]
        L.otgref_joints_finished_with_velocity.argtypes = [C.c_void_p]
        L.otgref_cart_create.restype = C.c_void_p
        L.otgref_cart_create.argtypes = [C.c_double, dp, dp, dp]
        L.otgref_cart_destroy.argtypes = [C.c_void_p]
        L.otgref_cart_set_limits.argtypes = [C.c_void_p, dp]
        L.otgref_cart_reinitialize.argtypes = [C.c_void_p, dp, dp, C.c_int]
        L.otgref_cart_step.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, ip]
        L.otgref_cart_finished_with_velocity.argtypes = [C.c_void_p]
        L.otgref_calculate.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, C.c_int, dp, dp, dp, dp, C.c_int, dp, dp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class JointsOtgRef:
    """one OTG_joints instance (acceleration-limited), /root/reference/src/helper_modules/OTG_joints.cpp"""

    def __init__(self, q0, dt, vmax, amax):
        q0 = np.ascontiguousarray(q0, float)
        self.n = q0.size
        vm = np.ascontiguousarray(np.broadcast_to(np.asarray(vmax, float), (self.n,)))
        am = np.ascontiguousarray(np.broadcast_to(np.asarray(amax, float), (self.n,)))
        self._h = lib().otgref_joints_create(self.n, float(dt), _p(q0), _p(vm), _p(am))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().otgref_joints_destroy(self._h)
            self._h = None

    def reinitialize(self, q0):
        q0 = np.ascontiguousarray(q0, float)
        lib().otgref_joints_reinitialize(self._h, _p(q0))

    def set_limits(self, vmax, amax):
        vm = np.ascontiguousarray(np.broadcast_to(np.asarray(vmax, float), (self.n,)))
        am = np.ascontiguousarray(np.broadcast_to(np.asarray(amax, float), (self.n,)))
        lib().otgref_joints_set_limits(self._h, _p(vm), _p(am))

    def set_jerk_limits(self, vmax, amax, jmax):
        """JointTask::enableInternalOtgJerkLimited's setters (the re-initialisation that may precede them is reinitialize())"""
        vm, am, jm = (np.ascontiguousarray(np.broadcast_to(np.asarray(x, float), (self.n,))) for x in (vmax, amax, jmax))
        lib().otgref_joints_set_jerk_limits(self._h, _p(vm), _p(am), _p(jm))

    def step(self, goal_pos, goal_vel=None):
        gp = np.ascontiguousarray(goal_pos, float)
        gv = np.zeros(self.n) if goal_vel is None else np.ascontiguousarray(goal_vel, float)
        pos, vel, acc = np.empty(self.n), np.empty(self.n), np.empty(self.n)
        gr = C.c_int(0)
        res = lib().otgref_joints_step(self._h, _p(gp), _p(gv), _p(pos), _p(vel), _p(acc), C.byref(gr))
        return pos, vel, acc, bool(gr.value), res

    @property
    def finished_with_velocity(self):
        return lib().otgref_joints_finished_with_velocity(self._h)


class CartesianOtgRef:
    """one OTG_6dof_cartesian instance (acceleration-limited), /root/reference/src/helper_modules/OTG_6dof_cartesian.cpp.
    limits = (max linear velocity, max linear acceleration, max angular velocity, max angular acceleration)"""

    def __init__(self, pos0, R0, dt, limits=(0.3, 2.0, np.pi / 3, 2 * np.pi)):
        p, R, lim = (np.ascontiguousarray(x, float) for x in (pos0, np.asarray(R0).reshape(9), limits))
        self._h = lib().otgref_cart_create(float(dt), _p(p), _p(R), _p(lim))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().otgref_cart_destroy(self._h)
            self._h = None

    def set_limits(self, limits):
        lim = np.ascontiguousarray(limits, float)
        lib().otgref_cart_set_limits(self._h, _p(lim))

    def set_jerk_limits(self, limits6):
        """(max linear velocity, acceleration, jerk, max angular velocity, acceleration, jerk): MotionForceTask::enableInternalOtgJerkLimited"""
        lim = np.ascontiguousarray(limits6, float)
        assert lim.size == 6
        lib().otgref_cart_set_jerk_limits(self._h, _p(lim))

    def reinitialize(self, pos, R, mask=3):
        p, Rm = np.ascontiguousarray(pos, float), np.ascontiguousarray(np.asarray(R).reshape(9), float)
        lib().otgref_cart_reinitialize(self._h, _p(p), _p(Rm), int(mask))

    def step(self, goal_pos, goal_R, goal_linvel=None, goal_angvel=None):
        """-> (desired[24] in the motion-force goal layout, goal_reached, ruckig result)"""
        gp = np.ascontiguousarray(goal_pos, float)
        gR = np.ascontiguousarray(np.asarray(goal_R).reshape(9), float)
        gv = np.zeros(3) if goal_linvel is None else np.ascontiguousarray(goal_linvel, float)
        gw = np.zeros(3) if goal_angvel is None else np.ascontiguousarray(goal_angvel, float)
        out = np.zeros(24)
        gr = C.c_int(0)
        res = lib().otgref_cart_step(self._h, _p(gp), _p(gR), _p(gv), _p(gw), _p(out), C.byref(gr))
        return out, bool(gr.value), res

    @property
    def finished_with_velocity(self):
        return lib().otgref_cart_finished_with_velocity(self._h)


def calculate(p0, v0, pf, vf, vmax, amax, sync=2, times=()):
    """raw state-to-state trajectory: dict(result, duration, t[dim,7], a[dim,7], brake[dim,2], samples[ns,3,dim])"""
    arrs = [np.ascontiguousarray(x, float) for x in (p0, v0, pf, vf, vmax, amax)]
    n = arrs[0].size
    dur = C.c_double(0)
    t, a, br = np.zeros((n, 7)), np.zeros((n, 7)), np.zeros((n, 2))
    tm = np.ascontiguousarray(times, float)
    smp = np.zeros((tm.size, 3, n))
    r = lib().otgref_calculate(n, *[_p(x) for x in arrs], int(sync), C.byref(dur), _p(t), _p(a), _p(br), tm.size, _p(tm), _p(smp))
    return dict(result=r, duration=dur.value, t=t, a=a, brake=br, samples=smp)

"""ctypes binding of include/saip.h.  Fails loudly when libsaip.so has not been built."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "libsaip.so")
SOURCES = ["csrc/saip_engine.cpp", "csrc/saip_comm.cpp", "csrc/saip_kernel_wg.hip", "csrc/saip_kernel_lane.hip", "csrc/saip_kernel_lane_lean.hip", "csrc/saip_kernel_oct.hip", "csrc/saip_kernel_octjf.hip", "csrc/saip_kernel_wave.hip", "csrc/saip_otg.hip", "csrc/saip_dynamics.hip", "csrc/saip_dynamics_oct.hip"]
HEADERS = ["csrc/saip_device.h", "csrc/saip_law.h", "csrc/saip_wg_linalg.h", "csrc/saip_wg_cycle.h", "csrc/saip_fk.h", "csrc/saip_oct_common.h", "csrc/saip_wave_prims.h", "csrc/saip_dynamics_oct_body.h", "../include/saip.h"]

SAIP_OK, SAIP_ERR_INVALID_ARGUMENT, SAIP_ERR_UNSUPPORTED, SAIP_ERR_NO_DEVICE, SAIP_ERR_DEVICE, SAIP_ERR_ORDER = range(6)
NAME_LEN = 48


class SaipError(RuntimeError):
    pass


class SaipUnsupported(NotImplementedError):
    pass


class SaipNoDevice(RuntimeError):
    pass


class LinkDesc(C.Structure):
    _fields_ = [("name", C.c_char * NAME_LEN), ("joint_type", C.c_int), ("origin_xyz", C.c_double * 3),
                ("origin_rpy", C.c_double * 3), ("axis", C.c_double * 3), ("mass", C.c_double), ("com", C.c_double * 3),
                ("inertia", C.c_double * 6), ("q_lower", C.c_double), ("q_upper", C.c_double),
                ("velocity_limit", C.c_double), ("effort_limit", C.c_double)]


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc cross-compiles for gfx950 (works without a GPU). Output stays in-tree: sai-primitives_amd/libsaip.so"""
    srcs = [os.path.join(_DIR, s) for s in SOURCES]
    deps = srcs + [os.path.join(_DIR, h) for h in HEADERS]
    exact_lib = os.path.join(_DIR, "libsaip_octexact.so")
    if not force and os.path.exists(LIB_PATH) and os.path.exists(exact_lib) and all(min(os.path.getmtime(LIB_PATH), os.path.getmtime(exact_lib)) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    extra = os.environ.get("SAIP_EXTRA_HIPCC_FLAGS", "").split()  # diagnostic builds only (e.g. -DSAIP_STAMP)
    # One object per source, then one link.  The cycle kernels run at one wavefront per SIMD (512 registers per lane), so instruction
    # level parallelism is the only latency hiding there is: they are built with the max-ilp machine scheduling strategy (measured
    # -2.4 % cfg2 / -3.8 % cfg3 against the default); the OTG and dynamics kernels are faster with the default strategy (integrate
    # 33.6 vs 40.0 us).
    sched = {"csrc/saip_kernel_lane.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
             "csrc/saip_kernel_wg.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
             "csrc/saip_kernel_oct.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
             "csrc/saip_kernel_octjf.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
    objs = []
    procs = []
    # test-only variant of the eight-lane cycle kernel in which EVERY instance takes the exact eigen fallbacks (they are rare on real
    # data): linked into libsaip_octexact.so next to the product library, loaded by tests/test_gpu_oct.py through SAIP_LIB
    oct_rel = "csrc/saip_kernel_oct.hip"
    oct_exact_obj = os.path.join(_DIR, "build", "saip_kernel_oct_exact.o")
    os.makedirs(os.path.dirname(oct_exact_obj), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-c", "-DSAIP_OCT_FORCE_EXACT", "-o", oct_exact_obj] + sched[oct_rel] + extra + [os.path.join(_DIR, oct_rel)]
    procs.append((cmd, subprocess.Popen(cmd)))
    jf_rel = "csrc/saip_kernel_octjf.hip"   # the joint-first eight-lane kernel has the same kind of fallback: same treatment
    jf_exact_obj = os.path.join(_DIR, "build", "saip_kernel_octjf_exact.o")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-c", "-DSAIP_OCT_FORCE_EXACT", "-o", jf_exact_obj] + sched[jf_rel] + extra + [os.path.join(_DIR, jf_rel)]
    procs.append((cmd, subprocess.Popen(cmd)))
    for rel in SOURCES:
        obj = os.path.join(_DIR, "build", os.path.basename(rel) + ".o")
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-c", "-o", obj] + sched.get(rel, []) + extra + [os.path.join(_DIR, rel)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs)
    exact_objs = [oct_exact_obj if os.path.basename(o) == "saip_kernel_oct.hip.o" else (jf_exact_obj if os.path.basename(o) == "saip_kernel_octjf.hip.o" else o) for o in objs]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", exact_lib] + exact_objs)
    # objects of one-off experiments (tools/oct_variant.sh and friends leave build/*_var_*.o behind) are not part of the product: they would
    # travel to the GPU box with every gpurun snapshot
    keep = {os.path.basename(o) for o in objs + [oct_exact_obj, jf_exact_obj]}
    for f in os.listdir(os.path.join(_DIR, "build")):
        if f.endswith(".o") and f not in keep:
            os.remove(os.path.join(_DIR, "build", f))
    return LIB_PATH


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("SAIP_LIB", LIB_PATH)  # A/B measurements of two builds inside one GPU session
    if not os.path.exists(path):
        raise SaipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(the engine has no CPU path)")
    L = C.CDLL(path)
    dp, vp, ip = C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int)
    u8p = C.POINTER(C.c_ubyte)
    sig = {
        "saip_model_create_serial_chain": (C.c_int, [C.POINTER(LinkDesc), C.c_int, C.POINTER(vp)]),
        "saip_model_destroy": (None, [vp]),
        "saip_model_dof": (C.c_int, [vp]),
        "saip_model_link_index": (C.c_int, [vp, C.c_char_p]),
        "saip_model_joint_limits": (C.c_int, [vp, dp, dp, dp, dp]),
        "saip_batch_create": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
        "saip_batch_destroy": (None, [vp]),
        "saip_batch_size": (C.c_int, [vp]),
        "saip_batch_ld": (C.c_int, [vp]),
        "saip_batch_set_leading_dimension": (C.c_int, [vp, C.c_int]),
        "saip_batch_add_motion_force_task": (C.c_int, [vp, C.c_char_p, C.c_char_p, dp, dp, dp, C.c_int, dp, C.c_int, C.c_double, ip]),
        "saip_batch_add_joint_task": (C.c_int, [vp, C.c_char_p, dp, C.c_int, C.c_double, ip]),
        "saip_batch_finalize": (C.c_int, [vp]),
        "saip_batch_task_count": (C.c_int, [vp]),
        "saip_batch_task_type": (C.c_int, [vp, C.c_int]),
        "saip_batch_task_name": (C.c_char_p, [vp, C.c_int]),
        "saip_batch_task_by_name": (C.c_int, [vp, C.c_char_p]),
        "saip_batch_task_dof": (C.c_int, [vp, C.c_int]),
        "saip_batch_goal_components": (C.c_int, [vp, C.c_int]),
        "saip_batch_get_task_projection": (C.c_int, [vp, C.c_int, dp, dp, ip]),
        "saip_batch_set_pos_control_gains": (C.c_int, [vp, C.c_int, dp, dp, dp, C.c_int]),
        "saip_batch_set_ori_control_gains": (C.c_int, [vp, C.c_int, dp, dp, dp, C.c_int]),
        "saip_batch_set_joint_gains": (C.c_int, [vp, C.c_int, dp, dp, dp, C.c_int]),
        "saip_batch_set_dynamic_decoupling_type": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_bie_threshold": (C.c_int, [vp, C.c_int, C.c_double]),
        "saip_batch_set_singularity_bounds": (C.c_int, [vp, C.c_int, C.c_double, C.c_double]),
        "saip_batch_set_internal_otg": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_singularity_handling": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_singularity_strategies": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_reinitialize_task": (C.c_int, [vp, C.c_int]),
        "saip_batch_get_current_pose_host": (C.c_int, [vp, C.c_int, dp, dp]),
        "saip_batch_reset_integrators": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_singularity_gains": (C.c_int, [vp, C.c_int, C.c_double, C.c_double, C.c_double]),
        "saip_batch_set_all_singularities_type1": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_type1_posture": (C.c_int, [vp, C.c_int, dp, C.c_int]),
        "saip_batch_set_otg_acceleration_limited": (C.c_int, [vp, C.c_int, dp, dp, C.c_int]),
        "saip_batch_set_otg_jerk_limited": (C.c_int, [vp, C.c_int, dp, dp, dp, C.c_int]),
        "saip_batch_get_desired_host": (C.c_int, [vp, C.c_int, dp]),
        "saip_batch_integrate": (C.c_int, [vp, C.c_double, C.c_int, dp, C.c_double]),
        "saip_batch_rollout_async": (C.c_int, [vp, C.c_int, C.c_double, C.c_int, dp, C.c_double]),
        "saip_batch_get_state_host": (C.c_int, [vp, dp, dp]),
        "saip_batch_set_torques_host": (C.c_int, [vp, dp]),
        "saip_batch_get_otg_status_host": (C.c_int, [vp, C.c_int, ip, ip]),
        "saip_batch_set_velocity_saturation": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_saturation_velocities": (C.c_int, [vp, C.c_int, dp, C.c_int]),
        "saip_batch_parametrize_force_motion_spaces": (C.c_int, [vp, C.c_int, C.c_int, dp, ip]),
        "saip_batch_parametrize_moment_rot_motion_spaces": (C.c_int, [vp, C.c_int, C.c_int, dp, ip]),
        "saip_batch_set_parametrization_in_compliant_frame": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_force_control_gains": (C.c_int, [vp, C.c_int, C.c_double, C.c_double, C.c_double]),
        "saip_batch_set_moment_control_gains": (C.c_int, [vp, C.c_int, C.c_double, C.c_double, C.c_double]),
        "saip_batch_set_closed_loop_force_control": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_closed_loop_moment_control": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_passivity": (C.c_int, [vp, C.c_int, C.c_int]),
        "saip_batch_set_force_control_parameters": (C.c_int, [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
        "saip_batch_set_control_to_sensor_transform": (C.c_int, [vp, C.c_int, dp, dp]),
        "saip_batch_enable_gravity_compensation": (C.c_int, [vp, C.c_int]),
        "saip_batch_enable_joint_limit_avoidance": (C.c_int, [vp, C.c_int]),
        "saip_batch_enable_torque_saturation": (C.c_int, [vp, C.c_int]),
        "saip_batch_set_integrator_tracking": (C.c_int, [vp, C.c_int]),
        "saip_batch_set_flagged_torque_policy": (C.c_int, [vp, C.c_int]),
        "saip_batch_set_flagged_recompute": (C.c_int, [vp, C.c_int]),
        "saip_batch_set_state_host": (C.c_int, [vp, dp, dp]),
        "saip_batch_set_goal_host": (C.c_int, [vp, C.c_int, dp]),
        "saip_batch_set_goal_field_host": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, dp]),
        "saip_batch_get_goal_host": (C.c_int, [vp, C.c_int, dp]),
        "saip_batch_reinitialize_tasks": (C.c_int, [vp]),
        "saip_batch_device_q": (vp, [vp]),
        "saip_batch_device_dq": (vp, [vp]),
        "saip_batch_device_goal": (vp, [vp, C.c_int]),
        "saip_batch_device_tau": (vp, [vp]),
        "saip_batch_device_status": (vp, [vp]),
        "saip_batch_bind_tau_device": (C.c_int, [vp, vp]),
        "saip_batch_stream": (vp, [vp]),
        "saip_batch_update_task_models": (C.c_int, [vp]),
        "saip_batch_compute_control_torques": (C.c_int, [vp, dp, u8p]),
        "saip_batch_step_async": (C.c_int, [vp]),
        "saip_batch_synchronize": (C.c_int, [vp]),
        "saip_batch_get_torques_host": (C.c_int, [vp, dp, u8p]),
        "saip_batch_get_task_nullspace_host": (C.c_int, [vp, C.c_int, dp]),
        "saip_batch_task_update_model": (C.c_int, [vp, C.c_int, dp]),
        "saip_batch_task_update_model_device": (C.c_int, [vp, C.c_int, vp]),
        "saip_batch_task_compute_torques": (C.c_int, [vp, C.c_int, dp, dp, u8p]),
        "saip_batch_task_compute_torques_device": (C.c_int, [vp, C.c_int, vp, vp]),
        "saip_batch_task_get_nullspaces_host": (C.c_int, [vp, C.c_int, dp, dp, dp]),
        "saip_batch_task_device_nullspace": (vp, [vp, C.c_int, C.c_int]),
        "saip_batch_task_device_torques": (vp, [vp, C.c_int]),
        "saip_batch_set_state_device": (C.c_int, [vp, vp, vp]),
        "saip_batch_wait_for": (C.c_int, [vp, vp]),
        "saip_batch_set_kernel": (C.c_int, [vp, C.c_int]),
        "saip_batch_kernel_name": (C.c_char_p, [vp]),
        "saip_batch_time_steps": (C.c_int, [vp, C.c_int, C.c_int, dp]),
        "saip_batch_dof": (C.c_int, [vp]),
        "saip_comm_probe": (C.c_int, [C.c_int]),
        "saip_comm_unique_id": (C.c_int, [vp]),
        "saip_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]),
        "saip_comm_destroy": (None, [vp]),
        "saip_comm_world": (C.c_int, [vp]),
        "saip_comm_rank": (C.c_int, [vp]),
        "saip_batch_all_gather_torques": (C.c_int, [vp, vp, vp]),
        "saip_batch_time_steps_begin": (C.c_int, [vp, C.c_int]),
        "saip_batch_time_steps_end": (C.c_int, [vp, dp]),
        "saip_batch_time_steps_gather": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, dp, dp]),
        "saip_multi_time_steps": (C.c_int, [vp, C.c_int, C.c_int, dp, dp]),
        "saip_multi_create": (C.c_int, [vp, C.c_int, ip, C.c_int, C.POINTER(vp)]),
        "saip_multi_size": (C.c_int, [vp]),
        "saip_multi_batch": (vp, [vp, C.c_int]),
        "saip_multi_finalize": (C.c_int, [vp]),
        "saip_multi_step_async": (C.c_int, [vp]),
        "saip_multi_all_gather_torques": (C.c_int, [vp]),
        "saip_multi_synchronize": (C.c_int, [vp]),
        "saip_multi_gathered_device": (vp, [vp, C.c_int]),
        "saip_multi_get_gathered_host": (C.c_int, [vp, C.c_int, dp]),
        "saip_multi_destroy": (None, [vp]),
        "saip_last_error": (C.c_char_p, []),
        "saip_version": (C.c_char_p, []),
        "saip_device_count": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    L._declared = sorted(sig)
    _LIB = L
    return L


def check(status: int):
    if status == SAIP_OK:
        return
    msg = lib().saip_last_error().decode()
    if status == SAIP_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)  # the reference throws std::invalid_argument
    if status == SAIP_ERR_UNSUPPORTED:
        raise SaipUnsupported(msg)
    if status == SAIP_ERR_NO_DEVICE:
        raise SaipNoDevice(msg)
    raise SaipError(f"[status {status}] {msg}")


def device_count() -> int:
    return lib().saip_device_count()

"""sai-primitives_amd: MI355X-native batched operational-space control engine.

Python side = a thin ctypes binding of the C-ABI in include/saip.h (`capi`) plus a mirror of the
reference's task/controller interface in a batched flavour (`SaiModel`, `MotionForceTask`, `JointTask`,
`RobotController`, same method names and error behaviour as /root/reference/src/RobotController.h:47-90,
tasks/MotionForceTask.h, tasks/JointTask.h).  All arithmetic happens in the HIP kernels of csrc/;
there is no CPU path: without the built library or without a GPU every compute call raises.
"""
from .capi import lib, build_library, SaipError, SaipUnsupported, SaipNoDevice, device_count  # noqa: F401
from .controller import (SaiModel, MotionForceTask, JointTask, RobotController, TaskType,  # noqa: F401
                         DynamicDecouplingType, PIDGains, load_robot_description)

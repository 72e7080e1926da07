"""Multi-GPU plumbing: robot instances are independent, so a batch shards by contiguous instance ranges
(one process per GPU, no data-path collective).  The only collective is the optional gather of the torque slabs
(RCCL all_gather over xGMI on GPUs, gloo in the CPU tests).  torch.distributed is used for rendezvous/collectives
only; the engine itself never touches torch."""
from __future__ import annotations

import os


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total: int, world: int, rank: int):
    """contiguous block of instances owned by `rank` (sizes differ by at most one)"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_ld(total: int, world: int) -> int:
    """common leading dimension of the per-rank [dof][ld] torque slabs: the engine's ld (multiple of 32) of the LARGEST shard, so that
    all_gather_into_tensor sees identical shapes on every rank even when the shards differ by one instance"""
    return (-(-total // world) + 31) // 32 * 32


def init_process_group(backend: str):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend)
    return dist


def gather_buffer(tau_local, world: int):
    """the [world, dof, ld] result buffer of gather_torques (allocate it outside a timed region)"""
    import torch
    return torch.empty((world,) + tuple(tau_local.shape), dtype=tau_local.dtype, device=tau_local.device)


def gather_torques(tau_local, world: int, out=None):
    """tau_local: torch tensor [dof, ld] (device for nccl, cpu for gloo) -> [world, dof, ld] on every rank.  Every rank's slab must
    have the same shape: ranks with uneven shards pad to common_ld(total, world) (see shard_ld)."""
    import torch.distributed as dist
    if world == 1:
        # a gather over one rank is the identity: the engine already wrote the slab, nothing to move.  The result ALIASES tau_local (it
        # changes with the next cycle) unless an `out` buffer is given, which then receives a copy like on several ranks
        if out is not None:
            out[0].copy_(tau_local)
            return out
        return tau_local.unsqueeze(0)
    if out is None:
        out = gather_buffer(tau_local, world)
    dist.all_gather_into_tensor(out.view((-1,) + tuple(tau_local.shape[1:])), tau_local.contiguous())  # concat along dim 0
    return out


class NativeGather:
    """the final torque gather done by the engine itself: C++ on RCCL (csrc/saip_comm.cpp), ncclAllGather on the batch's stream.
    torch.distributed only carries the 128-byte unique id from rank 0 to the other ranks (the launcher's rendezvous).
    Construction is collective and fails on ALL ranks or on none: every rank first probes what it needs locally (librccl, its device,
    on rank 0 the unique id) and the answers are reduced before anybody enters ncclCommInitRank -- a rank that failed on its own would
    otherwise leave the others blocked inside it.  The slabs must have one shape on every rank (same dof, same leading dimension:
    RobotController(..., leading_dimension=shard_ld(total, world)) for uneven shards); that is checked here as well."""

    def __init__(self, ctrl, world: int, rank: int, local_rank: int, device):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import capi
        L = capi.lib()
        ident = torch.zeros(128, dtype=torch.uint8)
        err = None
        try:
            capi.check(L.saip_comm_probe(int(local_rank)))
            if rank == 0:
                buf = (C.c_ubyte * 128)()
                capi.check(L.saip_comm_unique_id(buf))
                ident = torch.tensor(list(buf), dtype=torch.uint8)
        except Exception as e:  # noqa: BLE001  (the rank still takes part in the reductions below, or the other ranks hang in them)
            err = e
        shape = [float(L.saip_batch_dof(ctrl._h)), float(L.saip_batch_ld(ctrl._h))]
        if world > 1:
            on_dev = dist.get_backend() == "nccl"
            flags = torch.tensor([1.0 if err is not None else 0.0, shape[0], shape[1], -shape[0], -shape[1]], dtype=torch.float64)
            flags = flags.to(device) if on_dev else flags
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            flags = flags.cpu().tolist()
            if flags[0] != 0.0:
                raise RuntimeError(f"native RCCL gather unavailable on some rank (this rank: {err})")
            if flags[1] != -flags[3] or flags[2] != -flags[4]:
                raise RuntimeError(f"torque slabs differ across ranks (this rank: dof {shape[0]:.0f}, ld {shape[1]:.0f}; largest {flags[1]:.0f}, {flags[2]:.0f}): "
                                   "give every rank's controller leading_dimension=shard_ld(total, world)")
            t = ident.to(device) if on_dev else ident
            dist.broadcast(t, src=0)
            ident = t.cpu()
        elif err is not None:
            raise RuntimeError(f"native RCCL gather unavailable ({err})")
        buf = (C.c_ubyte * 128)(*ident.tolist())
        self._comm = C.c_void_p()
        capi.check(L.saip_comm_create(int(local_rank), int(world), int(rank), buf, C.byref(self._comm)))
        self._ctrl, self._L = ctrl, L

    def world(self) -> int:
        """the number of ranks RCCL counts in the communicator (ncclCommCount)"""
        return int(self._L.saip_comm_world(self._comm))

    def gather(self, out):
        """out: torch tensor [world, dof, ld] on this rank's GPU; asynchronous on the engine stream"""
        from . import capi
        capi.check(self._L.saip_batch_all_gather_torques(self._ctrl._h, self._comm, out.data_ptr()))
        return out

    def time_steps(self, steps: int, out, every_step: bool):
        """(elapsed_ms, gather_ms): `steps` cycles + the all-gather (behind the last cycle, or behind every cycle) event-timed on the engine stream"""
        return self._ctrl.timeStepsGather(steps, self._comm, out.data_ptr(), every_step)

    def close(self):
        if self._comm:
            self._L.saip_comm_destroy(self._comm)
            self._comm = None


def max_over_ranks(value: float, device) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class MultiController:
    """ONE process driving n GPUs through the engine's own C++ multi-device layer (saip_multi_*, csrc/saip_comm.cpp: one batch + stream +
    communicator per device, ncclCommInitAll, grouped ncclAllGather) -- no torch, no torch.distributed, no launcher.  Every device gets the
    same task stack (built from workloads.py-style specs) and a contiguous shard of the instances."""

    def __init__(self, description, task_specs, batch_per_device: int, devices, leading_dimension=None):
        import ctypes as C
        from . import capi
        from .controller import SaiModel, RobotController, tasks_from_specs
        L = capi.lib()
        self._L, self._m = L, C.c_void_p()
        self.devices = [int(d) for d in devices]
        self.robots = [SaiModel(description, batch_per_device, d) for d in self.devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        capi.check(L.saip_multi_create(self.robots[0]._h, int(batch_per_device), arr, len(self.devices), C.byref(self._m)))
        self.controllers, self.tasks = [], []
        try:
            for i, robot in enumerate(self.robots):
                objs = tasks_from_specs(robot, task_specs)
                self.controllers.append(RobotController(robot, objs, leading_dimension=leading_dimension, _adopt=L.saip_multi_batch(self._m, i)))
                self.tasks.append(objs)
            capi.check(L.saip_multi_finalize(self._m))   # finalizes every batch, then ncclCommInitAll and the gather buffers
            for c in self.controllers:
                c._after_finalize()
        except Exception:
            self.close()
            raise

    def world(self) -> int:
        return int(self._L.saip_multi_size(self._m))

    def step_async(self):
        from . import capi
        capi.check(self._L.saip_multi_step_async(self._m))

    def all_gather(self):
        from . import capi
        capi.check(self._L.saip_multi_all_gather_torques(self._m))

    def synchronize(self):
        from . import capi
        capi.check(self._L.saip_multi_synchronize(self._m))

    def time_steps(self, steps: int, every_step_gather: bool):
        """(elapsed_ms, gather_ms): HIP events on every device's stream, maximum over the devices (saip_multi_time_steps)"""
        import ctypes as C
        from . import capi
        e, g = C.c_double(0.0), C.c_double(0.0)
        capi.check(self._L.saip_multi_time_steps(self._m, int(steps), 1 if every_step_gather else 0, C.byref(e), C.byref(g)))
        return e.value, g.value

    def gathered(self, i: int):
        """[n_devices][B][dof] host copy of the gathered torques as device i holds them"""
        import numpy as np
        import ctypes as C
        from . import capi
        n, dof, B = self.world(), self.robots[0].dof(), self.robots[0].batch_size
        out = np.empty((n, dof, B))
        capi.check(self._L.saip_multi_get_gathered_host(self._m, int(i), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out.transpose(0, 2, 1)

    def close(self):
        if getattr(self, "_m", None):
            for c in self.controllers:
                c._h = None           # the batches belong to the multi object
            self._L.saip_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        self.close()

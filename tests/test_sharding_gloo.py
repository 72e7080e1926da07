"""world_size-2 test of the N>1 path on CPU (gloo): contiguous instance shards, no data-path collective,
one all_gather of the torque slabs, max-over-ranks timing.  The GPU engine cannot run here, so each rank's
shard is computed by the CPU oracle (allowed in tests/): what is under test is the sharding/gather plumbing
bench.py uses (sai-primitives_amd/sharding.py) and that shard r of the seeded workload is exactly the r-th block
of the global stream."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_uneven(rank, world, port, total, out_dir):
    """shards that differ by one instance: every rank pads its slab to sharding.shard_ld(total, world) so that the collective sees equal shapes"""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from sai_primitives_amd import sharding
    dist = sharding.init_process_group("gloo")
    lo, hi = sharding.shard_range(total, world, rank)
    ld = sharding.shard_ld(total, world)
    slab = torch.zeros((7, ld), dtype=torch.float64)
    slab[:, :hi - lo] = torch.arange(lo, hi, dtype=torch.float64)[None, :] + 1000.0 * torch.arange(7, dtype=torch.float64)[:, None]
    gathered = sharding.gather_torques(slab, world, out=sharding.gather_buffer(slab, world))
    np.save(os.path.join(out_dir, f"uneven_{rank}.npy"), gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, per_rank, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import workloads as W
    from oracle import Oracle
    from sai_primitives_amd import sharding

    r, lr, w = sharding.dist_env()
    assert (r, w) == (rank, world)
    dist = sharding.init_process_group("gloo")
    d = W.make_inputs(2, per_rank, shard=rank)          # what bench.py does per rank
    tau, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"])
    assert st.sum() == 0
    ld = (per_rank + 31) // 32 * 32                      # engine layout: [dof][ld]
    slab = torch.zeros((7, ld), dtype=torch.float64)
    slab[:, :per_rank] = torch.from_numpy(tau.T.copy())
    gathered = sharding.gather_torques(slab, world)      # [world, dof, ld] on every rank
    t = sharding.max_over_ranks(1.0 + rank, torch.device("cpu"))
    assert t == float(world)
    np.save(os.path.join(out_dir, f"gathered_{rank}.npy"), gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_and_gather(tmp_path):
    import torch.multiprocessing as mp
    import workloads as W
    from oracle import Oracle
    from sai_primitives_amd import sharding

    world, per_rank = 2, 24
    mp.spawn(_worker, args=(world, _free_port(), per_rank, str(tmp_path)), nprocs=world, join=True)
    g0 = np.load(tmp_path / "gathered_0.npy")
    g1 = np.load(tmp_path / "gathered_1.npy")
    assert np.array_equal(g0, g1)                        # every rank holds the whole result
    # single-process reference over the global batch: ranks own contiguous blocks of the same stream
    d = W.make_inputs(2, world * per_rank)
    ref, _ = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"])
    for r in range(world):
        lo, hi = sharding.shard_range(world * per_rank, world, r)
        assert (lo, hi) == (r * per_rank, (r + 1) * per_rank)
        assert np.array_equal(g0[r][:, :per_rank].T, ref[lo:hi])


def test_uneven_shards_gather(tmp_path):
    import torch.multiprocessing as mp
    from sai_primitives_amd import sharding
    world, total = 2, 65                                 # 33 + 32 instances: engine ld 64 and 32, common ld 64
    assert sharding.shard_ld(total, world) == 64 and sharding.shard_ld(65536, 8) == 8192 and sharding.shard_ld(5, 8) == 32
    mp.spawn(_worker_uneven, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    g = np.load(tmp_path / "uneven_0.npy")
    assert np.array_equal(g, np.load(tmp_path / "uneven_1.npy")) and g.shape == (2, 7, 64)
    for r in range(world):
        lo, hi = sharding.shard_range(total, world, r)
        assert np.array_equal(g[r][3, :hi - lo], np.arange(lo, hi) + 3000.0)


def test_shard_range_covers_everything_once():
    from sai_primitives_amd import sharding
    for total, world in [(65536, 8), (4097, 8), (10, 3), (5, 8)]:
        seen = []
        for r in range(world):
            lo, hi = sharding.shard_range(total, world, r)
            seen += list(range(lo, hi))
        assert seen == list(range(total))


def test_uneven_shards_share_one_leading_dimension():
    """the native gather (saip_batch_all_gather_torques) moves dof * ld doubles per rank: shards that differ by one instance must be created
    with the largest shard's leading dimension (saip_batch_set_leading_dimension), which NativeGather checks across ranks"""
    import pytest
    from sai_primitives_amd import sharding
    from sai_primitives_amd.controller import controller_from_specs
    import workloads as W
    total, world = 4097, 8            # shards of 513 and 512 instances: default ld 544 and 512
    ld = sharding.shard_ld(total, world)
    assert ld == 544
    tasks = W.config_tasks(2)
    for rank in (0, 7):
        lo, hi = sharding.shard_range(total, world, rank)
        _, ctrl, _ = controller_from_specs("panda_arm", tasks, hi - lo, device=-1, leading_dimension=ld)
        assert ctrl.devicePointers()["ld"] == ld
    with pytest.raises(ValueError):
        controller_from_specs("panda_arm", tasks, 513, device=-1, leading_dimension=512)   # smaller than the batch
    with pytest.raises(ValueError):
        controller_from_specs("panda_arm", tasks, 513, device=-1, leading_dimension=550)   # not a multiple of 32


class _StubLib:
    """saip_comm_* of the C-ABI replaced by recorders (there is no RCCL and no device here): what is under test is NativeGather's collective
    agree-or-fall-back logic around them -- the probe answers and slab shapes are reduced over the process group BEFORE anybody would enter
    ncclCommInitRank, and rank 0's unique id reaches every rank"""

    def __init__(self, real, rank, fail_probe_on=None):
        self._real, self._rank, self._fail = real, rank, fail_probe_on
        self.created_with = None

    def __getattr__(self, name):   # everything else (saip_batch_dof, saip_batch_ld, ...) is the real library
        return getattr(self._real, name)

    def saip_comm_probe(self, device):
        return 3 if self._rank == self._fail else 0      # SAIP_ERR_NO_DEVICE

    def saip_comm_unique_id(self, buf):
        for i in range(128):
            buf[i] = (37 * i + 11) % 256
        return 0

    def saip_comm_create(self, device, world, rank, buf, out):
        self.created_with = (device, world, rank, bytes(buf))
        return 0

    def saip_comm_world(self, comm):
        return self.created_with[1]

    def saip_comm_destroy(self, comm):
        pass


def _worker_native(rank, world, port, fail_on, ld_rank1, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import workloads as W
    from sai_primitives_amd import sharding, capi
    from sai_primitives_amd.controller import controller_from_specs
    dist = sharding.init_process_group("gloo")
    ld = ld_rank1 if rank == 1 else 64
    _, ctrl, _ = controller_from_specs("panda_arm", W.config_tasks(2), 40, device=-1, leading_dimension=ld)
    stub = _StubLib(capi.lib(), rank, fail_probe_on=fail_on)
    capi.lib = lambda: stub                               # (this process only: mp.spawn children are fresh interpreters)
    outcome = "ok"
    try:
        ng = sharding.NativeGather(ctrl, world, rank, rank, torch.device("cpu"))
        assert ng.world() == world
        assert stub.created_with[:3] == (rank, world, rank)
        assert stub.created_with[3] == bytes((37 * i + 11) % 256 for i in range(128))   # rank 0's id, on every rank
    except RuntimeError as e:
        outcome = "fallback: " + str(e)[:60]
        assert stub.created_with is None                  # nobody entered the communicator creation
        # ... and the fallback bench.py takes still works on the same process group
        slab = torch.full((7, 64), float(rank), dtype=torch.float64)
        got = sharding.gather_torques(slab, world)
        assert all(bool((got[r] == r).all()) for r in range(world))
    with open(os.path.join(out_dir, f"native_{rank}.txt"), "w") as f:
        f.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_on,ld_rank1,expect", [(None, 64, "ok"), (1, 64, "fallback"), (None, 96, "fallback")])
def test_native_gather_agrees_or_falls_back_on_all_ranks(tmp_path, fail_on, ld_rank1, expect):
    """all ranks create the native communicator, or none does: a probe that fails on ONE rank, or slabs of different shapes, make EVERY rank
    raise before saip_comm_create (which would otherwise block the healthy ranks inside ncclCommInitRank)"""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker_native, args=(world, _free_port(), fail_on, ld_rank1, str(tmp_path)), nprocs=world, join=True)
    outcomes = [open(tmp_path / f"native_{r}.txt").read() for r in range(world)]
    assert all(o.startswith(expect) for o in outcomes), outcomes

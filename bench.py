#!/usr/bin/env python3
"""Headline benchmark: control-cycles/sec for BASELINE.json configs[1]
(4096 x Panda 7-DoF, full MotionForceTask + full JointTask in its nullspace) per GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts the N ranks itself, see _spawn_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus 2 --dry-run                     (no GPU: ranks, shards and the gather shapes rehearsed over gloo)
    python bench.py --config 4 --gpus 8                    (BASELINE config 4: 65536 instances = 8192 per GPU)

One "step" = one control cycle (updateModel + updateControllerTaskModels + computeControlTorques) of every instance of
the rank's batch = one launch of the cycle kernel, inputs already resident in HBM.  Weak scaling: every rank owns
4096 instances (rank r = instances [4096 r, 4096 (r+1)) of the seeded stream); no data-path collective; ONE RCCL
all_gather of the final torque slabs at the end of the timed region (north_star: "RCCL ... only for the final torque gather").
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector = FP64 matrix peak (FP32 vector 157.3 TF / 2, MI355X_MICROARCH.md:41)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md:36 (spec; 6290 measured)


def cpu_baseline(d, seconds_multi=10.0, seconds_single=4.0, passes=1000, warm=100):
    """The CPU oracle timed on this box's host cores: the LITERAL restatement of the reference path (SVDs, duplicate Jacobians,
    the JLA model every cycle) and its LEAN variant (closed forms behind certificates, oracle.set_lean), each on all cores and on one thread;
    `passes` batch-steps after `warm` warm-up steps, MEDIAN pass time.  Bounded: every leg runs on a prefix of the workload sized so that its
    `warm + passes` steps take about 10 s (all cores) or 4 s (one thread).  This is the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    from oracle import Oracle
    cores = min(os.cpu_count() or 1, 16)
    B = d["q"].shape[0]

    def leg(lean, nthreads):
        o = Oracle(d["model"], d["tasks"])
        o.set_lean(lean)
        def slab(bs):
            return (np.ascontiguousarray(d["q"][:bs].T), np.ascontiguousarray(d["dq"][:bs].T), [np.ascontiguousarray(g[:bs].T) for g in d["goals"]])
        # all cores: the whole batch per pass (short passes do not spread over the cores: a pass of 256 instances on 8 threads measured SLOWER than
        # on one), fewer passes when 1000 of them do not fit the budget; one thread: a prefix sized to the budget
        seconds = seconds_multi if nthreads > 1 else seconds_single
        q, dq, goals = slab(B if nthreads > 1 else min(B, 256))
        o.step_soa(q, dq, goals, nthreads=nthreads)
        one = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            o.step_soa(q, dq, goals, nthreads=nthreads)
            one = min(one, max(time.perf_counter() - t0, 1e-5))
        bs, n_pass = q.shape[1], passes
        if nthreads > 1:
            n_pass = int(max(20, min(passes, seconds / one / 1.1)))
        else:
            bs = int(min(B, max(1, bs / one * seconds / (passes + warm))))
            q, dq, goals = slab(bs)
        for _ in range(max(2, n_pass // 10)):
            o.step_soa(q, dq, goals, nthreads=nthreads)
        ts = np.empty(n_pass)
        for i in range(n_pass):
            t0 = time.perf_counter()
            o.step_soa(q, dq, goals, nthreads=nthreads)
            ts[i] = time.perf_counter() - t0
        return bs / float(np.median(ts)), (bs, n_pass)

    lit, bs_lit = leg(False, cores)
    lean, bs_lean = leg(True, cores)
    lit1, bs1 = leg(False, 1)
    lean1, bs1l = leg(True, 1)
    return {"value": lit, "unit": "control-cycles/s", "cores": cores, "kind": "port", "statistic": "median pass time; passes/10 warm-up passes before each leg",
            "lean_value": lean, "single_thread_value": lit1, "lean_single_thread_value": lean1,
            "sample": f"literal CPU restatement of the reference path (oracle/saip_oracle.c, gcc -O2, pthreads over instances) on {bs_lit[0]} instances "
                      f"x {bs_lit[1]} passes; lean variant (no SVD where certified, no duplicate Jacobians, no unused JLA model) on {bs_lean[0]} x {bs_lean[1]}; "
                      f"single-thread legs on the first {bs1[0]} / {bs1l[0]} instances x {bs1[1]} passes; not the Eigen/sai-model binary (cannot be built: Eigen3, "
                      f"sai-model, RBDL absent)"}


def _spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here -- fresh child processes, one per GPU, with
    the environment torch.distributed.run would give them -- BEFORE anything in this process touches a GPU (the parent never does).
    Rank 0's stdout (the JSON line) is passed through; the exit code is the worst of the ranks'."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for line in out.decode().splitlines():   # stdout carries the JSON line only; anything else a library printed there goes to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def dry_run(args, rank, world):
    """CPU rehearsal of the N-rank run (no GPU, no kernel): every rank builds the configuration-only controller of its shard, the ranks
    rendezvous over gloo, gather slabs of the real shape [dof][ld] and check them; rank 0 prints the line with n_gpus = the number of ranks
    the process group counts.  tests/test_bench_cli.py runs it with 2 ranks."""
    import numpy as np
    import torch
    import workloads as W
    from sai_primitives_amd import sharding
    from sai_primitives_amd.controller import controller_from_specs
    if world > 1:
        sharding.init_process_group("gloo")
    import torch.distributed as dist
    B = args.batch
    model = W.load_robot(W.CONFIGS[args.config]["robot"])
    tasks = W.config_tasks(args.config)
    ld = sharding.shard_ld(B * world, world)
    robot, ctrl, _ = controller_from_specs(model.name, tasks, B, device=-1, leading_dimension=ld)   # host logic only: the engine has no CPU path
    n = robot.dof()
    assert ctrl.devicePointers()["ld"] == ld
    slab = torch.full((n, ld), float(rank + 1), dtype=torch.float64)
    got = sharding.gather_torques(slab, world, out=sharding.gather_buffer(slab, world))
    ok = all(bool((got[r] == r + 1).all()) for r in range(world))
    counted = dist.get_world_size() if world > 1 else 1
    ok = ok and counted == args.gpus
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if counted != args.gpus:
            raise SystemExit(f"bench: --gpus {args.gpus} but the process group counts {counted} ranks")
        print(json.dumps({
            "metric": f"control-cycles/sec (whole node), {n}-DoF {len(tasks)}-task hierarchy, batch={B}", "value": None, "unit": "control-cycles/s",
            "n_gpus": counted, "dry_run": True, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {W.CONFIGS[args.config]['desc']}", "batch_per_gpu": B, "global_batch": B * world,
                       "shards": [[r * B, (r + 1) * B] for r in range(world)], "gather_shape": [world, n, ld],
                       "parallelism": f"instances sharded x{world}, no data-path collective, one final torque all_gather", "gather": "gloo (dry run)"},
            "valid": ok}))
    if not ok:
        raise SystemExit("bench --dry-run: gather check failed")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default: the configuration's: 4096 for configs 2, 3, 5; 8192 = 65536 / 8 for config 4)")
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--dry-run", action="store_true", help="no GPU: rehearse ranks, shards and gather shapes over gloo")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 general workgroup-per-instance, 2 lane-per-instance")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", type=float, default=None, help="measured HBM bytes per launch (from a separate rocprofv3 --pmc pass); "
                    "default: the committed PMC summary profiles/pmc_traffic.json when it matches kernel/config/batch")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process only starts the ranks (it must not touch a GPU itself: the ranks are fresh processes)
        if not args.dry_run:
            import torch
            have = torch.cuda.device_count()   # (does not initialise the GPU)
            if have < args.gpus:
                raise SystemExit(f"bench: --gpus {args.gpus} but only {have} GPU(s) are visible; no line is printed for a run that is not the one asked for")
        sys.exit(_spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import workloads as W
    import sai_primitives_amd as sp
    from sai_primitives_amd import sharding
    from sai_primitives_amd.controller import controller_from_specs

    rank, local_rank, world = sharding.dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.batch is None:
        args.batch = 8192 if args.config == 4 else W.CONFIGS[args.config]["batch"]
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        sharding.init_process_group("nccl")
        import torch.distributed as dist

    B = args.batch
    d = W.make_inputs(args.config, B, shard=rank)
    robot, ctrl, _ = controller_from_specs(d["model"].name, d["tasks"], B, device=local_rank)
    ctrl.setKernel(args.kernel)
    n = robot.dof()
    ld = ctrl.devicePointers()["ld"]
    tau_t = torch.zeros((n, ld), dtype=torch.float64, device=dev)   # engine writes torques straight into the gather buffer
    gather_out = sharding.gather_buffer(tau_t, world)               # allocated here: no allocator call inside the timed region
    torch.cuda.synchronize()                                        # the zero fill runs on torch's stream, the engine on its own
    ctrl.bindTauDevice(tau_t.data_ptr())
    robot.setQ(d["q"])
    robot.setDq(d["dq"])
    robot.updateModel()
    ctrl.setGoals(d["goals"])

    # the final gather is the engine's own (C++ on RCCL, csrc/saip_comm.cpp); torch.distributed is the rendezvous (unique id, barrier,
    # max over ranks) and the fallback if the native communicator cannot be set up
    native, gather_kind = None, "none (one rank)"
    if world > 1:
        ok_native = 1.0
        try:
            native = sharding.NativeGather(ctrl, world, rank, local_rank, dev)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: native RCCL gather unavailable ({e}); falling back to torch.distributed", file=sys.stderr)
            ok_native = 0.0
        if sharding.max_over_ranks(1.0 - ok_native, dev) != 0.0:   # all ranks or none
            if native is not None:
                native.close()
            native = None
        gather_kind = "engine: ncclAllGather (RCCL) on the engine stream" if native is not None else "torch.distributed all_gather_into_tensor (RCCL)"
    # n_gpus is what the communicator itself counts (ncclCommCount of the engine's communicator, else the process group's size), never --gpus
    counted = native.world() if native is not None else (dist.get_world_size() if world > 1 else 1)
    if counted != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but the communicator counts {counted} ranks: no line is printed")

    def final_gather():
        if world == 1:                    # one rank: there is nothing to gather (config.gather says so); no copy is issued
            return tau_t.unsqueeze(0)
        if native is not None:
            native.gather(gather_out)
            ctrl.synchronize()
            return gather_out
        return sharding.gather_torques(tau_t, world, out=gather_out)

    # untimed warmup: W cycles, and one gather so that the collective's first-use costs (communicator setup, lazy kernel loading:
    # 0.2 ms and more) do not land in the timed region
    ctrl.timeSteps(max(args.warmup, 1), 0)
    final_gather()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_ms = ctrl.timeSteps(args.steps, 0)              # K launches on the engine stream, HIP events around them
    gathered = final_gather()                          # the final torque gather (one RCCL all-gather for world > 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dev)
    ev_ms = sharding.max_over_ranks(ev_ms, dev)

    # correctness of what was timed: finite torques, no flagged instance, gather holds this rank's slab
    tau = ctrl.getTorques()
    ok = bool(np.isfinite(tau).all() and (ctrl.status & 1).sum() == 0)  # bit 0 = flagged (torques held or NaN by policy); bits 1..3 are informational (truncated / POPC window / blended)
    ok = ok and bool(torch.equal(gathered[rank], tau_t))
    # cross-check of the timed kernel against the engine's other, independently written kernel (general workgroup-per-instance
    # path) on the same resident inputs: catches a wrong-but-finite result without touching the oracle
    xerr = None
    stateful = any(np.any(np.asarray(t.get(k, 0.0)) != 0) for t in d["tasks"] for k in ("ki", "ki_pos", "ki_ori"))  # integrators advance per step
    if not stateful and args.kernel != 1 and ctrl.kernelName() != "" and not ctrl.kernelName().startswith("saip_cycle_wg"):
        timed_kernel = ctrl.kernelName()
        ctrl.setKernel(1)
        ctrl.stepAsync()
        ctrl.synchronize()
        tau_wg = ctrl.getTorques()
        ctrl.setKernel(args.kernel)
        ctrl.stepAsync()
        ctrl.synchronize()
        xerr = W.torque_error(tau, tau_wg)
        ok = ok and xerr < 1e-7 and ctrl.kernelName() == timed_kernel

    ok = sharding.max_over_ranks(0.0 if ok else 1.0, dev) == 0.0   # valid only if every rank's slab is
    if rank == 0:
        traffic = args.traffic
        traffic_src = "--traffic" if traffic is not None else None
        if traffic is None:
            # FETCH_SIZE / WRITE_SIZE come from separate `rocprofv3 --pmc` passes of this same command (gpurun refuses PMC + tracing
            # in one run); their per-launch means are committed under profiles/ and keyed by config/batch/kernel variant.
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pt = json.load(f)
                key = f"cfg{args.config}_B{B}"
                if key in pt and pt[key]["kernel"] == ctrl.kernelName():
                    traffic = pt[key]["hbm_bytes_per_launch"]
                    traffic_src = pt[key]["source"]
            except (OSError, KeyError, ValueError):
                pass
        total_cycles = B * world * args.steps
        value = total_cycles / elapsed
        flops = W.algorithmic_flops(d["model"], d["tasks"])
        byts = W.algorithmic_bytes(d["model"], d["tasks"])
        launch_s = ev_ms * 1e-3 / args.steps
        ach_tf = flops * B / launch_s / 1e12
        ach_gbs = byts * B / launch_s / 1e9
        out = {
            "metric": f"control-cycles/sec (whole node), {n}-DoF {len(d['tasks'])}-task hierarchy, batch={B}",
            "value": value, "value_event_timed": B * world * args.steps / (ev_ms * 1e-3), "unit": "control-cycles/s", "n_gpus": counted, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {W.CONFIGS[args.config]['desc']}", "batch_per_gpu": B,
                       "global_batch": B * world, "robot": d["model"].name, "dof": n,
                       "tasks": [t["name"] for t in d["tasks"]], "parallelism": f"instances sharded x{world}, no data-path collective, "
                       "one final torque all_gather", "gather": gather_kind, "kernel": ctrl.kernelName()},
            "roofline": {"bound": "mfma", "note": "FP64 compute bound (FP64 vector pipe; its peak equals the FP64 MFMA peak on MI355X); HBM never binds",
                         "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP64_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_flops_per_cycle": flops, "units_per_launch": B,
                         "avg_launch_us": launch_s * 1e6, "kernel": ctrl.kernelName()},
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_cycle": byts, "traffic": traffic},
            "max_abs_tau": float(np.abs(tau).max()), "valid": ok, "cross_kernel_max_rel_err": xerr,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(d)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench: invalid torques")


if __name__ == "__main__":
    main()

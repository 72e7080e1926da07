#!/usr/bin/env python3
"""Headline benchmark: control-cycles/sec for BASELINE.json configs[1]
(4096 x Panda 7-DoF, full MotionForceTask + full JointTask in its nullspace) per GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts the N ranks itself, see _spawn_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N --launcher multi              (ONE torch-free process drives the N GPUs through the engine's saip_multi_* layer)
    python bench.py --gpus 2 --dry-run                     (no GPU: ranks, shards and the gather shapes rehearsed over gloo)
    python bench.py --config 4 --gpus 8                    (BASELINE config 4: 65536 instances = 8192 per GPU)
    python bench.py --gpus N --gather every-step           (an all-gather behind EVERY cycle instead of one behind the last)

One "step" = one control cycle (updateModel + updateControllerTaskModels + computeControlTorques) of every instance of
the rank's batch = one launch of the cycle kernel, inputs already resident in HBM.  Weak scaling: every rank owns
4096 instances (rank r = instances [4096 r, 4096 (r+1)) of the seeded stream); no data-path collective; ONE RCCL
all_gather of the final torque slabs at the end of the timed region (north_star: "RCCL ... only for the final torque gather")
-- `--gather every-step` puts one behind every cycle instead (what a consumer that takes the torques each control period
sees); either way the gathers are event-timed on their own as well (`config.gather_us` = one gather).
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


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _host_threads():
    """the hardware threads this process may run on (the affinity mask: a container's CPU set), = hardware_concurrency() on an unrestricted box"""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return max(1, os.cpu_count() or 1)


def _cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max, v1 cfs quota), None = unlimited / unknown"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, p = float(f.read()), float(g.read())
            return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(d, tau_gpu=None, seconds_multi=10.0, seconds_single=4.0, passes=1000, warm=100):
    """The CPU oracle timed on this box's host cores: the LITERAL restatement of the reference path (SVDs, duplicate Jacobians,
    the JLA model every cycle) and its LEAN variant (closed forms behind certificates, oracle.set_lean), each on the thread count that is
    fastest on this box (scanned up to the hardware threads the process may use -- BASELINE.md 4: hardware_concurrency(); count, CPU model, the
    container's CPU quota and the scan itself are in the line) and on one thread; `passes` batch-steps after `warm`
    warm-up steps, MEDIAN pass time.  Bounded: every leg runs on a prefix of the workload sized so that its `warm + passes` steps take about
    10 s (all threads) or 4 s (one thread).  With `tau_gpu` (the torques the timed kernel left, [B][dof]) it also reports the second half of
    BASELINE.json's metric: max |tau - tau_ref| relative, per joint torque, against the oracle on the same inputs (outside any timed
    region).  This is the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import workloads as W
    from oracle import Oracle
    B = d["q"].shape[0]
    # How many threads: hardware_concurrency() is what BASELINE.md 4 asks for, but a container's CPU share can be far below the hardware threads it
    # sees (the 1-GPU box: 256 visible, a quota of 16 -- 256 threads measured 4x SLOWER than 16 there).  So the count is MEASURED: a short scan of the
    # literal variant over thread counts up to the affinity mask, the fastest one is used for the timed legs and the whole scan is in the line.
    hw = min(_host_threads(), 256)
    quota = _cgroup_cpu_quota()
    scan = {}
    o = Oracle(d["model"], d["tasks"])
    qs, dqs, gs = np.ascontiguousarray(d["q"].T), np.ascontiguousarray(d["dq"].T), [np.ascontiguousarray(g.T) for g in d["goals"]]
    for t in sorted({t for t in (4, 8, 16, 32, 64, 128, hw) if t <= hw}):
        o.step_soa(qs, dqs, gs, nthreads=t)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            o.step_soa(qs, dqs, gs, nthreads=t)
            best = min(best, time.perf_counter() - t0)
        scan[t] = B / best
    cores = max(scan, key=scan.get)

    def leg(lean, nthreads):
        o = Oracle(d["model"], d["tasks"])
        o.set_lean(lean)
        def slab(bs):
            return (np.ascontiguousarray(d["q"][:bs].T), np.ascontiguousarray(d["dq"][:bs].T), [np.ascontiguousarray(g[:bs].T) for g in d["goals"]])
        # all threads: the whole batch per pass (short passes do not spread over the cores: a pass of 256 instances on 8 threads measured SLOWER than
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

    err = None
    if tau_gpu is not None:
        ref, st = Oracle(d["model"], d["tasks"]).step(d["q"], d["dq"], d["goals"], nthreads=cores)
        ok = (st & 1) == 0
        err = float(W.torque_error(tau_gpu[ok], ref[ok])) if ok.any() else None
    lit, bs_lit = leg(False, cores)
    lean, bs_lean = leg(True, cores)
    lit1, bs1 = leg(False, 1)
    lean1, bs1l = leg(True, 1)
    return {"value": lit, "unit": "control-cycles/s", "cores": cores, "cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "affinity_threads": hw,
            "cgroup_cpu_quota": quota, "thread_scan": {str(k): v for k, v in scan.items()}, "kind": "port",
            "statistic": "median pass time; passes/10 warm-up passes before each leg",
            "lean_value": lean, "single_thread_value": lit1, "lean_single_thread_value": lean1, "max_rel_err_vs_oracle": err,
            "sample": f"literal CPU restatement of the reference path (oracle/saip_oracle.c, gcc -O2, pthreads over instances, {cores} threads) on {bs_lit[0]} instances "
                      f"x {bs_lit[1]} passes; lean variant (no SVD where certified, no duplicate Jacobians, no unused JLA model) on {bs_lean[0]} x {bs_lean[1]}; "
                      f"single-thread legs on the first {bs1[0]} / {bs1l[0]} instances x {bs1[1]} passes; not the Eigen/sai-model binary (cannot be built: Eigen3, "
                      f"sai-model, RBDL absent)"}


def _visible_gpus():
    """GPUs this process could open, counted WITHOUT the HIP / HSA runtime (the launcher parent must not initialise a GPU before it starts the
    ranks): the KFD topology's nodes with SIMDs, cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when it cannot be told."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def _spawn_ranks(n: int, timeout_s: float) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here -- fresh child processes, one per GPU, with
    the environment torch.distributed.run would give them -- BEFORE anything in this process touches a GPU (the parent never does).
    Rank 0's stdout (the JSON line) is passed through, the other ranks' stdout goes to this process's stderr (so does everybody's stderr).
    All children are polled: the first one that exits non-zero -- or the overall timeout -- ends the others (a rank that left early would
    otherwise keep the rest inside init_process_group / ncclCommInitRank for ever); the exit code is the worst of the ranks'."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)   # (drain the pipe: a full pipe would block rank 0)
    reader.start()
    deadline = time.monotonic() + timeout_s
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f"rank {bad[0]} exited with code {codes[bad[0]]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = f"no result within {timeout_s:.0f} s"
            break
        time.sleep(0.1)
    if failed is not None:
        print(f"[bench] {failed}: ending the other ranks; no line is printed", file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5.0)
    codes = [p.returncode if p.returncode is not None else 1 for p in procs]
    if failed is None:
        for line in b"".join(out_chunks).decode().splitlines():   # stdout carries the JSON line only; anything else a library printed there goes to stderr
            print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
        sys.stdout.flush()
        return max(abs(c) for c in codes)
    return max([abs(c) for c in codes if c != 0] + [1])


def _metric(n, ntasks, B):
    return f"control-cycles/sec (whole node), {n}-DoF {ntasks}-task hierarchy, batch={B}"


def _parallelism(world, gather_mode):
    what = "one final torque all_gather" if gather_mode == "final" else "a torque all_gather behind every cycle"
    return f"instances sharded x{world}, no data-path collective, {what}"


def dry_run(args, rank, world):
    """CPU rehearsal of the N-rank run (no GPU, no kernel): every rank builds the configuration-only controller of its shard, the ranks
    rendezvous over gloo, gather slabs of the real shape [dof][ld] -- once, or once per step with --gather every-step -- and check them; rank 0
    prints the line with n_gpus = the number of ranks the process group counts.  tests/test_bench_cli.py runs it with 2 ranks."""
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
    ngather = min(args.steps, 8) if args.gather == "every-step" else 1
    ok, t_g = True, 0.0
    out = None
    for it in range(ngather):
        slab = torch.full((n, ld), float(rank + 1 + 10 * it), dtype=torch.float64)
        out = sharding.gather_buffer(slab, world) if out is None else out
        t0 = time.perf_counter()
        got = sharding.gather_torques(slab, world, out=out)
        t_g += time.perf_counter() - t0
        ok = ok and all(bool((got[r] == r + 1 + 10 * it).all()) for r in range(world))
    counted = dist.get_world_size() if world > 1 else 1
    ok = ok and counted == args.gpus
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if counted != args.gpus:
            raise SystemExit(f"bench: --gpus {args.gpus} but the process group counts {counted} ranks")
        print(json.dumps({
            "metric": _metric(n, len(tasks), B), "value": None, "unit": "control-cycles/s",
            "n_gpus": counted, "dry_run": True, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {W.CONFIGS[args.config]['desc']}", "batch_per_gpu": B, "global_batch": B * world,
                       "shards": [[r * B, (r + 1) * B] for r in range(world)], "gather_shape": [world, n, ld],
                       "parallelism": _parallelism(world, args.gather), "gather": "gloo (dry run)", "gather_mode": args.gather,
                       "gathers": ngather, "gather_us": t_g / ngather * 1e6, "launcher": "ranks"},
            "valid": ok}))
    if not ok:
        raise SystemExit("bench --dry-run: gather check failed")


def run_multi(args):
    """--launcher multi: ONE process, no torch, no torch.distributed: the engine's C++ multi-device layer (saip_multi_*: a batch, a stream and an
    RCCL communicator per device, grouped ncclAllGather) driven through ctypes.  With --dry-run (no GPU) the shards' configuration-only
    controllers are built in this process and the line reports the shapes a real run would gather."""
    import numpy as np
    import workloads as W
    import sai_primitives_amd as sp
    from sai_primitives_amd import sharding
    from sai_primitives_amd.controller import controller_from_specs
    world = args.gpus
    B = args.batch
    tasks = W.config_tasks(args.config)
    model = W.load_robot(W.CONFIGS[args.config]["robot"])
    ld = sharding.shard_ld(B * world, world)
    every = args.gather == "every-step"
    base = {"unit": "control-cycles/s", "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic"}
    cfg = {"workload": f"config {args.config}: {W.CONFIGS[args.config]['desc']}", "batch_per_gpu": B, "global_batch": B * world,
           "parallelism": _parallelism(world, args.gather), "gather_mode": args.gather, "launcher": "multi (one process, saip_multi_*, no torch)"}
    if args.dry_run:
        ctrls = [controller_from_specs(model.name, tasks, B, device=-1, leading_dimension=ld) for _ in range(world)]
        n = ctrls[0][0].dof()
        ok = all(c[1].devicePointers()["ld"] == ld for c in ctrls)
        cfg.update({"shards": [[r * B, (r + 1) * B] for r in range(world)], "gather_shape": [world, n, ld], "gather": "none (dry run)"})
        print(json.dumps(dict(base, metric=_metric(n, len(tasks), B), value=None, n_gpus=len(ctrls), dry_run=True, config=cfg, valid=ok)))
        if not ok:
            raise SystemExit("bench --dry-run: shard check failed")
        return
    have = sp.device_count()
    if have < world:
        raise SystemExit(f"bench: --gpus {world} but only {have} GPU(s) are visible; no line is printed for a run that is not the one asked for")
    mc = sharding.MultiController(model.name, tasks, B, list(range(world)), leading_dimension=ld)
    if mc.world() != world:
        raise SystemExit(f"bench: --gpus {world} but the engine drives {mc.world()} devices: no line is printed")
    shards = [W.make_inputs(args.config, B, shard=r) for r in range(world)]
    for r, (robot, ctrl) in enumerate(zip(mc.robots, mc.controllers)):
        ctrl.setKernel(args.kernel)
        robot.setQ(shards[r]["q"])
        robot.setDq(shards[r]["dq"])
        robot.updateModel()
        ctrl.setGoals(shards[r]["goals"])
    n = mc.robots[0].dof()
    for _ in range(max(args.warmup, 1)):
        mc.step_async()
    if world > 1:
        mc.all_gather()
    mc.synchronize()
    t0 = time.perf_counter()
    ev_ms, g_ms = mc.time_steps(args.steps, every)     # K cycles on every device (+ the grouped all-gather), HIP events on every stream
    elapsed = time.perf_counter() - t0
    ngather = (args.steps if every else 1) if world > 1 else 0
    taus = [c.getTorques() for c in mc.controllers]
    ok = all(bool(np.isfinite(t).all()) for t in taus) and all(int((c.status & 1).sum()) == 0 for c in mc.controllers)
    if world > 1:
        for i in range(world):
            got = mc.gathered(i)
            ok = ok and all(np.array_equal(got[r], taus[r]) for r in range(world))
    d = shards[0]
    flops = W.algorithmic_flops(d["model"], d["tasks"])
    byts = W.algorithmic_bytes(d["model"], d["tasks"])
    launch_s = (ev_ms - g_ms) * 1e-3 / args.steps
    kname = mc.controllers[0].kernelName()
    mfma = kname.startswith("saip_cycle_wave")
    cfg.update({"robot": d["model"].name, "dof": n, "tasks": [t["name"] for t in d["tasks"]], "kernel": kname,
                "gather": "engine: grouped ncclAllGather (RCCL, ncclCommInitAll)" if world > 1 else "none (one device)",
                "gathers": ngather, "gather_us": (g_ms / ngather * 1e3) if ngather else 0.0})
    out = dict(base, metric=_metric(n, len(d["tasks"]), B), value=B * world * args.steps / elapsed,
               value_event_timed=B * world * args.steps / (ev_ms * 1e-3), n_gpus=mc.world(), ms_per_step=elapsed * 1e3 / args.steps, config=cfg,
               roofline=_roofline(flops, B, launch_s, kname, mfma, None, None), roofline_hbm=_roofline_hbm(byts, B, launch_s, None),
               max_abs_tau=float(max(np.abs(t).max() for t in taus)), valid=ok, cross_kernel_max_rel_err=None)
    mc.close()
    if not args.no_cpu_baseline:
        stateful = any(np.any(np.asarray(t.get(k, 0.0)) != 0) for t in d["tasks"] for k in ("ki", "ki_pos", "ki_ori"))
        cb = cpu_baseline(d, tau_gpu=None if stateful else taus[0])
        out["cpu_baseline"] = cb
        out["max_rel_err_vs_oracle"] = cb["max_rel_err_vs_oracle"]
        if cb["max_rel_err_vs_oracle"] is not None:
            out["valid"] = ok = ok and cb["max_rel_err_vs_oracle"] < 1e-5
    print(json.dumps(out))
    if not ok:
        raise SystemExit("bench: invalid torques")


def _roofline(flops, B, launch_s, kname, mfma, traffic, traffic_src):
    ach_tf = flops * B / launch_s / 1e12
    return {"bound": "mfma" if mfma else "fp64-valu",
            "note": ("FP64 matrix cores (v_mfma_f64_16x16x4_f64) carry the n x n products" if mfma else
                     "FP64 vector pipe: this kernel issues no MFMA (n <= 8 fills no tile)") + "; the FP64 vector and FP64 MFMA peaks are equal on MI355X; HBM never binds",
            "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP64_PEAK_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_src, "algorithmic_flops_per_cycle": flops, "units_per_launch": B,
            "avg_launch_us": launch_s * 1e6, "kernel": kname}


def _roofline_hbm(byts, B, launch_s, traffic):
    ach_gbs = byts * B / launch_s / 1e9
    return {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_cycle": byts, "traffic": traffic}


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
    ap.add_argument("--gather", choices=("final", "every-step"), default="final",
                    help="final (default; north_star's wording): ONE torque all-gather behind the last timed cycle; every-step: one behind every cycle")
    ap.add_argument("--launcher", choices=("ranks", "multi"), default="ranks",
                    help="ranks (default): one process per GPU (torch.distributed rendezvous, the engine's RCCL gather); multi: one torch-free process "
                         "drives all GPUs through saip_multi_*")
    ap.add_argument("--spawn-timeout", type=float, default=1800.0, help="seconds the self-started ranks get before they are ended")
    ap.add_argument("--traffic", type=float, default=None, help="measured HBM bytes per launch (from a separate rocprofv3 --pmc pass); "
                    "default: the committed PMC summary profiles/pmc_traffic.json when it matches kernel/config/batch")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.launcher == "multi":
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != 1:
            raise SystemExit("--launcher multi is ONE process for all GPUs: do not start it under a multi-rank launcher")
        import workloads as W
        if args.batch is None:
            args.batch = 8192 if args.config == 4 else W.CONFIGS[args.config]["batch"]
        return run_multi(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process only starts the ranks (it must not touch a GPU itself: the ranks are fresh processes)
        if not args.dry_run:
            have = _visible_gpus()   # (from the KFD topology in /sys: no HIP / HSA call in this process)
            if have is not None and have < args.gpus:
                raise SystemExit(f"bench: --gpus {args.gpus} but only {have} GPU(s) are visible; no line is printed for a run that is not the one asked for")
        sys.exit(_spawn_ranks(args.gpus, args.spawn_timeout))

    import numpy as np
    import torch
    import workloads as W
    import sai_primitives_amd as sp  # noqa: F401
    from sai_primitives_amd import sharding
    from sai_primitives_amd.controller import controller_from_specs

    rank, local_rank, world = sharding.dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.batch is None:
        args.batch = 8192 if args.config == 4 else W.CONFIGS[args.config]["batch"]
    if args.dry_run:
        # (rehearsal hooks for tests/test_bench_cli.py: a rank that leaves, or hangs, before the rendezvous)
        if os.environ.get("SAIP_BENCH_FAIL_RANK") == str(rank):
            raise SystemExit(f"bench: rank {rank} told to fail (SAIP_BENCH_FAIL_RANK)")
        if os.environ.get("SAIP_BENCH_HANG_RANK") == str(rank):
            time.sleep(3600)
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench: rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible; no line is printed")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        sharding.init_process_group("nccl")
        import torch.distributed as dist

    B = args.batch
    every = args.gather == "every-step"
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

    # the gather is the engine's own (C++ on RCCL, csrc/saip_comm.cpp); torch.distributed is the rendezvous (unique id, barrier,
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

    def one_gather():
        if world == 1:                    # one rank: there is nothing to gather (config.gather says so); no copy is issued
            return tau_t.unsqueeze(0)
        if native is not None:
            native.gather(gather_out)
            ctrl.synchronize()
            return gather_out
        return sharding.gather_torques(tau_t, world, out=gather_out)

    def timed_region():
        """K cycles + the gather(s): (event ms of the whole region on the engine stream, ms of the gathers alone, the gathered slabs)"""
        if world == 1:
            # K launches between two HIP events, nothing to gather, and NO wait in here: the torch.cuda.synchronize() that closes the timed region is
            # the one wait (each runtime wait is its own marker round trip; a 20-step region notices a second one); the event time is read behind it
            if os.environ.get("SAIP_BENCH_TWO_WAITS"):   # A/B knob: the engine's own wait, then torch's (how it was up to round 3)
                return ctrl.timeSteps(args.steps, 0), 0.0, tau_t.unsqueeze(0)
            ctrl.timeStepsBegin(args.steps)
            return None, 0.0, tau_t.unsqueeze(0)
        if native is not None:
            e_ms, g_ms = native.time_steps(args.steps, gather_out, every)        # launches and all-gather(s) on the engine stream, events around each
            return e_ms, g_ms, gather_out
        # fallback (torch.distributed's collective lives on torch's stream): the gathers are timed by the host clock around their own waits
        if not every:
            e_ms = ctrl.timeSteps(args.steps, 0)
            tg = time.perf_counter()
            got = one_gather()
            torch.cuda.synchronize()
            g_ms = (time.perf_counter() - tg) * 1e3
            return e_ms + g_ms, g_ms, got
        e_ms = g_ms = 0.0
        got = None
        for _ in range(args.steps):
            e_ms += ctrl.timeSteps(1, 0)
            tg = time.perf_counter()
            got = one_gather()
            torch.cuda.synchronize()
            g_ms += (time.perf_counter() - tg) * 1e3
        return e_ms + g_ms, g_ms, got

    # untimed warmup: W cycles, and one gather so that the collective's first-use costs (communicator setup, lazy kernel loading:
    # 0.2 ms and more) do not land in the timed region
    ctrl.timeSteps(max(args.warmup, 1), 0)
    one_gather()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_ms, g_ms, gathered = timed_region()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if ev_ms is None:
        ev_ms = ctrl.timeStepsEnd()
    elapsed = sharding.max_over_ranks(elapsed, dev)
    ev_ms = sharding.max_over_ranks(ev_ms, dev)
    g_ms = sharding.max_over_ranks(g_ms, dev)
    ngather = (args.steps if every else 1) if world > 1 else 0

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
    out = None
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
        launch_s = (ev_ms - g_ms) * 1e-3 / args.steps   # the kernel alone: the gathers' own event time is not the kernel's
        kname = ctrl.kernelName()
        mfma = kname.startswith("saip_cycle_wave")      # the only cycle kernel that issues v_mfma_f64 (chains of 9..32 dof); SQ_INSTS_MFMA = 0 for the others
        out = {
            "metric": _metric(n, len(d["tasks"]), B),
            "value": value, "value_event_timed": B * world * args.steps / (ev_ms * 1e-3), "unit": "control-cycles/s", "n_gpus": counted, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {W.CONFIGS[args.config]['desc']}", "batch_per_gpu": B,
                       "global_batch": B * world, "robot": d["model"].name, "dof": n,
                       "tasks": [t["name"] for t in d["tasks"]], "parallelism": _parallelism(world, args.gather),
                       "gather": gather_kind, "gather_mode": args.gather, "gathers": ngather,
                       "gather_us": (g_ms / ngather * 1e3) if ngather else 0.0, "launcher": "ranks", "kernel": kname},
            "roofline": _roofline(flops, B, launch_s, kname, mfma, traffic, traffic_src),
            "roofline_hbm": _roofline_hbm(byts, B, launch_s, traffic),
            "max_abs_tau": float(np.abs(tau).max()), "valid": ok, "cross_kernel_max_rel_err": xerr,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if native is not None:
        native.close()
    if rank == 0:
        # the CPU path beside every point (north_star): on rank 0, AFTER the process group is gone -- the other ranks have left, so the host
        # cores are the baseline's alone (ranks spinning in a barrier would take them)
        if not args.no_cpu_baseline:
            cb = cpu_baseline(d, tau_gpu=None if stateful else tau)
            out["cpu_baseline"] = cb
            out["max_rel_err_vs_oracle"] = cb["max_rel_err_vs_oracle"]   # the second half of BASELINE.json's metric (tolerance 1e-5)
            if cb["max_rel_err_vs_oracle"] is not None:
                out["valid"] = ok = ok and cb["max_rel_err_vs_oracle"] < 1e-5
        print(json.dumps(out))
    if not ok:
        raise SystemExit("bench: invalid torques")


if __name__ == "__main__":
    main()

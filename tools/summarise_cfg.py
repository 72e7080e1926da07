#!/usr/bin/env python3
"""gpurun_out/prof_cfg<C>_<tag>/ (tools/profile_cfg.sh) -> profiles/<RND>_kernel_stats_<kernel>_cfg<C>_<tag>.csv (rocprofv3's own --stats table),
<RND>_kernel_durations_<kernel>_cfg<C>_<tag>.json (per kernel, from the raw dispatch trace of the same run: launches, MEAN and MEDIAN duration,
5th / 95th percentile, min / max / standard deviation -- one outlier dispatch moves the mean of a 7 us kernel by several per cent, the median
says what the kernel takes), <RND>_pmc_<kernel>_cfg<C>_<tag>.json, <RND>_bench_<kernel>_cfg<C>_<tag>.json, and the entry of
profiles/pmc_traffic.json (the per-launch HBM traffic bench.py reports) when FETCH_SIZE and WRITE_SIZE were collected.
   python tools/summarise_cfg.py <config> <tag>            (RND = $SAIP_ROUND, default r04)"""
import csv, glob, json, os, re, shutil, statistics, sys

RND = os.environ.get("SAIP_ROUND", "r04")

cfg, tag = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_cfg{cfg}_{tag}"
bench = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])
KERNEL = bench["config"]["kernel"]
SHORT = re.sub(r"[^a-z0-9]+", "", KERNEL.replace("saip_cycle_", ""))
B = bench["config"]["batch_per_gpu"]
stats = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], f"profiles/{RND}_kernel_stats_{SHORT}_cfg{cfg}_{tag}.csv")
# mean AND median per kernel from the raw dispatch trace of the --stats run
durs = {}
for f in glob.glob(f"{src}/stats/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        durs.setdefault(row["Kernel_Name"], []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
dsum = {}
for k, v in durs.items():
    v.sort()
    pct = lambda q: v[min(len(v) - 1, int(q * len(v)))]
    dsum[k] = dict(launches=len(v), mean_us=sum(v) / len(v), median_us=statistics.median(v), p05_us=pct(0.05), p95_us=pct(0.95), min_us=v[0], max_us=v[-1],
                   stdev_us=statistics.pstdev(v) if len(v) > 1 else 0.0)
if dsum:
    json.dump(dict(command=f"rocprofv3 --kernel-trace --stats -- python3 bench.py --config {cfg} --steps {bench['steps']} (tools/profile_cfg.sh)",
                   bench_avg_launch_us=bench["roofline"]["avg_launch_us"], kernels=dsum),
              open(f"profiles/{RND}_kernel_durations_{SHORT}_cfg{cfg}_{tag}.json", "w"), indent=1)
pmc = {}
base = KERNEL.split("<")[0]
for d in sorted(glob.glob(f"{src}/pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        vals = {}
        for row in csv.DictReader(open(f)):
            if base in row["Kernel_Name"] and "list" not in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in vals.items():
            pmc[k] = dict(mean_per_launch=sum(v) / len(v), launches=len(v))
if pmc:
    json.dump(pmc, open(f"profiles/{RND}_pmc_{SHORT}_cfg{cfg}_{tag}.json", "w"), indent=1)
shutil.copy(f"{src}/bench.json", f"profiles/{RND}_bench_{SHORT}_cfg{cfg}_{tag}.json")
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fk, wk = pmc["FETCH_SIZE"]["mean_per_launch"], pmc["WRITE_SIZE"]["mean_per_launch"]
    path = "profiles/pmc_traffic.json"
    try:
        pt = json.load(open(path))
    except (OSError, ValueError):
        pt = {}
    pt[f"cfg{cfg}_B{B}"] = {"kernel": KERNEL, "hbm_bytes_per_launch": (2 * fk + wk) * 1024, "fetch_size_kib": fk, "write_size_kib": wk,
                            "source": f"profiles/{RND}_pmc_{SHORT}_cfg{cfg}_{tag}.json: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, "
                                      f"tools/profile_cfg.sh) of `python3 bench.py --config {cfg} --steps 20 --warmup 5 --no-cpu-baseline`; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, "
                                      "FETCH_SIZE doubled per MI355X_MICROARCH.md:298"}
    json.dump(pt, open(path, "w"), indent=1)
print(KERNEL, json.dumps({k: round(v["mean_per_launch"], 1) for k, v in pmc.items()}, indent=1))
for k, v in sorted(dsum.items(), key=lambda kv: -kv[1]["mean_us"] * kv[1]["launches"])[:4]:
    print(f"{k[:70]:70s} n={v['launches']:5d} mean {v['mean_us']:8.3f} us  median {v['median_us']:8.3f} us  p95 {v['p95_us']:8.3f}  max {v['max_us']:8.3f}")
print("bench.py's own HIP-event figure:", round(bench["roofline"]["avg_launch_us"], 3), "us per launch; frac", round(bench["roofline"]["frac"], 4))

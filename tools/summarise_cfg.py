#!/usr/bin/env python3
"""gpurun_out/prof_cfg<C>_<tag>/ (tools/profile_cfg.sh) -> profiles/rNN_kernel_stats_<kernel>_cfg<C>_<tag>.csv, r02_pmc_<kernel>_cfg<C>_<tag>.json,
r02_bench_<kernel>_cfg<C>_<tag>.json; for config 2 also profiles/pmc_traffic.json (the per-launch HBM traffic bench.py reports).
   python tools/summarise_cfg.py <config> <tag>"""
import csv, glob, json, os, re, shutil, sys

RND = os.environ.get("SAIP_ROUND", "r03")

cfg, tag = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_cfg{cfg}_{tag}"
bench = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])
KERNEL = bench["config"]["kernel"]
SHORT = re.sub(r"[^a-z0-9]+", "", KERNEL.replace("saip_cycle_", ""))
B = bench["config"]["batch_per_gpu"]
stats = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], f"profiles/{RND}_kernel_stats_{SHORT}_cfg{cfg}_{tag}.csv")
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
for f in stats[:1]:
    print(open(f).read()[:900])

#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/profile_round.sh) -> profiles/r01_kernel_stats_lane_cfg2_<tag>.csv, r01_pmc_lane_cfg2_<tag>.json,
r01_bench_lane_cfg2_<tag>.json and profiles/pmc_traffic.json (the per-launch HBM traffic bench.py reports)."""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "v5"
src = f"gpurun_out/prof_{tag}"
KERNEL = "saip_cycle_lane"
if os.path.exists(f"{src}/bench.json"):  # the kernel the bench line was measured on (saip_cycle_oct since v8)
    try:
        KERNEL = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])["config"]["kernel"]
    except (ValueError, KeyError, IndexError):
        pass
SHORT = KERNEL.replace("saip_cycle_", "")

stats = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], f"profiles/r01_kernel_stats_{SHORT}_cfg2_{tag}.csv")
pmc = {}
for d in sorted(glob.glob(f"{src}/pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        vals = {}
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in vals.items():
            pmc[k] = dict(mean_per_launch=sum(v) / len(v), launches=len(v))
json.dump(pmc, open(f"profiles/r01_pmc_{SHORT}_cfg2_{tag}.json", "w"), indent=1)
if os.path.exists(f"{src}/bench.json"):
    shutil.copy(f"{src}/bench.json", f"profiles/r01_bench_{SHORT}_cfg2_{tag}.json")
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fk, wk = pmc["FETCH_SIZE"]["mean_per_launch"], pmc["WRITE_SIZE"]["mean_per_launch"]
    json.dump({"cfg2_B4096": {"kernel": KERNEL, "hbm_bytes_per_launch": (2 * fk + wk) * 1024, "fetch_size_kib": fk, "write_size_kib": wk,
                              "source": f"profiles/r01_pmc_{SHORT}_cfg2_{tag}.json: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, "
                                        "tools/profile_round.sh) of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline`; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, "
                                        "FETCH_SIZE doubled per MI355X_MICROARCH.md:298"}},
              open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: round(v["mean_per_launch"], 1) for k, v in pmc.items()}, indent=1))

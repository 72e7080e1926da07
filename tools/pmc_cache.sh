#!/bin/bash
# instruction / scalar cache behaviour of one configuration (separate PMC passes; run on the GPU box through gpurun):
#   tools/pmc_cache.sh <config> [batch]
CFG=${1:-5}; B=${2:-4096}
OUT=gpurun_out/pmc_cache_cfg$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_INSTS_SMEM; do
	rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o run -- python3 bench.py --config $CFG --batch $B --steps 6 --warmup 2 --no-cpu-baseline > $OUT/$C.log 2>&1 || echo "counter $C failed"
done
python3 - <<'PY'
import csv, glob, os, sys
out = sorted(glob.glob("gpurun_out/pmc_cache_cfg*"))[-1]
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        vals = {}
        for row in csv.DictReader(open(f)):
            if "saip_cycle" in row["Kernel_Name"]:
                vals.setdefault((row["Kernel_Name"][:50], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
        for k, v in vals.items():
            print(k[0], k[1], "mean per launch", sum(v) / len(v), "launches", len(v))
PY

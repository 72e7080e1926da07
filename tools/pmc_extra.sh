#!/bin/bash
# extra PMC passes (one counter per run) for the headline kernel: where does the lone wavefront wait?
set -e
OUT=gpurun_out/pmc_extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in "$@"; do
	rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$C.log 2>&1 || echo "counter $C failed"
	python3 - "$OUT/$C" "$C" <<'PY'
import csv, glob, sys
vals = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "saip_cycle_lane" in row["Kernel_Name"] and row["Counter_Name"] == sys.argv[2]:
            vals.append(float(row["Counter_Value"]))
print(sys.argv[2], sum(vals) / len(vals) if vals else None, len(vals))
PY
done

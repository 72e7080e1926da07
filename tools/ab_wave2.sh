#!/bin/bash
# GPU box: wavefront kernel after a change: its tests, config 5 timing, scratch traffic (FETCH_SIZE / WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_wave.py -x -q -m gpu > gpurun_out/ab_wave2_tests.log 2>&1 || { tail -30 gpurun_out/ab_wave2_tests.log; exit 1; }
tail -2 gpurun_out/ab_wave2_tests.log
for rep in 1 2; do python bench.py --no-cpu-baseline --config 5 --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 5', round(d['roofline']['avg_launch_us'],2), 'us')"; done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/ab_wave2_$C -o run -- python3 bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("gpurun_out/ab_wave2_$C/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "saip_cycle_wave" in r["Kernel_Name"] and r["Counter_Name"]=="$C"]
import collections
print("$C", sum(v)/max(len(v),1)*0+ (sum(v)/ (len(v)/ (len(set())+1)) if False else sum(v)/len(v)), "KiB per counter row,", len(v), "rows")
PY
done

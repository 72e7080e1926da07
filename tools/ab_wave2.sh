#!/bin/bash
# GPU box: wavefront kernel after a change: its tests, config 5 timing, scratch traffic (FETCH_SIZE / WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_wave.py -x -q -m gpu > gpurun_out/ab_wave2_tests.log 2>&1 || { tail -30 gpurun_out/ab_wave2_tests.log; exit 1; }
tail -2 gpurun_out/ab_wave2_tests.log
for rep in 1 2; do python bench.py --no-cpu-baseline --config 5 --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 5', round(d['roofline']['avg_launch_us'],2), 'us')"; done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/ab_wave2_$C -o run -- python3 bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_wave2_$C.log 2>&1 || echo "rocprofv3 --pmc $C failed: see gpurun_out/ab_wave2_$C.log"
  python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("gpurun_out/ab_wave2_$C/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "saip_cycle_wave" in r["Kernel_Name"] and r["Counter_Name"]=="$C"]
print("$C", sum(v)/len(v) if v else float("nan"), "KiB per launch,", len(v), "launches")
PY
done

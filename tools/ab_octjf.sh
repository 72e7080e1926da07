#!/bin/bash
# GPU box: tests of the joint-first eight-lane kernel (both forms), then config 6 at B = 4096 in the two- and one-wavefront form and on the lane kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_octjf.py -x -q -m gpu > gpurun_out/octjf_tests.log 2>&1 || { tail -30 gpurun_out/octjf_tests.log; exit 1; }
tail -1 gpurun_out/octjf_tests.log
SAIP_OCT_NO_DUO=1 python -m pytest tests/test_gpu_octjf.py -x -q -m gpu > gpurun_out/octjf_tests_noduo.log 2>&1 || { tail -30 gpurun_out/octjf_tests_noduo.log; exit 1; }
tail -1 gpurun_out/octjf_tests_noduo.log
for MODE in duo noduo; do
  if [ $MODE = noduo ]; then export SAIP_OCT_NO_DUO=1; else unset SAIP_OCT_NO_DUO; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/octjf_$MODE -o run -- python3 bench.py --config 6 --batch 4096 --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/octjf_$MODE.json 2> gpurun_out/octjf_$MODE.err
  head -3 gpurun_out/octjf_$MODE/run_kernel_stats.csv | cut -c1-140
done

#!/bin/bash
# GPU box: tests of the joint-first eight-lane kernel, then config 6 at B = 4096 on it and on the lane kernel (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_octjf.py -x -q -s -m gpu > gpurun_out/octjf_tests.log 2>&1 || { tail -30 gpurun_out/octjf_tests.log; exit 1; }
tail -3 gpurun_out/octjf_tests.log
for K in 0 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/octjf_k$K -o run -- python3 bench.py --config 6 --batch 4096 --kernel $K --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/octjf_k$K.json 2> gpurun_out/octjf_k$K.err
  head -4 gpurun_out/octjf_k$K/run_kernel_stats.csv | cut -c1-160
done

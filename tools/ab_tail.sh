#!/bin/bash
# GPU box: the in-kernel slow tail of the eight-lane kernels (round 4) against the list launch it replaces (SAIP_NO_SLOW_TAIL=1):
# all GPU tests in both modes, then configs 3 / 6 / 12 (and 2, 14 as controls) per launch and per step
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -x -q -m gpu > gpurun_out/ab_tail_tests.log 2>&1 || { tail -40 gpurun_out/ab_tail_tests.log; exit 1; }
tail -1 gpurun_out/ab_tail_tests.log
for rep in 1 2; do for cfg in 3 6 12 2 14; do for mode in tail list; do
  if [ $mode = list ]; then export SAIP_NO_SLOW_TAIL=1; else unset SAIP_NO_SLOW_TAIL; fi
  python bench.py --no-cpu-baseline --config $cfg --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode cfg$cfg', d['config'].get('kernel'), round(d['roofline']['avg_launch_us'],3), 'us/launch', 'step', round(d['ms_per_step']*1e3,3), 'us valid', d['valid'])"
done; done; done 2>&1 | tee gpurun_out/ab_tail.log

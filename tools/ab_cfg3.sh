#!/bin/bash
# GPU box: partial-task stacks on the eight-lane kernel after a change -- parity tests, then config 3 in both forms against the variants given
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -x -q -m gpu > gpurun_out/ab_cfg3_tests.log 2>&1 || { tail -30 gpurun_out/ab_cfg3_tests.log; exit 1; }
tail -1 gpurun_out/ab_cfg3_tests.log
for rep in 1 2; do for lib in sai-primitives_amd/libsaip.so "$@"; do for mode in duo noduo; do
  if [ $mode = noduo ]; then export SAIP_OCT_NO_DUO=1; else unset SAIP_OCT_NO_DUO; fi
  SAIP_LIB=$lib python bench.py --no-cpu-baseline --config 3 --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $mode cfg3', round(d['roofline']['avg_launch_us'],3), 'us', 'step', round(d['ms_per_step']*1e3,3), 'valid', d['valid'])"
done; done; done
unset SAIP_OCT_NO_DUO
for lib in sai-primitives_amd/libsaip.so "$@"; do
  SAIP_LIB=$lib python bench.py --no-cpu-baseline --config 2 --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg2', round(d['roofline']['avg_launch_us'],3), 'us valid', d['valid'])"
done

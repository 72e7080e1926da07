#!/bin/bash
# GPU box: eight-lane kernels after a change -- their parity tests, then the headline configuration in both forms and config 6
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -x -q -m gpu > gpurun_out/ab_head_tests.log 2>&1 || { tail -30 gpurun_out/ab_head_tests.log; exit 1; }
tail -2 gpurun_out/ab_head_tests.log
bash tools/ab_variants.sh sai-primitives_amd/libsaip.so "$@" 2>&1 | tee gpurun_out/ab_head.log
python bench.py --no-cpu-baseline --config 6 --batch 4096 --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 6', round(d['roofline']['avg_launch_us'],3), 'us', d['config']['kernel'])"
python bench.py --no-cpu-baseline --config 2 --batch 8192 --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2 at 8192', round(d['roofline']['avg_launch_us'],3), 'us', d['config']['kernel'])"

#!/bin/bash
# A/B of library builds over the secondary configurations (3, 12, 14, 15 at 4096 instances):  tools/ab_cfgs.sh lib1.so lib2.so
cd "$GRAFT_REPO_ROOT"
for CFG in 3 12 14 15; do for lib in "$@"; do
  SAIP_LIB=$lib python bench.py --no-cpu-baseline --config $CFG --batch 4096 --steps 300 --warmup 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $CFG $(basename $lib)', round(d['roofline']['avg_launch_us'],2), 'us per step valid', d['valid'])"
done; done

#!/bin/bash
# A/B: where the eight-lane kernels stop paying against the lane kernel (batch sweep past 8192): HIP-event time per launch, ms per step
# usage: ab_cliff.sh "<configs>" "<batches>"
cd "$GRAFT_REPO_ROOT"
for C in ${1:-2}; do
for B in ${2:-8192 10240 12288 16384 20480 24576 32768}; do
	for m in 8192 1000000; do
		SAIP_OCT_MAX_BATCH=$m python3 bench.py --config $C --batch $B --steps ${3:-300} --warmup ${4:-50} --no-cpu-baseline 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg $C B=$B max=$m', d['config']['kernel'], 'launch_us', round(d['roofline']['avg_launch_us'],2), 'step_us', round(d['ms_per_step']*1e3,2), 'M/s', round(d['value']/1e6,1), 'valid', d['valid'])"
	done
done
done

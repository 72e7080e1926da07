#!/bin/bash
# same-session comparison lane (kernel 2) vs oct (kernel 3) over batch sizes: tools/oct_sweep.sh "<batches>"
for B in $1; do
	for k in 2 3; do
		python bench.py --no-cpu-baseline --batch $B --kernel $k --steps 600 --warmup 100 2>/dev/null |
			python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B $B kernel', d['config']['kernel'], round(d['roofline']['avg_launch_us'],3), 'us', round(d['value']/1e6,1), 'M/s valid', d['valid'], 'xerr', d.get('cross_kernel_max_rel_err'))"
	done
done

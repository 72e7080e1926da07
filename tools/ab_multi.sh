#!/bin/bash
# same-session comparison of several builds: tools/ab_multi.sh "<configs>" lib1.so lib2.so ...
CFGS=$1; shift
for rep in 1 2; do
	for lib in "$@"; do
		for c in $CFGS; do
			SAIP_LIB=$lib python bench.py --no-cpu-baseline --config $c --steps 600 --warmup 100 2>/dev/null |
				python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) cfg$c', round(d['roofline']['avg_launch_us'],3), 'us valid', d['valid'])"
		done
	done
done

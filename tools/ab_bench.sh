#!/bin/bash
# A/B of two builds of libsaip.so inside ONE GPU session (box-to-box variance is ~10 %, so only same-session numbers compare):
#   tools/ab_bench.sh path/to/libsaip_old.so path/to/libsaip_new.so [configs...]
OLD=$1; NEW=$2; shift 2
CFGS=${@:-2 3 6}
for rep in 1 2; do
	for lib in "$OLD" "$NEW"; do
		for c in $CFGS; do
			SAIP_LIB=$lib python bench.py --no-cpu-baseline --config $c --steps 600 --warmup 100 2>/dev/null |
				python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) cfg$c', round(d['roofline']['avg_launch_us'],3), 'us valid', d['valid'])"
		done
	done
done
for lib in "$OLD" "$NEW"; do
	SAIP_LIB=$lib python bench.py --no-cpu-baseline --batch 65536 --steps 300 --warmup 50 2>/dev/null |
		python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) B=65536', round(d['roofline']['avg_launch_us'],3), 'us valid', d['valid'])"
done

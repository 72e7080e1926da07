#!/bin/bash
# A/B of builds of libsaip.so on config 5 inside ONE GPU session:  tools/ab_wave.sh lib1.so lib2.so ...
for rep in 1 2; do
	for lib in "$@"; do
		SAIP_LIB=$lib python bench.py --no-cpu-baseline --config 5 --steps 40 --warmup 5 2>/dev/null |
			python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) cfg5', round(d['roofline']['avg_launch_us'],1), 'us valid', d['valid'], 'err', d.get('cross_kernel_max_rel_err'))"
	done
done

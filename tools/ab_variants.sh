for rep in 1 2; do for v in "" _var_prev _var_nofront; do
  lib=sai-primitives_amd/libsaip$v.so
  SAIP_LIB=$lib python bench.py --no-cpu-baseline --config 2 --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['avg_launch_us'],3), 'us valid', d['valid'])"
done; done

# A/B of library variants inside one GPU session: tools/ab_variants.sh lib1.so lib2.so ...   (config 2; also SAIP_OCT_NO_DUO=1 for each)
for rep in 1 2; do for lib in "$@"; do for mode in duo noduo; do
  if [ $mode = noduo ]; then export SAIP_OCT_NO_DUO=1; else unset SAIP_OCT_NO_DUO; fi
  SAIP_LIB=$lib python bench.py --no-cpu-baseline --config 2 --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $mode', round(d['roofline']['avg_launch_us'],3), 'us valid', d['valid'])"
done; done; done

for rep in 1 2; do
for v in old new; do
for c in 2 3 6; do
SAIP_LIB=$PWD/sai-primitives_amd/libsaip_$v.so python bench.py --no-cpu-baseline --config $c --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v cfg$c', round(d['roofline']['avg_launch_us'],3), d['valid'])"
done; done; done

# batch sweep, two-wavefront vs one-wavefront form of the lean eight-lane kernel (one GPU session)
for B in 512 2048 4096 6144 8192; do
  for mode in duo noduo; do
    if [ $mode = noduo ]; then export SAIP_OCT_NO_DUO=1; else unset SAIP_OCT_NO_DUO; fi
    python bench.py --no-cpu-baseline --config 2 --batch $B --steps 400 --warmup 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B $mode', round(d['roofline']['avg_launch_us'],3), 'us', round(d['value']/1e6,1), 'M/s valid', d['valid'], d['roofline']['kernel'])"
  done
done

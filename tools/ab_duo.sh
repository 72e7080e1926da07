# A/B inside one GPU session: two-wavefront form vs one-wavefront form (SAIP_OCT_NO_DUO=1) vs an older library if given
for rep in 1 2; do
  for mode in duo noduo; do
    for c in 2 14; do
      if [ $mode = noduo ]; then export SAIP_OCT_NO_DUO=1; else unset SAIP_OCT_NO_DUO; fi
      python bench.py --no-cpu-baseline --config $c --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode cfg$c', round(d['roofline']['avg_launch_us'],3), 'us valid', d['valid'], d['roofline']['kernel'])"
    done
  done
done
unset SAIP_OCT_NO_DUO
for lib in "$@"; do SAIP_LIB=$lib python bench.py --no-cpu-baseline --config 2 --steps 600 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['avg_launch_us'],3), 'us valid', d['valid'])"; done

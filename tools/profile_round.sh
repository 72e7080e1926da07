#!/bin/bash
# Measurement pass for the headline configuration (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 200 --warmup 20`      -> kernel average durations
#   2. separate rocprofv3 --pmc passes (one counter group per run; gpurun refuses PMC together with other tracing)
#   3. the default bench line
# Everything lands under gpurun_out/prof_$TAG; tools/summarise_profile.py turns it into the committed profiles/ files.
set -e
TAG=${1:-v5}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR; do
	rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1 || echo "counter $C failed"
done
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json

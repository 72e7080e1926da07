#!/bin/bash
# Measurement pass for one configuration (run on the GPU box through gpurun):
#   tools/profile_cfg.sh <config> <tag> [steps] [pmc: 0|1]        (SAIP_PROFILE_BATCH=N: instances per GPU instead of the configuration's own)
#   1. rocprofv3 --kernel-trace --stats of `bench.py --config C --steps S`         -> kernel average durations
#   2. (pmc = 1) separate rocprofv3 --pmc passes, one counter per run (gpurun refuses PMC together with other tracing)
#   3. the bench line itself
# Everything lands under gpurun_out/prof_cfg<C>_<tag>; tools/summarise_cfg.py turns it into the committed profiles/r04_* files.
set -e
CFG=${1:-2}; TAG=${2:-v1}; STEPS=${3:-200}; PMC=${4:-1}
# counters of the PMC passes (one rocprofv3 run each); override with SAIP_PMC_COUNTERS="A B C"
COUNTERS=${SAIP_PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD}
OUT=gpurun_out/prof_cfg${CFG}_$TAG
BATCH=${SAIP_PROFILE_BATCH:+--batch $SAIP_PROFILE_BATCH}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WARM=$((STEPS / 10 + 2))
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --config $CFG $BATCH --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/stats.log 2>&1
if [ "$PMC" = "1" ]; then
	for C in $COUNTERS; do
		rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o run -- python3 bench.py --config $CFG $BATCH --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1 || echo "counter $C failed"
	done
fi
python3 bench.py --config $CFG $BATCH --steps $STEPS --warmup $WARM > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json

#!/bin/bash
# round 4: one clean measurement pass per configuration on the code at HEAD (rocprofv3 directly in front of `python3 bench.py`):
#   tools/profile_r04.sh <tag> "<configs with PMC>" "<configs with kernel stats only>"
# kernel trace + stats (mean AND median per kernel from the raw dispatch trace: tools/summarise_cfg.py), PMC passes for the first list,
# the bench line with the CPU baseline.  The box is warmed first: the first profiled run of a session is noisy.
TAG=${1:-v1}; FULL=${2:-"2 3 5"}; LIGHT=${3:-"4 6 12 14"}
export SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
python3 bench.py --no-cpu-baseline --steps 3000 > /dev/null 2>&1
for C in $FULL; do
	S=400; [ "$C" = "5" ] && S=60
	bash tools/profile_cfg.sh $C $TAG $S 1 > gpurun_out/profile_r04_cfg$C.log 2>&1; tail -c 160 gpurun_out/profile_r04_cfg$C.log; echo
done
for C in $LIGHT; do
	SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE" bash tools/profile_cfg.sh $C $TAG 400 1 > gpurun_out/profile_r04_cfg$C.log 2>&1; tail -c 160 gpurun_out/profile_r04_cfg$C.log; echo
done
# the driver's own form of the headline line (20 steps) and the default line
mkdir -p gpurun_out/prof_cfg2_$TAG
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_cfg2_$TAG/bench_steps20.json 2> /dev/null
python3 bench.py > gpurun_out/prof_cfg2_$TAG/bench_default.json 2> /dev/null
tail -c 300 gpurun_out/prof_cfg2_$TAG/bench_steps20.json

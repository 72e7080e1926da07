#!/bin/bash
# round-3 measurement pass B: the headline configuration (all counters of the review list + MFMA), config 5 on the wavefront kernel, config 6
TAG=${1:-v19}
bash tools/profile_cfg.sh 2 $TAG 200 1 > gpurun_out/profile_r03_cfg2.log 2>&1; tail -c 300 gpurun_out/profile_r03_cfg2.log; echo
python3 bench.py --config 2 --steps 20 --warmup 5 > gpurun_out/prof_cfg2_$TAG/bench_steps20.json 2> /dev/null
export SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU"
bash tools/profile_cfg.sh 5 v2 60 1 > gpurun_out/profile_r03_cfg5.log 2>&1; tail -c 300 gpurun_out/profile_r03_cfg5.log; echo

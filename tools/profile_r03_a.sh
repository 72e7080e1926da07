#!/bin/bash
# round-3 measurement pass A: configs 3, 6, 12, 14 at 4096 instances (kernel stats + the PMC counters of the review list)
export SAIP_PROFILE_BATCH=4096
export SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES"
for C in "$@"; do bash tools/profile_cfg.sh $C r3 200 1 > gpurun_out/profile_r03_cfg$C.log 2>&1; tail -c 200 gpurun_out/profile_r03_cfg$C.log; echo; done

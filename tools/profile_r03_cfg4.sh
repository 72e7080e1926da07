#!/bin/bash
# round 3: config 4's per-GPU shard (8192 instances: the one-wavefront form, one chip-load) and the config-2 stack at 16384 (two rounds of the eight-lane kernel)
export SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
bash tools/profile_cfg.sh 4 v1 400 1 > gpurun_out/profile_r03_cfg4.log 2>&1; tail -c 200 gpurun_out/profile_r03_cfg4.log; echo
SAIP_PROFILE_BATCH=16384 bash tools/profile_cfg.sh 2 B16384 400 1 > gpurun_out/profile_r03_cfg2_B16384.log 2>&1; tail -c 200 gpurun_out/profile_r03_cfg2_B16384.log; echo

#!/bin/bash
# round-3 measurement pass C (final code): config 5 (wavefront kernel), config 6 (joint-first eight-lane kernel), the 20-step headline line
export SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"
bash tools/profile_cfg.sh 5 v3 60 1 > gpurun_out/profile_r03_cfg5.log 2>&1; tail -c 200 gpurun_out/profile_r03_cfg5.log; echo
export SAIP_PROFILE_BATCH=4096
export SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES"
bash tools/profile_cfg.sh 6 r3b 200 1 > gpurun_out/profile_r03_cfg6.log 2>&1; tail -c 200 gpurun_out/profile_r03_cfg6.log; echo
unset SAIP_PROFILE_BATCH
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_steps20_v20.json 2> /dev/null; tail -c 300 gpurun_out/bench_steps20_v20.json
python3 bench.py > gpurun_out/bench_default_v20.json 2> /dev/null

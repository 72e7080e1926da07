#!/bin/bash
# round 3, final code: headline (tag v26; driver-style 20-step line and the default line) and config 4's shard (v3)
export SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
python3 bench.py --no-cpu-baseline --steps 3000 > /dev/null 2>&1  # (warm the box: the first profiled run of a session is noisy)
bash tools/profile_cfg.sh 2 v26 400 1 > gpurun_out/profile_r03_cfg2.log 2>&1; tail -c 200 gpurun_out/profile_r03_cfg2.log; echo
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_cfg2_v26/bench_steps20.json 2> /dev/null
python3 bench.py > gpurun_out/prof_cfg2_v26/bench_default.json 2> /dev/null
bash tools/profile_cfg.sh 4 v3 400 1 > gpurun_out/profile_r03_cfg4.log 2>&1; tail -c 200 gpurun_out/profile_r03_cfg4.log; echo

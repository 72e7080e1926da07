#!/bin/bash
# round-3 last measurement pass (after the two-wavefront hand-over change): headline with the full counter list + config 14 (same instantiation)
python3 bench.py --no-cpu-baseline --steps 2000 > /dev/null 2>&1  # (warm the box: the first profiled run of a session is noisy)
bash tools/profile_cfg.sh 2 v24 400 1 > gpurun_out/profile_r03_cfg2.log 2>&1; tail -c 200 gpurun_out/profile_r03_cfg2.log; echo
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_cfg2_v24/bench_steps20.json 2> /dev/null
python3 bench.py > gpurun_out/prof_cfg2_v24/bench_default.json 2> /dev/null
export SAIP_PROFILE_BATCH=4096
export SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES"
bash tools/profile_cfg.sh 14 v5 200 1 > gpurun_out/profile_r03_cfg14.log 2>&1; tail -c 120 gpurun_out/profile_r03_cfg14.log; echo

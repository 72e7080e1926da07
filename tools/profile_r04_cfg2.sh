#!/bin/bash
# round 4: the headline configuration once more, the way that gave clean numbers in round 3 (v25): the box warmed by an UNRECORDED profiled run of
# the same command (the first rocprofv3 run of a session is slow and noisy: median 7.5 against 6.7 us in pass b), then the recorded passes at two
# run lengths -- 400 steps (what the other configurations use) and 2000 (bench.py's default) -- and the bench lines themselves.
TAG=${1:-c}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --no-cpu-baseline --steps 3000 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_warm -o run -- python3 bench.py --config 2 --steps 2000 --warmup 200 --no-cpu-baseline > /dev/null 2>&1
rm -rf gpurun_out/prof_warm
SAIP_PMC_COUNTERS="FETCH_SIZE WRITE_SIZE" bash tools/profile_cfg.sh 2 ${TAG}2000 2000 1 > gpurun_out/profile_r04_cfg2_2000.log 2>&1; tail -c 120 gpurun_out/profile_r04_cfg2_2000.log; echo
bash tools/profile_cfg.sh 2 $TAG 400 0 > gpurun_out/profile_r04_cfg2.log 2>&1; tail -c 120 gpurun_out/profile_r04_cfg2.log; echo
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_cfg2_$TAG/bench_steps20.json 2> /dev/null
python3 bench.py > gpurun_out/prof_cfg2_$TAG/bench_default.json 2> /dev/null

#!/bin/bash
# GPU box: fixed overhead of a 20-step timed region: engine wait polling (SAIP_SPIN_WAIT_US) x runtime active wait (ROC_ACTIVE_WAIT_TIMEOUT, us)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for AW in 0 200 5000; do  # (measured: no effect on this runtime)
for SPIN in 2000 0; do
  ROC_ACTIVE_WAIT_TIMEOUT=$AW SAIP_SPIN_WAIT_US=$SPIN python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('active wait $AW spin $SPIN: 20 steps', round(d['ms_per_step']*1e3,3), 'us per step,', round(d['roofline']['avg_launch_us'],3), 'us per launch, value', round(d['value']/1e6,1), 'M/s valid', d['valid'])"
done; done; done
ROC_ACTIVE_WAIT_TIMEOUT=5000 SAIP_SPIN_WAIT_US=0 python tools/bench_overhead_probe.py 2>/dev/null | tail -9

#!/bin/bash
# A/B: grid of the device-side slow-path launch (empty list on config 3): kernel time of saip_cycle_wg_list by rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in 256 64 16 4; do  # (needs the SAIP_LIST_GRID knob of launch_cycle_wg_list, removed again after the measurement)
	export SAIP_LIST_GRID=$g
	rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lg_$g -o run -- python3 bench.py --config 3 --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/lg_$g.log 2>&1
	echo "grid $g: $(grep -h 'wg_list\|cycle_oct' gpurun_out/lg_$g/*kernel_stats.csv gpurun_out/lg_$g/*/*kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | tr '\n' ' ')  $(python3 -c "import json;d=json.loads(open('gpurun_out/lg_$g.log').read().strip().splitlines()[-1]);print('ms_per_step',round(d['ms_per_step']*1000,2),'us')")"
done

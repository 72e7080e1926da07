#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE calibration (see fetch_calib.hip) + the reciprocal-estimate accuracy check
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/calib; mkdir -p $OUT
tools/ubench/rcp_accuracy > $OUT/rcp_accuracy.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o run -- tools/ubench/fetch_calib > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o run -- tools/ubench/fetch_calib > $OUT/w.log 2>&1
python3 tools/ubench/fetch_calib_report.py $OUT $OUT/fetch_calib.json > $OUT/fetch_calib.txt
cat $OUT/rcp_accuracy.txt $OUT/fetch_calib.txt

#!/bin/bash
# GPU box: A/B of the single-Newton-step variant of oct_rcp / oct_rsqrt (tools/oct_variant.sh newton1 -DSAIP_OCT_EXP_NEWTON1)
cd "$GRAFT_REPO_ROOT"
V=sai-primitives_amd/libsaip_var_newton1.so
bash tools/ab_variants.sh sai-primitives_amd/libsaip.so $V > gpurun_out/ab_newton.log 2>&1
SAIP_LIB=$V python -m pytest tests/test_gpu_oct.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/ab_newton_tests.log 2>&1
tail -3 gpurun_out/ab_newton_tests.log; cat gpurun_out/ab_newton.log

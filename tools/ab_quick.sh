#!/bin/bash
# GPU box: quick A/B of library variants on the headline configuration (both forms) -- no tests; tools/ab_quick.sh lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT"
bash tools/ab_variants.sh "$@" 2>&1 | tee gpurun_out/ab_quick.log

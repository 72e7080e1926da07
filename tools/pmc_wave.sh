#!/bin/bash
# where a wavefront of the config-5 kernel spends its cycles: issue / wait / cache counters, one rocprofv3 --pmc pass each
#   tools/pmc_wave.sh [tag] [lib]           -> gpurun_out/pmc_wave_<tag>/summary.txt
TAG=${1:-a}; LIB=${2:-sai-primitives_amd/libsaip.so}
OUT=gpurun_out/pmc_wave_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SAIP_LIB=$LIB
for C in ${SAIP_PMC_COUNTERS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE}; do
	rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o run -- python3 bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/$C.log 2>&1 || echo "counter $C failed"
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        vals = {}
        for row in csv.DictReader(open(f)):
            if "saip_cycle_wave" in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in vals.items():
            print(f"{k:32s} {sum(v) / len(v):16.1f} per launch ({len(v)} launches)")
PY

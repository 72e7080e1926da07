cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_sizes.py tests/test_gpu_wave.py tests/test_gpu_oct.py tests/test_gpu_octjf.py tests/test_gpu_dynamics.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_multi.sh "2 3 4" sai-primitives_amd/libsaip_var_noremap.so sai-primitives_amd/libsaip.so
bash tools/ab_wave.sh sai-primitives_amd/libsaip_var_noremap.so sai-primitives_amd/libsaip.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in sai-primitives_amd/libsaip_var_noremap.so sai-primitives_amd/libsaip.so; do for cfg in 2 5; do for C in FETCH_SIZE WRITE_SIZE; do
  SAIP_LIB=$lib rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/remap_$C -o run -- python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("gpurun_out/remap_$C/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if ("saip_cycle_wave" in r["Kernel_Name"] or "saip_cycle_oct" in r["Kernel_Name"]) and r["Counter_Name"]=="$C"]
print("$lib cfg$cfg $C", sum(v)/len(v) if v else float("nan"), "KiB per launch,", len(v), "launches")
PY
  rm -rf gpurun_out/remap_$C
done; done; done

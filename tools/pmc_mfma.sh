#!/bin/bash
# MFMA utilisation of every kernel from SQ counters: one rocprofv3 --pmc pass (SQ block, 8 slots; kernel-trace only,
# as gpurun requires). MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES or duration x clock) -- tools/pmc_mfma_summary.py.
# Output: gpurun_out/pmc_mfma/*.csv, gpurun_out/pmc_mfma.json
set -u
REPO=$(pwd)
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > "$REPO/gpurun_out/counters_list.txt" 2>&1
python3 "$REPO/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --latency-frames 5 --tune-cache /tmp/tune.json > /dev/null 2>&1
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf "$REPO/gpurun_out/pmc_mfma_$tag"
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_mfma_$tag" -o pmc -- \
    python3 "$REPO/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --latency-frames 5 --tune-cache /tmp/tune.json > "$REPO/gpurun_out/pmc_mfma_$tag.log" 2>&1
  echo "$tag exit $?"
done
cd "$REPO"
python3 tools/pmc_mfma_summary.py gpurun_out/pmc_mfma_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/pmc_mfma_SQ_INSTS_MFMA > gpurun_out/pmc_mfma.json && head -c 3000 gpurun_out/pmc_mfma.json

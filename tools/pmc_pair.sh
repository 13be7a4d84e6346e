#!/bin/bash
# SQ counters of the head-pair kernel (and everything else in the frame): MFMA busy, waits, LDS activity / bank conflicts.
# usage (on the GPU box): tools/pmc_pair.sh [bench args]; output gpurun_out/pmc_pair_<n>/, summary gpurun_out/pmc_pair.json
set -u
REPO=$(pwd)
export TMPDIR=/tmp; cd /tmp
n=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU"; do
  n=$((n+1))
  rm -rf "$REPO/gpurun_out/pmc_pair_$n"
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_pair_$n" -o pmc -- \
    python3 "$REPO/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --latency-frames 5 "$@" > "$REPO/gpurun_out/pmc_pair_$n.log" 2>&1
  echo "pass $n exit $?"
done
cd "$REPO"
python3 tools/pmc_mfma_summary.py gpurun_out/pmc_pair_1 gpurun_out/pmc_pair_2 > gpurun_out/pmc_pair.json

#!/bin/bash
# HBM-side traffic of every kernel from the L2's fabric counters, one rocprofv3 --pmc pass per counter (TCC slots do
# not fit FETCH_SIZE and WRITE_SIZE together: MI355X_MICROARCH.md, rocprofv3 PMC slots).
#   tools/pmc_traffic.sh [precision [size [variant]]]     (defaults: fp16 640 A)
# Output: gpurun_out/pmc_<counter>_<key>/ and the merged table gpurun_out/pmc_traffic.json, keyed by configuration
# ("variant:precision:size") -- copy it to profiles/rNN/pmc_traffic.json, where bench.py looks `roofline.traffic` up.
set -u
PREC=${1:-fp16}; SIZE=${2:-640}; VAR=${3:-A}
KEY="$VAR:$PREC:$SIZE"; TAG="${VAR}_${PREC}_${SIZE}"
REPO=$(pwd)
export TMPDIR=/tmp; cd /tmp
ARGS="--precision $PREC --size $SIZE --variant $VAR --no-cpu-baseline --latency-frames 5 --tune-cache /tmp/tune_$TAG.json"
# tactic cache first (no tuning launches inside the counted runs)
python3 "$REPO/bench.py" --steps 50 --warmup 10 $ARGS > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$REPO/gpurun_out/pmc_${c}_$TAG"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_${c}_$TAG" -o pmc -- \
    python3 "$REPO/bench.py" --steps 100 --warmup 20 $ARGS > "$REPO/gpurun_out/pmc_${c}_$TAG.log" 2>&1
  echo "$c exit $?"
done
cd "$REPO"
python3 tools/pmc_summary.py "gpurun_out/pmc_FETCH_SIZE_$TAG" "gpurun_out/pmc_WRITE_SIZE_$TAG" --key "$KEY" --merge gpurun_out/pmc_traffic.json

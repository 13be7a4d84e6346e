#!/bin/bash
# HBM-side traffic of every kernel from the L2's fabric counters, one rocprofv3 --pmc pass per counter (TCC slots do
# not fit FETCH_SIZE and WRITE_SIZE together: MI355X_MICROARCH.md, rocprofv3 PMC slots). Output: gpurun_out/pmc/*.csv
set -u
REPO=$(pwd)
export TMPDIR=/tmp; cd /tmp
# tactic cache first (no tuning launches inside the counted runs)
python3 "$REPO/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --latency-frames 5 --tune-cache /tmp/tune.json > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$REPO/gpurun_out/pmc_$c"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_$c" -o pmc -- \
    python3 "$REPO/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --latency-frames 5 --tune-cache /tmp/tune.json > "$REPO/gpurun_out/pmc_$c.log" 2>&1
  echo "$c exit $?"; ls "$REPO/gpurun_out/pmc_$c" | head
done
cd "$REPO"
python3 tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json && head -c 1500 gpurun_out/pmc_traffic.json

#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (and of --precision strict): per-kernel average durations,
# the numbers bench.py's live HIP-event timings are checked against. Output: gpurun_out/stats_<tag>/ -> copy the
# *_kernel_stats.csv into profiles/rNN/.
set -u
REPO=$(pwd)
export TMPDIR=/tmp; cd /tmp
for PREC in fp16 strict; do
  ARGS="--precision $PREC --no-cpu-baseline --latency-frames 20 --tune-cache /tmp/tune_$PREC.json"
  python3 "$REPO/bench.py" --steps 50 --warmup 10 $ARGS > /dev/null 2>&1     # tactic cache: no tuning launches in the traced run
  rm -rf "$REPO/gpurun_out/stats_$PREC"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/stats_$PREC" -o stats -- \
    python3 "$REPO/bench.py" --steps 2000 --warmup 200 $ARGS > "$REPO/gpurun_out/stats_$PREC.json" 2> "$REPO/gpurun_out/stats_$PREC.err"
  echo "$PREC exit $?"
  find "$REPO/gpurun_out/stats_$PREC" -name "*kernel_stats.csv" | head -2
done

#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (and of --precision strict): per-kernel average durations,
# the numbers bench.py's live HIP-event timings are checked against. Output: gpurun_out/stats_<tag>/ -> copy the
# *_kernel_stats.csv into profiles/rNN/.
# The default command keeps TWO frames in flight on two streams: kernels of the two frames overlap on the chip and their traced
# durations stretch (the dominant pair: 15.6 us traced against 13.4 us event-timed in the serial frame). The third pass traces the
# same command at ONE frame in flight (UNINA_IN_FLIGHT=1): that is the number `roofline.avg_launch_us` has to agree with.
set -u
REPO=$(pwd)
export TMPDIR=/tmp; cd /tmp
for PREC in fp16 strict serial; do
  unset UNINA_IN_FLIGHT
  if [ $PREC = serial ]; then export UNINA_IN_FLIGHT=1; PREC=fp16; TAG=fp16_serial; else TAG=$PREC; fi
  ARGS="--precision $PREC --no-cpu-baseline --latency-frames 20 --tune-cache /tmp/tune_$PREC.json"
  python3 "$REPO/bench.py" --steps 50 --warmup 10 $ARGS > /dev/null 2>&1     # tactic cache: no tuning launches in the traced run
  rm -rf "$REPO/gpurun_out/stats_$TAG"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/stats_$TAG" -o stats -- \
    python3 "$REPO/bench.py" --steps 2000 --warmup 200 $ARGS > "$REPO/gpurun_out/stats_$TAG.json" 2> "$REPO/gpurun_out/stats_$TAG.err"
  echo "$TAG exit $?"
  find "$REPO/gpurun_out/stats_$TAG" -name "*kernel_trace.csv" -delete
  find "$REPO/gpurun_out/stats_$TAG" -name "*kernel_stats.csv" | head -2
done

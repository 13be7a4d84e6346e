#!/bin/bash
# One GPU-box visit: parity tests -> smoke -> bench [-> rocprofv3 kernel trace]. Each step runs under its own
# timeout; a step that times out (124/137) ends the visit (no further GPU work after a hang). Logs: gpurun_out/.
set -u
REPO=$(pwd)
mkdir -p gpurun_out
run() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "== $name"; timeout -k 10 "$secs" "$@" > "$REPO/gpurun_out/$name.log" 2>&1; local rc=$?
  echo "== $name exit $rc"; grep -v amdgpu.ids "$REPO/gpurun_out/$name.log" | tail -n "${TAILN:-15}"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit $rc; fi
  return $rc
}
[ -n "${SKIP_TESTS:-}" ] || run tests 420 python -m pytest tests -m gpu -q --timeout 300 ${PYTEST_ARGS:-}
[ -n "${SKIP_TESTS:-}" ] || run smoke 120 python -c 'import __graft_entry__ as g; g.smoke()'
run bench 300 python bench.py --gpus 1 --tune-cache /tmp/tune.json ${BENCH_ARGS:-}
if [ -n "${PROFILE:-}" ]; then
  export TMPDIR=/tmp; cd /tmp
  rm -rf "$REPO/gpurun_out/prof"
  TAILN=5 run rocprof 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o trace --output-format csv -- python3 "$REPO/bench.py" --gpus 1 --steps 300 --warmup 50 --no-cpu-baseline --latency-frames 50 --tune-cache /tmp/tune.json
  cd "$REPO"; find gpurun_out/prof -name '*stats*' | head; 
fi
exit 0

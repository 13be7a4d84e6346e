#!/bin/bash
# One GPU-box visit: parity tests -> smoke -> bench. Each step runs under its own timeout; a step that
# times out (124/137) ends the visit (no further GPU work after a hang). Logs land in gpurun_out/.
set -u
mkdir -p gpurun_out
run() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "== $name"; timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "== $name exit $rc"; tail -n "${TAILN:-15}" "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit $rc; fi
  return $rc
}
run tests 420 python -m pytest tests -m gpu -q -x --timeout 300 ${PYTEST_ARGS:-}
run smoke 120 python -c 'import __graft_entry__ as g; g.smoke()'
run bench 300 python bench.py --gpus 1 ${BENCH_ARGS:-}
exit 0

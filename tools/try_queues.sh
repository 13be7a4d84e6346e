#!/bin/bash
for q in "" 2 4 8 16; do for f in 2 3; do
  echo "== GPU_MAX_HW_QUEUES=${q:-default} in_flight=$f"
  ( [ -n "$q" ] && export GPU_MAX_HW_QUEUES=$q; UNINA_IN_FLIGHT=$f python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --latency-frames 100 --streams 1 --tune-cache /tmp/tune.json 2>&1 | grep -v amdgpu | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fps', d['value'], 'lat p50', d['latency_ms']['p50'])" )
done; done

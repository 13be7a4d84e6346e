#!/bin/bash
# graph parallelism x HW queues: serial latency is the figure of merit here
for q in 8 16; do for s in 1 2 3 4; do
  echo -n "GPU_MAX_HW_QUEUES=$q streams=$s: "
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 600 --warmup 100 --no-cpu-baseline --latency-frames 200 --streams $s --tune-cache /tmp/tune.json 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('fps', d['value'], 'lat p50', d['latency_ms']['p50'], 'p99', d['latency_ms']['p99'])"
done; done

#!/bin/bash
# compares graph parallelism settings (fps / latency) on one box
for s in 1 2 3 4; do
  echo "== streams $s"
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --latency-frames 200 --streams $s --tune-cache /tmp/tune.json 2>&1 | grep -v amdgpu | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fps', d['value'], 'lat p50', d['latency_ms']['p50'], 'p99', d['latency_ms']['p99'], 'sum_ops_ms', d['roofline']['sum_of_ops_ms'])"
done

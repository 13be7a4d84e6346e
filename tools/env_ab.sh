#!/bin/bash
# same-box A/B of environment switches: tools/env_ab.sh "A=1" "B=0 C=1" ...   (2 interleaved rounds; "X=0" = defaults)
for round in 1 2; do
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --latency-frames 500 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$e round $round:', d['value'], 'fps  p50', d['latency_ms']['p50'], 'p99', d['latency_ms']['p99'], ' sum_of_ops', r['sum_of_ops_ms'], ' dom', r['kernel'][:28], r['avg_launch_us'])"
done
done

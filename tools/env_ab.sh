for round in 1 2; do
for e in "X=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HIP_FORCE_DEV_KERNARG=0"; do
  env $e timeout -k 10 200 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --latency-frames 500 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e round $round:', d['value'], 'fps  p50', d['latency_ms']['p50'], 'p99', d['latency_ms']['p99'])"
done
done

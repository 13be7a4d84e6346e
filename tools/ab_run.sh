#!/bin/bash
# same-box A/B of variant libraries: tools/ab_run.sh NAME [NAME...]  ("main" = libunina_mi355.so); 3 interleaved rounds
R=$(pwd)
for round in 1 2 3; do
for n in "$@"; do
  lib=$R/unina-yolo-dla_amd/libunina_$n.so; [ "$n" = main ] && lib=$R/unina-yolo-dla_amd/libunina_mi355.so
  UNINA_LIB=$lib timeout -k 10 200 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --latency-frames 500 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n round $round:', d['value'], 'fps  p50', d['latency_ms']['p50'], ' sum_of_ops', d['roofline']['sum_of_ops_ms'], ' dom', d['roofline']['avg_launch_us'])"
done
done

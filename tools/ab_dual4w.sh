for v in 0 3 0 3; do
echo "UNINA_DUAL_4W=$v"
UNINA_DUAL_4W=$v timeout -k 10 120 python tools/profile_ops.py 2>/dev/null | grep -E "conv_dual_head3x3|sum of ops"
UNINA_DUAL_4W=$v timeout -k 10 200 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['latency_ms']['p50'])"
done

timeout -k 10 300 python tools/block_phases.py > gpurun_out/block_phases_v2.txt 2>&1; cat gpurun_out/block_phases_v2.txt | grep -v amdgpu.ids; for w in 0 1 0 1; do UNINA_L2_WARM=$w timeout -k 10 200 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('warm=$w', d['value'], d['latency_ms']['p50'], d['roofline']['sum_of_ops_ms'], [(k['kernel'][:40],k['us_per_frame']) for k in d['roofline']['top_kernels'][:6]])
"; done

#!/bin/bash
# All bench configurations back to back on one box -> gpurun_out/bench_configs.jsonl (one JSON line each).
set -e
out=gpurun_out/bench_configs.jsonl
mkdir -p gpurun_out; : > $out
python bench.py --steps 2000 --warmup 200 | tail -1 >> $out
python bench.py --steps 2000 --warmup 200 --precision strict --no-cpu-baseline | tail -1 >> $out
python bench.py --steps 2000 --warmup 200 --precision int8 --no-cpu-baseline | tail -1 >> $out
python bench.py --steps 500 --warmup 50 --precision fp32 --no-cpu-baseline | tail -1 >> $out
python bench.py --steps 1000 --warmup 100 --size 1280 --no-cpu-baseline | tail -1 >> $out
python bench.py --steps 2000 --warmup 200 --variant B --no-cpu-baseline | tail -1 >> $out
python bench.py --steps 2000 --warmup 200 --variant B --precision int8 --no-cpu-baseline | tail -1 >> $out
echo done

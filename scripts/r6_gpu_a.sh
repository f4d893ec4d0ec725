#!/bin/bash
# round 6, first GPU call: the whole -m gpu suite with durations (to tier it), smoke, the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 2700 python -m pytest tests/ -q -m gpu --durations=80 -x > $O/r6a_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r6a_gpu.log | tail -100
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r6a_bench.json 2> $O/r6a_bench.err; echo "bench rc=$?"; cut -c1-1500 $O/r6a_bench.json
timeout 300 python bench.py --dtype fp16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --traffic none > $O/r6a_bench_fp16.json 2> $O/r6a_bench_fp16.err; echo "bench fp16 rc=$?"; cut -c1-600 $O/r6a_bench_fp16.json

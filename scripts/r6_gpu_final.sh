#!/bin/bash
# round 6, final validation on the final build: the driver's three steps (pytest -m gpu, smoke(), the default bench command)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/r6_final_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r6_final_gpu.log | tail -4 | cut -c1-200
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/r06_smoke.log; echo "smoke rc=$?"; tail -5 $O/r06_smoke.log | cut -c1-300
( time timeout 900 python bench.py > $O/r06_bench_bf16.json 2> $O/r06_bench_bf16.err ) 2>&1 | grep real; cut -c1-300 $O/r06_bench_bf16.json

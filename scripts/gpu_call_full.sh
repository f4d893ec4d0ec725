#!/bin/bash
# the driver's round-end command: every -m gpu test, then smoke(), then the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/full_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -v "MIOpen\|amdgpu.ids" gpurun_out/full_gpu.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; echo "bench rc=$?"; cat gpurun_out/full_bench.json

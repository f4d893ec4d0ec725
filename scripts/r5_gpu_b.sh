#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/ -q -m gpu -x > gpurun_out/r5b_full_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -v "MIOpen\|amdgpu.ids" gpurun_out/r5b_full_gpu.log | tail -25
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -14 > gpurun_out/r5b_smoke.log; cat gpurun_out/r5b_smoke.log

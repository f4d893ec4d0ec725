#!/bin/bash
# the driver's round-end command: every -m gpu test, then smoke()
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r2_full_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -v "MIOpen\|amdgpu.ids" gpurun_out/r2_full_gpu.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6

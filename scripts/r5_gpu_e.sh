#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 200 python scripts/r5_slab_ab.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5e_slab.log 2>&1; cat gpurun_out/r5e_slab.log
timeout 2000 python -m pytest tests/ -q -m gpu -x > gpurun_out/r5e_full_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -v "MIOpen\|amdgpu.ids" gpurun_out/r5e_full_gpu.log | tail -8

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "in8 or conv_in_3_128" 2>&1 | grep -v "amdgpu.ids" | tail -15 ) > gpurun_out/r5d_ops.log 2>&1; tail -6 gpurun_out/r5d_ops.log
( timeout 400 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "golden or (handle_matches and kl_488_small)" 2>&1 | grep -v "amdgpu.ids" | tail -8 ) > gpurun_out/r5d_e2e.log 2>&1; tail -4 gpurun_out/r5d_e2e.log
( timeout 300 python scripts/r5_ab2.py 3 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5d_ab2.log 2>&1; cat gpurun_out/r5d_ab2.log
( timeout 200 python scripts/r5_slab_ab.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5d_slab.log 2>&1; cat gpurun_out/r5d_slab.log

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "temporal_block or split_independent" > gpurun_out/r2x_ops.log 2>&1; echo "ops rc=$?"; grep -v amdgpu.ids gpurun_out/r2x_ops.log | tail -8
for v in 1 0; do echo "== VT_TBLOCK_V3=$v"; VT_TBLOCK_V3=$v MB_ONLY=tblock timeout 120 python scripts/conv_microbench.py 2>&1 | grep "tblock fused"; done
timeout 120 python scripts/tblock_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2x_t3prof.txt; grep "step 9" gpurun_out/r2x_t3prof.txt

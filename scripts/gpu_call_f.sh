#!/bin/bash
# round 3, call F: two-group fused temporal block
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "temporal_block or split_independent or tblock" -x > $O/f_ops.log 2>&1; echo "ops rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/f_ops.log | tail -15
timeout 300 python scripts/tblock_profile.py > $O/f_tblock_stamps.txt 2>&1; grep "ms per launch\|bit for bit\|Error\|error" $O/f_tblock_stamps.txt

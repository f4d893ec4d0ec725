#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --timeout 600 -k "lerp" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu --timeout 600 -k "v11 or v1_1 or tiled" -x 2>&1 | tail -3
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-400

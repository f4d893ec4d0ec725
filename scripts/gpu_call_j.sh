#!/bin/bash
# round-2 GPU call J: A/B experiments: 4-wave 256x256 igemm tile (512 registers), zero-data power check, other configs
export TMPDIR=/tmp
mkdir -p gpurun_out
for tile in 0 4256; do
  echo "VT_CONV_TILE=$tile"; VT_CONV_TILE=$tile MB_ONLY="L1 spatial" timeout 120 python scripts/conv_microbench.py 2>&1 | grep "L1 \|L2 \|dec up" | tee -a gpurun_out/r2j_mb.log
  VT_CONV_TILE=$tile MB_ONLY="dec up" timeout 120 python scripts/conv_microbench.py 2>&1 | grep "L1 \|L2 \|dec up" | tee -a gpurun_out/r2j_mb.log
  VT_CONV_TILE=$tile MB_ONLY="L2 spatial" timeout 120 python scripts/conv_microbench.py 2>&1 | grep "L1 \|L2 \|dec up" | tee -a gpurun_out/r2j_mb.log
done
echo "zero-data A/B (WS 3x3 + tblock)"; for z in 0 1; do MB_ZERO=$z MB_LN=0 MB_ONLY="L0 spatial" timeout 120 python scripts/conv_microbench.py 2>&1 | grep "L0 spatial\|tblock" | tee -a gpurun_out/r2j_mb.log; done
VT_CONV_TILE=4256 timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "test_conv_large or conv3d_333_256" > gpurun_out/r2j_ops.log 2>&1; echo "ops(4256) rc=$?"; tail -3 gpurun_out/r2j_ops.log
timeout 400 python scripts/other_configs_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2j_other.log

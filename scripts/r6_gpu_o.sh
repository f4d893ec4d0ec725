#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
VIDTOK_AMD_LIB=$PWD/ab_libs/libvidtok_amd_l13.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --timeout 600 -k "temporal_block" -x 2>&1 | tail -3
for rep in 1 2; do
for lib in vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_l13.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 120 python scripts/c128_time.py bf16 2>&1 | grep -v amdgpu.ids | sed 's/| conv3x3.*//'
done
done | tee $O/r06_c128_variants9.txt

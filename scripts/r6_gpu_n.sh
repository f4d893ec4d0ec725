#!/bin/bash
# round 6: issue priorities in the fused temporal block (tp1 = group 1 static priority 2, no toggles; tp2 = the same + its GEMM at 3)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for rep in 1 2; do
for lib in vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_tp1.so ab_libs/libvidtok_amd_tp2.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 120 python scripts/c128_time.py bf16 2>&1 | grep -v amdgpu.ids | sed 's/| conv3x3.*//'
done
done | tee $O/r06_c128_variants8.txt

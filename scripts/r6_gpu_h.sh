#!/bin/bash
# round 6, eighth GPU call: store policy (nt) and start stagger variants of the two weight-stationary kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for rep in 1 2; do
for lib in vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_tb_nt.so ab_libs/libvidtok_amd_ws_nt.so ab_libs/libvidtok_amd_tb_stag.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 120 python scripts/c128_time.py bf16 2>&1 | grep -v amdgpu.ids
done
done | tee $O/r06_c128_variants3.txt

#!/bin/bash
# round 6, twelfth GPU call: split phase of conv3x3_ws2 with the head at issue priority 0 (group 1's phase at 2): head = 16 / 24 / 32 / 40
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for rep in 1 2; do
for lib in ab_libs/libvidtok_amd_ws_nt.so ab_libs/libvidtok_amd_p16.so vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_p32.so ab_libs/libvidtok_amd_p40.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 120 python scripts/c128_time.py bf16 2>&1 | grep -v amdgpu.ids | sed 's/tblock_pair (zero[^|]*| tblock_pair (rep[^|]*| //'
done
done | tee $O/r06_c128_variants6.txt
timeout 200 python scripts/ws2_profile.py > $O/r06_ws2_iteration_cycles.txt 2>&1; head -20 $O/r06_ws2_iteration_cycles.txt | cut -c1-250

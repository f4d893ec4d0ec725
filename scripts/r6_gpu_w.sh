#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --timeout 600 -k "temporal_block" -x 2>&1 | tail -2
for rep in 1 2; do
for lib in ab_libs/libvidtok_amd_nohold.so vidtok_amd/libvidtok_amd.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 120 python scripts/c128_time.py bf16 2>&1 | grep -v amdgpu.ids | sed 's/| conv3x3.*//'
done
done | tee $O/r06_c128_variants10.txt
timeout 200 python scripts/tblock_profile.py > $O/r06_tblock_pair_phase_cycles_hold.txt 2>&1; grep -v amdgpu $O/r06_tblock_pair_phase_cycles_hold.txt | awk 'NR<=3 || (NR>=18 && NR<=20)' | cut -c1-250

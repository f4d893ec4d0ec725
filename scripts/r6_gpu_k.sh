#!/bin/bash
# round 6, eleventh GPU call: conv3x3_ws2 with group 0's MFMA phase split across the end-of-iteration barrier (head = 16 / 24 / 32 / 48 MFMAs)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --timeout 600 -k "weight_stationary or streaming or conv_ln or test_conv" -x > $O/r6k_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/r6k_ops.log | cut -c1-250
for rep in 1 2; do
for lib in ab_libs/libvidtok_amd_ws_nt.so ab_libs/libvidtok_amd_h16.so ab_libs/libvidtok_amd_h24.so vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_h48.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 120 python scripts/c128_time.py bf16 2>&1 | grep -v amdgpu.ids | sed 's/tblock_pair (zero[^|]*| tblock_pair (rep[^|]*| //'
done
done | tee $O/r06_c128_variants5.txt
timeout 200 python scripts/ws2_profile.py > $O/r06_ws2_iteration_cycles.txt 2>&1; head -20 $O/r06_ws2_iteration_cycles.txt | cut -c1-250
for lib in ab_libs/libvidtok_amd_base.so vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_base.so vidtok_amd/libvidtok_amd.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/r06_step_variants5.txt

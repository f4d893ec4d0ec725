#!/bin/bash
# round 6, tenth GPU call: conv3x3_ws2 with its first fragments requested in front of the barriers; conv_nt_mb; full operator tests
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu --timeout 600 -x > $O/r6j_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/r6j_ops.log | cut -c1-250
for rep in 1 2; do
for lib in ab_libs/libvidtok_amd_ws_nt.so vidtok_amd/libvidtok_amd.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 120 python scripts/c128_time.py bf16 2>&1 | grep -v amdgpu.ids
done
done | tee $O/r06_c128_variants4.txt
timeout 200 python scripts/ws2_profile.py > $O/r06_ws2_iteration_cycles.txt 2>&1; head -20 $O/r06_ws2_iteration_cycles.txt | cut -c1-250
for lib in ab_libs/libvidtok_amd_base.so ab_libs/libvidtok_amd_igemm_nt.so vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_base.so vidtok_amd/libvidtok_amd.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/r06_step_variants4.txt

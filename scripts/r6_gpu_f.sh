#!/bin/bash
# round 6, sixth GPU call: the widest level's two kernels under library variants (same box): base = start of the session; fd4/5/6 = two register
# sets (round-5 structure) + diet + fragment prefetch distance 4/5/6; ss_fd3/4/8 = one register set + FD 3/4/8; shipped = one set + FD 6 + sub-tile K walk in ws2
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --timeout 600 -k "temporal_block or weight_stationary or tblock" > $O/r6f_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/r6f_ops.log | cut -c1-250
for rep in 1 2; do
for lib in ab_libs/libvidtok_amd_base.so ab_libs/libvidtok_amd_fd4.so ab_libs/libvidtok_amd_fd6.so ab_libs/libvidtok_amd_ss_fd3.so ab_libs/libvidtok_amd_ss_fd4.so vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_ss_fd8.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 120 python scripts/c128_time.py bf16 2>&1 | grep -v amdgpu.ids
done
done | tee $O/r06_c128_variants.txt
VIDTOK_AMD_LIB=$PWD/ab_libs/libvidtok_amd_base.so timeout 120 python scripts/c128_time.py f16 2>&1 | grep -v amdgpu.ids | tee -a $O/r06_c128_variants.txt
timeout 120 python scripts/c128_time.py f16 2>&1 | grep -v amdgpu.ids | tee -a $O/r06_c128_variants.txt
for lib in ab_libs/libvidtok_amd_base.so vidtok_amd/libvidtok_amd.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/r06_step_variants.txt
timeout 200 python scripts/tblock_profile.py > $O/r06_tblock_pair_phase_cycles.txt 2>&1; grep "per launch" $O/r06_tblock_pair_phase_cycles.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu --timeout 600 > $O/r6f_ops_all.log 2>&1; echo "ops all rc=$?"; tail -8 $O/r6f_ops_all.log | cut -c1-250

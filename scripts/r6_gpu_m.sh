#!/bin/bash
# round 6, thirteenth GPU call: head sizes 4 / 8 / 12 for the LayerNorm variants of conv3x3_ws2; whole step; the full GPU suite with durations
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for rep in 1 2; do
for lib in ab_libs/libvidtok_amd_ws_nt.so ab_libs/libvidtok_amd_q4.so ab_libs/libvidtok_amd_q8.so ab_libs/libvidtok_amd_q12.so vidtok_amd/libvidtok_amd.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 120 python scripts/c128_time.py bf16 2>&1 | grep -v amdgpu.ids | sed 's/tblock_pair (zero[^|]*| tblock_pair (rep[^|]*| //'
done
done | tee $O/r06_c128_variants7.txt
for lib in ab_libs/libvidtok_amd_base.so vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_base.so vidtok_amd/libvidtok_amd.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/r06_step_variants7.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=25 --timeout 900 > $O/r6m_full_gpu.log 2>&1; echo "full gpu rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r6m_full_gpu.log | tail -40 | cut -c1-200

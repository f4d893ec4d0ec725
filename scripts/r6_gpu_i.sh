#!/bin/bash
# round 6, ninth GPU call: whole step -- base vs shipped (diet + one-set temporal block + nt stores in the two weight-stationary kernels) vs the same
# with nt stores in every 16-bit epilogue of the implicit-GEMM kernel; per-group breakdowns of the last two; e2e subset on the shipped build
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for rep in 1 2; do
for lib in ab_libs/libvidtok_amd_base.so vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_igemm_nt.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done
done 2>&1 | tee $O/r06_step_variants3.txt
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none --no-extras > /dev/null 2> $O/r06_breakdown_shipped.txt
VIDTOK_AMD_LIB=$PWD/ab_libs/libvidtok_amd_igemm_nt.so timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none --no-extras > /dev/null 2> $O/r06_breakdown_igemm_nt.txt
timeout 300 python bench.py --dtype fp16 --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16 shipped', d['value'], d['ms_per_step'])" | tee -a $O/r06_step_variants3.txt
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -m gpu --durations=10 --timeout 900 -x > $O/r6i_e2e.log 2>&1; echo "e2e rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r6i_e2e.log | tail -25 | cut -c1-250

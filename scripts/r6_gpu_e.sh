#!/bin/bash
# round 6, fifth GPU call: VALU rate calibration with 1-4 waves per SIMD; the LayerNorm "diet" (one-pass moments, folded SiLU) against the
# round's base library on the same box (ab_libs/libvidtok_amd_base.so = the build before it); operator tests and the e2e subset on the new build
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 120 scripts/build/valu_rate_bench > $O/r06_valu_rate_bench.txt 2>&1; echo "valu bench rc=$?"; cat $O/r06_valu_rate_bench.txt
for rep in 1 2; do
  for lib in ab_libs/libvidtok_amd_base.so vidtok_amd/libvidtok_amd.so; do
    VIDTOK_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], 'z_rel', d.get('parity',{}))"
  done
done 2>&1 | tee $O/r06_ln_diet_ab.txt
VIDTOK_AMD_LIB=$PWD/ab_libs/libvidtok_amd_base.so timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none --no-extras > /dev/null 2> $O/r06_breakdown_base.txt
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none --no-extras > /dev/null 2> $O/r06_breakdown_diet.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x --timeout 600 > $O/r6e_ops.log 2>&1; echo "ops rc=$?"; tail -8 $O/r6e_ops.log | cut -c1-250
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -m gpu --durations=15 --timeout 900 > $O/r6e_e2e.log 2>&1; echo "e2e rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r6e_e2e.log | tail -40 | cut -c1-250

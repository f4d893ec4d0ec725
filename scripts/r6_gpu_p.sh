#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for rep in 1 2 3; do
for lib in vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_resboth.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done
done 2>&1 | tee $O/r06_step_variants8.txt
VIDTOK_AMD_LIB=$PWD/ab_libs/libvidtok_amd_resboth.so timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none --no-extras > /dev/null 2> $O/r06_breakdown_resboth.txt
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none --no-extras > /dev/null 2> $O/r06_breakdown_shipped2.txt

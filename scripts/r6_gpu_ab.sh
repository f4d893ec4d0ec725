#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for nt in 64 400 1000 16; do
  VT_CONV_NT_MB=$nt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 conv_nt_mb=$nt', d['value'], d['ms_per_step'])"
done
done 2>&1 | tee gpurun_out/r06_nt_threshold_ab.txt

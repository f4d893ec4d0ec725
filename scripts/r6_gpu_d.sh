#!/bin/bash
# round 6, fourth GPU call: VALU rate calibration (scripts/valu_rate_bench.hip), conv_stagger A/B on the bench step
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 120 scripts/build/valu_rate_bench > $O/r06_valu_rate_bench.txt 2>&1; echo "valu bench rc=$?"; cat $O/r06_valu_rate_bench.txt
for s in 0 2400 0 1200 4800; do
  VT_CONV_STAGGER=$s timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv_stagger=$s', d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/r06_conv_stagger_ab.txt

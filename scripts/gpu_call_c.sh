#!/bin/bash
# round 3, call C: two-group weight-stationary kernel (conv_ws = 2), K-step schedule 4
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "weight_stationary or split_independent or 8wave_schedules" > $O/c_ops.log 2>&1; echo "ops rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/c_ops.log | tail -25
for S in 4; do VT_CONV_SCHED=$S timeout 200 python scripts/conv_profile.py > $O/c_stamps_s$S.txt 2>&1; echo "stamps sched $S:"; grep -A8 "average step" $O/c_stamps_s$S.txt | head -24; done
for CFG in "2 1" "4 1" "2 2" "4 2"; do set -- $CFG
  VT_CONV_SCHED=$1 VT_CONV_WS=$2 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --breakdown > $O/c_bench_s$1_w$2.json 2> $O/c_bench_s$1_w$2.txt
  echo "sched=$1 ws=$2: $(python -c "import json,sys; d=json.load(open('$O/c_bench_s$1_w$2.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1 | tail -1)"
  grep "K=  1152  x  9\|K=  4608  x  2\|K=  9216" $O/c_bench_s$1_w$2.txt
done

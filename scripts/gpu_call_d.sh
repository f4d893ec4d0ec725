#!/bin/bash
# round 3, call D: conv_ws2 (4x16 tiles, LDS-DMA residual), stamps, A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "weight_stationary or split_independent" > $O/d_ops.log 2>&1; echo "ops rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/d_ops.log | tail -25
timeout 200 python scripts/ws2_profile.py > $O/d_ws2_stamps.txt 2>&1; cat $O/d_ws2_stamps.txt | grep -v amdgpu.ids | head -40
for CFG in "2 1 0" "2 2 0" "2 1 1"; do set -- $CFG
  VT_CONV_SCHED=$1 VT_CONV_WS=$2 VT_CONV_X_NT=$3 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --breakdown > $O/d_bench_s$1_w$2_nt$3.json 2> $O/d_bench_s$1_w$2_nt$3.txt
  echo "sched=$1 ws=$2 x_nt=$3: $(python -c "import json,sys; d=json.load(open('$O/d_bench_s$1_w$2_nt$3.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1 | tail -1)"
  grep "K=  1152  x  9\|K=  4608  x  2\|K=  9216" $O/d_bench_s$1_w$2_nt$3.txt
done

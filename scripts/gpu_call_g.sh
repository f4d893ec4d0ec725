#!/bin/bash
# round 3, call G: A/B of the K-step schedules after the loop clean-up
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv" > $O/g_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/g_ops.log
for CFG in "1 2 2" "2 2 2" "0 2 2"; do set -- $CFG
  VT_CONV_SCHED=$1 VT_CONV_WS=$2 VT_TBLOCK_FUSED=$3 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --breakdown > $O/g_bench_s$1_w$2_t$3.json 2> $O/g_bench_s$1_w$2_t$3.txt
  echo "sched=$1 ws=$2 tblock=$3: $(python -c "import json,sys; d=json.load(open('$O/g_bench_s$1_w$2_t$3.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1 | tail -1)"
  grep "K=  4608  x  2\|K=  9216  x  2\|K=  2304  x  3\|K= 13824  x  9" $O/g_bench_s$1_w$2_t$3.txt
done

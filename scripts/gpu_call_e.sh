#!/bin/bash
# round 3, call E: conv_ws2 after the issue-slot diet (tile cursor, bit-formula pixel map, buffer stores, no pins)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "weight_stationary or split_independent" --count 1 > $O/e_ops.log 2>&1 || timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "weight_stationary or split_independent" > $O/e_ops.log 2>&1; echo "ops rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/e_ops.log | tail -5
for i in 1 2 3 4 5 6; do timeout 200 python scripts/ws2_debug.py 2>&1 | grep "mismatches" | tr '\n' ';'; echo; done
timeout 200 python scripts/ws2_profile.py > $O/e_ws2_stamps.txt 2>&1
for CFG in "2 2"; do set -- $CFG
  VT_CONV_SCHED=$1 VT_CONV_WS=$2 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --breakdown > $O/e_bench_s$1_w$2.json 2> $O/e_bench_s$1_w$2.txt
  echo "sched=$1 ws=$2: $(python -c "import json,sys; d=json.load(open('$O/e_bench_s$1_w$2.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1 | tail -1)"
  grep "K=  1152  x  9\|K=  4608  x  2\|K=  9216" $O/e_bench_s$1_w$2.txt
done

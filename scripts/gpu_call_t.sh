#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for sc in 1 0; do echo "== VT_CONV_SCHED=$sc"; VT_CONV_SCHED=$sc timeout 120 python scripts/conv_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2t_convprof_s$sc.txt; grep -A5 "average step" gpurun_out/r2t_convprof_s$sc.txt; done
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "forced_256 or conv_large or pointer" > gpurun_out/r2t_ops.log 2>&1; echo "ops rc=$?"; grep -v amdgpu.ids gpurun_out/r2t_ops.log | tail -5
for sc in 1 0; do echo "== VT_CONV_SCHED=$sc"; VT_CONV_SCHED=$sc timeout 200 python scripts/conv_microbench.py 2>&1 | grep -v amdgpu.ids | grep "L1\|L2\|L3\|mid\|dec\|total"; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SCHED=1:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
VT_CONV_SCHED=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SCHED=0:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"

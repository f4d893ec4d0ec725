#!/bin/bash
# round 3, call H: chunk state inside the fused temporal block (v1.1 tiling), tiled end-to-end cases, tiled throughput
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "temporal_block or split_independent" > $O/h_ops.log 2>&1; echo "ops rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/h_ops.log | tail -12
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x -k "tiled or v11 or temporal_blocks" > $O/h_e2e.log 2>&1; echo "e2e rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/h_e2e.log | tail -12
for F in 1 0; do
  VT_TBLOCK_FUSED=$F timeout 600 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --traffic none > $O/h_bench_cfg4_f$F.json 2> $O/h_bench_cfg4_f$F.txt; echo "cfg4 tblock_fused=$F rc=$?: $(python -c "import json; d=json.load(open('$O/h_bench_cfg4_f$F.json')); print(d['value'], d['ms_per_step'], d['config']['workload'])" 2>&1 | tail -1)"
done

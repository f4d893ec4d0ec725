#!/bin/bash
# round 3, call A: new operator variants (K-step schedule 2, LN256 epilogue v1, option switches) + A/B timing
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x -m gpu > $O/a_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/a_ops.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "graph_cache or tiled_chunks or fused_and_unfused or matches_cpu_oracle or golden" > $O/a_e2e.log 2>&1; echo "e2e rc=$?"; tail -5 $O/a_e2e.log
timeout 600 python -m pytest tests/test_video_io.py tests/test_metrics.py -q -m gpu > $O/a_vio.log 2>&1; echo "vio rc=$?"; tail -3 $O/a_vio.log
for S in 1 2; do for V in 0 1; do
  VT_CONV_SCHED=$S VT_CONV_LN256_V=$V timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --breakdown > $O/a_bench_s${S}_v${V}.json 2> $O/a_bench_s${S}_v${V}.txt
  echo "sched=$S ln256_v=$V: $(python -c "import json,sys; d=json.load(open('$O/a_bench_s${S}_v${V}.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1 | tail -1)"
done; done
VT_CONV_SCHED=2 VT_CONV_LN256_V=1 VT_CONV_TILE_MIN=64 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none > $O/a_bench_tm64.json 2>/dev/null; python -c "import json; d=json.load(open('$O/a_bench_tm64.json')); print('tile_min 64:', d['value'])"

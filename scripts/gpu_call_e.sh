#!/bin/bash
# round-2 GPU call E: conv_ws128 with the conflict-free lane->pixel mapping; accumulator placement A/B; video I/O loop test
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_video_io.py -m gpu -q -x -k "weight_stationary or temporal_block or reconstruction" > gpurun_out/r2e_ops.log 2>&1; echo "ops rc=$?"; tail -4 gpurun_out/r2e_ops.log
VT_WS_ACC=0 timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "weight_stationary or temporal_block" > gpurun_out/r2e_ops_acc0.log 2>&1; echo "ops(acc0) rc=$?"; tail -2 gpurun_out/r2e_ops_acc0.log
for acc in 1 0; do for ln in 1 0; do
  echo "VT_WS_ACC=$acc MB_LN=$ln"; VT_WS_ACC=$acc VT_CONV_WS=1 MB_LN=$ln MB_ONLY="L0 spatial" timeout 150 python scripts/conv_microbench.py 2>&1 | grep "L0 spatial\|tblock" | tee -a gpurun_out/r2e_mb.log
done; done
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err; echo "bench rc=$?"; cat gpurun_out/r2e_bench.json; grep -v amdgpu.ids gpurun_out/r2e_bench.err | head -8

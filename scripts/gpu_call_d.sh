#!/bin/bash
# round-2 GPU call D: conv_ws128 v3 (row phase sliced into the MFMA shadows), video I/O kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_video_io.py -m gpu -q -x -s -k "weight_stationary or video or frames or reconstruction" > gpurun_out/r2d_ops.log 2>&1; echo "ops rc=$?"; grep -v "amdgpu.ids" gpurun_out/r2d_ops.log | tail -15
VT_CONV_WS=1 MB_LN=1 MB_ONLY="L0 spatial" timeout 150 python scripts/conv_microbench.py 2>&1 | grep "L0 spatial" | tee gpurun_out/r2d_mb.log
VT_CONV_WS=1 MB_LN=0 MB_ONLY="L0 spatial" timeout 150 python scripts/conv_microbench.py 2>&1 | grep "L0 spatial" | tee -a gpurun_out/r2d_mb.log
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "golden or (matches_cpu_oracle and 488_4chn)" > gpurun_out/r2d_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 gpurun_out/r2d_e2e.log
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err; echo "bench rc=$?"; cat gpurun_out/r2d_bench.json; grep -v amdgpu.ids gpurun_out/r2d_bench.err | head -8

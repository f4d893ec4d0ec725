#!/bin/bash
# round-2 GPU call I: fused temporal block v2 (OUT / LN1 row jobs sliced into the GEMMs' MFMA shadows)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "temporal_block or weight_stationary" > gpurun_out/r2i_ops.log 2>&1; echo "ops rc=$?"; tail -4 gpurun_out/r2i_ops.log
MB_ONLY="tblock" timeout 150 python scripts/conv_microbench.py 2>&1 | grep "tblock" | tee gpurun_out/r2i_mb.log
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "golden or (matches_cpu_oracle and 488_4chn) or graph_cache" > gpurun_out/r2i_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 gpurun_out/r2i_e2e.log
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err; echo "bench rc=$?"; cat gpurun_out/r2i_bench.json; grep -v amdgpu.ids gpurun_out/r2i_bench.err | head -6

#!/bin/bash
# round-2 GPU call C: weight-stationary kernel v2 (row phase pipelined under the next tile's K loop), engine graph cache, fixed bf16 gates
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "weight_stationary or temporal_block" > gpurun_out/r2c_ops.log 2>&1; echo "ops rc=$?"; tail -15 gpurun_out/r2c_ops.log
VT_CONV_WS=1 MB_LN=1 MB_ONLY="L0 spatial" timeout 150 python scripts/conv_microbench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c_mb.log
VT_CONV_WS=1 MB_LN=0 MB_ONLY="L0 spatial" timeout 150 python scripts/conv_microbench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2c_mb.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -s -k "graph_cache or autocast or (matches_cpu_oracle and 32768) or (matches_cpu_oracle and 488_4chn) or golden" > gpurun_out/r2c_e2e.log 2>&1; echo "e2e rc=$?"; grep -v MIOpen gpurun_out/r2c_e2e.log | tail -25
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err; echo "bench rc=$?"; cat gpurun_out/r2c_bench.json; grep -v amdgpu.ids gpurun_out/r2c_bench.err | head -16

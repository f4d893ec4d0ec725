#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "narrow or avgpool or test_conv[" > gpurun_out/r2o_ops.log 2>&1; echo "ops rc=$?"; grep -v amdgpu.ids gpurun_out/r2o_ops.log | tail -25
for nw in 1 0; do VT_CONV_NARROW=$nw MB_ONLY=conv_out timeout 120 python scripts/conv_microbench.py 2>&1 | grep -v amdgpu.ids | tail -3; done
timeout 400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "rare or tiled_chunks or graph_cache or golden" > gpurun_out/r2o_e2e.log 2>&1; echo "e2e rc=$?"; grep -v amdgpu.ids gpurun_out/r2o_e2e.log | tail -25
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none > gpurun_out/r2o_bench.json 2> gpurun_out/r2o_bench.err; echo "bench rc=$?"; cat gpurun_out/r2o_bench.json; grep "N=   3" gpurun_out/r2o_bench.err
timeout 300 python scripts/other_configs_bench.py > gpurun_out/r2o_other.log 2>&1; grep -v amdgpu.ids gpurun_out/r2o_other.log | tail -6
timeout 200 python scripts/gemm_reference.py > gpurun_out/r2o_gemm.log 2>&1; grep -v amdgpu.ids gpurun_out/r2o_gemm.log | tail -12

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "temporal_block or split_independent" > gpurun_out/r2p_ops.log 2>&1; echo "ops rc=$?"; grep -v amdgpu.ids gpurun_out/r2p_ops.log | tail -15
MB_ONLY=tblock timeout 120 python scripts/conv_microbench.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none > gpurun_out/r2p_bench.json 2> gpurun_out/r2p_bench.err; echo "bench rc=$?"; cat gpurun_out/r2p_bench.json; grep "K=   768" gpurun_out/r2p_bench.err

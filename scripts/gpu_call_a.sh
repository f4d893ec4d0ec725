#!/bin/bash
# round-2 GPU call A: new parity tests + bench line (see gpurun_out/r2a_*)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > gpurun_out/r2a_ops.log 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/r2a_ops.log
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_metrics.py -m gpu -q -s > gpurun_out/r2a_e2e.log 2>&1; echo "e2e rc=$?"; tail -5 gpurun_out/r2a_e2e.log
timeout 600 python bench.py --steps 10 --warmup 3 --breakdown > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err; echo "bench rc=$?"; cat gpurun_out/r2a_bench.json; tail -45 gpurun_out/r2a_bench.err

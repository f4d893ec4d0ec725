#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 python scripts/r5_streams.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5c_streams.log 2>&1; cat gpurun_out/r5c_streams.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5c_bench.json 2> gpurun_out/r5c_bench.err; echo "bench rc=$?"; cat gpurun_out/r5c_bench.json; tail -5 gpurun_out/r5c_bench.err

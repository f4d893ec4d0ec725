#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 700 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "tap_skip or split_k or flash_attention" 2>&1 | grep -v "amdgpu.ids" | tail -8 ) > gpurun_out/r5a_ops.log 2>&1
( timeout 700 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "autocast or golden or (matches_cpu_oracle and fsq)" 2>&1 | grep -v "amdgpu.ids" | tail -8 ) > gpurun_out/r5a_e2e.log 2>&1
( timeout 500 python -m pytest tests/test_capi.py -q -x -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -8 ) > gpurun_out/r5a_capi.log 2>&1
( timeout 400 python scripts/r5_ab.py 2 2>&1 | grep -v "amdgpu.ids" ) > gpurun_out/r5a_ab.log 2>&1
tail -4 gpurun_out/r5a_ops.log gpurun_out/r5a_e2e.log gpurun_out/r5a_capi.log; cat gpurun_out/r5a_ab.log

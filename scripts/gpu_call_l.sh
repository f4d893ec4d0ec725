#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python scripts/determinism_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2l_det.log

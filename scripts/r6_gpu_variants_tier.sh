#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m "gpu and variants" --durations=10 --timeout 1200 > gpurun_out/r6s_variants.log 2>&1; echo "variants rc=$?"; grep -v "MIOpen\|amdgpu.ids" gpurun_out/r6s_variants.log | tail -18 | cut -c1-200

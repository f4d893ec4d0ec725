#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python scripts/determinism_check.py 2>&1 | grep "tblock" | tee gpurun_out/r2m_det.log
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "split_independent or gather or temporal_block or weight_stationary" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "full_size_properties or rare or golden" 2>&1 | tail -4
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/r2m_bench.json 2> gpurun_out/r2m_bench.err; echo "bench rc=$?"; cat gpurun_out/r2m_bench.json

#!/bin/bash
# round-2 GPU call K: FSQ aux kernel rewrite, v1.1 chunk assembly without torch.cat, other-configs timing
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_video_io.py -m gpu -q -x -k "fsq or gather or lerp or frames or reconstruction or copy" > gpurun_out/r2k_ops.log 2>&1; echo "ops rc=$?"; tail -4 gpurun_out/r2k_ops.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "golden or v11_long_video_tiled or (matches_cpu_oracle and v1_1) or (matches_cpu_oracle and 32768) or graph_cache" > gpurun_out/r2k_e2e.log 2>&1; echo "e2e rc=$?"; tail -4 gpurun_out/r2k_e2e.log
timeout 400 python scripts/other_configs_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2k_other.log

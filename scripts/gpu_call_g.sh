#!/bin/bash
# round-2 GPU call G: two-sweep row phase (no loads and stores of a wave in flight together), tblock prefetch fence
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_ws2
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_video_io.py -m gpu -q -x -k "weight_stationary or temporal_block or reconstruction" > gpurun_out/r2g_ops.log 2>&1; echo "ops rc=$?"; tail -4 gpurun_out/r2g_ops.log
for ln in 1 0; do
  echo "MB_LN=$ln"; VT_CONV_WS=1 MB_LN=$ln MB_ONLY="L0 spatial" timeout 150 python scripts/conv_microbench.py 2>&1 | grep "L0 spatial\|tblock" | tee -a gpurun_out/r2g_mb.log
done
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err; echo "bench rc=$?"; cat gpurun_out/r2g_bench.json; grep -v amdgpu.ids gpurun_out/r2g_bench.err | head -8
export VT_CONV_WS=1 MB_LN=0
bash scripts/pmc_conv2.sh "L0 spatial" $PWD/gpurun_out/pmc_ws2
python scripts/pmc_summary.py gpurun_out/pmc_ws2 | tee gpurun_out/pmc_ws2/summary.txt

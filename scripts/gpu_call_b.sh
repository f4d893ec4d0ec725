#!/bin/bash
# round-2 GPU call B: weight-stationary 3x3 kernel + fused temporal block: parity, micro-benchmarks, model-level check, bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "weight_stationary or temporal_block or fsq_aux" > gpurun_out/r2b_ops.log 2>&1; echo "ops rc=$?"; tail -15 gpurun_out/r2b_ops.log
for ws in 1 0; do
  VT_CONV_WS=$ws MB_LN=1 MB_ONLY="L0" timeout 150 python scripts/conv_microbench.py > gpurun_out/r2b_mb_ws$ws.log 2>&1; cat gpurun_out/r2b_mb_ws$ws.log
done
timeout 420 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "golden or (matches_cpu_oracle and 488_4chn) or (matches_cpu_oracle and 32768)" > gpurun_out/r2b_e2e.log 2>&1; echo "e2e rc=$?"; tail -8 gpurun_out/r2b_e2e.log
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err; echo "bench rc=$?"; cat gpurun_out/r2b_bench.json; tail -45 gpurun_out/r2b_bench.err
VT_CONV_WS=0 VIDTOK_AMD_FUSE_TBLOCK=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2b_bench_old.json 2> gpurun_out/r2b_bench_old.err; echo "bench(old kernels) rc=$?"; cat gpurun_out/r2b_bench_old.json

#!/bin/bash
# what profiles/rNN_bench_bf16{.json,_conv_breakdown.txt,_kernel_stats.md} are made from: bench line with live PMC traffic +
# CPU baseline, per-layer breakdown, rocprofv3 kernel table (run on the GPU box: gpurun -- bash scripts/collect_profiles.sh)
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2z_bench_bf16.json 2> gpurun_out/r2z_bench_bf16.err; echo "bench bf16 rc=$?"; cat gpurun_out/r2z_bench_bf16.json
timeout 200 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none > /dev/null 2> gpurun_out/r2z_breakdown_bf16.txt; echo "breakdown rc=$?"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2z_prof -o bench -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --traffic none > $R/gpurun_out/r2z_prof.log 2>&1); echo "rocprof rc=$?"
DB=$(find gpurun_out/r2z_prof -name "*.db" | head -1); rm -f gpurun_out/r2z_kernel_stats.md; python scripts/rocprof_summary.py "$DB" gpurun_out/r2z_kernel_stats.md; head -9 gpurun_out/r2z_kernel_stats.md | cut -c1-160
find gpurun_out/r2z_prof -name "*.db" -delete

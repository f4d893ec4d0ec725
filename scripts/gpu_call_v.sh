#!/bin/bash
# round-2 final profile collection: bench lines (bf16 with live PMC traffic + CPU baseline, fp32), per-layer breakdown,
# rocprofv3 kernel table, the other BASELINE configs
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2v_bench_bf16.json 2> gpurun_out/r2v_bench_bf16.err; echo "bench bf16 rc=$?"; cat gpurun_out/r2v_bench_bf16.json
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none > /dev/null 2> gpurun_out/r2v_breakdown_bf16.txt; echo "breakdown rc=$?"
timeout 400 python bench.py --dtype fp32 --steps 5 --warmup 2 --breakdown --no-cpu-baseline --traffic none > gpurun_out/r2v_bench_fp32.json 2> gpurun_out/r2v_breakdown_fp32.txt; echo "bench fp32 rc=$?"; cat gpurun_out/r2v_bench_fp32.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2v_prof -o bench -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --traffic none > $R/gpurun_out/r2v_prof.log 2>&1); echo "rocprof rc=$?"
DB=$(find gpurun_out/r2v_prof -name "*.db" | head -1); echo "db=$DB"; rm -f gpurun_out/r2v_kernel_stats.md; python scripts/rocprof_summary.py "$DB" gpurun_out/r2v_kernel_stats.md; head -12 gpurun_out/r2v_kernel_stats.md | cut -c1-200
find gpurun_out/r2v_prof -name "*.db" -size +30M -delete
timeout 400 python scripts/other_configs_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2v_other.txt; cat gpurun_out/r2v_other.txt

#!/bin/bash
export TMPDIR=/tmp
P=r05
O=gpurun_out
R=/root/repo
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "golden or handle_matches or graph_cache or full_size or tiled" 2>&1 | grep -v amdgpu.ids | tail -5 ) > $O/r5i_e2e.log 2>&1; tail -3 $O/r5i_e2e.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${P}_bench_bf16.json 2> $O/${P}_bench_bf16.err; echo "bench bf16 rc=$?"; cut -c1-330 $O/${P}_bench_bf16.json
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --no-extras > /dev/null 2> $O/${P}_bench_bf16_conv_breakdown.txt; echo "breakdown rc=$?"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/${P}_prof -o bench -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --traffic none --no-extras > $R/$O/${P}_prof.log 2>&1); echo "rocprof rc=$?"
DB=$(find $O/${P}_prof -name "*.db" | head -1); rm -f $O/${P}_bench_bf16_kernel_stats.md; python scripts/rocprof_summary.py "$DB" $O/${P}_bench_bf16_kernel_stats.md; head -12 $O/${P}_bench_bf16_kernel_stats.md | cut -c1-150
rm -rf $O/${P}_prof

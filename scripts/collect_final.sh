#!/bin/bash
# round-end subset of collect_profiles.sh (the kernels of the benchmark path are unchanged since the full r04 collection: the counter /
# stamp files stay): the three bench lines, per-group breakdowns, rocprofv3 kernel tables, smoke log
export TMPDIR=/tmp
P=${1:-r04}
mkdir -p gpurun_out
O=gpurun_out
R=/root/repo
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${P}_bench_bf16.json 2> $O/${P}_bench_bf16.err; echo "bench bf16 rc=$?"; cut -c1-300 $O/${P}_bench_bf16.json
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --no-extras > /dev/null 2> $O/${P}_bench_bf16_conv_breakdown.txt; echo "breakdown rc=$?"
timeout 400 python bench.py --dtype fp32 --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none --no-extras > $O/${P}_bench_fp32.json 2> $O/${P}_bench_fp32_conv_breakdown.txt; echo "fp32 rc=$?"; cut -c1-200 $O/${P}_bench_fp32.json
timeout 600 python bench.py --dtype bf16x3 --steps 10 --warmup 3 --breakdown --no-cpu-baseline --no-extras > $O/${P}_bench_bf16x3.json 2> $O/${P}_bench_bf16x3_conv_breakdown.txt; echo "bf16x3 rc=$?"; cut -c1-200 $O/${P}_bench_bf16x3.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/${P}_prof_x3 -o bench -- python $R/bench.py --dtype bf16x3 --steps 3 --warmup 3 --no-cpu-baseline --traffic none --no-extras > $R/$O/${P}_prof_x3.log 2>&1); echo "rocprof x3 rc=$?"
DB=$(find $O/${P}_prof_x3 -name "*.db" | head -1); rm -f $O/${P}_bench_bf16x3_kernel_stats.md; python scripts/rocprof_summary.py "$DB" $O/${P}_bench_bf16x3_kernel_stats.md; head -6 $O/${P}_bench_bf16x3_kernel_stats.md | cut -c1-160
rm -rf $O/${P}_prof_x3
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/${P}_prof -o bench -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --traffic none --no-extras > $R/$O/${P}_prof.log 2>&1); echo "rocprof rc=$?"
DB=$(find $O/${P}_prof -name "*.db" | head -1); rm -f $O/${P}_bench_bf16_kernel_stats.md; python scripts/rocprof_summary.py "$DB" $O/${P}_bench_bf16_kernel_stats.md; head -10 $O/${P}_bench_bf16_kernel_stats.md | cut -c1-160
rm -rf $O/${P}_prof
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/${P}_smoke.log; tail -12 $O/${P}_smoke.log

#!/bin/bash
export TMPDIR=/tmp
P=r05
O=gpurun_out
R=/root/repo
mkdir -p $O
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/${P}_prof -o bench -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --traffic none --no-extras > $R/$O/${P}_prof.log 2>&1); echo "rocprof rc=$?"
DB=$(find $O/${P}_prof -name "*.db" | head -1); rm -f $O/${P}_bench_bf16_kernel_stats.md; python scripts/rocprof_summary.py "$DB" $O/${P}_bench_bf16_kernel_stats.md; cut -c1-150 $O/${P}_bench_bf16_kernel_stats.md
rm -rf $O/${P}_prof
tail -2 $O/${P}_prof.log | cut -c1-300
timeout 2000 python -m pytest tests/ -q -m gpu > $O/r5g_full_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r5g_full_gpu.log | tail -6

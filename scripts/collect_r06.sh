#!/bin/bash
# what profiles/r06_* are made from (GPU box: gpurun -- bash scripts/collect_r06.sh): the bench line with live PMC traffic, CPU baseline sweep and
# the extras; per-group breakdowns; the fp32 and split-bf16 lines; rocprofv3 kernel tables (bench step, tiled configs[4] pass); HBM-side traffic and
# SQ counters (separate --pmc passes, counters only with --kernel-trace); cycle stamps of the fused temporal block; smoke log
export TMPDIR=/tmp
P=r06
mkdir -p gpurun_out
O=gpurun_out
R=/root/repo
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${P}_bench_bf16.json 2> $O/${P}_bench_bf16.err; echo "bench bf16 rc=$?"; cut -c1-400 $O/${P}_bench_bf16.json
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --no-extras > /dev/null 2> $O/${P}_bench_bf16_conv_breakdown.txt; echo "breakdown rc=$?"
timeout 400 python bench.py --dtype fp32 --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none --no-extras > $O/${P}_bench_fp32.json 2> $O/${P}_bench_fp32_conv_breakdown.txt; echo "fp32 rc=$?"; cut -c1-200 $O/${P}_bench_fp32.json
timeout 600 python bench.py --dtype bf16x3 --steps 10 --warmup 3 --breakdown --no-cpu-baseline --no-extras > $O/${P}_bench_bf16x3.json 2> $O/${P}_bench_bf16x3_conv_breakdown.txt; echo "bf16x3 rc=$?"; cut -c1-200 $O/${P}_bench_bf16x3.json
timeout 400 python bench.py --dtype fp16 --steps 20 --warmup 5 --breakdown --no-cpu-baseline --no-extras > $O/${P}_bench_fp16.json 2> $O/${P}_bench_fp16_conv_breakdown.txt; echo "fp16 rc=$?"; cut -c1-200 $O/${P}_bench_fp16.json
# (the activation restream A/B of VERDICT r5 #4: scripts/r6_gpu_c.sh -> profiles/r06_activation_restream_ab.txt)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/${P}_prof -o bench -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --traffic none --no-extras > $R/$O/${P}_prof.log 2>&1); echo "rocprof rc=$?"
DB=$(find $O/${P}_prof -name "*.db" | head -1); rm -f $O/${P}_bench_bf16_kernel_stats.md; python scripts/rocprof_summary.py "$DB" $O/${P}_bench_bf16_kernel_stats.md; head -30 $O/${P}_bench_bf16_kernel_stats.md | cut -c1-170
rm -rf $O/${P}_prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/${P}_prof_t -o tiled -- python $R/scripts/tiled_pass.py 3 > $R/$O/${P}_prof_t.log 2>&1); echo "rocprof tiled rc=$?"
DB=$(find $O/${P}_prof_t -name "*.db" | head -1); rm -f $O/${P}_tiled_kernel_stats.md; python scripts/rocprof_summary.py "$DB" $O/${P}_tiled_kernel_stats.md; head -8 $O/${P}_tiled_kernel_stats.md | cut -c1-170
rm -rf $O/${P}_prof_t
bash scripts/pmc_bench.sh $R/$O/${P}_pmc_traffic bf16; python scripts/pmc_traffic.py $O/${P}_pmc_traffic $O/${P}_conv_traffic_pmc.json 2>&1 | tail -3
cd /tmp
for pass in "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  set -- $pass; name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/${P}_sq/$name -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --traffic none --no-extras > $R/$O/${P}_sq_$name.log 2>&1; echo "$name rc=$?"
done
cd $R
python scripts/pmc_summary.py $O/${P}_sq > $O/${P}_bench_bf16_sq_pmc.txt 2>&1; head -30 $O/${P}_bench_bf16_sq_pmc.txt
timeout 200 python scripts/tblock_profile.py > $O/${P}_tblock_pair_phase_cycles.txt 2>&1; grep "per launch" $O/${P}_tblock_pair_phase_cycles.txt
timeout 200 python scripts/ws2_profile.py > $O/${P}_ws2_iteration_cycles.txt 2>&1; head -4 $O/${P}_ws2_iteration_cycles.txt | cut -c1-200
timeout 120 python scripts/c128_time.py bf16 > $O/${P}_c128_time.txt 2>&1; timeout 120 python scripts/c128_time.py f16 >> $O/${P}_c128_time.txt 2>&1; cat $O/${P}_c128_time.txt
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/${P}_smoke.log; tail -14 $O/${P}_smoke.log
rm -rf $O/${P}_pmc_traffic/*/*.db $O/${P}_sq/*/*.db

#!/bin/bash
# what profiles/rNN_* are made from (run on the GPU box: gpurun -- bash scripts/collect_profiles.sh rNN):
#   bench line with live PMC traffic + CPU baseline, per-layer breakdown, rocprofv3 kernel table, fp32 line, HBM traffic per
#   launch group, SQ counters of the bench run, cycle stamps of the three persistent / tiled kernels, micro-benchmarks
export TMPDIR=/tmp
P=${1:-r04}
mkdir -p gpurun_out
O=gpurun_out
R=/root/repo
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${P}_bench_bf16.json 2> $O/${P}_bench_bf16.err; echo "bench bf16 rc=$?"; cut -c1-400 $O/${P}_bench_bf16.json
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --no-extras > /dev/null 2> $O/${P}_bench_bf16_conv_breakdown.txt; echo "breakdown rc=$?"
timeout 400 python bench.py --dtype fp32 --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none --no-extras > $O/${P}_bench_fp32.json 2> $O/${P}_bench_fp32_conv_breakdown.txt; echo "fp32 rc=$?"; cut -c1-300 $O/${P}_bench_fp32.json
# split-bf16 mode ("bf16x3"): bench line with live PMC traffic, per-layer breakdown, launch time of the 8-wave tile by what is left in the K step
timeout 600 python bench.py --dtype bf16x3 --steps 10 --warmup 3 --breakdown --no-cpu-baseline --no-extras > $O/${P}_bench_bf16x3.json 2> $O/${P}_bench_bf16x3_conv_breakdown.txt; echo "bf16x3 rc=$?"; cut -c1-300 $O/${P}_bench_bf16x3.json
MODE=bf16x3 timeout 200 python scripts/conv_profile.py > $O/${P}_igemm_step_cycles_bf16x3.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/${P}_prof_x3 -o bench -- python $R/bench.py --dtype bf16x3 --steps 3 --warmup 3 --no-cpu-baseline --traffic none --no-extras > $R/$O/${P}_prof_x3.log 2>&1); echo "rocprof x3 rc=$?"
DB=$(find $O/${P}_prof_x3 -name "*.db" | head -1); rm -f $O/${P}_bench_bf16x3_kernel_stats.md; python scripts/rocprof_summary.py "$DB" $O/${P}_bench_bf16x3_kernel_stats.md; head -8 $O/${P}_bench_bf16x3_kernel_stats.md | cut -c1-160
rm -rf $O/${P}_prof_x3
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/${P}_prof -o bench -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --traffic none --no-extras > $R/$O/${P}_prof.log 2>&1); echo "rocprof rc=$?"
DB=$(find $O/${P}_prof -name "*.db" | head -1); rm -f $O/${P}_bench_bf16_kernel_stats.md; python scripts/rocprof_summary.py "$DB" $O/${P}_bench_bf16_kernel_stats.md; head -12 $O/${P}_bench_bf16_kernel_stats.md | cut -c1-160
find $O/${P}_prof -name "*.db" -delete
# HBM-side traffic per launch (two PMC passes) and SQ counters (two passes), counters only with --kernel-trace
bash scripts/pmc_bench.sh $R/$O/${P}_pmc_traffic bf16; python scripts/pmc_traffic.py $O/${P}_pmc_traffic $O/${P}_conv_traffic_pmc.json 2>&1 | tail -3
cd /tmp
for pass in "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  set -- $pass; name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/${P}_sq/$name -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --traffic none --no-extras > $R/$O/${P}_sq_$name.log 2>&1; echo "$name rc=$?"
done
cd $R
python scripts/pmc_summary.py $O/${P}_sq > $O/${P}_bench_bf16_sq_pmc.txt 2>&1; head -40 $O/${P}_bench_bf16_sq_pmc.txt
# cycle stamps
timeout 200 python scripts/ws2_profile.py > $O/${P}_ws2_iteration_cycles.txt 2>&1
timeout 200 python scripts/tblock_profile.py > $O/${P}_tblock_pair_phase_cycles.txt 2>&1; grep "per launch" $O/${P}_tblock_pair_phase_cycles.txt
for s in 1 2; do VT_CONV_SCHED=$s timeout 200 python scripts/conv_profile.py > $O/${P}_igemm_step_cycles_sched$s.txt 2>&1; done
python scripts/x3_accuracy.py vidtok_fsq_causal_488_32768 17 256 1 > $O/${P}_bf16x3_accuracy.txt 2>&1; python scripts/x3_accuracy.py vidtok_kl_causal_488_4chn 17 256 1 >> $O/${P}_bf16x3_accuracy.txt 2>&1
# micro-benchmarks (standalone binaries built in the container: hipcc --offload-arch=gfx950 -O3 -o scripts/build/<name> scripts/<name>.hip)
for b in mfma_phase_bench mfma_tile_bench hbm_bw_bench; do [ -x scripts/build/$b ] && timeout 120 scripts/build/$b > $O/${P}_$b.txt 2>&1; done
rm -rf $O/${P}_prof $O/${P}_pmc_traffic/*/*.db $O/${P}_sq/*/*.db

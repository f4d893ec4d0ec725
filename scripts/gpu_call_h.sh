#!/bin/bash
# round-2 GPU call H: the numbers and profiles that get committed: bench lines (bf16 with cpu baseline, fp32), rocprofv3 kernel
# trace of the bench command, PMC traffic passes, SQ counters of one bench pass
export TMPDIR=/tmp
mkdir -p gpurun_out/final_r2
O=$PWD/gpurun_out/final_r2
timeout 600 python bench.py --steps 10 --warmup 3 --breakdown > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?"; cat $O/bench_bf16.json
timeout 600 python bench.py --steps 5 --warmup 3 --dtype fp32 --breakdown --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench fp32 rc=$?"; cat $O/bench_fp32.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bench -- python /root/repo/bench.py --steps 3 --warmup 3 --no-cpu-baseline --traffic none > $O/prof_bf16.log 2>&1); echo "rocprof rc=$?"
ls $O/prof_bf16 | head; DB=$(find $O/prof_bf16 -name "*.db" | head -1); [ -n "$DB" ] && python scripts/rocprof_summary.py $DB $O/kernel_stats.md && head -12 $O/kernel_stats.md
bash scripts/pmc_bench.sh $O/pmc_traffic bf16; python scripts/pmc_traffic.py $O/pmc_traffic $O/conv_traffic_pmc.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq/sq1 -o p -- python /root/repo/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-graph --traffic none > $O/pmc_sq1.log 2>&1); python scripts/pmc_summary.py $O/pmc_sq | tee $O/pmc_sq_summary.txt
# keep the merge small: drop raw traces
find $O -name "*.db" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +5M -delete; find $O -name "*counter_collection.csv" -size +8M -delete; du -sh $O

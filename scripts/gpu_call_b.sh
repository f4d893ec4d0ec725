#!/bin/bash
# round 3, call B: schedule 3, cycle stamps of schedules 1-3, per-layer traffic table
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "8wave_schedules or ln256" > $O/b_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/b_ops.log
timeout 300 python -m pytest tests/test_video_io.py -q -m gpu -k "replays_reference" > $O/b_vio.log 2>&1; echo "vio rc=$?"; tail -3 $O/b_vio.log
for S in 1 2 3; do VT_CONV_SCHED=$S timeout 200 python scripts/conv_profile.py > $O/b_stamps_s$S.txt 2>&1; echo "stamps sched $S:"; grep -A8 "average step" $O/b_stamps_s$S.txt | head -24; done
for S in 2 3; do
  VT_CONV_SCHED=$S timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --breakdown > $O/b_bench_s${S}.json 2> $O/b_bench_s${S}.txt
  echo "sched=$S: $(python -c "import json,sys; d=json.load(open('$O/b_bench_s${S}.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1 | tail -1)"
done
VT_CONV_SCHED=2 timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --traffic pmc --breakdown > $O/b_bench_traffic.json 2> $O/b_bench_traffic.txt; grep "traffic\|measured" $O/b_bench_traffic.txt | head -60
python -c "import json; d=json.load(open('$O/b_bench_traffic.json')); print(json.dumps(d['roofline']['hbm'], indent=0)[:3000]); print(d['roofline']['traffic'], d['roofline']['traffic_source'])"

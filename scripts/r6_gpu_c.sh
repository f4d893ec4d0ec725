#!/bin/bash
# round 6, third GPU call: the activation side of the traffic question (times, then shader clocks under the profiler), the end-to-end tests with durations, smoke
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
R=/root/repo
timeout 300 python scripts/activation_restream_ab.py 10 > $O/r06_activation_restream_ab.txt 2>&1; echo "restream A/B rc=$?"; grep -v amdgpu.ids $O/r06_activation_restream_ab.txt
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/$O/r06_ab_clock -o p -- python $R/scripts/activation_restream_ab.py 10 > $R/$O/r06_ab_clock.log 2>&1); echo "restream clock rc=$?"
python scripts/ab_clock_summary.py $O/r06_ab_clock 10 > $O/r06_ab_clock.txt 2>&1; cat $O/r06_ab_clock.txt | tail -25
find $O/r06_ab_clock -name "*.db" -delete; find $O/r06_ab_clock -name "*.csv" -size +2M -delete
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_capi.py tests/test_metrics.py tests/test_video_io.py tests/test_distributed_cpu.py -q -m gpu --durations=50 --timeout 900 > $O/r6c_e2e.log 2>&1; echo "e2e rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r6c_e2e.log | tail -75 | cut -c1-220
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/r6c_smoke.log; tail -14 $O/r6c_smoke.log

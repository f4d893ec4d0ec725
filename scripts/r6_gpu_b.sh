#!/bin/bash
# round 6, second GPU call: which fp16 e2e case stalled (verbose, per-test timeout), the loader-fed tile (tests, then its bench), durations of the operator tests
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 700 python -m pytest tests/test_gpu_e2e.py -v -s --timeout 300 -k "test_matches_cpu_oracle and (shape5 or shape6 or shape7 or shape8 or shape9)" > $O/r6b_e2e_f16.log 2>&1; echo "e2e f16 rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r6b_e2e_f16.log | grep -E "PASS|FAIL|rel|rate|usable|Timeout|Error|passed|failed" | tail -30 | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x --timeout 120 -k "tr256" > $O/r6b_tr256.log 2>&1; echo "tr256 tests rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r6b_tr256.log | tail -25 | cut -c1-220
timeout 400 python scripts/tr256_bench.py 10 > $O/r6b_tr256_bench.txt 2>&1; echo "tr256 bench rc=$?"; cat $O/r6b_tr256_bench.txt | grep -v amdgpu.ids
timeout 400 python scripts/tr256_bench.py 10 2 > $O/r6b_tr256o_bench.txt 2>&1; echo "tr256 overlapped bench rc=$?"; cat $O/r6b_tr256o_bench.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -m gpu --durations=60 --timeout 300 > $O/r6b_ops.log 2>&1; echo "ops rc=$?"; grep -v "MIOpen\|amdgpu.ids" $O/r6b_ops.log | tail -80 | cut -c1-200

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "rare" > gpurun_out/r2n_rare.log 2>&1; echo "rare rc=$?"; grep -v "amdgpu.ids" gpurun_out/r2n_rare.log | tail -30
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "test_conv" > gpurun_out/r2n_ops.log 2>&1; echo "ops rc=$?"; tail -8 gpurun_out/r2n_ops.log
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "golden or (matches_cpu_oracle and 488_4chn) or full_size_properties" > gpurun_out/r2n_e2e.log 2>&1; echo "e2e rc=$?"; tail -4 gpurun_out/r2n_e2e.log
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none > gpurun_out/r2n_bench.json 2> gpurun_out/r2n_bench.err; echo "bench rc=$?"; cat gpurun_out/r2n_bench.json; grep -v amdgpu.ids gpurun_out/r2n_bench.err | head -14
VT_CONV_FUSE_LN256=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no LN256 fusion:', d['value'], d['ms_per_step'])"

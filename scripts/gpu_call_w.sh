#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --traffic none 2> gpurun_out/r2w_bd.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TILE_MIN=128:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; grep "M=   20480\|M=   81920\|M=   10240\|M=   40960" gpurun_out/r2w_bd.txt
VT_CONV_TILE_MIN=384 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TILE_MIN=384:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "golden or full_size_properties or matches_cpu_oracle" 2>&1 | tail -3

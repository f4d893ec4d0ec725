#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V3=1:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
VT_TBLOCK_V3=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V3=0:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
timeout 400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "golden or full_size_properties or bf16 or graph_cache or num_codebooks" 2>&1 | tail -3

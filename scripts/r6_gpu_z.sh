#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "streaming" 2>&1 | tail -2
for rep in 1 2; do
for nt in 0 64 16; do
  VT_CONV_NT_MB=$nt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 conv_nt_mb=$nt', d['value'], d['ms_per_step'])"
done
done 2>&1 | tee $O/r06_nt_stores_ab.txt
for nt in 0 64; do
  VT_CONV_NT_MB=$nt timeout 300 python bench.py --dtype bf16x3 --steps 10 --warmup 3 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3 conv_nt_mb=$nt', d['value'], d['ms_per_step'])"
done 2>&1 | tee -a $O/r06_nt_stores_ab.txt
for nt in 0 64; do
  VT_CONV_NT_MB=$nt timeout 300 python bench.py --dtype fp16 --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16 conv_nt_mb=$nt', d['value'], d['ms_per_step'])"
done 2>&1 | tee -a $O/r06_nt_stores_ab.txt
for nt in 0 64; do echo -n "tiled conv_nt_mb=$nt: "; VT_CONV_NT_MB=$nt timeout 300 python scripts/tiled_pass.py 4 2>&1 | grep -v amdgpu.ids; done | tee -a $O/r06_nt_stores_ab.txt

#!/bin/bash
# round 6: cache policy of the K loop's LDS-DMA pieces: wnt = weight pieces nt (evict first), xnt = activation pieces nt
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for lib in vidtok_amd/libvidtok_amd.so ab_libs/libvidtok_amd_wnt.so ab_libs/libvidtok_amd_xnt.so; do
  VIDTOK_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done
done 2>&1 | tee gpurun_out/r06_dma_policy_ab.txt

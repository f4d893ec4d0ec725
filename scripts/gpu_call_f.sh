#!/bin/bash
# round-2 GPU call F: SQ counters of the weight-stationary kernels (where do the cycles of a tile go?)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_ws
export VT_CONV_WS=1 MB_LN=0
bash scripts/pmc_conv2.sh "L0 spatial" $PWD/gpurun_out/pmc_ws
python scripts/pmc_summary.py gpurun_out/pmc_ws | tee gpurun_out/pmc_ws/summary.txt

#!/bin/bash
# A/B of option conv_half256 (K bound of the 128 x 256 half-tile form) on the bench step, per launch group
mkdir -p gpurun_out
for k in 0 1024 1536 2304; do
  echo "=== VT_CONV_HALF256=$k" >> gpurun_out/half_ab.txt
  VT_CONV_HALF256=$k python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --traffic none --breakdown 2>&1 | grep -v "traffic per step\|measured " >> gpurun_out/half_ab.txt
done

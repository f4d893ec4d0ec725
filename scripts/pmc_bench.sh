#!/bin/bash
# HBM-side traffic of the conv kernel over one bench run: two rocprofv3 PMC passes (FETCH_SIZE costs 3 of
# the 4 TCC slots, WRITE_SIZE 2 -- they cannot share a pass), counters only with --kernel-trace.
# usage: scripts/pmc_bench.sh <outdir> [bf16|fp32]   (then scripts/pmc_traffic.py <outdir> profiles/rNN_conv_traffic.json)
set -u
OUT="$1"; DT="${2:-bf16}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -o p -- \
    python /root/repo/bench.py --dtype $DT --steps 2 --warmup 1 --no-cpu-baseline --no-graph --traffic none --no-extras > "$OUT/$c.log" 2>&1
done

#!/bin/bash
# two SQ counter passes over one layer of the conv micro-benchmark.  usage: scripts/pmc_conv2.sh "<MB_ONLY filter>" <outdir>
set -u
FILTER="$1"; OUT="$2"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  MB_ONLY="$FILTER" timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
    python /root/repo/scripts/conv_microbench.py > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU

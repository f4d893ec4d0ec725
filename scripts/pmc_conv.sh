#!/bin/bash
# PMC passes over the conv micro-benchmark (one rocprofv3 run per counter set; counters only with
# --kernel-trace, as the gpurun rules require).  usage: scripts/pmc_conv.sh "<MB_ONLY filter>" <outdir>
set -u
FILTER="$1"; OUT="$2"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  MB_ONLY="$FILTER" timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
    python /root/repo/scripts/conv_microbench.py > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM
run tcc1 TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE

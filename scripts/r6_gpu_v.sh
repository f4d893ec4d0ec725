#!/bin/bash
# counters of the memory path under the fused temporal block alone (where do its stores stall?)
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd /tmp
i=0
for pass in "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE" \
            "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY" \
            "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCP_TA_ADDR_STALL_CYCLES" \
            "TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/r06_tb_pmc/p$i -o p -- python $R/scripts/tblock_only.py > $O/r06_tb_pmc_$i.log 2>&1; echo "pass $i rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r06_tb_pmc/p*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tblock' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(f"{k:40s} mean per launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
find gpurun_out/r06_tb_pmc -name "*.db" -delete

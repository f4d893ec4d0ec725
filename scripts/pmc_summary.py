"""Summarise the rocprofv3 PMC CSVs written by scripts/pmc_conv.sh: per counter, the mean over the
dispatches of the conv kernel (counter values are per dispatch, summed over XCDs/SEs by rocprofv3)."""
import csv
import glob
import os
import sys
from collections import defaultdict


FAMILIES = ("conv_igemm", "conv3x3_ws2", "conv3x3_ws128", "conv3d_narrow", "conv_in8", "tblock_pair", "flash_attn", "layernorm_act")


def main(root):
    out = {}
    for f in sorted(glob.glob(os.path.join(root, "*", "p_counter_collection.csv"))):
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            fam = next((t for t in FAMILIES if t in kn), None)
            if fam is None:
                continue
            agg[(fam, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[f"{k[0]}:{k[1]}"] = (sum(v) / len(v), len(v))
    kt = glob.glob(os.path.join(root, "sq1", "p_kernel_trace.csv"))
    if kt:
        for fam in FAMILIES:
            d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if fam in r["Kernel_Name"]]
            if d:
                out[f"{fam}:kernel_ns(avg)"] = (sum(d) / len(d), len(d))
    for k in sorted(out):
        print(f"{k:52s} {out[k][0]:18.1f}   (n={out[k][1]})")
    return out


if __name__ == "__main__":
    main(sys.argv[1])

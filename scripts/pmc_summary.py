"""Summarise the rocprofv3 PMC CSVs written by scripts/pmc_conv.sh: per counter, the mean over the
dispatches of the conv kernel (counter values are per dispatch, summed over XCDs/SEs by rocprofv3)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    out = {}
    for f in sorted(glob.glob(os.path.join(root, "*", "p_counter_collection.csv"))):
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "conv_igemm" not in r["Kernel_Name"] and "conv_stream" not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = (sum(v) / len(v), len(v))
    kt = glob.glob(os.path.join(root, "sq1", "p_kernel_trace.csv"))
    if kt:
        d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))
             if "conv_igemm" in r["Kernel_Name"] or "conv_stream" in r["Kernel_Name"]]
        out["kernel_ns(avg)"] = (sum(d) / len(d), len(d))
    for k in sorted(out):
        print(f"{k:28s} {out[k][0]:18.1f}   (n={out[k][1]})")
    return out


if __name__ == "__main__":
    main(sys.argv[1])

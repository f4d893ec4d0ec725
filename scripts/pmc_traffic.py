"""Turn the two PMC passes of scripts/pmc_bench.sh into the per-launch HBM-side traffic of the conv kernel.

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived metrics) and count the L2's memory-side requests
(Infinity-Cache hits included).  gfx950 correction (MI355X_MICROARCH.md, section HBM): FETCH_SIZE reports
exactly half of the bytes of wide (16 B/lane) streaming reads -- which is what the LDS-DMA operand
gather issues -- so it is doubled; WRITE_SIZE is used as reported (uncalibrated)."""
import csv
import glob
import json
import os
import sys


FAMILIES = ("conv_igemm", "conv3x3_ws2", "conv3x3_ws128", "conv3d_narrow", "conv_in8", "tblock_pair", "flash_attn")   # the MFMA kernels of the path (bench.py roofline.kernel)


def collect(path, counter):
    vals = []
    for f in glob.glob(os.path.join(path, counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(t in r["Kernel_Name"] for t in FAMILIES) and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


def main(root, out=None):
    fetch, write = collect(root, "FETCH_SIZE"), collect(root, "WRITE_SIZE")
    n = min(len(fetch), len(write))
    res = {
        "kernel": "conv_igemm_glds_kernel + conv3x3_ws2_kernel + conv_in8_kernel + conv3d_narrow_kernel + tblock_pair_kernel + flash_attn_kernel", "launches_sampled": n,
        "fetch_kib_reported_per_launch": sum(fetch) / max(1, len(fetch)),
        "write_kib_per_launch": sum(write) / max(1, len(write)),
        "fetch_correction": 2.0,
        "traffic_bytes_per_launch": (2.0 * sum(fetch) / max(1, len(fetch)) + sum(write) / max(1, len(write))) * 1024.0,
        "note": "mean over every conv launch of the traced bench passes (warm-up + timed + event-timed steps)",
    }
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:3])

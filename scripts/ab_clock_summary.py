"""Shader clock per arm of scripts/activation_restream_ab.py from a `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv` run of it:
the profiled launches come in the script's order (per case: 2 repetitions x 5 arms x (3 + reps) launches), so dispatch i of the instrumented kernel
belongs to arm (i // (3 + reps)) % 5.  clock = GRBM_GUI_ACTIVE / duration (MI355X_MICROARCH.md, DVFS give-back); GRBM_GUI_ACTIVE is reported summed over
the 8 XCDs.  usage: python scripts/ab_clock_summary.py <dir with p_counter_collection.csv / p_kernel_trace.csv> <reps>"""
import csv
import glob
import os
import sys

ARMS = ["(i)   activations from an L2-resident live patch", "(ii)  as shipped", "(iii) activation pieces = zero fills",
        "      weights L2-resident, live", "      both operands L2-resident, live"]


def main(root, reps):
    cc = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(cc[0])) if "conv_igemm" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    if kt and rows and "Start_Timestamp" not in rows[0]:       # the timestamps live in the kernel trace: join on the dispatch id
        ts = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt[0]))}
        for r in rows:
            r["Start_Timestamp"], r["End_Timestamp"] = ts[r["Dispatch_Id"]]["Start_Timestamp"], ts[r["Dispatch_Id"]]["End_Timestamp"]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # one un-instrumented launch (ops.conv, which records the descriptor) precedes the arms of each case: the instrumented kernel has another name
    prof = [r for r in rows if "Lb1ELi" in r["Kernel_Name"] or "true" in r["Kernel_Name"]]
    per = 3 + reps
    n_case = 2 * len(ARMS) * per
    for c in range(len(prof) // n_case):
        print(f"case {c}:")
        for a, name in enumerate(ARMS):
            sel = []
            for rep in range(2):
                base = c * n_case + (rep * len(ARMS) + a) * per
                sel += prof[base + 3: base + per]
            ns = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in sel]
            cyc = [float(r["Counter_Value"]) / 8.0 for r in sel]
            ghz = sum(c_ / n for c_, n in zip(cyc, ns)) / len(sel)
            print(f"    {name:50s} {sum(ns) / len(ns) / 1e6:7.3f} ms under the profiler   clock {ghz:5.3f} GHz   (n = {len(sel)})")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)

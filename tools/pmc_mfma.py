#!/usr/bin/env python3
"""MFMA-busy fraction and clock of the conv kernels from a `rocprofv3 --pmc GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace` run of bench.py (RGFM_OVERLAP=0).

  tools/pmc_mfma.py <dir>     (expects <dir>/*/*_counter_collection.csv and *_kernel_trace.csv)

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); clk = (GRBM_GUI_ACTIVE / 8) / duration.
The 20 longest launches of each conv kernel are averaged (short launches read high, MI355X_MICROARCH.md)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    cc = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    kt = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    vals = defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(cc)):
        did = int(r["Dispatch_Id"])
        vals[did][r["Counter_Name"]] = vals[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[did] = r["Kernel_Name"]
    by = defaultdict(list)
    for did, v in vals.items():
        n = names[did]
        if "conv_mfma" in n and did in dur and "GRBM_GUI_ACTIVE" in v:
            by[n.split("(")[0].replace("void rgfm::", "")].append((dur[did], v))
    out = []
    for n, lst in sorted(by.items()):
        lst.sort(key=lambda t: -t[0])
        top = lst[:20]
        act = sum(v["GRBM_GUI_ACTIVE"] for _, v in top) / 8.0
        busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for _, v in top)
        wave = sum(v.get("SQ_WAVE_CYCLES", 0.0) for _, v in top)
        wait = sum(v.get("SQ_WAIT_ANY", 0.0) for _, v in top)
        t = sum(dv for dv, _ in top)
        out.append({"kernel": n, "launches_averaged": len(top), "avg_us": t / len(top) / 1e3, "clk_GHz": act / t,
                    "mfma_busy": busy / (act * 1024.0), "wait_any": wait / wave if wave else None})
    print(json.dumps({"what": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY; 20 longest "
                              "launches of each conv kernel; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)",
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    main()

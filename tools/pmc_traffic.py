#!/usr/bin/env python3
"""HBM traffic of the conv_mfma launches from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE: separate passes, MI355X_MICROARCH.md "HBM").

  tools/pmc_traffic.py <dir_with_FETCH_pass> <dir_with_WRITE_pass> [B] > profiles/rNN_conv_traffic_<mode>.json

The output carries the sha256 of the conv kernel sources (bench.kernel_sources_sha): bench.py reports
`roofline.traffic` from the file only while those sources are unchanged.

Corrections applied as the guide prescribes for gfx950: FETCH_SIZE (KB) under-reports wide
16-B-per-lane streaming reads by exactly 2x -> doubled; WRITE_SIZE (KB) is exact.
Prints totals over the conv class and the algorithmic bytes (each conv reads its input(s),
residual and weights once and writes its output + statistics once)."""
import csv
import glob
import json
import sys
from collections import defaultdict

import os  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from trace_layers import unet_convs  # noqa: E402
from bench import kernel_sources_sha  # noqa: E402


def load(d, counter):
    vals = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0])):
        if r["Counter_Name"] == counter:
            vals[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    return vals, names


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    fv, fn = load(fd, "FETCH_SIZE")
    wv, wn = load(wd, "WRITE_SIZE")
    conv_f = [fv[d] for d in sorted(fv) if "conv_mfma" in fn[d]]
    conv_w = [wv[d] for d in sorted(wv) if "conv_mfma" in wn[d]]
    mn, sv = unet_convs(1, 32, 32, (1, 2)), unet_convs(3, 32, 64, (1, 2, 2))
    per_step = len(mn) + len(sv)
    # last Euler step of the guided main loop (batch B)
    f_step = conv_f[-per_step:]
    w_step = conv_w[-per_step:]
    rd = 2.0 * 1024.0 * sum(f_step)
    wr = 1024.0 * sum(w_step)
    alg = 0.0
    for (name, mode, S, cin, cout, sk) in sv + mn:
        s_in = S * 2 if mode == 1 else (S // 2 if mode == 2 else S)
        alg += 4.0 * B * (s_in * s_in * cin + S * S * cout)          # input + output
        alg += 4.0 * B * S * S * (sk if sk else (cout if "conv2" in name else 0))  # skip-conv input / identity residual
        alg += 4.0 * (9 * cin + sk) * cout                          # weights
        alg += 4.0 * B * (S * S // 64 if S * S > 64 else 1) * cout * 2  # GroupNorm partial statistics
    out = {
        "what": f"conv_mfma launches of one guided Euler step, batch {B} (MNIST32 + SVHN nets, {per_step} launches)",
        "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr, "algorithmic_bytes": alg,
        "ratio": (rd + wr) / alg,
        "per_launch_avg_bytes": (rd + wr) / per_step,
        "corrections": "FETCH_SIZE KB x1024 x2 (gfx950 wide-read under-count), WRITE_SIZE KB x1024",
        "kernel_sources_sha256": kernel_sources_sha(),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

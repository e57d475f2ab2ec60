#!/usr/bin/env python3
"""Per-layer efficiency from a rocprofv3 kernel trace of bench.py.

Reconstructs the conv_mfma launch sequence of one SVHN and one MNIST32 U-Net
evaluation (same walk as csrc/rgfm_host.h UNetRun::run), attaches algorithmic
FLOPs to each launch, and prints achieved TFLOP/s per layer for one main-loop
step (B rows) of the trace.  Usage: trace_layers.py <kernel_trace.csv> [B]"""
import csv
import sys
from collections import defaultdict


def unet_convs(in_ch, size, mc, mult, nres=2):
    """[(name, mode, S_out, cin, cout, skipk)] in launch order (conv_mfma only)."""
    out = []
    ch, S = mc, size
    skips = [ch]

    def res(name, cin, cout, S):
        out.append((name + ".conv1", 0, S, cin, cout, 0))
        out.append((name + ".conv2", 0, S, cout, cout, cin if cin != cout else 0))

    e = 0
    for l, m in enumerate(mult):
        for _ in range(nres):
            res(f"enc{e}", ch, mc * m, S)
            ch = mc * m
            skips.append(ch)
            e += 1
        if l < len(mult) - 1:
            S //= 2
            out.append((f"down{l}", 1, S, ch, ch, 0))
            skips.append(ch)
    res("mid0", ch, ch, S)
    res("mid1", ch, ch, S)
    d = 0
    for l in range(len(mult) - 1, -1, -1):
        for _ in range(nres + 1):
            res(f"dec{d}", ch + skips.pop(), mc * mult[l], S)
            ch = mc * mult[l]
            d += 1
        if l > 0:
            S *= 2
            out.append((f"up{l}", 2, S, ch, ch, 0))
    return out


def main():
    path = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "conv_mfma" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])))
    rows.sort()
    mn = unet_convs(1, 32, 32, (1, 2))
    sv = unet_convs(3, 32, 64, (1, 2, 2))
    per_step = len(mn) + len(sv)
    # the trace ends with the guided main loop: its last per_step conv launches are one Euler step
    # (MNIST32 net then SVHN net) at batch B
    start = len(rows) - per_step
    seq = rows[start:start + per_step]
    tot_fl = tot_t = tot_roof = tot_by = 0.0
    agg = defaultdict(lambda: [0.0, 0.0, 0.0])
    # per-layer roofline against the MEASURED ceilings of the bench line (roofline.measured_ceilings): the f16-MFMA
    # loop / 3 products, and the float4 copy rate; bytes = input + residual or skip source + output, once each
    mf = float(sys.argv[3]) if len(sys.argv) > 3 else 565.0   # TFLOP/s fp32-equivalent
    bw = float(sys.argv[4]) if len(sys.argv) > 4 else 5.9     # TB/s (rgfm_ubench_hbm_copy: eight loads in flight per thread)
    print(f"{'layer':16s} {'S':>3s} {'cin':>4s} {'cout':>4s} {'grid':>10s} {'us':>8s} {'TF/s':>7s} {'MB':>6s} {'roof us':>8s} {'bound':>5s} {'frac':>5s}")
    # launch order inside a step: SVHN net first (side stream), then the MNIST net
    for (name, mode, S, cin, cout, sk), (t0, t1, kn, gx, gy, gz) in zip([("s." + a[0],) + a[1:] for a in sv] +
                                                                      [("m." + a[0],) + a[1:] for a in mn], seq):
        fl = 2.0 * B * S * S * cout * (9 * cin + sk)
        us = (t1 - t0) / 1e3
        s_in = S if mode == 0 else (2 * S if mode == 1 else S // 2)
        by = 4.0 * B * (s_in * s_in * cin + S * S * cout)
        if name.endswith(".conv2"):
            by += 4.0 * B * S * S * (sk if sk else cout)  # 1x1-skip source, or the identity residual
        # an Upsample conv launched as four parity classes (grid z = 4) EXECUTES 4 / 9 of its algorithmic products: the
        # TF/s column stays algorithmic, its matrix roofline is that of the executed work (name marked *)
        t2 = mode == 2 and gz == 4
        if t2:
            name += "*"
        t_m, t_h = fl * (4.0 / 9.0 if t2 else 1.0) / (mf * 1e6), by / (bw * 1e6)
        roof = max(t_m, t_h)
        tot_fl += fl
        tot_t += us
        tot_roof += roof
        tot_by += by
        agg[(S, mode)][0] += fl
        agg[(S, mode)][1] += us
        agg[(S, mode)][2] += roof
        print(f"{name:16s} {S:3d} {cin:4d} {cout:4d} {gx:6d}x{gy:<3d} {us:8.1f} {fl / us / 1e6:7.1f} {by / 1e6:6.0f} {roof:8.1f} "
              f"{'mfma' if t_m >= t_h else 'hbm':>5s} {roof / us:5.2f}")
    print(f"step total: {tot_fl / 1e12:.3f} TFLOP in {tot_t / 1e3:.2f} ms = {tot_fl / tot_t / 1e6:.1f} TF/s")
    print(f"per-layer roofline (max of {mf:.0f} TFLOP/s fp32-equivalent and {bw:.2f} TB/s, both measured): {tot_roof / 1e3:.2f} ms "
          f"-> the step runs at {tot_roof / tot_t:.2f} of it; algorithmic bytes {tot_by / 1e9:.2f} GB")
    span = (seq[-1][1] - seq[0][0]) / 1e6
    print(f"wall span of these launches: {span:.2f} ms")
    for k in sorted(agg):
        fl, us, roof = agg[k]
        print(f"  S={k[0]:2d} mode={k[1]}: {us / 1e3:7.2f} ms  {fl / us / 1e6:6.1f} TF/s  ({100 * us / tot_t:.1f}% of conv time)  {roof / us:.2f} of its roofline")


if __name__ == "__main__":
    main()

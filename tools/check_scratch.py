#!/usr/bin/env python3
"""Register / scratch table of the library's kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(csrc/Makefile writes them to build/*.res), and the build gate VERDICT r3 item 8 asked for: a conv instantiation
that the library dispatches may use scratch only if it is on the allow-list below, with the reason stated.

    tools/check_scratch.py build/*.res          (exit 1 on a violation; --table prints every conv kernel)
"""
import re
import subprocess
import sys

# instantiation (demangled prefix) -> (max scratch bytes per lane, why it is tolerated)
ALLOW = {
    # the 128-register cut of the tile stream: 10 dwords saved before / restored after the K loop of a tile (tile-boundary
    # values), none inside it (tools/kbench: scratch traffic per tile = 7 stores + 9 loads against ~40 k cycles of K loop)
    "conv_mfma_hx2q_kernel<4, false, 2, 1>": (40, "tile-boundary values, outside the K loop"),
    "conv_mfma_hx2q_kernel<5, false, 2, 1>": (40, "tile-boundary values, outside the K loop"),
    # built but not dispatched (conv_hx2q_supported keeps Cout = 64 skip layers on hx2p)
    "conv_mfma_hx2q_kernel<4, true, 2, 1>": (104, "not dispatched"),
    "conv_mfma_hx2q_kernel<5, true, 2, 1>": (104, "not dispatched"),
    # four-wave 64-channel workgroups: under-filled launches only (MC pre-phase of the MNIST net's 16x16 level, small batches)
    "conv_mfma_hx2p_kernel<2, 0, 2, false>": (44, "under-filled launches only; prologue values"),
    "conv_mfma_hx2p_kernel<2, 2, 2, false>": (44, "under-filled launches only; prologue values"),
    "conv_mfma_hx2p_kernel<2, 0, 2, true>": (44, "under-filled launches only; prologue values"),
    # the P-format producer twin of the two-tile kernel (MNIST net's 16x16 conv1 layers): 3 dwords around the epilogue
    "conv_mfma_hx2p_kernel<2, 0, 0, true>": (12, "epilogue of the P-format producer"),
    # the Winograd kernel: up to four dwords of the epilogue's addressing saved in the prologue, restored behind the K loop
    "conv_mfma_hx2w_kernel<4>": (16, "epilogue addressing, outside the K loop"),
    "conv_mfma_hx2w_kernel<5>": (16, "epilogue addressing, outside the K loop"),
}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    table = "--table" in sys.argv
    rows = []
    for f in files:
        name = vg = None
        for ln in open(f, errors="replace"):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                name = m.group(1)
            m = re.search(r" VGPRs: (\d+)", ln)
            if m:
                vg = int(m.group(1))
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", ln)
            if m and name:
                rows.append((name, vg, int(m.group(1))))
                name = None
    if not rows:
        print("check_scratch: no resource remarks found", file=sys.stderr)
        return 1
    names = demangle([r[0] for r in rows])
    bad = 0
    for (mn, vg, sc), dn in zip(rows, names):
        short = re.sub(r"^void rgfm::", "", dn)
        short = re.sub(r"\(.*$", "", short)
        conv = "conv_mfma" in short
        if table and conv:
            print(f"{short:60s} vgpr {vg:4d} scratch {sc}")
        if sc > 0:
            lim = ALLOW.get(short)
            if lim is None or sc > lim[0]:
                print(f"check_scratch: {short}: {sc} bytes of scratch per lane" + (f" (allowed: {lim[0]})" if lim else " (not on the allow-list)"),
                      file=sys.stderr)
                bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Per-launch durations and gaps of ONE Euler step of the SVHN net at a small row count (the per-rank pre-phase of a multi-GPU
run), from a rocprofv3 kernel trace.  Two parts:
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/small_rows_trace.py run 32
  python3 tools/small_rows_trace.py show DIR"""
import csv
import glob
import sys

if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, '.')
    from ratio_guided_multimodal_fm_amd import _engine, models as M
    from ratio_guided_multimodal_fm_amd.synth import load_synth
    dev = torch.device("cuda:0")
    fs = load_synth(M.FlowMatchingUNetSVHN(), 1).eval().to(dev)
    nb = int(sys.argv[2])
    yy = torch.randn(nb, 3, 32, 32, device=dev)
    for _ in range(2):
        _engine._sample_single(fs, yy.clone(), 20)
    torch.cuda.synchronize()
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"])))
    rows.sort()
    # the last step: from the last conv_in launch but one to the last conv_out
    idx = [i for i, r in enumerate(rows) if "conv_in" in r[2]]
    a = idx[-1]
    b = max(i for i, r in enumerate(rows) if "conv_out" in r[2])
    step = rows[a:b + 1]
    tot = step[-1][1] - step[0][0]
    busy = sum(r[1] - r[0] for r in step)
    print(f"{len(step)} launches, span {tot / 1e3:.1f} us, sum of durations {busy / 1e3:.1f} us, gaps {(tot - busy) / 1e3:.1f} us")
    prev = None
    for r in step:
        gap = (r[0] - prev) / 1e3 if prev else 0.0
        prev = r[1]
        name = r[2].replace("rgfm::", "").replace("(rgfm::ConvArgs, int)", "").replace("void ", "")[:60]
        print(f"{name:60s} wgs {r[3] // r[6]:4d}x{r[4]}x{r[5]} thr {r[6]:4d}  {(r[1] - r[0]) / 1e3:7.1f} us  gap {gap:5.1f}")

"""Where a sampling call's time goes, and what one rank of an N-GPU run does (development tool, one GPU).

  python tools/phase_split.py [--json out.json]

* MC pre-phase (N_mc rows per net) vs guided main loop (B rows) of BASELINE configs[2];
* the per-step rate of the two nets against the number of rows per launch;
* PROJECTION of the weak-scaling curve of bench.py (512 rows per rank, configs[3] at 8 ranks): one rank of a W-rank
  run integrates N_mc / W pre-phase rows, evaluates the ratio net on them, and runs the guided loop on its 512 rows
  -- all of which this tool times on ONE GPU at exactly those shapes.  The collectives (one all_gather of the MC
  set, 4.2 MB; one gather of 8.4 MB per rank) are NOT executed here and are priced from the guide's xGMI figures
  (<= 0.2 ms per call at these sizes against a ~1.1 s call).  A projection, not a measurement of an 8-GPU node."""
import json
import sys
import time

import torch

sys.path.insert(0, '.')
from ratio_guided_multimodal_fm_amd import _engine, models as M  # noqa: E402
from ratio_guided_multimodal_fm_amd.synth import load_synth, paired_noise  # noqa: E402

dev = torch.device("cuda:0")
fm = load_synth(M.FlowMatchingUNetMNIST(32), 0).eval().to(dev)
fs = load_synth(M.FlowMatchingUNetSVHN(), 1).eval().to(dev)
rr = load_synth(M.RatioEstimatorMNISTSVHN(), 2).eval().to(dev)
B, N, S = 512, 256, 100
x0, y0, mx0, my0 = (v.to(dev) for v in paired_noise(42, B, N, (1, 32, 32), (3, 32, 32)))
PAIR_FLOP = 3.503947776e9  # one pair, one Euler step


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


out = {}
mx1, my1 = mx0.clone(), my0.clone()
tp = t(lambda: _engine.sample_two_streams(fm, mx0.clone(), fs, my0.clone(), S))
_engine.sample_two_streams(fm, mx1, fs, my1, S)
r = rr._engine.eval(mx1, my1, "ratio")
tm = t(lambda: _engine.sample_pair(fm, fs, x0.clone(), y0.clone(), mx1, my1, r, S, 0.5))
print(f"pre-phase ({N} rows x 2 nets, {S} steps): {1e3 * tp:.1f} ms = {N * S * PAIR_FLOP / tp / 1e12:.1f} TFLOP/s; "
      f"main loop ({B} rows): {1e3 * tm:.1f} ms = {B * S * PAIR_FLOP / tm / 1e12:.1f} TFLOP/s")
out["prephase_ms"], out["mainloop_ms"] = 1e3 * tp, 1e3 * tm
rows = {}
for nb in (32, 64, 128, 256, 384, 512, 768):
    xx, yy = torch.randn(nb, 1, 32, 32, device=dev), torch.randn(nb, 3, 32, 32, device=dev)
    tt = t(lambda: _engine.sample_two_streams(fm, xx.clone(), fs, yy.clone(), 20))
    rows[nb] = 1e3 * tt / 20
    print(f"  unguided two nets, {nb} rows, 20 steps: {1e3 * tt / 20:.2f} ms/step = {nb * PAIR_FLOP / (tt / 20) / 1e12:.1f} TFLOP/s")
out["ms_per_step_by_rows"] = rows

print("projection of bench.py --gpus W (512 rows per rank; gamma 1.0 as configs[3]):")
tm1 = t(lambda: _engine.sample_pair(fm, fs, x0.clone(), y0.clone(), mx1, my1, r, S, 1.0))
proj = {}
base = None
for W in (1, 2, 4, 8):
    n = N // W
    a, b = mx0[:n].clone(), my0[:n].clone()
    tpw = t(lambda: _engine.sample_two_streams(fm, mx0[:n].clone(), fs, my0[:n].clone(), S))
    _engine.sample_two_streams(fm, a, fs, b, S)
    trw = t(lambda: rr._engine.eval(a, b, "ratio"), reps=5)
    call = tpw + trw + tm1 + (0.0 if W == 1 else 0.0004)  # + two latency-bound collectives (priced, not run)
    ips = W * B / call
    base = base or ips
    proj[W] = {"prephase_rows": n, "prephase_ms": 1e3 * tpw, "ratio_ms": 1e3 * trw, "mainloop_ms": 1e3 * tm1,
               "call_ms": 1e3 * call, "images_per_s": ips, "speedup_vs_1": ips / base}
    print(f"  W={W}: pre-phase {n:3d} rows {1e3 * tpw:7.1f} ms + ratio {1e3 * trw:5.2f} ms + main loop {1e3 * tm1:7.1f} ms "
          f"= {1e3 * call:7.1f} ms per call -> {ips:7.1f} paired images/s ({ips / base:.2f}x of W=1)")
out["weak_scaling_projection"] = proj
if "--json" in sys.argv:
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
        json.dump(out, f, indent=1)

"""Where a sampling call's time goes: MC pre-phase (N_mc rows per net) vs guided main loop (B rows), and how the
per-step rate of the two nets scales with the number of rows (development tool)."""
import sys, time, torch
sys.path.insert(0, '.')
from ratio_guided_multimodal_fm_amd import _engine, models as M
from ratio_guided_multimodal_fm_amd.synth import load_synth, paired_noise
dev = torch.device("cuda:0")
fm = load_synth(M.FlowMatchingUNetMNIST(32), 0).eval().to(dev)
fs = load_synth(M.FlowMatchingUNetSVHN(), 1).eval().to(dev)
rr = load_synth(M.RatioEstimatorMNISTSVHN(), 2).eval().to(dev)
x0, y0, mx0, my0 = (v.to(dev) for v in paired_noise(42, 512, 256, (1, 32, 32), (3, 32, 32)))
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        t0=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    return sorted(ts)[len(ts)//2]
mx1, my1 = mx0.clone(), my0.clone()
tp = t(lambda: _engine.sample_two_streams(fm, mx0.clone(), fs, my0.clone(), 100))
_engine.sample_two_streams(fm, mx1, fs, my1, 100)
r = rr._engine.eval(mx1, my1, "ratio")
tm = t(lambda: _engine.sample_pair(fm, fs, x0.clone(), y0.clone(), mx1, my1, r, 100, 0.5))
fl_pre = 256*100*3.503947776e9; fl_main = 512*100*3.503947776e9
print(f"pre-phase (256 rows x 2 nets, 100 steps): {1e3*tp:.1f} ms = {fl_pre/tp/1e12:.1f} TFLOP/s; main loop (512 rows): {1e3*tm:.1f} ms = {fl_main/tm/1e12:.1f} TFLOP/s")
for nb in (128, 256, 384, 512, 768):
    xx, yy = torch.randn(nb,1,32,32,device=dev), torch.randn(nb,3,32,32,device=dev)
    tt = t(lambda: _engine.sample_two_streams(fm, xx.clone(), fs, yy.clone(), 20))
    print(f"  unguided two nets, {nb} rows, 20 steps: {1e3*tt/20:.2f} ms/step = {nb*3.503947776e9/(tt/20)/1e12:.1f} TFLOP/s")

"""Probe: does splitting a 512-row evaluation of the two nets into two 256-row halves on separate streams (four
streams in all) beat the single 512-row call?  (development experiment; result in DESIGN.md)"""
import copy, sys, time
import torch
sys.path.insert(0, '.')
from ratio_guided_multimodal_fm_amd import _engine, models as M
from ratio_guided_multimodal_fm_amd.synth import load_synth

dev = torch.device("cuda:0")
fm = load_synth(M.FlowMatchingUNetMNIST(32), 0).eval().to(dev)
fs = load_synth(M.FlowMatchingUNetSVHN(), 1).eval().to(dev)
fm2, fs2 = copy.deepcopy(fm), copy.deepcopy(fs)
S = 20
x, y = torch.randn(512, 1, 32, 32, device=dev), torch.randn(512, 3, 32, 32, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def whole():
    _engine.sample_two_streams(fm, x.clone(), fs, y.clone(), S)

def halves():
    xa, ya, xb, yb = x[:256].clone(), y[:256].clone(), x[256:].clone(), y[256:].clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(sA):
        _engine.sample_two_streams(fm, xa, fs, ya, S)
    with torch.cuda.stream(sB):
        _engine.sample_two_streams(fm2, xb, fs2, yb, S)

for name, fn in (("one 512-row call", whole), ("two 256-row calls on two stream pairs", halves), ("one 512-row call", whole),
                 ("two 256-row calls on two stream pairs", halves)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"{name}: {1e3 * t / S:.2f} ms per step")

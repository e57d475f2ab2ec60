#!/usr/bin/env python3
"""Error of the two-plane fp16 product (conv_mfma_hx2*.hip) measured in numpy -- the numbers DESIGN.md section 4
quotes instead of an asserted bound.

    a' = 16 a = a_h + a_l,  a_h = fp16(a'), a_l = fp16(a' - a_h)       (activation: S_A = 16)
    w' = s_w w = w_h + w_l, s_w = 2^k with max|w'| in [2^13, 2^14)      (weights: one scale per conv)
    a w ~= (a_l w_h + a_h w_l + a_h w_h) / (16 s_w)                     (three exact fp16 x fp16 products, fp32 sums)

Prints the relative error of single products and of K-term dot products against float64 for (i) activations as the
convs see them -- silu of a unit-variance normalised value --, (ii) the raw residual stream at several magnitudes
(the low side of the window), so that the flag thresholds can be read off."""
import numpy as np

rng = np.random.default_rng(0)


def split16(v):
    h = v.astype(np.float16)
    l = (v - h.astype(np.float32)).astype(np.float16)
    return h, l


def hx2_products(a, w):
    """a, w float32 arrays -> the kernel's approximation of a * w, evaluated in float64 from the fp16 planes."""
    sw = np.float32(2.0 ** (13 - np.floor(np.log2(np.abs(w).max()))))
    ah, al = split16(np.float32(16.0) * a)
    wh, wl = split16(sw * w)
    f = lambda x: x.astype(np.float64)
    return (f(al) * f(wh) + f(ah) * f(wl) + f(ah) * f(wh)) / (16.0 * float(sw))


def report(tag, a, w, K=576):
    exact = a.astype(np.float64) * w.astype(np.float64)
    got = hx2_products(a, w)
    nz = exact != 0
    rel = np.abs(got - exact)[nz] / np.abs(exact)[nz]
    q = lambda p: np.log2(max(np.quantile(rel, p), 1e-300))
    n = (a.size // K) * K
    de, dg = exact[:n].reshape(-1, K).sum(1), got[:n].reshape(-1, K).sum(1)
    # dot products: error relative to the root-sum-square of the terms (what an fp32 accumulation is judged against)
    scale = np.sqrt((exact[:n].reshape(-1, K) ** 2).sum(1))
    drel = np.abs(dg - de) / scale
    # the same dot product with fp32 operands and fp32 sequential accumulation (the reference's arithmetic class)
    f32 = np.zeros(n // K, dtype=np.float32)
    a32, w32 = a[:n].reshape(-1, K), w[:n].reshape(-1, K)
    for k in range(K):
        f32 = f32 + a32[:, k] * w32[:, k]
    frel = np.abs(f32.astype(np.float64) - de) / scale
    print(f"{tag:44s} product rel err: median 2^{q(0.5):6.1f}  99.9% 2^{q(0.999):6.1f}  max 2^{q(1.0):6.1f} | "
          f"K={K} dot / rss: hx2 median {np.median(drel):.2e} max {drel.max():.2e}; fp32 median {np.median(frel):.2e} max {frel.max():.2e}")


N = 576 * 4000
z = rng.standard_normal(N).astype(np.float32)
silu = (z / (1.0 + np.exp(-z))).astype(np.float32)
w = (0.05 * rng.standard_normal(N)).astype(np.float32)
report("GroupNorm+SiLU activations (normalised path)", silu, w)
for lg in (0, -4, -8, -10, -12, -14, -20):
    report(f"raw residual stream, values N(0,1) x 2^{lg}", (z * np.float32(2.0 ** lg)).astype(np.float32), w)

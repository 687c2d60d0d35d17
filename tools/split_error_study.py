"""Numerical study for DESIGN section 7.1a (CPU, numpy): error of an fp32 GEMM row computed as (a) plain float32, (b) the
shipped three-plane bf16 split with six products, (c) a two-plane fp16 split with three products, all accumulated in float32,
against float64.  Operands: LayerNorm-like activations (unit variance, optionally scaled) and 1/sqrt(K)-uniform weights."""
import json, sys
import numpy as np

rng = np.random.default_rng(0)
f32 = np.float32


def bf16(x):                       # round-to-nearest-even to bfloat16, kept as float32
    u = x.astype(f32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(f32)


def split_bf16x3(x):
    hi = bf16(x); mid = bf16(x - hi); lo = bf16(x - hi - mid)
    return hi, mid, lo


def split_f16x2(x, scale):
    xs = (x * f32(scale)).astype(f32)
    hi = xs.astype(np.float16).astype(f32)
    lo = (xs - hi).astype(np.float16).astype(f32)
    return hi, lo


def mm(a, b):                      # float32 accumulation
    return a.astype(f32) @ b.astype(f32).T


def study(rows, K, cols, act_scale, sa, sb):
    A = (rng.standard_normal((rows, K)) * act_scale).astype(f32)
    W = ((rng.random((cols, K)) * 2 - 1) / np.sqrt(K)).astype(f32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    den = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T          # sum |a||w|: the natural error scale
    out = {}
    out["fp32"] = mm(A, W)
    ah, am, al = split_bf16x3(A); wh, wm, wl = split_bf16x3(W)
    out["bf16x3_6prod"] = ((((mm(al, wh) + mm(ah, wl)) + mm(am, wm)) + mm(am, wh)) + mm(ah, wm)) + mm(ah, wh)
    h, l = split_f16x2(A, sa); g, m_ = split_f16x2(W, sb)
    out["f16x2_3prod"] = (((mm(l, g) + mm(h, m_)) + mm(h, g)) / f32(sa * sb)).astype(f32)
    res = {}
    for k, v in out.items():
        e = np.abs(v.astype(np.float64) - ref) / den
        res[k] = {"mean": float(e.mean()), "max": float(e.max())}
    return res


if __name__ == "__main__":
    rows = 512
    table = {}
    for name, K, act_scale, sa, sb in [("K432 act~1 scales 1/128", 432, 1.0, 1, 128), ("K432 act~1 scales 16/128", 432, 1.0, 16, 128),
                                       ("K864 act~1 scales 1/128", 864, 1.0, 1, 128), ("K432 act~1e-3 scales 1/128", 432, 1e-3, 1, 128),
                                       ("K432 act~30 scales 1/128", 432, 30.0, 1, 128)]:
        table[name] = study(rows, K, 432, act_scale, sa, sb)
    print(json.dumps(table, indent=1))

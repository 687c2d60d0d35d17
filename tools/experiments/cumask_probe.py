"""Probe (not in the product): hipExtStreamCreateWithCUMask on MI355X -- which mask bit is which XCC / CU, and what the plane GEMM launches cost
when the chip is split between independent streams (each on its own CUs, de-phased) instead of one stream on all of it."""
import os, sys, ctypes, collections, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = torch.device("cuda:0")
W = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libwhere.so"))

def masked_stream(bits):
    words = (ctypes.c_uint * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    st = ctypes.c_void_p()
    rc = W.masked_stream(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)

def where(stream, nwg=2048):
    out = torch.zeros(2 * nwg, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    W.where_launch(ctypes.c_void_p(out.data_ptr()), nwg, 20, ctypes.c_void_p(stream.cuda_stream))
    torch.cuda.synchronize()
    o = out.cpu().view(nwg, 2)
    xcc = (o[:, 0] & 15).tolist()
    hw = o[:, 1].tolist()
    cus = collections.Counter((x, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15) for x, h in zip(xcc, hw))
    return collections.Counter(xcc), len(cus)

res = {}
full = torch.cuda.Stream(device=dev)
print("full stream:", where(full))
masks = {"low128": set(range(128)), "even": set(range(0, 256, 2)), "mod8lt4": {i for i in range(256) if i % 8 < 4}, "first32": set(range(32)),
         "mod8eq0": set(range(0, 256, 8)), "mod16lt8": {i for i in range(256) if i % 16 < 8}}
streams = {}
for k, bits in masks.items():
    try:
        streams[k] = masked_stream(bits)
        xc, ncu = where(streams[k])
        print(k, dict(xc), "distinct CUs", ncu)
        res[k] = {"xcc_histogram": dict(xc), "distinct_cus": ncu}
    except AssertionError as e:
        print(k, "failed", e)

# ---- GEMM launches: one stream on the whole chip vs independent streams on disjoint CU sets
C = 432
def problem(rows):
    x = torch.randn(rows, C, device=dev)
    img, bnd = lib.planes_from_f32(x)
    g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    lnb = lib.ln_bound(g1, b1)
    msg_img, msg_b = lib.planes_from_f32(torch.randn(rows, C, device=dev))
    hid_img, hid_b = lib.planes_from_f32(torch.randn(rows, 2 * C, device=dev))
    o_img = torch.zeros_like(img); o_b = torch.zeros(rows, device=dev); o32 = torch.empty(rows, 3 * C, device=dev)
    h_img = torch.zeros_like(hid_img); h_b = torch.zeros(rows, device=dev)
    ang = torch.rand(rows, C // 2, device=dev); cosT, sinT = ang.cos().contiguous(), ang.sin().contiguous()
    pk3 = lib.pack_weight_planes(torch.randn(3 * C, C, device=dev) / C ** 0.5, 3, C)
    pk1 = lib.pack_weight_planes(torch.randn(C, C, device=dev) / C ** 0.5, 1, C)
    pk0 = lib.pack_weight_planes(torch.randn(2 * C, 2 * C, device=dev) / (2 * C) ** 0.5, 2, C)
    pk2 = lib.pack_weight_planes(torch.randn(C, 2 * C, device=dev) / (2 * C) ** 0.5, 1, C)
    def layer():
        lib.linear_planes(rows, C, 3, img, bnd, C, pk3, lib.PL_F32, out=o32, ldo=3 * C, blk_stride=C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C)
        lib.linear_planes(rows, C, 1, img, bnd, C, pk1, lib.PL_LN, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, lnb=lnb)
        lib.linear_planes(rows, C, 2, img, bnd, C, pk0, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=C, out_image=h_img, out_image_k=2 * C, out_bound=h_b, relu=True)
        lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, pk2, lib.PL_LN, out=o32, ldo=3 * C, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=x, ldr=C, bound_resid=bnd, lnb=lnb)
    return layer

import time
def run(parts, reps=12):
    """parts: list of (stream, layer callable, delay launches); every stream runs `reps` layer chains; wall time of all"""
    for st, f, _ in parts:
        with torch.cuda.stream(st):
            f(); f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for st, f, _ in parts:
            with torch.cuda.stream(st):
                f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

TOTAL = 65536 * 2            # rows of two 128-pair batches (what a bench step holds)
out = {}
whole = [problem(65536) for _ in range(2)]
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
out["one stream, whole chip, 2 x 65536 rows in sequence"] = run([(s1, whole[0], 0), (s1, whole[1], 0)])
out["two streams, whole chip each (the bench today)"] = run([(s1, whole[0], 0), (s2, whole[1], 0)])
for name, nsplit in (("low128", 2), ("mod8lt4", 2), ("mod16lt8", 2)):
    if name not in streams: continue
    bits = masks[name]
    other = masked_stream(set(range(256)) - bits)
    print("complement of", name, where(other))
    out["two streams on disjoint halves (%s | complement), 65536 rows each" % name] = run([(streams[name], whole[0], 0), (other, whole[1], 0)])
del whole
# eight streams, one XCC each (if the mask has an XCC-per-bit-class form), 16384 rows each
for name, cls in (("bit % 8", lambda i, k: i % 8 == k), ("bit // 32", lambda i, k: i // 32 == k)):
    try:
        sts = [masked_stream({i for i in range(256) if cls(i, k)}) for k in range(8)]
        print(name, "stream 0 ->", where(sts[0]))
        probs = [problem(16384) for _ in range(8)]
        out["eight streams (%s == k), 16384 rows each" % name] = run([(sts[k], probs[k], 0) for k in range(8)])
        fs = [torch.cuda.Stream(device=dev) for _ in range(8)]
        out["eight unmasked streams, 16384 rows each"] = run([(fs[k], probs[k], 0) for k in range(8)])
        sts4 = [masked_stream({i for i in range(256) if cls(i, 2 * k) or cls(i, 2 * k + 1)}) for k in range(4)]
        probs4 = [problem(32768) for _ in range(4)]
        out["four streams (%s in {2k, 2k+1}), 32768 rows each" % name] = run([(sts4[k], probs4[k], 0) for k in range(4)])
        del probs, probs4
    except AssertionError as e:
        print(name, "failed", e)
for k, v in out.items(): print("%-90s %9.1f us per layer chain of %d rows" % (k, v, TOTAL))
json.dump({"masks": res, "us_per_layer_chain_131072_rows": out}, open(os.path.join(os.environ.get("OUT", "gpurun_out"), "r04_cumask_probe.json"), "w"), indent=1)

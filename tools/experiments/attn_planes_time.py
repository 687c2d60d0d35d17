"""(XCD=ab: interleaved A/B of the XCD-aware workgroup dealing, DR_ATTN_XCD = 1 / 0, three rounds, outputs compared bitwise.)
Time dr_attention_planes alone (HIP events, images built once) at loop shapes: cfg2 (128 pairs x 2 sides, 256 x 256, d 108), cfg3 (8 pairs,
512 x 512, d 132), cfg5 self / cross segments (2048 / 1024 keys, d 64).  TFLOP/s = 4 L S C per segment / time (fp32-equivalent)."""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = "cuda:0"
raw = lib.raw()
cases = [("cfg2 self, 256 segments of 256x256, d108", 256, 256, 256, 4, 108), ("cfg3 self, 16 segments of 512x512, d132", 16, 512, 512, 4, 132),
         ("cfg5 image self, 8 segments of 2048x2048, d64", 8, 2048, 2048, 4, 64), ("cfg5 point self, 8 x 1024x1024, d64", 8, 1024, 1024, 4, 64),
         ("cfg5 cross image<-points, 8 x 2048x1024, d64", 8, 2048, 1024, 4, 64), ("cfg5 cross points<-image, 8 x 1024x2048, d64", 8, 1024, 2048, 4, 64),
         ("cfg5 one pair image self 2048x2048", 1, 2048, 2048, 4, 64)]
only = os.environ.get("CASE")
res = []
for name, P, Lq, Lk, H, d in cases:
    if only and only not in name:
        continue
    dp = (d + 15) // 16 * 16
    torch.manual_seed(0)
    mk = lambda L: torch.randn(P * L, H * dp, device=dev)
    q, k, v = mk(Lq), mk(Lk), mk(Lk)
    qi, qb = lib.planes_from_f32(q)
    kb_in = k.view(P, -1).abs().amax(1).repeat_interleave(Lk)
    vb_in = v.view(P, -1).abs().amax(1).repeat_interleave(Lk)
    ki, kb = lib.planes_from_f32_bounded(k, kb_in)
    vi, vb = lib.planes_from_f32_bounded(v, vb_in)
    oi = torch.zeros(raw.dr_plane_image_bytes(P * Lq, H * dp), dtype=torch.uint8, device=dev)
    ob = torch.zeros(P * Lq, device=dev)
    run = lambda: lib.check(raw.dr_attention_planes(P, Lq, Lk, H, d, lib.ptr(qi), lib.ptr(qb), lib.ptr(ki), lib.ptr(kb), lib.ptr(vi), lib.ptr(vb), None, None,
                                                    lib.ptr(oi), lib.ptr(ob), lib.stream_of(q)))
    def timed():
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    fl = 4.0 * P * Lq * Lk * H * d
    if os.environ.get("XCD") == "ab":
        raw.dr_debug_enable_env(1)
        t = {"1": [], "0": []}
        outs = {}
        for rnd in range(3):
            for m in ("1", "0"):
                os.environ[os.environ.get("ABKNOB", "DR_ATTN_XCD")] = m
                oi.zero_()
                t[m].append(round(timed(), 1))
                outs[m] = oi.clone()
        if os.environ.get("ABKNOB", "DR_ATTN_XCD") == "DR_ATTN_XCD":
            assert torch.equal(outs["1"], outs["0"]), "the dealing changed the result"
        res.append(dict(case=name, us_xcd_groups=t["1"], us_plain=t["0"], TFLOPs_xcd=fl / min(t["1"]) / 1e6, TFLOPs_plain=fl / min(t["0"]) / 1e6))
        print(res[-1], flush=True)
        continue
    us = timed()
    res.append(dict(case=name, us_per_launch=us, TFLOPs=fl / us / 1e6, workgroups=P * H * ((Lq + 127) // 128)))
    print(res[-1], flush=True)
print(json.dumps(res))

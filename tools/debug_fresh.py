import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
from diffreg_hip import synth, lib
from oracle import diffreg_oracle as orc
from tests.helpers import T, weights, pair, masks
from tests.test_loop_gpu import engine
DEV = "cuda:0"
variant, N, M, steps, mc, seed = "3dmatch", 160, 144, 4, 200, 77
v = synth.VARIANTS[variant]; W = weights(variant)
_, p = pair(variant, N, M, seed); ms, mt = masks(N, M)
trace = []
ref = orc.denoise_loop(W, v, p["f_s"], p["f_t"], p["p_s"], p["p_t"], ms, mt, p["x_T"], steps, mc, variant=variant, trace=trace)
eng = engine(variant, steps, mc)
out = eng.run(p["f_s"].to(DEV), p["f_t"].to(DEV), p["p_s"].to(DEV), p["p_t"].to(DEV), p["x_T"].to(DEV), trace=True)
for k in range(steps):
    d = (out["x0"][k, 0].cpu() - trace[k]["x0"][0]).abs()
    print(k, "x0 diff max %.3e n>1e-4 %d" % (d.max(), (d > 1e-4).sum()), "Rf diff %.2e" % (out["R_forwd"][k, 0].cpu() - trace[k]["R_forwd"][0]).abs().max(),
          "cond", out["cond"][k, 0].item(), trace[k]["cond"].item())
# single denoiser evaluation with identity warp
so, to, conf = eng.denoise_match(p["f_s"].to(DEV), p["f_t"].to(DEV), p["p_s"].to(DEV), p["p_t"].to(DEV))
hs, ht, pe_s, pe_t = orc.denoiser(W, v, p["f_s"], p["f_t"], p["p_s"], p["p_t"], ms, mt)
print("denoiser src diff %.3e tgt diff %.3e" % ((so.cpu() - hs).abs().max(), (to.cpu() - ht).abs().max()))
cref = orc.match_head(W, v, hs, ht, pe_s, pe_t, ms, mt)
print("conf diff %.3e" % (conf.cpu() - cref).abs().max())
# layer by layer
C, H = v["C"], v["H"]
half = lambda cs: (cs[0][..., 0::2].reshape(-1, C // 2).contiguous().to(DEV), cs[1][..., 0::2].reshape(-1, C // 2).contiguous().to(DEV))
cs_, ss_ = half(pe_s); ct_, st_ = half(pe_t)
pre = "denoising_transformer.layers.0."
tens = [W[pre + k].to(DEV) for k in lib._LAYER_KEYS]
a = lib.attention_layer(tens, C, H, p["f_s"].to(DEV), p["f_s"].to(DEV), cs_, ss_, cs_, ss_).cpu()
b = orc.attention_layer(W, pre, p["f_s"], p["f_s"], pe_s, pe_s, None, None, H)
print("self src layer diff %.3e" % (a - b).abs().max())
a = lib.attention_layer(tens, C, H, p["f_t"].to(DEV), p["f_t"].to(DEV), ct_, st_, ct_, st_).cpu()
b = orc.attention_layer(W, pre, p["f_t"], p["f_t"], pe_t, pe_t, None, None, H)
d = (a - b).abs()[0]
print("self tgt layer diff %.3e" % d.max(), "rows bad:", torch.nonzero(d.max(1)[0] > 1e-3).flatten().tolist()[:20])
a = lib.attention_layer(tens, C, H, p["f_s"].to(DEV), p["f_t"].to(DEV), cs_, ss_, ct_, st_).cpu()
b = orc.attention_layer(W, pre, p["f_s"], p["f_t"], pe_s, pe_t, None, None, H)
d = (a - b).abs()[0]
print("cross s<-t layer diff %.3e" % d.max(), "rows bad:", torch.nonzero(d.max(1)[0] > 1e-3).flatten().tolist()[:20])

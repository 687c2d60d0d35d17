"""Diagnostic (not a test): the device's matching-head backward on the stress head, step by step against a float64 evaluation of the same chain
(oracle functions + torch autograd on the CPU) on the reference's layer outputs.  For every step: the device's value end to end, and the step
alone on float64-exact inputs rounded to float32 -- which step spends the accuracy.  python tests/debug_stress_head_steps.py   (needs a GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip import lib, synth
from diffreg_hip.autograd import _tables
from oracle import diffreg_oracle as orc, train_oracle as tro
from models.pipeline import Pipeline
from tests.helpers import train_case, train_weights
from tests.test_models_api_gpu import StubBackbone, ref_like_config
DEV = "cuda:0"
GOLD = os.path.join(ROOT, "tests", "golden")
UP, G, HC = (np.load(os.path.join(GOLD, f)) for f in ("train_backward_upstream.npz", "train_forward.npz", "train_backward_head_control.npz"))
c = train_case("b1")
model = Pipeline(ref_like_config("3dmatch", 20, c["mc"]), backbone=StubBackbone())
sd = model.state_dict()
for k, a in train_weights("main").items():
    sd[k] = a
model.load_state_dict(sd); model = model.to(DEV)
tr, head = model.denoising_transformer, model.denoising_coarse_matching
sm, tm = c["src_mask"], c["tgt_mask"]
with torch.no_grad():
    src_pe, tgt_pe = tr.positional_encoding(torch.from_numpy(G["b1_src_warped"]).to(DEV)), tr.positional_encoding(c["p_t"].to(DEV))
cs, ss = _tables(src_pe); ct, st = _tables(tgt_pe)
C = head.src_proj.weight.shape[0]
iters = int(head.skh_iters)
# ---- float64 chain on the CPU
d = torch.float64
full = lambda t_: torch.repeat_interleave(t_.double().cpu(), 2, dim=-1) if t_.shape[-1] == C // 2 else t_.double().cpu()
pe_s64, pe_t64 = (full(cs), full(ss)), (full(ct), full(st))
s64 = torch.from_numpy(UP["branch_out_src32"]).to(d).requires_grad_(True)
t64 = torch.from_numpy(UP["branch_out_tgt32"]).to(d).requires_grad_(True)
W64 = head.src_proj.weight.detach().cpu().to(d).requires_grad_(True)
bin64 = head.bin_score.detach().cpu().to(d).requires_grad_(True)
a64 = orc.embed_pos(s64 @ W64.T, pe_s64) / C ** 0.5
b64 = orc.embed_pos(t64 @ W64.T, pe_t64) / C ** 0.5
sim64 = torch.einsum("bsc,btc->bst", a64, b64)
conf64 = orc.sinkhorn_conf(sim64, bin64, iters, sm, tm)
gt = torch.zeros_like(conf64); gt[0][c["matches"][0][0], c["matches"][0][1]] = 1
loss64 = tro.focal_loss(conf64, gt)
for x in (a64, b64, sim64, conf64):
    x.retain_grad()
loss64.backward()
print("float64 chain vs the reference module's float64 (h64): conf %.2e, up_src %.2e of max" % (
    np.abs(conf64.detach().numpy() - HC["h64_conf"]).max(), np.abs(s64.grad.numpy() - HC["h64_up_src"]).max() / np.abs(HC["h64_up_src"]).max()))
rel = lambda got, ref: float((got.double().cpu() - ref).abs().max() / ref.abs().max())
f32 = lambda x: x.detach().float().to(DEV).contiguous()
B, N, M = 1, s64.shape[1], t64.shape[1]
smd, tmd = sm.to(DEV), tm.to(DEV)
# ---- device, end to end
sf, tf, W = f32(s64), f32(t64), f32(W64)
spre, tpre = lib.linear(sf.reshape(N, C), W), lib.linear(tf.reshape(M, C), W)
a = lib.rotary(spre, cs, ss, scale=1.0 / C ** 0.5).view(B, N, C); b = lib.rotary(tpre, ct, st, scale=1.0 / C ** 0.5).view(B, M, C)
sim = lib.bmm_nt(a, b)
conf = lib.sinkhorn(sim, f32(bin64).reshape(1), iters, smd, tmd)
gconf = lib.focal_loss_backward(conf, gt.float().to(DEV))
gs, ga = lib.sinkhorn_backward(sim, f32(bin64), iters, smd, tmd, gconf)
print("END TO END   a %.2e  sim abs %.2e (max |sim| %.0f)  conf abs %.2e  gconf %.2e  gs %.2e" % (
    rel(a, a64.detach()), float((sim.double().cpu() - sim64.detach()).abs().max()), float(sim64.abs().max()),
    float((conf.double().cpu() - conf64.detach()).abs().max()), rel(gconf, conf64.grad), rel(gs, sim64.grad)))
# ---- every step alone on float64-exact inputs rounded to float32
conf_t = lib.sinkhorn(f32(sim64), f32(bin64).reshape(1), iters, smd, tmd)
print("STEP ALONE   sinkhorn forward on sim64: conf abs %.2e" % float((conf_t.double().cpu() - conf64.detach()).abs().max()))
gconf_t = lib.focal_loss_backward(f32(conf64), gt.float().to(DEV))
print("STEP ALONE   focal backward on conf64: %.2e of max" % rel(gconf_t, conf64.grad))
gs_t, _ = lib.sinkhorn_backward(f32(sim64), f32(bin64), iters, smd, tmd, f32(conf64.grad))
print("STEP ALONE   sinkhorn backward on (sim64, gconf64): %.2e of max" % rel(gs_t, sim64.grad))
gs_o, _ = tro.sinkhorn_backward(sim64.detach().float(), bin64.detach().float(), iters, sm, tm, conf64.grad.float())
print("             (the oracle's float32 adjoint recurrences on the same: %.2e ; torch autograd float32: see h32 control)" % rel(gs_o, sim64.grad))
trn = lambda x: x.transpose(-1, -2).contiguous()
g_a = lib.bmm_nt(f32(sim64.grad), trn(f32(b64)))
print("STEP ALONE   g_a = gs b on (gs64, b64): %.2e of max" % rel(g_a, a64.grad))
g_sp = lib.rotary(f32(a64.grad).reshape(N, C), cs, ss, inverse=True, scale=1.0 / C ** 0.5)
g_tp = lib.rotary(f32(b64.grad).reshape(M, C), ct, st, inverse=True, scale=1.0 / C ** 0.5)
g_src = lib.linear(g_sp, trn(W)).view(B, N, C)
print("STEP ALONE   rotary^T + g W on a64.grad: %.2e of max" % rel(g_src, s64.grad))
pad4 = lambda x: torch.nn.functional.pad(x, (0, (-x.shape[1]) % 4))
g_W = lib.linear(pad4(trn(g_sp)), pad4(trn(sf.reshape(N, C)))) + lib.linear(pad4(trn(g_tp)), pad4(trn(tf.reshape(M, C))))
print("STEP ALONE   weight gradient on (a64.grad, b64.grad): %.2e of max ; the two sides' maxima %.2e / %.2e against the sum's %.2e" % (
    rel(g_W, W64.grad), float(lib.linear(pad4(trn(g_sp)), pad4(trn(sf.reshape(N, C)))).abs().max()),
    float(lib.linear(pad4(trn(g_tp)), pad4(trn(tf.reshape(M, C)))).abs().max()), float(W64.grad.abs().max())))
# ---- the chain from the device's own conf, later steps exact: what the forward's deviation alone costs
conf_dev64 = conf.double().cpu().requires_grad_(True)
l2 = tro.focal_loss(conf_dev64, gt); l2.backward()
print("focal gradient at the device's conf (float64 arithmetic) vs at conf64: %.2e of max" % rel(conf_dev64.grad, conf64.grad))

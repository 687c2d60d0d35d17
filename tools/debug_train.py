"""training forward on the device vs the reference vectors: where the first deviation appears"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from diffreg_hip import synth
from oracle import train_oracle as tro, diffreg_oracle as orc
from tests.helpers import train_case, train_weights
from tests.test_models_api_gpu import StubBackbone, ref_like_config
from models.pipeline import Pipeline
DEV = "cuda:0"
G = np.load(os.path.join(ROOT, "tests/golden/train_forward.npz"))
tag = "b1"
c = train_case(tag)
B, N, M = c["B"], c["N"], c["M"]
v = synth.VARIANTS["3dmatch"]
W = train_weights()
model = Pipeline(ref_like_config("3dmatch", 20, c["mc"]), backbone=StubBackbone())
sd = model.state_dict()
for k, a in W.items(): sd[k] = a
model.load_state_dict(sd); model = model.to(DEV).train()
d = {}
fs, ft, ps, pt = (c[k].to(DEV) for k in ("f_s", "f_t", "p_s", "p_t"))
sm, tm = c["src_mask"].to(DEV), c["tgt_mask"].to(DEV)
a_s, a_t, pe_s, pe_t = model.coarse_transformer(fs, ft, ps, pt, sm, tm, d)
# oracle, step by step
C, H = v["C"], v["H"]
o_pe_s = orc.vol_pe(c["p_s"], C, v["origin"], v["voxel"]); o_pe_t = orc.vol_pe(c["p_t"], C, v["origin"], v["voxel"])
f_s, f_t = c["f_s"], c["f_t"]
pre = "coarse_transformer.layers."
f_s = orc.attention_layer(W, pre + "0.", f_s, f_s, o_pe_s, o_pe_s, c["src_mask"], c["src_mask"], H)
f_t = orc.attention_layer(W, pre + "0.", f_t, f_t, o_pe_t, o_pe_t, c["tgt_mask"], c["tgt_mask"], H)
f_s = orc.attention_layer(W, pre + "1.", f_s, f_t, o_pe_s, o_pe_t, c["src_mask"], c["tgt_mask"], H)
f_t = orc.attention_layer(W, pre + "1.", f_t, f_s, o_pe_t, o_pe_s, c["tgt_mask"], c["src_mask"], H)
conf = orc.match_head(W, v, f_s, f_t, o_pe_s, o_pe_t, c["src_mask"], c["tgt_mask"], prefix=pre + "2.0.")
pl = d["position_layers"][1]
print("positioning conf: hip vs oracle max", (pl["conf_matrix"].cpu() - conf).abs().max().item(), "conf max", conf.max().item())
R, t, Rf, tf, cond, ok = orc.procrustes(conf, c["p_s"], c["p_t"], c["src_mask"], c["tgt_mask"], v["sample_rate"], c["mc"])
print("positioning R: hip vs oracle", (pl["R_s2t_pred"].cpu() - R).abs().max().item(), "cond", float(pl["condition"][0]), float(cond[0]), "ok", pl["solution_mask"].cpu(), ok)
R2 = orc.procrustes(pl["conf_matrix"].cpu(), c["p_s"], c["p_t"], c["src_mask"], c["tgt_mask"], v["sample_rate"], c["mc"])[0]
print("oracle procrustes on the hip conf vs hip R", (pl["R_s2t_pred"].cpu() - R2).abs().max().item())
cf = conf.contiguous()
res = lib_proc = None
from diffreg_hip import lib
r = lib.procrustes(cf.to(DEV), ps, pt, sm, tm, 1.0, c["mc"], want_topk=True)
idx = r[6][0].cpu().long()
w, i, j = orc.topk_pairs(cf, 96)
ref_set = set((i[0] * M + j[0]).tolist()); got_set = set(idx.tolist())
print("hip R on oracle conf vs oracle R", (r[0].cpu() - R).abs().max().item(), "set diff", len(ref_set ^ got_set), "K", len(got_set), len(idx))
flat = cf.view(-1)
print("ref-only", sorted([(int(e), float(flat[e])) for e in ref_set - got_set])[:8])
print("hip-only", sorted([(int(e), float(flat[e])) for e in got_set - ref_set])[:8])
print("kth value", float(w[0, -1]), "count >= kth", int((flat >= w[0, -1]).sum()))

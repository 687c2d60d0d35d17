"""debug: the stress head's gradient chain taken apart (see tests/test_train_gpu.py::test_stress_head_gradient_chain_taken_apart)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip.autograd import _layers_of, matching_head_form, focal_loss
from models.pipeline import Pipeline
from tests.helpers import train_case, train_weights
from tests.test_models_api_gpu import StubBackbone, ref_like_config
DEV = "cuda:0"
GOLD = os.path.join(ROOT, "tests", "golden")
UP, GB, GPm, G = (np.load(os.path.join(GOLD, f)) for f in ("train_backward_upstream.npz", "train_backward.npz", "train_backward_params.npz", "train_forward.npz"))
c = train_case("b1")
model = Pipeline(ref_like_config("3dmatch", 20, c["mc"]), backbone=StubBackbone())
sd = model.state_dict()
for k, a in train_weights("main").items():
    sd[k] = a
model.load_state_dict(sd); model = model.to(DEV)
tr = model.denoising_transformer
sm, tm = c["src_mask"].to(DEV), c["tgt_mask"].to(DEV)
print("masks: src %d of %d, tgt %d of %d" % (int(sm.sum()), sm.numel(), int(tm.sum()), tm.numel()))
with torch.no_grad():
    src_pe, tgt_pe = tr.positional_encoding(torch.from_numpy(G["b1_src_warped"]).to(DEV)), tr.positional_encoding(c["p_t"].to(DEV))
def mx(a): return float(np.abs(np.asarray(a)).max())
# end to end
fs = (c["f_s"] * 0.5).to(DEV).requires_grad_(True); ft = (c["f_t"] * 0.5).to(DEV).requires_grad_(True)
s, t, _, _ = _layers_of(tr, fs, ft, src_pe, tgt_pe, sm, tm)
s.retain_grad(); t.retain_grad()
hat = matching_head_form(model.denoising_coarse_matching, s, t, src_pe, tgt_pe, sm, tm, tr.pe_type)
gt = torch.zeros_like(hat); gt[0][c["matches"][0][0].to(DEV), c["matches"][0][1].to(DEV)] = 1
focal_loss(hat, gt).backward()
print("upstream: max |up32| src %.3e tgt %.3e ; device's own upstream differs from the reference's by %.3e / %.3e ; ref32-ref64 %.3e / %.3e"
      % (mx(UP["branch_up_src32"]), mx(UP["branch_up_tgt32"]), mx(s.grad.cpu().numpy() - UP["branch_up_src32"]), mx(t.grad.cpu().numpy() - UP["branch_up_tgt32"]),
         mx(UP["branch_up_src32"] - UP["branch_up_src64"]), mx(UP["branch_up_tgt32"] - UP["branch_up_tgt64"])))
print("end to end: fs.grad max %.3e, |dev - ref32| %.3e, |dev - ref64| %.3e" % (mx(GB["branch_grad_src"]), mx(fs.grad.cpu().numpy() - GB["branch_grad_src"]), mx(fs.grad.cpu().numpy() - GPm["branch_grad_src64"])))
e2e = fs.grad.clone()
# masked rows of the reference's upstream
print("reference's upstream on masked rows: src %.3e tgt %.3e" % (mx(UP["branch_up_src32"][0][~c["src_mask"][0].numpy()]) if (~c["src_mask"]).any() else 0.0,
                                                                mx(UP["branch_up_tgt32"][0][~c["tgt_mask"][0].numpy()]) if (~c["tgt_mask"]).any() else 0.0))
for prm in model.parameters():
    prm.grad = None
fs2 = (c["f_s"] * 0.5).to(DEV).requires_grad_(True); ft2 = (c["f_t"] * 0.5).to(DEV).requires_grad_(True)
s2, t2, _, _ = _layers_of(tr, fs2, ft2, src_pe, tgt_pe, sm, tm)
((s2 * torch.from_numpy(UP["branch_up_src32"]).to(DEV)).sum() + (t2 * torch.from_numpy(UP["branch_up_tgt32"]).to(DEV)).sum()).backward()
print("from the reference's upstream: |dev - ref32| %.3e, |dev - ref64| %.3e ; vs the device's own end-to-end %.3e" % (
    mx(fs2.grad.cpu().numpy() - GB["branch_grad_src"]), mx(fs2.grad.cpu().numpy() - GPm["branch_grad_src64"]), mx((fs2.grad - e2e).cpu().numpy())))
d = np.abs(fs2.grad.cpu().numpy() - GB["branch_grad_src"])[0]
r = d.max(1)
print("rows with the largest deviation:", np.argsort(-r)[:8], r[np.argsort(-r)[:8]], "masked?", (~c["src_mask"][0].numpy())[np.argsort(-r)[:8]])

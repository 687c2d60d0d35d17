"""time of one training step (forward_train + loss + backward) behind a frozen stub backbone, B = 1, N = M = 256"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from diffreg_hip import synth
from tests.helpers import T, train_weights
from tests.test_models_api_gpu import StubBackbone, ref_like_config
from models.loss import MatchMotionLoss
from models.pipeline import Pipeline
DEV = "cuda:0"
N = M = int(os.environ.get("N", "256"))
C = synth.VARIANTS["3dmatch"]["C"]
model = Pipeline(ref_like_config("3dmatch", 20, 200.0), backbone=StubBackbone())
sd = model.state_dict()
for k, a in train_weights().items(): sd[k] = a
model.load_state_dict(sd); model = model.to(DEV).train()
pr = synth.make_pair(N, M, C, seed=50)
feats = (torch.cat([T(pr["src_feats"]), T(pr["tgt_feats"])], 0) * 0.5).to(DEV)
pts = torch.cat([T(pr["s_pcd"]), T(pr["t_pcd"])], 0).to(DEV)
def batch():
    return {"points": [None, None, pts, None], "_feats": feats, "src_mask": torch.ones(1, N, dtype=torch.bool, device=DEV), "tgt_mask": torch.ones(1, M, dtype=torch.bool, device=DEV),
            "src_ind_coarse_split": torch.arange(N, device=DEV), "tgt_ind_coarse_split": torch.arange(M, device=DEV), "src_ind_coarse": torch.arange(N, device=DEV),
            "tgt_ind_coarse": torch.arange(N, N + M, device=DEV), "coarse_matches": [T(pr["gt_matches"]).t().contiguous().to(DEV)],
            "batched_rot": T(pr["R_gt"]).float()[None].to(DEV), "batched_trn": T(pr["t_gt"]).float().view(1, 3, 1).to(DEV)}
crit = MatchMotionLoss(dict(focal_alpha=0.25, focal_gamma=2.0, pos_weight=1.0, neg_weight=1.0, motion_loss_type="L1", motion_weight=0.1, match_weight=1, match_type="sinkhorn",
                            positioning_type="procrustes", confidence_threshold_metric=0.05, mutual_nearest=False, inlier_thr=0.1, fmr_thr=0.05, registration_threshold=0.2, dataset="3dmatch"))
def step():
    for p in model.parameters(): p.grad = None
    info = crit.forward_train(model.forward_train(batch()))
    info["loss"].backward()
    return float(info["loss"].detach())
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = step()
torch.cuda.synchronize()
print("N = M = %d: %.1f ms per training step (forward + loss + backward), loss %.5f" % (N, (time.perf_counter() - t0) / 5 * 1e3, l))
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): crit(model(batch()))
    torch.cuda.synchronize()
print("value-only forward + loss: %.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3))

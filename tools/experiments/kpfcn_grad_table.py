"""per-tensor deviation of the device backbone gradients from the reference backbone's (tests/golden/kpfcn_coarse.npz)"""
import importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
from diffreg_hip import synth
g = np.load(os.path.join(ROOT, "tests", "golden", "kpfcn_coarse.npz"))
kp = {k[3:]: g[k] for k in g.files if k.startswith("kp:")}
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
sd = {k: T(v) for k, v in synth.make_kpfcn_weights(kp).items()}
b = synth.make_kpfcn_batch()
tb = dict(points=[T(p) for p in b["points"]], neighbors=[T(p) for p in b["neighbors"]], pools=[T(p) for p in b["pools"]], upsamples=[T(p) for p in b["upsamples"]], features=T(b["features"]))
spec = importlib.util.spec_from_file_location("dr_models_backbone", os.path.join(ROOT, "diff-reg_amd", "models", "backbone.py"))
mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
cfg = dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), KP_influence="linear", aggregation_mode="sum", deformable=False, use_batch_norm=True, fine_feature_dim=264)
net = mb.KPFCN(cfg); net.load_state_dict(sd, strict=False); net = net.cuda().train()
db = {k: [t.cuda() for t in v] if isinstance(v, list) else v.cuda() for k, v in tb.items()}
out = net(db, phase="coarse")
G = T(synth.hash_normal(77, 1, tuple(out.shape)).astype(np.float32)).cuda()
(out * G).sum().backward()
P = dict(net.named_parameters())
for k in sorted(x[6:] for x in g.files if x.startswith("gnorm:")):
    got = P[k].grad.double().reshape(-1).cpu()
    idx, val, gmax, norm = g["gidx:" + k], g["gval:" + k], float(g["gmax:" + k]), float(g["gnorm:" + k])
    print("%-48s norm rel %.2e   sampled max / gmax %.2e   shape %s" % (k, abs(float(got.norm()) - norm) / norm, float((got[T(idx)] - T(val).double()).abs().max()) / gmax, tuple(P[k].shape)))

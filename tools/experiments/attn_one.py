"""A few launches of the attention layer at a batch that selects the 128-query kernels: target of rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib, synth
lib.ensure_init()
P, L, C, H = int(os.environ.get("P", "64")), 256, 432, 4
W = {k: torch.from_numpy(a).cuda() for k, a in synth.make_weights(C, seed=7, head_gain=24.0).items()}
tens = [W["denoising_transformer.layers.0." + k] for k in lib._LAYER_KEYS]
x = torch.randn(P, L, C, device="cuda"); y = torch.randn(P, L, C, device="cuda")
ang = torch.rand(P * L, C // 2, device="cuda") * 6.28
cs, sn = ang.cos().contiguous(), ang.sin().contiguous()
for _ in range(int(os.environ.get("N", "5"))):
    out = lib.attention_layer(tens, C, H, x, y, cs, sn, cs, sn)
torch.cuda.synchronize()

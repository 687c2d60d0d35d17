"""2D-3D variant: dr_denoise_loop_2d3d against the reference-minted vectors.  Needs a GPU."""
import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import diffreg_oracle as orc
from tests.helpers import T, masks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def setup(N, M, seed, steps, mc):
    from diffreg_hip.engine import DenoiseEngine2D3D
    Wn = synth.make_weights_2d3d(seed=9, head_gain=16.0)
    W = {k: T(a) for k, a in Wn.items()}
    eng = DenoiseEngine2D3D(W, steps=steps, max_condition_num=mc, device=DEV)
    pr = synth.make_pair_2d3d(N, M, seed, weights=Wn)
    q = lambda k: T(pr[k])[None]
    return W, eng, q


@pytest.mark.parametrize("N,M,nv,mv,mv_da,steps,mc,seed,tag", [(96, 160, 90, 150, 141, 3, 200, 31, "n96x160_s3_masked"),
                                                                 (128, 192, 128, 192, 192, 10, 0, 32, "n128x192_s10_mc0")])
def test_2d3d_against_reference_vectors(golden, N, M, nv, mv, mv_da, steps, mc, seed, tag):
    g = golden("2d3d_loop_" + tag)
    W, eng, q = setup(N, M, seed, steps, mc)
    ms, mt = masks(N, M, nv, mv)
    mt_da = torch.arange(M)[None] < mv_da
    dmask = (ms.to(DEV), mt.to(DEV), mt_da.to(DEV))
    d = lambda k: q(k).to(DEV)
    # one fusion + matching evaluation on the un-warped points
    f_img, f_pcd, c0 = eng.fuse_and_match(d("img_feats"), d("img_dino"), d("img_pixels"), d("pcd_feats"), d("s_pcd"), dmask)
    assert np.abs(f_img[0].cpu().numpy() - g["f_img0"]).max() < 2e-4
    assert np.abs(f_pcd[0].cpu().numpy() - g["f_pcd0"]).max() < 2e-4
    assert np.abs(c0[0].cpu().numpy() - g["conf0"]).max() < 1e-4
    # the loop
    out = eng.run(d("img_feats"), d("img_dino"), d("img_pixels"), d("pcd_feats"), d("s_pcd"), d("t_pcd_da"), d("x_T"), dmask, trace=True)
    assert out["conf_matrix_pred"].dtype == torch.float64
    assert np.abs(out["R_forwd"][:, 0].cpu().numpy() - g["R_forwd"]).max() < 1e-4
    assert np.abs(out["t_forwd"][:, 0].cpu().numpy() - g["t_forwd"]).max() < 1e-4
    x0 = out["x0"][:, 0].cpu().numpy()
    assert (np.abs(x0[-1] - g["x0_last"]) > 1e-4).mean() <= 1e-3
    conf = out["conf_matrix_pred"][0].cpu().numpy()
    assert (np.abs(conf - g["conf"]) > 1e-4).mean() <= 1e-3
    rel = np.abs(conf - g["conf"]) / np.maximum(g["conf"], 1e-9)
    assert np.quantile(rel[g["conf"] > 1e-6], 0.999) < 5e-3
    # read-out = top-1 union of the library's own conf (bit-exact index work)
    cnt = int(out["match_count"][0])
    got = set(map(tuple, out["matches_padded"][0, :cnt, 1:].cpu().tolist()))
    assert got == set(map(tuple, orc.top1_union(out["conf_matrix_pred"][0].cpu())[:, 1:].tolist()))


def test_2d3d_two_pairs_equal_single_pairs():
    W, eng, q1 = setup(96, 160, 41, 2, 200)
    _, _, q2 = setup(96, 160, 42, 2, 200)
    cat = lambda k: torch.cat([q1(k), q2(k)]).to(DEV)
    both = eng.run(cat("img_feats"), cat("img_dino"), cat("img_pixels"), cat("pcd_feats"), cat("s_pcd"), cat("t_pcd_da"), cat("x_T"))
    c_both = both["conf_matrix_pred"].clone()
    for i, q in enumerate((q1, q2)):
        d = lambda k: q(k).to(DEV)
        one = eng.run(d("img_feats"), d("img_dino"), d("img_pixels"), d("pcd_feats"), d("s_pcd"), d("t_pcd_da"), d("x_T"))
        assert (one["conf_matrix_pred"][0] - c_both[i]).abs().max().item() < 1e-6


def test_cfg5_1024x2048_10_steps():
    """BASELINE configs[4] at its stated size: N = 1024 point nodes x M = 2048 image patches (tiles beyond the register-resident
    Sinkhorn / Procrustes paths), 10 denoise steps, padding masks on both sides and a different (non-trivial) tgt_mask_da for the
    warp, against the oracle step by step: the -inf persistence of masked entries (quirk Q8), the fp64 state from step 2 on and
    the Procrustes feedback all run through the large-tile kernels for the whole loop."""
    N, M, steps, mc = 1024, 2048, 10, 200
    W, eng, q = setup(N, M, 51, steps, mc)
    ms, mt = masks(N, M, 1000, 2000)
    mt_da = torch.arange(M)[None] < 1900
    d = lambda k: q(k).to(DEV)
    out = eng.run(d("img_feats"), d("img_dino"), d("img_pixels"), d("pcd_feats"), d("s_pcd"), d("t_pcd_da"), d("x_T"),
                  (ms.to(DEV), mt.to(DEV), mt_da.to(DEV)), trace=True)
    tr = []
    ref = orc.denoise_loop_2d3d(W, synth.VARIANTS["2d3d"], q("img_feats"), q("img_dino"), q("img_pixels"), q("pcd_feats"), q("s_pcd"),
                                q("t_pcd_da"), ms, mt, mt_da, q("x_T"), steps, mc, trace=tr)
    assert len(tr) == steps
    for k in range(steps):
        assert (out["R_forwd"][k, 0].cpu() - tr[k]["R_forwd"][0]).abs().max().item() < 1e-4, k
        assert (out["t_forwd"][k, 0].cpu() - tr[k]["t_forwd"][0]).abs().max().item() < 1e-4, k
        dx = (out["x0"][k, 0].cpu() - tr[k]["x0"][0]).abs()
        assert (dx > 1e-4).float().mean().item() <= 1e-3, (k, dx.max().item())
    # the state after 10 steps: entries masked by the warp helper (src_mask x tgt_mask_da, in place: quirk Q8) are not finite
    # in the oracle and here alike, the others agree
    xf, xr = out["x_final"][0].cpu(), ref["x_final"][0].double()
    valid = ms[0][:, None] & (mt[0] & mt_da[0])[None, :]
    assert torch.isfinite(xr[valid]).all() and not torch.isfinite(xr[~valid]).any()
    assert torch.equal(torch.isfinite(xf), torch.isfinite(xr))
    dxf = (xf[valid] - xr[valid]).abs()
    assert (dxf > 1e-4).double().mean().item() <= 1e-3, dxf.max().item()
    dc = (out["conf_matrix_pred"][0].cpu() - ref["conf_matrix_pred"][0]).abs()
    assert (dc > 1e-4).double().mean().item() <= 1e-3, dc.max().item()
    assert float(dc[valid].max()) < 5e-3

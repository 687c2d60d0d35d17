"""Parity of the individual HIP ops (through the C ABI) with the oracle.  Needs a GPU."""
import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import diffreg_oracle as orc
from tests.helpers import T, weights, pair, masks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def half_tables(cos, sin):
    """oracle tables [B,N,C] (each angle twice) -> the un-duplicated [B*N, C/2] layout of the library."""
    return cos[..., 0::2].reshape(-1, cos.shape[-1] // 2).contiguous(), sin[..., 0::2].reshape(-1, sin.shape[-1] // 2).contiguous()


@pytest.mark.parametrize("rows,ncols,K", [(256, 432, 432), (512, 1296, 432), (100, 864, 864), (37, 432, 864), (1, 528, 528),
                                           (2048, 256, 256), (300, 200, 432), (4096, 864, 864)])
def test_linear_matches_fp64(rows, ncols, K):
    from diffreg_hip import lib
    x = T(synth.hash_normal(3, rows + K, (rows, K))).float()
    W = T(synth.hash_uniform(4, ncols + K, (ncols, K))).float() / K ** 0.5
    ref = (x.double() @ W.double().T)
    got = lib.linear(x.to(DEV), W.to(DEV)).cpu()
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() < 2e-6 * scale * max(1.0, (K / 432) ** 0.5)
    got_relu = lib.linear(x.to(DEV), W.to(DEV), epilogue=1, scale=0.5).cpu()
    assert (got_relu.double() - 0.5 * ref.clamp_min(0)).abs().max().item() < 2e-6 * scale * max(1.0, (K / 432) ** 0.5)


@pytest.mark.parametrize("variant", ["3dmatch", "4dmatch"])
def test_vol_pe_and_rotary_linear(variant, golden):
    from diffreg_hip import lib
    v = synth.VARIANTS[variant]
    C = v["C"]
    _, p = pair(variant, 64, 48, 3)
    cos, sin = orc.vol_pe(p["p_s"], C, v["origin"], v["voxel"])
    hc, hs = half_tables(cos, sin)
    gc, gs = lib.vol_pe(p["p_s"][0].to(DEV), C, v["origin"], v["voxel"])
    assert (gc.cpu() - hc).abs().max().item() < 2e-6 and (gs.cpu() - hs).abs().max().item() < 2e-6
    g = golden(variant + "_pe")
    assert np.abs(gc.cpu().numpy()[:16] - g["cos"][:, 0::2]).max() < 5e-6   # half an ulp of a ~100 rad angle
    # warped: R p + t folded into the kernel (pipeline.py:306)
    R = T(synth._rodrigues(np.array([0.3, -0.5, 0.8]), 0.4)).float()[None]
    t = torch.tensor([[0.11], [-0.07], [0.05]])
    warped = (R @ p["p_s"].transpose(1, 2) + t).transpose(1, 2)
    cw, sw = orc.vol_pe(warped, C, v["origin"], v["voxel"])
    hcw, hsw = half_tables(cw, sw)
    gcw, gsw = lib.vol_pe(p["p_s"][0].to(DEV), C, v["origin"], v["voxel"], R=R.to(DEV), t=t.to(DEV), rows_per_pair=64)
    assert (gcw.cpu() - hcw).abs().max().item() < 5e-5          # 1 ulp of the warp * highest frequency
    # linear + rotary epilogue == rotary(x W^T)
    W = weights(variant)["denoising_transformer.layers.0.q_proj.weight"]
    ref = orc.rotary(p["f_s"] @ W.T, cos, sin)[0]
    got = lib.linear(p["f_s"][0].to(DEV), W.to(DEV), epilogue=2, cos=gc, sin=gs, rot_C=C).cpu()
    assert (got - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("variant", ["3dmatch", "4dmatch"])
def test_attention_layer(variant, golden):
    from diffreg_hip import lib
    v = synth.VARIANTS[variant]
    C, H = v["C"], v["H"]
    Wd = weights(variant)
    pre = "denoising_transformer.layers.1."
    tens = [Wd[pre + k].to(DEV) for k in lib._LAYER_KEYS]
    _, p = pair(variant, 64, 48, 3)
    fs, ft = p["f_s"], p["f_t"]
    pes = orc.vol_pe(p["p_s"], C, v["origin"], v["voxel"])
    pet = orc.vol_pe(p["p_t"], C, v["origin"], v["voxel"])
    cs, ss = [a.to(DEV) for a in half_tables(*pes)]
    ct, st = [a.to(DEV) for a in half_tables(*pet)]
    full, part = masks(64, 48), masks(64, 48, 50, 41)
    g = golden(variant + "_attn_layer")
    cases = dict(self_full=(fs, fs, cs, ss, cs, ss, full[0], full[0]), cross_full=(fs, ft, cs, ss, ct, st, full[0], full[1]),
                 self_mask=(fs, fs, cs, ss, cs, ss, part[0], part[0]), cross_mask=(fs, ft, cs, ss, ct, st, part[0], part[1]))
    # the 32-query kernel, then the 128-query (flash) kernel forced: on the f32-input MFMA and with split operands
    for flash_min, split in ((100000, -1), (1, 0), (1, 1)):
        lib.raw().dr_debug_attention_config(flash_min)
        lib.raw().dr_debug_attention_split(split)
        try:
            for name, (x, y, cx, sx, cy, sy, mx, my) in cases.items():
                got = lib.attention_layer(tens, C, H, x.to(DEV), y.to(DEV), cx, sx, cy, sy, mx.to(DEV), my.to(DEV)).cpu()
                err = np.abs(got[0].numpy() - g[name]).max()
                assert err < 1e-4, (name, flash_min, split, err)          # north_star tolerance: 1e-4 fp32
        finally:
            lib.raw().dr_debug_attention_config(-1)
            lib.raw().dr_debug_attention_split(-1)
    # two pairs in one call == two single calls
    _, p2 = pair(variant, 64, 48, 4)
    x2 = torch.cat([fs, p2["f_s"]]).to(DEV)
    y2 = torch.cat([ft, p2["f_t"]]).to(DEV)
    pes2 = orc.vol_pe(p2["p_s"], C, v["origin"], v["voxel"]); pet2 = orc.vol_pe(p2["p_t"], C, v["origin"], v["voxel"])
    cs2, ss2 = [a.to(DEV) for a in half_tables(*pes2)]; ct2, st2 = [a.to(DEV) for a in half_tables(*pet2)]
    both = lib.attention_layer(tens, C, H, x2, y2, torch.cat([cs, cs2]), torch.cat([ss, ss2]), torch.cat([ct, ct2]),
                               torch.cat([st, st2])).cpu()
    ref2 = orc.attention_layer(Wd, pre, p2["f_s"], p2["f_t"], pes2, pet2, None, None, H)
    assert np.abs(both[0].numpy() - g["cross_full"]).max() < 1e-4
    assert (both[1] - ref2[0]).abs().max().item() < 1e-4


@pytest.mark.parametrize("split", [0, 1, -1])
@pytest.mark.parametrize("Lx,Ly", [(200, 130), (128, 129), (33, 300)])
def test_attention_flash_ragged(Lx, Ly, split):
    """128-query kernel on lengths that are no multiple of its blocks (partial waves, partial key tiles, masks),
    two pairs per call, against the oracle layer.  split = -1: the default rule instead, which at these sizes is the 32-query
    kernel in its 8-wave form (few workgroups, more than four key tiles: one to three tiles per wave, key mask as ballot bits)."""
    from diffreg_hip import lib
    variant = "3dmatch"
    v = synth.VARIANTS[variant]
    C, H = v["C"], v["H"]
    Wd = weights(variant)
    pre = "denoising_transformer.layers.1."
    tens = [Wd[pre + k].to(DEV) for k in lib._LAYER_KEYS]
    xs, ys, pex, pey = [], [], [], []
    for seed in (11, 12):
        _, p = pair(variant, Lx, Ly, seed)
        xs.append(p["f_s"]); ys.append(p["f_t"])
        pex.append(orc.vol_pe(p["p_s"], C, v["origin"], v["voxel"])); pey.append(orc.vol_pe(p["p_t"], C, v["origin"], v["voxel"]))
    mx, my = masks(Lx, Ly, Lx - 7, Ly - 5)
    x2, y2 = torch.cat(xs).to(DEV), torch.cat(ys).to(DEV)
    cx = torch.cat([half_tables(*q)[0] for q in pex]).to(DEV); sx = torch.cat([half_tables(*q)[1] for q in pex]).to(DEV)
    cy = torch.cat([half_tables(*q)[0] for q in pey]).to(DEV); sy = torch.cat([half_tables(*q)[1] for q in pey]).to(DEV)
    m2x, m2y = torch.cat([mx, mx]).to(DEV), torch.cat([my, my]).to(DEV)
    lib.raw().dr_debug_attention_config(1 if split >= 0 else -1)
    lib.raw().dr_debug_attention_split(split)
    try:
        got = lib.attention_layer(tens, C, H, x2, y2, cx, sx, cy, sy, m2x, m2y).cpu()
    finally:
        lib.raw().dr_debug_attention_config(-1)
        lib.raw().dr_debug_attention_split(-1)
    for i in range(2):
        ref = orc.attention_layer(Wd, pre, xs[i], ys[i], pex[i], pey[i], mx, my, H)[0]
        valid = mx[0].bool()
        assert (got[i][valid] - ref[valid]).abs().max().item() < 1e-4


@pytest.mark.parametrize("variant", ["3dmatch", "4dmatch"])
def test_procrustes(variant, golden):
    from diffreg_hip import lib
    g = golden(variant + "_procrustes")
    C = synth.VARIANTS[variant]["C"]
    pr = synth.make_pair(128, 128, C, seed=5)
    gtm = np.zeros((128, 128)); gtm[pr["gt_matches"][:, 0], pr["gt_matches"][:, 1]] = 6.0
    sc = T(gtm + synth.hash_normal(5, 77, (128, 128)))[None].float()
    sm, tm = masks(128, 128, 120, 111)
    conf = orc.sinkhorn_conf(sc, torch.tensor(1.0), 3, sm, tm)
    ps, pt = T(pr["s_pcd"])[None], T(pr["t_pcd"])[None]
    for mc in (0, 40, 200):
        R, t, Rf, tf, cond, ok, idx = lib.procrustes(conf.to(DEV), ps.to(DEV), pt.to(DEV), sm.to(DEV), tm.to(DEV), 1.0, mc,
                                                     use_mask_len=(variant == "4dmatch"), want_topk=True)
        np.testing.assert_allclose(R.cpu().numpy(), g["R_%d" % mc], atol=1e-5)
        np.testing.assert_allclose(t.cpu().numpy(), g["t_%d" % mc], atol=1e-5)
        np.testing.assert_allclose(Rf.cpu().numpy(), g["Rf_%d" % mc], atol=1e-5)
        np.testing.assert_allclose(tf.cpu().numpy(), g["tf_%d" % mc], atol=1e-5)
        np.testing.assert_allclose(cond.cpu().numpy(), g["cond_%d" % mc], rtol=1e-4)
        assert np.array_equal(ok.cpu().numpy(), g["ok_%d" % mc])
    # the selected SET equals the reference's top-K set (bit-exact index work)
    K = idx.shape[1] if variant == "3dmatch" else int(max(120, 111) * 1.0)
    ref_idx = conf.view(1, -1).sort(descending=True, dim=1)[1][0, :K]
    vals = conf.view(-1)
    assert vals[idx[0, :K].cpu().long()].sort()[0].equal(vals[ref_idx].sort()[0])
    assert idx[0, :K].cpu().long().unique().numel() == K                                 # no duplicates


def test_procrustes_degenerate_and_batch():
    from diffreg_hip import lib
    # all-zero confidences: Sxy = 0 -> cond = nan -> identity (procrustes.py:87-91); and a batch of 3 pairs
    P, N, M = 3, 40, 56
    conf = torch.rand(P, N, M, generator=torch.Generator().manual_seed(1))
    conf[1] = 0.0
    ps, pt = torch.rand(P, N, 3), torch.rand(P, M, 3)
    sm, tm = torch.ones(P, N, dtype=torch.bool), torch.ones(P, M, dtype=torch.bool)
    R, t, Rf, tf, cond, ok = lib.procrustes(conf.to(DEV), ps.to(DEV), pt.to(DEV), sm.to(DEV), tm.to(DEV), 1.0, 1e9)
    for b in (0, 2):
        r = orc.procrustes(conf[b:b + 1], ps[b:b + 1], pt[b:b + 1], sm[b:b + 1], tm[b:b + 1], 1.0, 1e9)
        np.testing.assert_allclose(R[b].cpu().numpy(), r[0][0].numpy(), atol=2e-5)
        np.testing.assert_allclose(t[b].cpu().numpy(), r[1][0].numpy(), atol=2e-5)
        assert abs(cond[b].item() / r[4][0].item() - 1) < 1e-3
    assert not ok[1].item()
    assert torch.equal(Rf[1].cpu(), torch.eye(3)) and torch.equal(tf[1].cpu(), torch.zeros(3, 1))


def test_procrustes_thin_tile_more_than_1024_selected():
    """K = max(N, M) * sample_rate > 1024 on a tile of <= 4096 entries (neither candidate list was built and
    the fit read an uninitialised list): N = M = 64 with sample_rate = 20 selects 1280 of the 4096 entries (a thin N = 2048, M = 2
    tile reaches the same branch but its fit is rank deficient: nothing to compare)."""
    from diffreg_hip import lib
    P, N, M = 2, 64, 64
    g = torch.Generator().manual_seed(4)
    conf = torch.rand(P, N, M, generator=g)
    ps = torch.rand(P, N, 3, generator=g)
    pt = torch.rand(P, M, 3, generator=g)
    sm, tm = torch.ones(P, N, dtype=torch.bool), torch.ones(P, M, dtype=torch.bool)
    R, t, Rf, tf, cond, ok = lib.procrustes(conf.to(DEV), ps.to(DEV), pt.to(DEV), sm.to(DEV), tm.to(DEV), 20.0, 1e9)
    for b in range(P):
        r = orc.procrustes(conf[b:b + 1], ps[b:b + 1], pt[b:b + 1], sm[b:b + 1], tm[b:b + 1], 20.0, 1e9)
        np.testing.assert_allclose(R[b].cpu().numpy(), r[0][0].numpy(), atol=1e-4)
        np.testing.assert_allclose(t[b].cpu().numpy(), r[1][0].numpy(), atol=1e-4)


@pytest.mark.parametrize("N,M", [(1024, 2048), (1500, 1500), (300, 700)])
@pytest.mark.parametrize("kind", ["distinct", "quantised", "flat", "sparse"])
def test_procrustes_large_tile_exact_selection(N, M, kind):
    """Tiles beyond 256 x 256 select their K entries with the whole chip (three digit-histogram passes + an ordered take pass,
    procrustes.hip).  Bit-exact index work: the selected VALUES are torch.topk's (procrustes.py:66), no index twice, and among the
    entries equal to the K-th value the lowest indices are the ones taken (the rule the one-workgroup kernels follow; torch.topk
    leaves it implementation-defined).  `quantised` (16 levels) and `flat` overflow any bounded candidate list; `sparse` has fewer
    non-zero entries than K."""
    from diffreg_hip import lib
    P = 2
    g = torch.Generator().manual_seed(N + M)
    conf = torch.rand(P, N, M, generator=g)
    if kind == "quantised":
        conf = (conf * 16).floor() / 16
    elif kind == "flat":
        conf = torch.full((P, N, M), 1.0 / M)
    elif kind == "sparse":
        conf = torch.where(conf > 0.9997, conf, torch.zeros(()))
    ps = torch.rand(P, N, 3, generator=g)
    pt = torch.rand(P, M, 3, generator=g)
    sm, tm = torch.ones(P, N, dtype=torch.bool), torch.ones(P, M, dtype=torch.bool)
    sm[1, N - 37:] = False
    for use_len in (False, True):
        R, t, Rf, tf, cond, ok, idx = lib.procrustes(conf.to(DEV), ps.to(DEV), pt.to(DEV), sm.to(DEV), tm.to(DEV), 1.0, 1e9,
                                                     use_mask_len=use_len, want_topk=True)
        for b in range(P):
            K = max(N, M) if not use_len else max(int(sm[b].sum()), int(tm[b].sum()))
            flat = conf[b].reshape(-1)
            top = flat.topk(K)[0]
            got = idx[b, :K].cpu().long()
            assert got.unique().numel() == K
            assert flat[got].sort(descending=True)[0].equal(top)
            kth = top[-1]
            n_ties = int((top == kth).sum())
            tie_idx = (flat == kth).nonzero()[:, 0][:n_ties]                     # the lowest indices among the ties
            assert got[flat[got] == kth].sort()[0].equal(tie_idx)
            if kind == "flat":
                continue                                                           # rank-deficient fit: nothing to compare
            r = orc.procrustes(conf[b:b + 1], ps[b:b + 1], pt[b:b + 1], sm[b:b + 1], tm[b:b + 1], 1.0, 1e9,
                               variant="4dmatch" if use_len else "3dmatch") \
                if kind == "distinct" else None
            if r is not None:
                np.testing.assert_allclose(R[b].cpu().numpy(), r[0][0].numpy(), atol=1e-4)
                np.testing.assert_allclose(t[b].cpu().numpy(), r[1][0].numpy(), atol=1e-4)


@pytest.mark.parametrize("N,M", [(128, 128), (96, 80), (256, 256), (33, 500), (20, 100), (1024, 2048), (1500, 1500), (1000, 3)])
def test_top1_union(N, M):
    """From 32 rows on the arg-maxima come from row-block workgroups over the whole chip (stateops.hip), below that from one
    workgroup per pair; both list exactly the reference's set, first occurrence winning ties (`quantised`: 8 levels)."""
    from diffreg_hip import lib
    conf = T(synth.hash_u01(8, N * M, N * M)).reshape(1, N, M)
    ref = orc.top1_union(conf[0])
    got = lib.top1_union(conf.to(DEV))[0].cpu()
    assert torch.equal(got, ref)
    got32 = lib.top1_union(conf.float().to(DEV))[0].cpu()
    assert torch.equal(got32, orc.top1_union(conf[0].float()))
    quant = torch.stack([(conf[0] * 8).floor() / 8, conf[0].flip(0)])           # a batch of two different tiles
    got = lib.top1_union(quant.to(DEV))
    for b in range(2):
        ref = orc.top1_union(quant[b])
        assert torch.equal(got[b].cpu(), ref)

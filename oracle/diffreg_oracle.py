"""ORACLE -- CPU restatement of Diff-Reg's reverse-diffusion matching path.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (diff-reg_amd/) never does and fails loudly when the HIP library is missing.

The path's primitive arithmetic lives in PyTorch (pinned torch==1.11.0+cu113 by
Diff-Reg-3dmatch/eccv24_3d_env.yml:209, absent from /root/reference), so the restatement is
written over torch CPU tensors in the same dtypes the reference ends up using (fp32 first step,
fp64 state afterwards -- quirk Q2).  Citations are into /root/reference:
  3D/ = Diff-Reg-3dmatch/, 4D/ = Diff-Reg-4dmatch/.

PINNING: the reference holds no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, generated in the build container by
oracle/make_golden.py (imports /root/reference) and committed as tests/golden/*.npz;
tests/test_oracle_golden.py checks every function below against them.
"""
import math
import torch

F32 = torch.float32
F64 = torch.float64


# ------------------------------------------------------------------------------------------
# schedule  (3D/models/pipeline.py:83-93, 155-162, 229-232)
# ------------------------------------------------------------------------------------------
def diffusion_schedule(T=1000, s=0.008):
    """alphas_cumprod, sqrt(1/ac), sqrt(1/ac - 1), all float64 [T]."""
    grid = torch.linspace(0, T, T + 1, dtype=F64)
    g = torch.cos(((grid / T) + s) / (1 + s) * math.pi * 0.5) ** 2
    g = g / g[0]
    betas = torch.clip(1 - g[1:] / g[:-1], 0, 0.999)
    ac = torch.cumprod(1.0 - betas, dim=0)
    return ac, torch.sqrt(1.0 / ac), torch.sqrt(1.0 / ac - 1)


def time_pairs(steps, T=1000):
    """[(999,949),...,(49,0)] for steps=20 (quirk Q20); linspace is fp32 then truncated."""
    ts = torch.linspace(0, T - 1, steps=steps + 1).int().tolist()
    ts = ts[::-1]
    return list(zip(ts[:-1], ts[1:]))


def ddim_coefficients(ac, t, t_next, eta=1.0):
    """sigma, c of 3D/models/pipeline.py:248-252 as 0-d float64 tensors."""
    a, an = ac[t], ac[t_next]
    sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
    c = (1 - an - sigma ** 2).sqrt()
    return sigma, c, an.sqrt()


# ------------------------------------------------------------------------------------------
# A.1 Sinkhorn with dustbins  (3D/models/matching.py:61-93)
# ------------------------------------------------------------------------------------------
def sinkhorn_log(scores, alpha, iters, src_mask, tgt_mask):
    """scores [B,N,M] (masked entries already -inf) -> log assignment [B,N+1,M+1] in scores.dtype.
    Masks are mandatory: the reference's mask=None branch raises (quirk Q21, matching.py:65-67,79)."""
    B, N, M = scores.shape
    dt = scores.dtype
    alpha = torch.as_tensor(alpha).to(dt)
    # mask sums are int64, so `.log()` yields float32: the marginals are fp32 numbers even when the
    # state is fp64 (quirk Q22, 3D/models/matching.py:69-70,79-82); they are promoted on use.
    rows = src_mask.sum(1, keepdim=True)
    cols = tgt_mask.sum(1, keepdim=True)
    Z = torch.full((B, N + 1, M + 1), 0.0, dtype=dt)
    Z[:, :N, :M] = scores
    Z[:, :N, M] = alpha
    Z[:, N, :] = alpha
    nu0 = -(rows + cols).log()                                      # "norm"  [B,1] float32
    log_mu = torch.cat([nu0.expand(B, N), cols.log() + nu0], 1)     # [B,N+1]
    log_nu = torch.cat([nu0.expand(B, M), rows.log() + nu0], 1)     # [B,M+1]
    u = torch.zeros_like(log_mu)
    v = torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v[:, None, :], dim=2)
        v = log_nu - torch.logsumexp(Z + u[:, :, None], dim=1)      # uses the NEW u
    return Z + u[:, :, None] + v[:, None, :] - nu0[:, :, None]


def pair_mask(src_mask, tgt_mask):
    return src_mask[:, :, None] & tgt_mask[:, None, :]


def sinkhorn_conf(scores, alpha, iters, src_mask, tgt_mask):
    """mask -> Sinkhorn -> exp -> drop dustbins (3D/models/matching.py:207-216)."""
    if src_mask is not None:
        scores = scores.masked_fill(~pair_mask(src_mask, tgt_mask), float("-inf"))
    Z = sinkhorn_log(scores, alpha, iters, src_mask, tgt_mask)
    return Z.exp()[:, :-1, :-1].contiguous()


# ------------------------------------------------------------------------------------------
# A.3 volumetric rotary position code  (3D/models/position_encoding.py:16-35, 49-87)
# ------------------------------------------------------------------------------------------
def vol_pe(xyz, C, origin, voxel):
    """xyz [B,N,3] fp32 -> (cos,sin) each [B,N,C]; per axis C/6 frequencies, each duplicated."""
    B, N, _ = xyz.shape
    vox = (xyz - torch.tensor(origin, dtype=F32).view(1, 1, 3)) / voxel
    d = C // 3
    freq = torch.exp(torch.arange(0, d, 2, dtype=F32) * (-math.log(10000.0) / d)).view(1, 1, -1)
    cs, sn = [], []
    for a in range(3):
        ang = vox[..., a:a + 1] * freq                       # [B,N,d/2]
        cs.append(torch.cos(ang).repeat_interleave(2, dim=-1))
        sn.append(torch.sin(ang).repeat_interleave(2, dim=-1))
    return torch.cat(cs, -1), torch.cat(sn, -1)


def vol_pe_sinusoidal(xyz, C, origin, voxel):
    """pe_type 'sinusoidal' (3D/models/position_encoding.py:56-69): the same sinusoid bank as cat[sin x, cos x, sin y, cos y, sin z, cos z],
    each C / 6 wide, ADDED to the features (embed_pos, :43-44)"""
    vox = (xyz - torch.tensor(origin, dtype=F32).view(1, 1, 3)) / voxel
    d = C // 3
    freq = torch.exp(torch.arange(0, d, 2, dtype=F32) * (-math.log(10000.0) / d)).view(1, 1, -1)
    parts = []
    for a in range(3):
        ang = vox[..., a:a + 1] * freq
        parts += [torch.sin(ang), torch.cos(ang)]
    return torch.cat(parts, -1)


def position_code(cfg, xyz):
    """the code in the form cfg['pe_type'] uses: (cos, sin) tables for 'rotary' (default), one additive tensor for 'sinusoidal'"""
    if cfg.get("pe_type", "rotary") == "sinusoidal":
        return vol_pe_sinusoidal(xyz, cfg["C"], cfg["origin"], cfg["voxel"])
    return vol_pe(xyz, cfg["C"], cfg["origin"], cfg["voxel"])


def embed_pos(x, pe):
    """VolPE.embed_pos (position_encoding.py:38-46): rotary tables rotate, a sinusoidal code is added"""
    return rotary(x, *pe) if isinstance(pe, tuple) else x + pe


def rotary(x, cos, sin):
    """x*cos + swap(x)*sin with swap(x)[2k] = -x[2k+1], swap(x)[2k+1] = x[2k]."""
    sw = torch.stack([-x[..., 1::2], x[..., 0::2]], dim=-1).reshape(x.shape)
    return x * cos + sw * sin


# ------------------------------------------------------------------------------------------
# A.3 attention layer and denoiser  (3D/models/transformero.py:43-96, 151-233)
# ------------------------------------------------------------------------------------------
def layer_norm(x, g, b, eps=1e-5):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), g, b, eps)


def attention_layer(W, pre, x, y, pe_x, pe_y, mask_x, mask_y, H):
    """x [B,L,C] attends to y [B,S,C]; W maps state-dict names (prefix `pre`) to tensors."""
    B, L, C = x.shape
    S = y.shape[1]
    d = C // H
    # pe = (cos, sin): rotary, R w x (transformero.py:59-74); pe = tensor: sinusoidal, w (x + p) on q and k only (:50-57);
    # pe = None: a layer of the entangled form, no code inside (:246-252)
    add = pe_x is not None and not isinstance(pe_x, tuple)
    q = ((x + pe_x) if add else x) @ W[pre + "q_proj.weight"].T
    k = ((y + pe_y) if add else y) @ W[pre + "k_proj.weight"].T
    v = y @ W[pre + "v_proj.weight"].T
    if isinstance(pe_x, tuple):
        q, k = rotary(q, *pe_x), rotary(k, *pe_y)
    q, k, v = q.view(B, L, H, d), k.view(B, S, H, d), v.view(B, S, H, d)
    a = torch.einsum("blhd,bshd->blsh", q, k)
    if mask_y is not None:
        a = a.masked_fill(mask_x[:, :, None, None] & ~mask_y[:, None, :, None], float("-inf"))
    a = torch.softmax(a / d ** 0.5, dim=2)
    o = torch.einsum("blsh,bshd->blhd", a, v).reshape(B, L, C)
    g = layer_norm(o @ W[pre + "merge.weight"].T, W[pre + "norm1.weight"], W[pre + "norm1.bias"])
    h = torch.relu(torch.cat([x, g], 2) @ W[pre + "mlp.0.weight"].T)
    g = layer_norm(h @ W[pre + "mlp.2.weight"].T, W[pre + "norm2.weight"], W[pre + "norm2.bias"])
    return x + g


def denoiser(W, cfg, f_s, f_t, p_s, p_t, mask_s, mask_t, prefix="denoising_transformer."):
    """six layers self,cross,...; same weights for the src and the tgt call; in a cross layer tgt
    attends to the UPDATED src (quirk Q11)."""
    C, H = cfg["C"], cfg["H"]
    pe_s, pe_t = position_code(cfg, p_s), position_code(cfg, p_t)
    if cfg.get("entangled", False):          # transformero.py:234-254: the code enters the features once, the layers get none
        f_s, f_t = embed_pos(f_s, pe_s), embed_pos(f_t, pe_t)
        for l in range(cfg["n_layers"]):
            pre = prefix + "layers.%d." % l
            if l % 2 == 0:
                f_s = attention_layer(W, pre, f_s, f_s, None, None, mask_s, mask_s, H)
                f_t = attention_layer(W, pre, f_t, f_t, None, None, mask_t, mask_t, H)
            else:
                f_s = attention_layer(W, pre, f_s, f_t, None, None, mask_s, mask_t, H)
                f_t = attention_layer(W, pre, f_t, f_s, None, None, mask_t, mask_s, H)
        return f_s, f_t, pe_s, pe_t
    for l in range(cfg["n_layers"]):
        pre = prefix + "layers.%d." % l
        if l % 2 == 0:
            f_s = attention_layer(W, pre, f_s, f_s, pe_s, pe_s, mask_s, mask_s, H)
            f_t = attention_layer(W, pre, f_t, f_t, pe_t, pe_t, mask_t, mask_t, H)
        else:
            f_s = attention_layer(W, pre, f_s, f_t, pe_s, pe_t, mask_s, mask_t, H)
            f_t = attention_layer(W, pre, f_t, f_s, pe_t, pe_s, mask_t, mask_s, H)
    return f_s, f_t, pe_s, pe_t


# ------------------------------------------------------------------------------------------
# A.4 matching head  (3D/models/matching.py:164-219) -- src_proj on BOTH sides (quirk Q1)
# ------------------------------------------------------------------------------------------
def match_head(W, cfg, f_s, f_t, pe_s, pe_t, mask_s, mask_t, prefix="denoising_coarse_matching."):
    C = cfg["C"]
    Wp = W[prefix + "src_proj.weight"]
    a, b = f_s @ Wp.T, f_t @ Wp.T
    if not cfg.get("entangled", False):      # matching.py:181-183
        a, b = embed_pos(a, pe_s), embed_pos(b, pe_t)
    a, b = a / C ** 0.5, b / C ** 0.5
    if cfg.get("match_type", "sinkhorn") == "dual_softmax":      # matching.py:193-205
        s1 = torch.einsum("bsc,btc->bst", a, b) / cfg["dsmax_temperature"]
        if mask_s is None:
            return torch.softmax(s1, 1) * torch.softmax(s1, 2)
        s2 = s1.clone()
        s1 = s1.masked_fill(~mask_s[:, :, None], float("-inf"))
        s2 = s2.masked_fill(~mask_t[:, None, :], float("-inf"))
        return torch.softmax(s1, 1) * torch.softmax(s2, 2)
    sim = torch.einsum("bsc,btc->bst", a, b)
    return sinkhorn_conf(sim, W[prefix + "bin_score"], cfg["skh_iters"], mask_s, mask_t)


# ------------------------------------------------------------------------------------------
# A.2 top-K weighted Procrustes  (3D/models/procrustes.py:17-93; 4D differs at :61-62)
# ------------------------------------------------------------------------------------------
def kabsch(X, Y, w, eps=1e-4):
    """X,Y [B,K,3] fp32, w [B,K,1] fp32 -> R fp32 [B,3,3], t [B,3,1], cond fp64 [B]."""
    wn = w / (w.abs().sum(1, keepdim=True) + eps)
    mx = (wn * X).sum(1, keepdim=True)
    my = (wn * Y).sum(1, keepdim=True)
    S = ((Y - my).transpose(1, 2) @ (wn * (X - mx))).double()
    U, D, Vh = torch.linalg.svd(S)
    V = Vh.transpose(1, 2)
    cond = D.max(1)[0] / D.min(1)[0]
    fix = torch.eye(3, dtype=F64).repeat(X.shape[0], 1, 1)
    fix[:, 2, 2] = torch.linalg.det(U) * torch.linalg.det(V)
    R = (U @ (fix @ V.transpose(1, 2))).float()
    t = my.transpose(1, 2) - R @ mx.transpose(1, 2)
    return R, t, cond


def topk_pairs(conf, K):
    """the K largest entries of conf [B,N,M] (descending, as Tensor.sort) -> (w, i, j)."""
    B, N, M = conf.shape
    val, idx = conf.reshape(B, -1).sort(descending=True, dim=1)
    return val[:, :K], idx[:, :K] // M, idx[:, :K] % M


def procrustes(conf, p_s, p_t, mask_s, mask_t, sample_rate, max_cond, variant="3dmatch"):
    """-> R, t, R_forwd, t_forwd, cond, ok.  conf must be fp32 (the per-step call casts, :302)."""
    B, N, M = conf.shape
    if variant == "4dmatch":
        ls, lt = mask_s.sum(1).float(), mask_t.sum(1).float()
    else:
        ls = torch.full((B,), float(N))
        lt = torch.full((B,), float(M))
    entry_max = (torch.maximum(ls, lt) * sample_rate).int()
    K = int(entry_max.float().mean().int())
    w, i, j = topk_pairs(conf, K)
    w = w.clone()
    w[torch.arange(K).view(1, -1) >= entry_max[:, None]] = 0.0
    bi = torch.arange(B).view(-1, 1).expand(B, K)
    R, t, cond = kabsch(p_s[bi, i], p_t[bi, j], w[..., None])
    ok = cond < max_cond
    Rf, tf = R.clone(), t.clone()
    Rf[~ok] = torch.eye(3)
    tf[~ok] = 0.0
    return R, t, Rf, tf, cond, ok


# ------------------------------------------------------------------------------------------
# A.5 final read-out  (3D/models/pipeline.py:12-65 with k=1, mutual=False, threshold=None)
# ------------------------------------------------------------------------------------------
def top1_union(conf2d):
    """conf2d [N,M] -> int64 [K,3] rows [0,i,j] of the row-argmax UNION column-argmax set."""
    N, M = conf2d.shape
    hit = torch.zeros(N, M, dtype=torch.bool)
    hit[torch.arange(N), conf2d.argmax(1)] = True
    hit[conf2d.argmax(0), torch.arange(M)] = True
    ij = hit.nonzero()
    return torch.cat([torch.zeros(len(ij), 1, dtype=torch.int64), ij], 1)


def mutual_match(conf, thr):
    """4D read-out used by the tester: > thr AND row max AND column max (3D/models/matching.py:126-143)."""
    m = (conf > thr) & (conf == conf.max(2, keepdim=True)[0]) & (conf == conf.max(1, keepdim=True)[0])
    return m.nonzero()


# ------------------------------------------------------------------------------------------
# the loop  (3D/models/pipeline.py:221-283, 287-309; 4D/models/pipeline.py:156-197)
# ------------------------------------------------------------------------------------------
def warp_from_matrix(W, cfg, x, p_s, p_t, mask_s, mask_t, max_cond, variant):
    """steps 2-5 of SURVEY Appendix A; `x` is masked IN PLACE like the reference (:296)."""
    x.masked_fill_(~pair_mask(mask_s, mask_t), float("-inf"))
    Z = sinkhorn_log(x, W["denoising_coarse_matching.bin_score"], cfg["skh_iters"], mask_s, mask_t)
    conf = Z.exp()[:, :-1, :-1].contiguous().float()
    R, t, Rf, tf, cond, ok = procrustes(conf, p_s, p_t, mask_s, mask_t, cfg["sample_rate"],
                                        max_cond, variant)
    warped = (Rf.float() @ p_s.transpose(1, 2) + tf.float()).transpose(1, 2)
    return warped, dict(conf=conf, R=R, t=t, R_forwd=Rf, t_forwd=tf, cond=cond, ok=ok)


def denoise_loop(W, cfg, f_s, f_t, p_s, p_t, mask_s, mask_t, x_T, steps, max_cond,
                 variant="3dmatch", noise=None, trace=None):
    """Reverse diffusion over the N x M matching matrix.

    f_s [1,N,C], f_t [1,M,C], p_s [1,N,3], p_t [1,M,3] fp32; masks bool; x_T [1,N,M] fp32;
    noise [steps,1,N,M] is the per-step xi (added only by the 4D variant).
    Returns dict(conf_matrix_pred, match_pred (3D only), x_final, per-step records in `trace`).
    """
    ac, sra, srm1 = diffusion_schedule()
    x = x_T.clone()
    bin_score = W["denoising_coarse_matching.bin_score"]
    for k, (t, tn) in enumerate(time_pairs(steps)):
        if variant == "3dmatch":
            x = x - x.min()                                                    # :239
        warped, rec = warp_from_matrix(W, cfg, x, p_s, p_t, mask_s, mask_t, max_cond, variant)
        hs, ht, pe_s, pe_t = denoiser(W, cfg, f_s, f_t, warped, p_t.float(), mask_s, mask_t)
        x0 = match_head(W, cfg, hs, ht, pe_s, pe_t, mask_s, mask_t)            # fp32
        # `extract` yields [1,1,1] float64 tensors -> eps is float64; the 0-d alphas below do NOT
        # promote, so x0*sqrt(a_next) is a float32 product (3D/models/pipeline.py:75-79,246-256)
        eps = (sra[t].view(1, 1, 1) * x - x0) / srm1[t].view(1, 1, 1)
        sigma, c, sqrt_an = ddim_coefficients(ac, t, tn)
        xi_dtype = x.dtype                      # randn_like(x): fp32 at step 1, fp64 afterwards (Q2)
        x = x0 * sqrt_an + c * eps
        if variant == "4dmatch":
            x = x + sigma * noise[k].to(xi_dtype)
        if trace is not None:
            rec.update(x0=x0, x=x.clone(), warped=warped)
            trace.append(rec)
    out = dict(x_final=x)
    if variant == "3dmatch":
        s = x - x.min()
        s = s.masked_fill(~pair_mask(mask_s, mask_t), float("-inf"))
        conf = sinkhorn_log(s, bin_score, cfg["skh_iters"], mask_s, mask_t).exp()[:, :-1, :-1].contiguous()
        out["conf_matrix_pred"] = conf
        out["match_pred"] = top1_union(conf[0])
    else:
        out["conf_matrix_pred"] = torch.sigmoid(x)
    return out


# ------------------------------------------------------------------------------------------
# consumer-side metric used for IR parity  (3D/models/loss.py:383-410)
# ------------------------------------------------------------------------------------------
def inlier_ratio(match_pred, p_s, p_t, R_gt, t_gt, thr=0.1):
    """fraction of predicted matches whose GT-aligned distance is < thr."""
    if len(match_pred) == 0:
        return 0.0
    i, j = match_pred[:, 1], match_pred[:, 2]
    moved = p_s[0, i].double() @ torch.as_tensor(R_gt, dtype=F64).T + torch.as_tensor(t_gt, dtype=F64)
    d = (moved - p_t[0, j].double()).norm(dim=1)
    return float((d < thr).double().mean())


# ==========================================================================================
# 2D-3D variant (row a10).  2D3D/ = Diff-Reg-2d3d/, EXP/ = its experiments/2d3dmatr.rgbdv2.stage4.level3.stage1/
# ==========================================================================================
def fourier_embedding(p, L=10):
    """[..., D] -> [..., (2L+1) D]: [p | per level l: sin(2^l p) (D values), cos(2^l p) (D values)]
    (2D3D/vision3d/layers/embedding.py:75-100 with use_pi=False, use_input=True)."""
    shape = p.shape[:-1]
    D = p.shape[-1]
    x = p.reshape(-1, 1, D)
    fac = (2.0 ** torch.arange(0, L).float()).view(1, -1, 1)
    th = fac * x
    emb = torch.cat([torch.sin(th), torch.cos(th)], dim=-1).reshape(*shape, 2 * L * D)
    return torch.cat([p, emb], dim=-1)


def linear(x, W, pre):
    w = W[pre + ".weight"]              # (x follows the weights' dtype: the float64 evaluation keeps float32 positions / embeddings as inputs)
    return x.to(w.dtype) @ w.T + W[pre + ".bias"]


def transformer_layer_2d3d(W, pre, x, y, H):
    """post-LN layer of 2D3D/vision3d/layers/transformer.py:58-158 (MultiHeadAttention), :161-220 (AttentionLayer),
    :223-238 (AttentionOutput), :241-301 (TransformerLayer); no masks are passed on the path (EXP/model.py:658-664)."""
    B, Lq, C = x.shape
    S = y.shape[1]
    d = C // H
    q = linear(x, W, pre + "attention.attention.q_token_layer").view(B, Lq, H, d).transpose(1, 2)
    k = linear(y, W, pre + "attention.attention.k_token_layer").view(B, S, H, d).transpose(1, 2)
    v = linear(y, W, pre + "attention.attention.v_token_layer").view(B, S, H, d).transpose(1, 2)
    a = torch.softmax(torch.einsum("bhnc,bhmc->bhnm", q, k) / d ** 0.5, dim=-1)
    h = torch.matmul(a, v).transpose(1, 2).reshape(B, Lq, C)
    z = layer_norm(linear(h, W, pre + "attention.linear") + x, W[pre + "attention.norm.weight"], W[pre + "attention.norm.bias"])
    f = linear(torch.relu(linear(z, W, pre + "output.expand")), W, pre + "output.squeeze")
    return layer_norm(z + f, W[pre + "output.norm.weight"], W[pre + "output.norm.bias"])


def fusion_module(W, cfg, img_feats, img_dino, img_pixels, pcd_feats, pcd_points, prefix="denoising_transformer."):
    """CrossModalFusionModule.forward (EXP/fusion_module.py:61-107): same weights for the image and the point call;
    in a cross block the points attend to the UPDATED image tokens."""
    H = cfg["H"]
    img = torch.relu(torch.cat([linear(img_feats, W, prefix + "img_in_proj"), linear(img_dino, W, prefix + "img_in_proj_dino")], -1))
    img = linear(img, W, prefix + "img_in_proj_all") + linear(fourier_embedding(img_pixels), W, prefix + "img_emb_proj")
    pts = pcd_points - pcd_points.mean(dim=1)                       # fusion_module.py:56 (mean over the nodes, B = 1)
    pcd = linear(pcd_feats, W, prefix + "pcd_in_proj") + linear(fourier_embedding(pts), W, prefix + "pcd_emb_proj")
    for l in range(cfg["n_layers"]):
        pre = prefix + "transformer.%d." % l
        if l % 2 == 0:
            img = transformer_layer_2d3d(W, pre, img, img, H)
            pcd = transformer_layer_2d3d(W, pre, pcd, pcd, H)
        else:
            img = transformer_layer_2d3d(W, pre, img, pcd, H)
            pcd = transformer_layer_2d3d(W, pre, pcd, img, H)
    return linear(img, W, prefix + "out_proj"), linear(pcd, W, prefix + "out_proj")


def match_head_2d3d(W, cfg, f_src, f_tgt, mask_s, mask_t, prefix="denoising_coarse_matching."):
    """EXP/matching.py:91-147: src_proj on both sides (Q1), no position code, / sqrt(C), Sinkhorn."""
    C = f_src.shape[-1]
    Wp = W[prefix + "src_proj.weight"]
    a = (f_src @ Wp.T) / C ** 0.5
    b = (f_tgt @ Wp.T) / C ** 0.5
    return sinkhorn_conf(torch.einsum("bsc,btc->bst", a, b), W[prefix + "bin_score"], cfg["skh_iters"], mask_s, mask_t)


def denoise_loop_2d3d(W, cfg, img_feats, img_dino, img_pixels, pcd_feats, s_pcd, t_pcd_da, mask_s, mask_t, mask_t_da, x_T,
                      steps, max_cond, trace=None):
    """EXP/model.py:637-694 + get_warped_from_noising_matching3D3D (:830-846).  src = point nodes [1,N,.], tgt = image
    patches [1,M,.]; the warp uses the depth-back-projected patch centres t_pcd_da and their mask; no min-shift;
    deterministic update; Procrustes K from the mask sums (EXP/procrustes.py:61-62)."""
    ac, sra, srm1 = diffusion_schedule()
    x = x_T.clone()
    bin_score = W["denoising_coarse_matching.bin_score"]
    for k, (t, tn) in enumerate(time_pairs(steps)):
        x.masked_fill_(~pair_mask(mask_s, mask_t_da), float("-inf"))
        Z = sinkhorn_log(x, bin_score, cfg["skh_iters"], mask_s, mask_t_da)
        conf = Z.exp()[:, :-1, :-1].contiguous().float()
        R, tt, Rf, tf, cond, ok = procrustes(conf, s_pcd, t_pcd_da, mask_s, mask_t_da, cfg["sample_rate"], max_cond, "4dmatch")
        warped = (Rf.float() @ s_pcd.transpose(1, 2) + tf.float()).transpose(1, 2)
        f_img, f_pcd = fusion_module(W, cfg, img_feats, img_dino, img_pixels, pcd_feats, warped)
        x0 = match_head_2d3d(W, cfg, f_pcd, f_img, mask_s, mask_t)
        eps = (sra[t].view(1, 1, 1) * x - x0) / srm1[t].view(1, 1, 1)
        sigma, c, sqrt_an = ddim_coefficients(ac, t, tn)
        x = x0 * sqrt_an + c * eps
        if trace is not None:
            trace.append(dict(x0=x0, R_forwd=Rf, t_forwd=tf, cond=cond, x=x.clone()))
    s = x.masked_fill(~pair_mask(mask_s, mask_t), float("-inf"))
    conf = sinkhorn_log(s, bin_score, cfg["skh_iters"], mask_s, mask_t).exp()[:, :-1, :-1].contiguous()
    return dict(conf_matrix_pred=conf, match_pred=top1_union(conf[0]), x_final=x)

"""Deterministic synthetic scene pairs and weights for the reverse-diffusion matching path.

Everything here is derived from an integer hash (splitmix64 finaliser) of (seed, stream, index),
so the same bits are produced on every box and with every numpy/torch version -- weights and
inputs never have to be committed, only reference OUTPUTS are stored under tests/golden/.

Shapes follow the reference's coarse level:
  s_pcd [N,3], t_pcd [M,3]   superpoints inside vol_bnds (3D/configs/test/3dmatch.yaml:46)
  src_feats [N,C], tgt_feats [M,C]  backbone features (matched points share a base vector)
  x_T [N,M]  initial noise of the reverse process (3D/models/pipeline.py:224)
"""
import math
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_u01(seed, stream, n):
    """n doubles in (0,1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        z = _mix(idx ^ base)
        z = _mix(z + base)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / (1 << 53))


def hash_bits(seed, stream, n):
    """n 53-bit integers (uint64): the integers under hash_u01.  The correspondence RANSAC samples with them
    (csrc/metrics.hip::hash_bits is the same arithmetic on the device)."""
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        z = _mix(idx ^ base)
        z = _mix(z + base)
    return z >> np.uint64(11)


def hash_uniform(seed, stream, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * hash_u01(seed, stream, n)).reshape(shape)


def hash_normal(seed, stream, shape):
    n = int(np.prod(shape))
    u1 = hash_u01(seed, 2 * stream + 1000003, n)
    u2 = hash_u01(seed, 2 * stream + 1000004, n)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)).reshape(shape)


# --------------------------------------------------------------------------------------------
# model dimensions of the three trees
# --------------------------------------------------------------------------------------------
VARIANTS = {
    # 3D/configs/test/3dmatch.yaml:23,35-36,42-51
    "3dmatch": dict(C=432, H=4, voxel=0.08, origin=(-3.6, -2.4, 1.14), skh_iters=3, bin_score=1.0,
                    sample_rate=1.0, n_layers=6),
    # 4D/configs/test/4dmatch.yaml:24,44,50,53
    "4dmatch": dict(C=528, H=4, voxel=0.04, origin=(-3.6, -2.4, 1.14), skh_iters=3, bin_score=1.0,
                    sample_rate=1.0, n_layers=6),
}

VARIANTS["2d3d"] = dict(C=256, H=4, skh_iters=3, bin_score=1.0, sample_rate=1.0, n_layers=6, img_dim=512, dino_dim=1024,
                        pcd_dim=512)     # EXP/config.py:116-139

LAYER_TYPES = ("self", "cross", "self", "cross", "self", "cross")  # 3D/models/pipeline.py:142


def make_weights(C, seed=7, n_layers=6, head_gain=1.0, dtype=np.float32):
    """Weights of the denoising transformer + matching head in the reference's state-dict layout
    (SURVEY.md section 8b): hash-uniform * 1/sqrt(fan_in), LayerNorm gamma=1, beta=0."""
    w = {}
    st = 0

    def lin(out_f, in_f, gain=1.0):
        nonlocal st
        st += 1
        a = gain * math.sqrt(3.0 / in_f)
        return hash_uniform(seed, st, (out_f, in_f), -a, a).astype(dtype)

    for l in range(n_layers):
        p = "denoising_transformer.layers.%d." % l
        w[p + "q_proj.weight"] = lin(C, C)
        w[p + "k_proj.weight"] = lin(C, C)
        w[p + "v_proj.weight"] = lin(C, C)
        w[p + "merge.weight"] = lin(C, C)
        w[p + "mlp.0.weight"] = lin(2 * C, 2 * C)
        w[p + "mlp.2.weight"] = lin(C, 2 * C)
        st += 1
        w[p + "norm1.weight"] = (1.0 + 0.1 * hash_uniform(seed, st, (C,))).astype(dtype)
        st += 1
        w[p + "norm1.bias"] = (0.05 * hash_uniform(seed, st, (C,))).astype(dtype)
        st += 1
        w[p + "norm2.weight"] = (1.0 + 0.1 * hash_uniform(seed, st, (C,))).astype(dtype)
        st += 1
        w[p + "norm2.bias"] = (0.05 * hash_uniform(seed, st, (C,))).astype(dtype)
    w["denoising_coarse_matching.src_proj.weight"] = lin(C, C, head_gain)
    w["denoising_coarse_matching.tgt_proj.weight"] = lin(C, C)  # allocated, never used (Q1)
    w["denoising_coarse_matching.bin_score"] = np.asarray(1.0, dtype=dtype)
    return w


def make_weights_coarse(C, seed=17, head_gain=1.0, dtype=np.float32):
    """Weights of the NON-denoising branch of the training forward (row f3; 3D/models/pipeline.py:142-147): coarse_transformer with
    layer_types [self, cross, positioning, self, cross] (layers.2 = [Matching, SoftProcrustesLayer]) and coarse_matching, in the
    reference's state-dict layout.  Same distributions as make_weights."""
    w = {}
    st = 0

    def lin(out_f, in_f, gain=1.0):
        nonlocal st
        st += 1
        a = gain * math.sqrt(3.0 / in_f)
        return hash_uniform(seed, st, (out_f, in_f), -a, a).astype(dtype)

    for l in (0, 1, 3, 4):
        p = "coarse_transformer.layers.%d." % l
        w[p + "q_proj.weight"] = lin(C, C)
        w[p + "k_proj.weight"] = lin(C, C)
        w[p + "v_proj.weight"] = lin(C, C)
        w[p + "merge.weight"] = lin(C, C)
        w[p + "mlp.0.weight"] = lin(2 * C, 2 * C)
        w[p + "mlp.2.weight"] = lin(C, 2 * C)
        for nm, base, amp in (("norm1.weight", 1.0, 0.1), ("norm1.bias", 0.0, 0.05), ("norm2.weight", 1.0, 0.1), ("norm2.bias", 0.0, 0.05)):
            st += 1
            w[p + nm] = (base + amp * hash_uniform(seed, st, (C,))).astype(dtype)
    for p in ("coarse_transformer.layers.2.0.", "coarse_matching."):
        w[p + "src_proj.weight"] = lin(C, C, head_gain)
        w[p + "tgt_proj.weight"] = lin(C, C)
        w[p + "bin_score"] = np.asarray(1.0, dtype=dtype)
    return w


def _rodrigues(axis, theta):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(theta) * K + (1 - math.cos(theta)) * (K @ K)


def make_pair(N, M, C, seed, overlap=0.6, feat_noise=0.3, pos_noise=0.01, max_angle_deg=45.0,
              dtype=np.float32):
    """One synthetic scene pair."""
    lo = np.array([-3.6, -2.4, 1.14]) + 0.05
    hi = np.array([1.093, 0.78, 2.92]) - 0.05
    s_pcd = lo + (hi - lo) * hash_u01(seed, 11, N * 3).reshape(N, 3)
    axis = hash_normal(seed, 12, (3,))
    theta = math.radians(max_angle_deg) * float(hash_u01(seed, 13, 1)[0])
    R = _rodrigues(axis, theta)
    t = hash_uniform(seed, 14, (3,), -0.5, 0.5)
    # rotate about the scene centre so the target stays inside the volume
    ctr = 0.5 * (lo + hi)
    n_ov = int(round(overlap * min(N, M)))
    perm_src = np.argsort(hash_u01(seed, 15, N))       # which src points have a partner
    perm_tgt = np.argsort(hash_u01(seed, 16, M))       # where the partner sits in tgt
    t_pcd = lo + (hi - lo) * hash_u01(seed, 17, M * 3).reshape(M, 3)
    moved = (s_pcd[perm_src[:n_ov]] - ctr) @ R.T + ctr + 0.1 * t
    t_pcd[perm_tgt[:n_ov]] = moved + pos_noise * hash_normal(seed, 18, (n_ov, 3))
    t_eff = ctr - R @ ctr + 0.1 * t          # t_pcd = R s_pcd + t_eff on the overlap
    base_s = hash_normal(seed, 19, (N, C))
    base_t = hash_normal(seed, 20, (M, C))
    base_t[perm_tgt[:n_ov]] = base_s[perm_src[:n_ov]]
    src_feats = base_s + feat_noise * hash_normal(seed, 21, (N, C))
    tgt_feats = base_t + feat_noise * hash_normal(seed, 22, (M, C))
    x_T = hash_normal(seed, 23, (N, M))
    gt = np.stack([perm_src[:n_ov], perm_tgt[:n_ov]], 1).astype(np.int64)
    return dict(s_pcd=s_pcd.astype(dtype), t_pcd=t_pcd.astype(dtype),
                src_feats=src_feats.astype(dtype), tgt_feats=tgt_feats.astype(dtype),
                x_T=x_T.astype(dtype), R_gt=R, t_gt=t_eff, gt_matches=gt)


def step_noise(N, M, seed, n_steps, dtype=np.float32):
    """Per-step noise xi_k (only the 4D variant adds sigma*xi, 4D/models/pipeline.py:188-190)."""
    return np.stack([hash_normal(seed, 100 + k, (N, M)) for k in range(n_steps)]).astype(dtype)


# --------------------------------------------------------------------------------------------
# 2D-3D variant (Diff-Reg-2d3d): CrossModalFusionModule + Matching weights, image/point-cloud pairs
# --------------------------------------------------------------------------------------------
def make_weights_2d3d(seed=9, head_gain=1.0, emb_gain=0.1, dtype=np.float32):
    v = VARIANTS["2d3d"]
    C = v["C"]
    w = {}
    st = [0]

    def lin(name, out_f, in_f, gain=1.0, bias=True):
        st[0] += 1
        a = gain * math.sqrt(3.0 / in_f)
        w[name + ".weight"] = hash_uniform(seed, st[0], (out_f, in_f), -a, a).astype(dtype)
        if bias:
            st[0] += 1
            w[name + ".bias"] = (0.1 * gain * hash_uniform(seed, st[0], (out_f,))).astype(dtype)

    def norm(name):
        st[0] += 1
        w[name + ".weight"] = (1.0 + 0.1 * hash_uniform(seed, st[0], (C,))).astype(dtype)
        st[0] += 1
        w[name + ".bias"] = (0.05 * hash_uniform(seed, st[0], (C,))).astype(dtype)

    p = "denoising_transformer."
    lin(p + "img_emb_proj", C, 42, emb_gain); lin(p + "pcd_emb_proj", C, 63, emb_gain)
    lin(p + "img_in_proj", C, v["img_dim"]); lin(p + "img_in_proj_dino", C, v["dino_dim"]); lin(p + "img_in_proj_all", C, 2 * C)
    lin(p + "pcd_in_proj", C, v["pcd_dim"]); lin(p + "out_proj", C, C)
    for l in range(v["n_layers"]):
        q = p + "transformer.%d." % l
        for nm in ("q_token_layer", "k_token_layer", "v_token_layer"):
            lin(q + "attention.attention." + nm, C, C)
        lin(q + "attention.linear", C, C); norm(q + "attention.norm")
        lin(q + "output.expand", 2 * C, C); lin(q + "output.squeeze", C, 2 * C); norm(q + "output.norm")
    lin("denoising_coarse_matching.src_proj", C, C, head_gain, bias=False)
    lin("denoising_coarse_matching.tgt_proj", C, C, bias=False)
    w["denoising_coarse_matching.bin_score"] = np.asarray(1.0, dtype=dtype)
    return w


def make_pair_2d3d(N, M, seed, weights=None, overlap=0.6, tok_noise=0.05, dtype=np.float32):
    """N point-cloud nodes (src) and M image patches (tgt): patch pixel coordinates [M,2] (normalised), the
    depth-back-projected patch centres t_pcd_da [M,3] and backbone features.  With `weights` (make_weights_2d3d)
    the point features of matched nodes are solved (least squares through pcd_in_proj) so that their input TOKEN
    equals the matched patch's token: the random-weight fusion module then yields correlated features and a
    structured matching matrix, like a trained model would."""
    v = VARIANTS["2d3d"]
    base = make_pair(N, M, 64, seed, overlap=overlap)
    rngs = lambda s, shape: hash_normal(seed, s, shape)
    gi, gj = base["gt_matches"][:, 0], base["gt_matches"][:, 1]
    img_feats = rngs(47, (M, v["img_dim"]))
    img_dino = rngs(48, (M, v["dino_dim"]))
    pcd_feats = rngs(49, (N, v["pcd_dim"]))
    if weights is not None:
        W = {k: a.astype(np.float64) for k, a in weights.items() if k.startswith("denoising_transformer.") and "_in_proj" in k}
        p = "denoising_transformer."
        u = np.concatenate([img_feats @ W[p + "img_in_proj.weight"].T + W[p + "img_in_proj.bias"],
                            img_dino @ W[p + "img_in_proj_dino.weight"].T + W[p + "img_in_proj_dino.bias"]], 1)
        tok_img = np.maximum(u, 0) @ W[p + "img_in_proj_all.weight"].T + W[p + "img_in_proj_all.bias"]     # [M, C]
        target = tok_img[gj] + tok_noise * rngs(50, (len(gj), v["C"])) - W[p + "pcd_in_proj.bias"]
        pcd_feats[gi] = target @ np.linalg.pinv(W[p + "pcd_in_proj.weight"]).T
    pix = hash_uniform(seed, 46, (M, 2), 0.0, 1.0)
    return dict(s_pcd=base["s_pcd"], t_pcd_da=base["t_pcd"], img_pixels=pix.astype(dtype), img_feats=img_feats.astype(dtype),
                img_dino=img_dino.astype(dtype), pcd_feats=pcd_feats.astype(dtype), x_T=base["x_T"],
                gt_matches=base["gt_matches"], R_gt=base["R_gt"], t_gt=base["t_gt"])


# ==========================================================================================
# KPFCN backbone inputs: a stacked src|tgt cloud with the per-layer index
# arrays the reference's collate function produces (3D/datasets/dataloader.py:13-68, 247-325):
# grid-subsampled points per layer, radius neighbours (sorted by distance, padded with the
# "shadow" index = number of support points), pools (layer l+1 queries -> layer l supports) and
# upsamples (layer l queries -> layer l+1 supports, nearest first).  Brute-force numpy: the
# point of these arrays is to be a fixed, reproducible input of the backbone, not a fast collate.
# ==========================================================================================
KPFCN_ARCH = ["simple", "resnetb", "resnetb_strided", "resnetb", "resnetb", "resnetb_strided", "resnetb", "resnetb",
              "resnetb_strided", "resnetb", "resnetb", "nearest_upsample", "unary", "nearest_upsample", "unary",
              "nearest_upsample", "unary"]                       # 3D/configs/models.py:3-21
KPFCN_CFG = dict(num_layers=4, in_points_dim=3, first_feats_dim=256, first_subsampling_dl=0.025, in_feats_dim=1,
                 conv_radius=2.5, num_kernel_points=15, KP_extent=2.0, coarse_feature_dim=432)   # 3D/configs/test/3dmatch.yaml:3-26


def _grid_subsample(pts, dl):
    """barycentre of the points of every occupied voxel of size dl, voxels in lexicographic order"""
    key = np.floor(pts / dl).astype(np.int64)
    key -= key.min(0)
    dims = key.max(0) + 1
    lin = (key[:, 0] * dims[1] + key[:, 1]) * dims[2] + key[:, 2]
    order = np.argsort(lin, kind="stable")
    lin_s, pts_s = lin[order], pts[order]
    starts = np.flatnonzero(np.r_[True, lin_s[1:] != lin_s[:-1]])
    sums = np.add.reduceat(pts_s.astype(np.float64), starts, axis=0)
    cnt = np.diff(np.r_[starts, len(lin_s)])[:, None]
    return (sums / cnt).astype(np.float32)


def _radius_neighbors(q, s, q_len, s_len, radius, limit):
    """[len(q), limit] int64 indices into the stacked support cloud, nearest first, shadow index = len(s)"""
    out = np.full((len(q), limit), len(s), dtype=np.int64)
    q0 = s0 = 0
    for ql, sl in zip(q_len, s_len):
        ss = s[s0:s0 + sl].astype(np.float64)
        for c0 in range(0, ql, 1024):                            # query blocks: bounded memory for large clouds
            qq = q[q0 + c0:q0 + min(ql, c0 + 1024)].astype(np.float64)
            d2 = ((qq[:, None, :] - ss[None, :, :]) ** 2).sum(-1)
            idx = np.argsort(d2, axis=1, kind="stable")[:, :limit]
            dd = np.take_along_axis(d2, idx, 1)
            idx = np.where(dd < radius * radius, idx + s0, len(s))
            out[q0 + c0:q0 + c0 + len(qq), :idx.shape[1]] = idx
        q0 += ql; s0 += sl
    return out


def make_kpfcn_batch(n_src=1400, n_tgt=1200, seed=0, limit=(28, 28, 30, 32), extent=1.0):
    """points / neighbors / pools / upsamples / stack_lengths of a 4-layer KPFCN batch + features [N0,1] = 1
    (extent scales the sheet: the number of occupied voxels, hence of layer-0 points, grows with extent^2)"""
    cfg = KPFCN_CFG
    clouds = []
    for c, n in enumerate((n_src, n_tgt)):
        u = hash_uniform(seed, 40 + c, (n, 3), 0.0, 1.0)
        # a folded sheet inside a 0.9 x 0.7 x 0.4 box: surface-like sampling, ~25-30 neighbours at the first radius
        x, y = 0.9 * extent * u[:, 0], 0.7 * extent * u[:, 1]
        z = 0.15 * np.sin(4.0 * x + c) * np.cos(3.0 * y) + 0.02 * u[:, 2] + 0.2
        clouds.append(np.stack([x, y, z], 1).astype(np.float32))
    dl = cfg["first_subsampling_dl"]
    pts = [_grid_subsample(c, dl) for c in clouds]
    points, lengths = [np.concatenate(pts)], [[len(p) for p in pts]]
    for l in range(1, cfg["num_layers"]):
        dl *= 2
        pts = [_grid_subsample(p, dl) for p in pts]
        points.append(np.concatenate(pts)); lengths.append([len(p) for p in pts])
    neighbors, pools, upsamples = [], [], []
    r = cfg["first_subsampling_dl"] * cfg["conv_radius"]
    for l in range(cfg["num_layers"]):
        neighbors.append(_radius_neighbors(points[l], points[l], lengths[l], lengths[l], r, limit[l]))
        if l + 1 < cfg["num_layers"]:
            pools.append(_radius_neighbors(points[l + 1], points[l], lengths[l + 1], lengths[l], r, limit[l]))
            upsamples.append(_radius_neighbors(points[l], points[l + 1], lengths[l], lengths[l + 1], 2 * r, limit[l + 1]))
        r *= 2
    return dict(points=points, neighbors=neighbors, pools=pools, upsamples=upsamples, stack_lengths=lengths,
                features=np.ones((len(points[0]), 1), np.float32))


def make_kpfcn_weights(kernel_points, seed=3):
    """state dict of models.backbone.KPFCN (3D/models/backbone.py:8-118) in the reference's names, hash-generated
    (kaiming-like 1/sqrt(fan_in) scale); kernel_points[name] come from the reference's own disposition file."""
    cfg = KPFCN_CFG
    sd, ctr = {}, [0]

    def w(shape, fan_in):
        ctr[0] += 1
        return (hash_uniform(seed, 1000 + ctr[0], shape) / np.sqrt(fan_in)).astype(np.float32)

    def kpconv(pre, cin, cout):
        sd[pre + "weights"] = w((cfg["num_kernel_points"], cin, cout), cin * 4)
        sd[pre + "kernel_points"] = kernel_points[pre + "kernel_points"]

    layer, in_dim, out_dim = 0, cfg["in_feats_dim"], cfg["first_feats_dim"]
    skip_dims, bi = [], 0
    for bi, block in enumerate(KPFCN_ARCH):
        if any(t in block for t in ("pool", "strided", "upsample", "global")):
            skip_dims.append(in_dim)
        if "upsample" in block:
            break
        pre = "encoder_blocks.%d." % bi
        if block == "simple":
            kpconv(pre + "KPConv.", in_dim, out_dim // 2)
        else:
            if in_dim != out_dim // 4:
                sd[pre + "unary1.mlp.weight"] = w((out_dim // 4, in_dim), in_dim)
            kpconv(pre + "KPConv.", out_dim // 4, out_dim // 4)
            sd[pre + "unary2.mlp.weight"] = w((out_dim, out_dim // 4), out_dim // 4)
            if in_dim != out_dim:
                sd[pre + "unary_shortcut.mlp.weight"] = w((out_dim, in_dim), in_dim)
        in_dim = out_dim // 2 if "simple" in block else out_dim
        if "pool" in block or "strided" in block:
            layer += 1; out_dim *= 2
    sd["coarse_out.weight"] = w((cfg["coarse_feature_dim"], in_dim // 2, 1), in_dim // 2)
    sd["coarse_out.bias"] = w((cfg["coarse_feature_dim"],), 4.0)
    start = bi
    for di, block in enumerate(KPFCN_ARCH[start:]):
        if di > 0 and "upsample" in KPFCN_ARCH[start + di - 1]:
            in_dim += skip_dims[layer]
        if block == "unary":
            sd["decoder_blocks.%d.mlp.weight" % di] = w((out_dim, in_dim), in_dim)
        in_dim = out_dim
        if "upsample" in block:
            layer -= 1; out_dim //= 2
        if di == 1:
            break                                                # the coarse phase returns after decoder block 1 (backbone.py:153-158)
    return sd


# --------------------------------------------------------------------------------------------
# evaluation-harness inputs: predicted matches, 4DMatch-style metric points, Redwood info matrices
# --------------------------------------------------------------------------------------------
def make_matches(pair, seed, inlier_frac=0.5, n_extra=0):
    """A match_pred-like list [K,3] int64 (0, i, j) for a make_pair() scene: one row per source point (plus n_extra
    rows that repeat source points, like the column arg-max half of the top-1 union), a fraction of the overlapping
    points matched to their true partner, everything else to a hash-chosen target.  Sorted row-major like nonzero()."""
    N, M = pair["s_pcd"].shape[0], pair["t_pcd"].shape[0]
    j = (hash_u01(seed, 31, N + n_extra) * M).astype(np.int64)
    i = np.concatenate([np.arange(N), (hash_u01(seed, 32, n_extra) * N).astype(np.int64)])
    gt = pair["gt_matches"]
    keep = hash_u01(seed, 33, len(gt)) < inlier_frac / max(len(gt) / N, 1e-9)
    j[gt[keep, 0]] = gt[keep, 1]
    key = np.unique(i * M + j)
    return np.stack([np.zeros_like(key), key // M, key % M], 1).astype(np.int64)


def make_metric_points(pair, seed, n_raw=2000, n_metric=600, flow_scale=0.05, dtype=np.float32):
    """4DMatch-style extras for compute_nrfmr (3D/lib/tester.py:150-210): a raw source cloud scattered around the
    superpoints, a smooth scene flow on it, the indices of the metric points, and the coarse flow of the superpoints."""
    s = pair["s_pcd"].astype(np.float64)
    N = len(s)
    owner = (hash_u01(seed, 41, n_raw) * N).astype(np.int64)
    raw = s[owner] + 0.04 * hash_normal(seed, 42, (n_raw, 3))
    A = flow_scale * hash_normal(seed, 43, (3, 3))
    b = flow_scale * hash_normal(seed, 44, (3,))
    flow = lambda x: np.sin(x @ A.T) * 0.5 + b
    metric_index = np.sort(np.argsort(hash_u01(seed, 45, n_raw))[:n_metric]).astype(np.int64)
    return dict(raw_pcd=raw.astype(dtype), raw_flow=flow(raw).astype(dtype), metric_index=metric_index,
                coarse_flow=flow(s).astype(dtype))


def make_info(seed):
    """A symmetric positive-definite 6x6 `gt.info` matrix of the Redwood registration benchmark (loss.py:27-44)."""
    B = hash_normal(seed, 51, (6, 6))
    return B @ B.T * 50.0 + 200.0 * np.eye(6)


def deform_targets(pair, coarse_flow, dtype=np.float32):
    """t_pcd of a non-rigid scene: the partners of the overlapping source points sit at R (s + flow) + t."""
    tp = pair["t_pcd"].astype(np.float64).copy()
    gt = pair["gt_matches"]
    s = pair["s_pcd"].astype(np.float64) + coarse_flow.astype(np.float64)
    tp[gt[:, 1]] = s[gt[:, 0]] @ pair["R_gt"].T + pair["t_gt"]
    return tp.astype(dtype)


def make_kpfcn_bn_biases(keys_shapes, seed=5):
    """the `bias` parameters of the BatchNormBlocks of a KPFCN built with use_batch_norm = False (3D/models/blocks.py:430-446), hash-generated:
    keys_shapes = [(state-dict key, shape)], in sorted key order"""
    return {k: (hash_uniform(seed, 7000 + i, tuple(shape)) * 0.2).astype(np.float32) for i, (k, shape) in enumerate(sorted(keys_shapes))}

"""Shared builders for the parity tests: synthetic pairs/weights as torch tensors."""
import numpy as np
import torch

from diffreg_hip import synth

HEAD_GAIN = 24.0   # must match oracle/make_golden.py
HEAD_GAIN_SOFT = 3.0   # the "soft" fixture family (oracle/make_golden.py): matching logits O(10) -- checkpoint-like scale, nothing ill-conditioned


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def weights(variant, family="main"):
    C = synth.VARIANTS[variant]["C"]
    return {k: T(v) for k, v in synth.make_weights(C, seed=7, head_gain=HEAD_GAIN if family == "main" else HEAD_GAIN_SOFT).items()}


def pair(variant, N, M, seed):
    C = synth.VARIANTS[variant]["C"]
    p = synth.make_pair(N, M, C, seed=seed)
    return p, dict(f_s=T(p["src_feats"])[None], f_t=T(p["tgt_feats"])[None], p_s=T(p["s_pcd"])[None],
                   p_t=T(p["t_pcd"])[None], x_T=T(p["x_T"])[None])


def assert_match_list_is_the_references(got, g, conf_err):
    """match_pred (3D/models/pipeline.py:12-65, 275-280: row arg-maxima united with column arg-maxima) is index work, so it is
    compared with the reference's list EXACTLY: the two lists must be equal as sets, except that an entry may differ where the
    arg-maximum itself is undecided -- a row / column whose best and second-best confidence are closer than 10 x the largest conf
    deviation (in practice: exact ties, e.g. rows whose mass sits on the dustbin and whose entries are bit-equal; torch's pick among
    equal values is implementation-defined).  An entry of the symmetric difference must lie in such a row or column AND be within that
    margin of its maximum.  -> (undecided rows, undecided columns)."""
    ref, conf = set(map(tuple, g["match_pred"].tolist())), g["conf"]
    tol = 10.0 * max(conf_err, 1e-12)
    srt_r, srt_c = np.sort(conf, 1), np.sort(conf, 0)
    und_r = set(np.nonzero(srt_r[:, -1] - srt_r[:, -2] <= tol)[0].tolist())
    und_c = set(np.nonzero(srt_c[-1] - srt_c[-2] <= tol)[0].tolist())
    am_r, am_c = conf.argmax(1), conf.argmax(0)
    for i in range(conf.shape[0]):
        if i not in und_r:
            assert (0, int(i), int(am_r[i])) in got, ("decided row arg-maximum missing", i)
    for j in range(conf.shape[1]):
        if j not in und_c:
            assert (0, int(am_c[j]), int(j)) in got, ("decided column arg-maximum missing", j)
    for (_, i, j) in got ^ ref:
        ok_r = i in und_r and conf[i, j] >= srt_r[i, -1] - tol
        ok_c = j in und_c and conf[i, j] >= srt_c[-1, j] - tol
        assert ok_r or ok_c, ("match lists differ at a decided entry", i, j, len(got ^ ref))
    return len(und_r), len(und_c)


def ref_match_list_2d3d(g, conf=None):
    """the 2D-3D fixtures store the reference's match list as match_i / match_j (+ the full conf for the small scenes; the cfg5-size ones are
    compact, so the caller supplies the matrix the arg-maxima's decidedness is judged on) -> the mapping assert_match_list_is_the_references reads"""
    mi, mj = np.asarray(g["match_i"]), np.asarray(g["match_j"])
    return {"match_pred": np.stack([np.zeros_like(mi), mi, mj], 1), "conf": np.asarray(g["conf"] if conf is None else conf)}


def masks(N, M, nv=None, mv=None):
    nv = N if nv is None else nv
    mv = M if mv is None else mv
    return torch.arange(N)[None] < nv, torch.arange(M)[None] < mv


def sinkhorn_case(N, M, nv, mv, dtype):
    sc = T(3.0 * synth.hash_normal(1, N * 1000 + M, (1, N, M))).to(dtype)
    sm, tm = masks(N, M, nv, mv)
    return sc.masked_fill(~(sm[:, :, None] & tm[:, None, :]), float("-inf")), sm, tm


# ------------------------------------------------------------------------------------------------
# evaluation-harness scenes; oracle/make_golden_metrics.py mints its vectors on exactly these inputs
# ------------------------------------------------------------------------------------------------
def metrics_scene(N, M, seed, inlier_frac=0.5, n_extra=100):
    """torch CPU tensors in the layout of the reference's `data` dict (B = 1): rigid (3DMatch) and non-rigid (4DMatch)
    targets, predicted matches, metric points, Redwood info matrix and pose estimates at growing distance from the truth."""
    import math
    p = synth.make_pair(N, M, 6, seed=seed)
    mp = synth.make_metric_points(p, seed)
    R, t = p["R_gt"], p["t_gt"]
    est = []
    for k, (ang, dt) in enumerate([(0.0, 0.0), (0.004, 0.01), (0.02, 0.05), (0.1, 0.3)]):
        dR = synth._rodrigues(synth.hash_normal(seed, 60 + k, (3,)), ang)
        est.append((T(dR @ R), T((t + dt * synth.hash_normal(seed, 70 + k, (3,))).reshape(3, 1))))   # float64 like Open3D's pose
    return dict(s_pcd=T(p["s_pcd"])[None], t_pcd=T(p["t_pcd"])[None], t_pcd4=T(synth.deform_targets(p, mp["coarse_flow"]))[None],
                rot=T(R.astype(np.float32))[None], trn=T(t.astype(np.float32).reshape(3, 1))[None],
                matches=T(synth.make_matches(p, seed, inlier_frac, n_extra)), raw_pcd=T(mp["raw_pcd"]), raw_flow=T(mp["raw_flow"]),
                metric_index=T(mp["metric_index"]), coarse_flow=T(mp["coarse_flow"]), info=synth.make_info(seed), est=est,
                pair=p)


def topk_case(name):
    """score matrices of the fine-matching kind (cosine similarities of patch features: distinct values in [-1, 1]) for
    batch_mutual_topk_select; the golden vectors of oracle/make_golden_metrics.py are minted on exactly these"""
    spec = {"patch64_k2_thr": dict(B=40, N=64, M=64, k=2, largest=True, threshold=0.75, mutual=True, masked=False),      # EXP/model.py:744-752
            "ragged_k3_or": dict(B=7, N=50, M=37, k=3, largest=True, threshold=None, mutual=False, masked=False),
            "smallest_k1": dict(B=5, N=128, M=96, k=1, largest=False, threshold=0.1, mutual=True, masked=False),
            "masked_k2": dict(B=12, N=64, M=48, k=2, largest=True, threshold=0.3, mutual=True, masked=True)}[name]
    B, N, M = spec["B"], spec["N"], spec["M"]
    seed = sum(map(ord, name))
    f = synth.hash_normal(seed, 1, (B, N, 16)); g = synth.hash_normal(seed, 2, (B, M, 16))
    g[:, :min(N, M)] = f[:, :min(N, M)] + 0.35 * g[:, :min(N, M)]          # correlated pairs: similarities close to 1 on a diagonal
    f /= np.linalg.norm(f, axis=2, keepdims=True); g /= np.linalg.norm(g, axis=2, keepdims=True)
    score = T(np.einsum("bnd,bmd->bnm", f, g).astype(np.float32))
    if not spec["largest"]:
        score = (1.0 - score) * 0.5
    rm = cm = None
    if spec["masked"]:
        rm = T(synth.hash_u01(seed, 3, B * N).reshape(B, N) > 0.2)
        cm = T(synth.hash_u01(seed, 4, B * M).reshape(B, M) > 0.2)
    return dict(score=score, k=spec["k"], largest=spec["largest"], threshold=spec["threshold"], mutual=spec["mutual"], row_masks=rm, col_masks=cm)


# ------------------------------------------------------------------------------------------------
# guard bands: an output / workspace buffer between two 64 KiB bands of 0xA5
# ------------------------------------------------------------------------------------------------
GUARD = 64 * 1024


def guarded(shape, dtype, device, fill=None):
    """-> (tensor of `shape` living between two 0xA5 bands, check()): check() raises when a kernel wrote outside the tensor."""
    n = int(np.prod(shape)) if len(shape) else 1
    esz = torch.empty((), dtype=dtype).element_size()
    nbytes = (n * esz + 255) // 256 * 256
    raw = torch.full((GUARD + nbytes + GUARD,), 0xA5, dtype=torch.uint8, device=device)
    t = raw[GUARD:GUARD + n * esz].view(dtype).view(*shape)
    if fill is not None:
        t.fill_(fill)

    def check():
        torch.cuda.synchronize()
        assert bool((raw[:GUARD] == 0xA5).all()) and bool((raw[GUARD + nbytes:] == 0xA5).all()), "guard band damaged"
    return t, check


# ------------------------------------------------------------------------------------------------
# forward half of the training branch; oracle/make_golden_train.py mints its vectors on exactly these inputs
# ------------------------------------------------------------------------------------------------
TRAIN_CASES = {"b1": (1, 96, 80, 50, 137, 200.0), "b2": (2, 64, 64, 60, 640, 200.0)}      # B, N, M, seed, time step, max_condition_num


def train_weights(family="main"):
    C = synth.VARIANTS["3dmatch"]["C"]
    gain = HEAD_GAIN if family == "main" else HEAD_GAIN_SOFT
    w = dict(synth.make_weights(C, seed=7, head_gain=gain))
    w.update(synth.make_weights_coarse(C, seed=17, head_gain=gain))
    return {k: T(v) for k, v in w.items()}


def train_case(tag):
    B, N, M, seed, ts, mc = TRAIN_CASES[tag]
    C = synth.VARIANTS["3dmatch"]["C"]
    prs = [synth.make_pair(N, M, C, seed=seed + b) for b in range(B)]
    randn = T(synth.hash_normal(seed, 900, (B, N, M)).astype(np.float32))
    randn[0, 0, 0] = 0.0
    st = lambda k: torch.stack([T(p[k]) for p in prs])
    return dict(B=B, N=N, M=M, ts=ts, mc=mc, f_s=st("src_feats"), f_t=st("tgt_feats"), p_s=st("s_pcd"), p_t=st("t_pcd"), randn=randn,
                matches=[T(p["gt_matches"]).t().contiguous() for p in prs],
                R_gt=torch.stack([T(p["R_gt"]).float() for p in prs]), t_gt=torch.stack([T(p["t_gt"]).float().view(3, 1) for p in prs]),
                src_mask=torch.ones(B, N, dtype=torch.bool), tgt_mask=torch.ones(B, M, dtype=torch.bool))


def train_branch_case(N, M, seed):
    """the case of the non-default training forms (oracle/make_golden_train_branches.py): one synthetic pair, the denoising branch's warped source =
    the source under the ground-truth pose (float32 torch arithmetic on the host, identical in the minting script and in the test)"""
    C = synth.VARIANTS["3dmatch"]["C"]
    p = synth.make_pair(N, M, C, seed=seed)
    R, t = T(p["R_gt"]).float()[None], T(p["t_gt"]).float().view(1, 3, 1)
    p_s = T(p["s_pcd"])[None]
    return dict(B=1, N=N, M=M, mc=200.0, f_s=T(p["src_feats"])[None], f_t=T(p["tgt_feats"])[None], p_s=p_s, p_t=T(p["t_pcd"])[None],
                warped=(torch.matmul(R, p_s.transpose(1, 2)) + t).transpose(1, 2).contiguous(), matches=[T(p["gt_matches"]).t().contiguous()],
                R_gt=R, t_gt=t, src_mask=torch.ones(1, N, dtype=torch.bool), tgt_mask=torch.ones(1, M, dtype=torch.bool))


def focal_case():
    """inputs of the stand-alone compute_correspondence_loss / compute_match_recall vectors"""
    P, N, M = 2, 40, 56
    conf = T(synth.hash_u01(5, 1, P * N * M).reshape(P, N, M).astype(np.float32))
    conf[0, 0, :4] = T(np.array([0.0, 1.0, 1e-7, 1 - 1e-8], dtype=np.float32))
    gt = torch.zeros(P, N, M)
    gi = torch.from_numpy(synth.hash_u01(5, 2, 60)).mul(P * N * M).long()
    gt.view(-1)[gi] = 1.0
    weight = T(synth.hash_u01(5, 3, P * N * M).reshape(P, N, M).astype(np.float32))
    return conf, gt, weight, gi


# ------------------------------------------------------------------------------------------------
# the patch-correspondence block behind the 2D-3D loop (row f4); oracle/make_golden_fine2d3d.py mints its vectors on exactly these inputs
# ------------------------------------------------------------------------------------------------
def fine2d3d_case(seed=5, C=128, n_img=48 * 64, n_pcd=3000, levels=((12, 16, 20), (6, 8, 80)), n_pcd_nodes=96, Kc=128, n_corr=150):
    """Two image levels of (h_c x w_c nodes, Ki pixels per patch), n_pcd_nodes point patches of Kc points (ragged: masks), n_corr node
    correspondences spread over the levels.  Fine features are unit vectors; a share of the points copies the feature of a pixel (+ noise)
    so that similarities beyond the 0.75 threshold exist."""
    u = lambda s, shape: synth.hash_normal(seed, s, shape)
    fi = u(1, (n_img, C)); fi /= np.linalg.norm(fi, axis=1, keepdims=True)
    fp = u(2, (n_pcd, C))
    twin = (synth.hash_u01(seed, 3, n_pcd) * n_img).astype(np.int64)                       # the pixel a point resembles
    has = synth.hash_u01(seed, 4, n_pcd) < 0.5
    fp[has] = fi[twin[has]] + 0.05 * fp[has]
    fp /= np.linalg.norm(fp, axis=1, keepdims=True)
    all_knn, totals, lev, tot = [], [], [], 0
    for li, (h, w, Ki) in enumerate(levels):
        nn_ = h * w
        all_knn.append(T((synth.hash_u01(seed, 10 + li, nn_ * Ki).reshape(nn_, Ki) * n_img).astype(np.int64)))
        totals.append(tot); tot += nn_
        lev.append(np.full(nn_, li, dtype=np.int64))
    pk = (synth.hash_u01(seed, 20, n_pcd_nodes * Kc).reshape(n_pcd_nodes, Kc) * n_pcd).astype(np.int64)
    # every point patch contains the twins of some pixels of image patches, so that real matches fall inside corresponding patches
    sizes = (synth.hash_u01(seed, 21, n_pcd_nodes) * (Kc - 40)).astype(np.int64) + 40
    pm = np.arange(Kc)[None, :] < sizes[:, None]
    pk[~pm] = n_pcd                                                                        # padded entries address the zero row
    img_nodes = (synth.hash_u01(seed, 30, n_corr) * tot).astype(np.int64)
    pcd_nodes = (synth.hash_u01(seed, 31, n_corr) * n_pcd_nodes).astype(np.int64)
    # plant matches: for correspondence c, make the first 12 valid points of its point patch twins of pixels of its image patch
    fp_t = fp.copy()
    lev_all = np.concatenate(lev)
    for c in range(n_corr):
        li = int(lev_all[img_nodes[c]]); loc = int(img_nodes[c] - totals[li])
        pix = all_knn[li][loc].numpy()
        for q in range(12):
            pt = pk[pcd_nodes[c], q]
            if pt < n_pcd:
                v = fi[pix[(7 * q + c) % len(pix)]] + 0.03 * u(40 + c % 7, (C,))
                fp_t[pt] = v / np.linalg.norm(v)
    return dict(img_node_corr_indices=T(img_nodes), pcd_node_corr_indices=T(pcd_nodes), img_node_levels=T(lev_all), all_img_total_nodes=totals,
                all_img_node_knn_indices=all_knn, pcd_node_knn_indices=T(pk), pcd_node_knn_masks=T(pm), img_feats_f=T(fi.astype(np.float32)),
                pcd_feats_f=T(fp_t.astype(np.float32)), img_points_f=T(u(50, (n_img, 3)).astype(np.float32)),
                img_pixels_f=T(u(51, (n_img, 2)).astype(np.float32)), pcd_points_f=T(u(52, (n_pcd, 3)).astype(np.float32)),
                pcd_pixels_f=T(u(53, (n_pcd, 2)).astype(np.float32)))


def train_backward_case(tag):
    """inputs of the matching-head backward vectors (oracle/make_golden_train.py, section "backward")"""
    P, N, M, nv, mv, seed = {"full": (2, 64, 48, 64, 48, 3), "masked": (2, 96, 80, 70, 61, 4), "big": (1, 256, 256, 256, 256, 5)}[tag]
    sc = T((3.0 * synth.hash_normal(seed, 700, (P, N, M))).astype(np.float32))
    gt = torch.zeros(P, N, M)
    k = min(nv, mv) // 2
    for b in range(P):
        i = T((synth.hash_u01(seed, 710 + b, k) * nv).astype(np.int64)); j = T((synth.hash_u01(seed, 720 + b, k) * mv).astype(np.int64))
        gt[b, i, j] = 1.0
        sc[b, i, j] += 6.0
    sm, tm = torch.arange(N)[None].expand(P, N) < nv, torch.arange(M)[None].expand(P, M) < mv
    return sc, gt, sm, tm


# ------------------------------------------------------------------------------------------------
# PnP-RANSAC scenes (row f4): 3D points in front of a camera, their pixels (h, w) with noise, a share of outliers
# ------------------------------------------------------------------------------------------------
def pnp_scene(seed, n=400, outliers=0.4, noise=0.7):
    u = lambda s, shape: synth.hash_normal(seed, s, shape)
    K = np.array([[585.0, 0.0, 320.0], [0.0, 585.0, 240.0], [0.0, 0.0, 1.0]])
    R = synth._rodrigues(u(1, (3,)), 0.5 * float(synth.hash_u01(seed, 2, 1)[0]) + 0.1)
    t = 0.3 * u(3, (3,))
    X = (synth.hash_u01(seed, 4, n * 3).reshape(n, 3) * 2 - 1) * np.array([1.2, 0.9, 0.8]) + np.array([0.0, 0.0, 3.0])
    Y = X @ R.T + t
    uv = np.stack([K[0, 0] * Y[:, 0] / Y[:, 2] + K[0, 2], K[1, 1] * Y[:, 1] / Y[:, 2] + K[1, 2]], 1) + noise * u(5, (n, 2))
    bad = synth.hash_u01(seed, 6, n) < outliers
    uv[bad] = synth.hash_u01(seed, 7, n * 2).reshape(n, 2)[bad] * np.array([640.0, 480.0])
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return X.astype(np.float32), uv[:, ::-1].astype(np.float32).copy(), K, T, ~bad       # pixels as (h, w) rows, like the reference's

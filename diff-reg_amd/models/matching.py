"""models.matching with the reference's interface (3D/models/matching.py): log_optimal_transport,
Matching, mutual_topk_select -- Sinkhorn / projection / similarity run in libdiffreg_hip."""
import torch
import torch.nn as nn

from diffreg_hip import lib
from models.position_encoding import half_tables


def mutual_topk_select(score_mat, k, largest, threshold, mutual, reduce_result=True):
    """3D/models/matching.py:6-59 / pipeline.py:12-65.  The (k=1, largest, no threshold, not mutual)
    case used by the pipeline read-out runs in dr_top1_union; other arguments use torch ops."""
    if k == 1 and largest and threshold is None and not mutual and reduce_result and score_mat.is_cuda:
        m = lib.top1_union(score_mat[None])[0]
        return m[:, 1], m[:, 2], score_mat[m[:, 1], m[:, 2]]
    num_rows, num_cols = score_mat.shape
    dev = score_mat.device
    row_idx = score_mat.topk(k=k, largest=largest, dim=1)[1]
    row_mat = torch.zeros_like(score_mat, dtype=torch.bool)
    row_mat[torch.arange(num_rows, device=dev).view(-1, 1).expand(-1, k), row_idx] = True
    col_idx = score_mat.topk(k=k, largest=largest, dim=0)[1]
    col_mat = torch.zeros_like(score_mat, dtype=torch.bool)
    col_mat[col_idx, torch.arange(num_cols, device=dev).view(1, -1).expand(k, -1)] = True
    corr = torch.logical_and(row_mat, col_mat) if mutual else torch.logical_or(row_mat, col_mat)
    if threshold is not None:
        corr = torch.logical_and(corr, score_mat > threshold if largest else score_mat < threshold)
    if reduce_result:
        r, c = torch.nonzero(corr, as_tuple=True)
        return r, c, score_mat[r, c]
    return corr


def log_optimal_transport(scores, alpha, iters, src_mask, tgt_mask):
    """[B,N,M] -> log assignment [B,N+1,M+1] in scores.dtype (3D/models/matching.py:61-93)."""
    if src_mask is None:
        raise AttributeError("log_optimal_transport needs masks (the reference's mask=None branch raises too, "
                             "matching.py:65-67,79)")
    return lib.sinkhorn(scores, alpha, iters, src_mask, tgt_mask, log_output=True,
                        strict=scores.dtype == torch.float64)


class Matching(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.match_type = config["match_type"]
        self.confidence_threshold = config["confidence_threshold"]
        d_model = config["feature_dim"]
        self.src_proj = nn.Linear(d_model, d_model, bias=False)
        self.tgt_proj = nn.Linear(d_model, d_model, bias=False)      # allocated, never used (quirk Q1)
        self.entangled = config["entangled"]
        if self.match_type == "dual_softmax":
            self.temperature = config["dsmax_temperature"]
        elif self.match_type == "sinkhorn":
            self.skh_init_bin_score = config["skh_init_bin_score"]
            self.skh_iters = config["skh_iters"]
            self.skh_prefilter = config["skh_prefilter"]
            self.bin_score = nn.Parameter(torch.tensor(self.skh_init_bin_score, requires_grad=True))
        else:
            raise NotImplementedError()

    @staticmethod
    @torch.no_grad()
    def get_match(conf_matrix, thr=0.0, mutual=True):
        """(index [K,3], mconf [K], mask [B,N,M]) of matching.py:126-143.  On the device: dr_mutual_match_* (one kernel; the
        read-out the 4DMatch tester applies to conf_matrix_pred, 4D/lib/tester.py:266); the list length K is data dependent, so
        the per-pair counts are read once to size it."""
        if conf_matrix.is_cuda and conf_matrix.dtype in (torch.float32, torch.float64) and max(conf_matrix.shape[1:]) <= 4096:
            B, N, M = conf_matrix.shape
            cap = N + M
            seg, mc, cnt, mask = lib.mutual_match(conf_matrix, thr, mutual, cap=cap, want_mask=True)
            if int(cnt.max()) > cap:                                   # more ties than any sane matrix has: exact, but slower
                seg, mc, cnt, mask = lib.mutual_match(conf_matrix, thr, mutual, cap=N * M, want_mask=True)
                cap = N * M
            keep = torch.arange(cap, device=conf_matrix.device)[None, :] < cnt[:, None]
            return seg[keep], mc[keep], mask.bool()
        mask = conf_matrix > thr
        if mutual:
            mask = mask * (conf_matrix == conf_matrix.max(dim=2, keepdim=True)[0]) \
                        * (conf_matrix == conf_matrix.max(dim=1, keepdim=True)[0])
        index = (mask == True).nonzero()  # noqa: E712
        mconf = conf_matrix[index[:, 0], index[:, 1], index[:, 2]]
        return index, mconf, mask

    get_topk_match = get_match

    @torch.no_grad()
    def forward(self, src_feats, tgt_feats, src_pe, tgt_pe, src_mask, tgt_mask, data, pe_type="rotary"):
        B, N, C = src_feats.shape
        M = tgt_feats.shape[1]
        W = self.src_proj.weight.detach()
        s_np = lib.linear(src_feats.reshape(B * N, C), W)
        t_np = lib.linear(tgt_feats.reshape(B * M, C), W)            # src_proj on both sides (Q1)
        data["src_feats_nopos"], data["tgt_feats_nopos"] = s_np.view(B, N, C), t_np.view(B, M, C)
        if self.entangled:                                            # matching.py:181: the features already carry the code
            s_pos, t_pos = s_np.view(B, N, C), t_np.view(B, M, C)
        elif pe_type == "rotary":
            # (the rotary form comes from the GEMM's epilogue: a second pass over the same small projection; inside the denoising
            #  loop -- dr_denoise_loop -- neither copy is materialised per step)
            cs, ss = half_tables(src_pe)
            ct, st = half_tables(tgt_pe)
            s_pos = lib.linear(src_feats.reshape(B * N, C), W, epilogue=2, cos=cs, sin=ss, rot_C=C).view(B, N, C)
            t_pos = lib.linear(tgt_feats.reshape(B * M, C), W, epilogue=2, cos=ct, sin=st, rot_C=C).view(B, M, C)
        elif pe_type == "sinusoidal":                                 # position_encoding.py:43-44: x + pe
            s_pos, t_pos = s_np.view(B, N, C) + src_pe, t_np.view(B, M, C) + tgt_pe
        else:
            raise KeyError(pe_type)
        data["src_feats"], data["tgt_feats"] = s_pos, t_pos
        a = s_pos / C ** 0.5
        b = t_pos / C ** 0.5
        sim = lib.bmm_nt(a, b)                                        # one strided-batch launch over the pairs
        if self.match_type == "dual_softmax":                         # matching.py:193-205
            conf = lib.dual_softmax(sim, self.temperature, src_mask, tgt_mask)
        else:
            conf = lib.sinkhorn(sim, self.bin_score, self.skh_iters, src_mask, tgt_mask, apply_mask=src_mask is not None)
        coarse_match, _, _ = self.get_match(conf, self.confidence_threshold)
        return conf, coarse_match

"""SoftProcrustesLayer with the reference's interface (3D/models/procrustes.py): top-K + weighted
Kabsch + 3x3 fp64 SVD run on the device in dr_procrustes_f32 (no .cpu() round trip)."""
import torch
import torch.nn as nn

from diffreg_hip import lib


class SoftProcrustesLayer(nn.Module):
    #: False = K from the padded sizes (Diff-Reg-3dmatch, 2d3d); True = K from the mask sums
    #: (Diff-Reg-4dmatch/models/procrustes.py:61-62, quirk Q17)
    use_mask_len = False

    def __init__(self, config):
        super().__init__()
        get = (lambda k: config[k]) if isinstance(config, dict) else (lambda k: getattr(config, k))
        self.sample_rate = get("sample_rate")
        self.max_condition_num = get("max_condition_num")

    @torch.no_grad()
    def forward(self, conf_matrix, src_pcd, tgt_pcd, src_mask, tgt_mask, strict_reference=False):
        """-> R, t, R_forwd, t_forwd, condition, solution_mask.

        A float64 conf makes the reference raise inside batch_weighted_procrustes and fall back to
        identity through its bare `except` (quirk Q3).  By default the well-defined value on
        float32(conf) is returned; strict_reference=True reproduces the identity fallback."""
        B = conf_matrix.shape[0]
        if conf_matrix.dtype == torch.float64:
            if strict_reference:
                R = torch.eye(3, dtype=torch.float64, device=conf_matrix.device)[None].repeat(B, 1, 1)
                t = torch.zeros(B, 3, 1, dtype=torch.float64, device=conf_matrix.device)
                cond = torch.zeros(B, dtype=torch.float64, device=conf_matrix.device)
                ok = cond < self.max_condition_num
                Rf, tf = R.clone(), t.clone()
                return R, t, Rf, tf, cond, ok
            conf_matrix = conf_matrix.float()
        return lib.procrustes(conf_matrix, src_pcd, tgt_pcd, src_mask, tgt_mask, self.sample_rate, self.max_condition_num,
                              use_mask_len=self.use_mask_len)

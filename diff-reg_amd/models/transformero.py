"""GeometryAttentionLayer / RepositioningTransformer with the reference's interface
(3D/models/transformero.py; Diff-Reg-4dmatch/models/transformer.py is the same file)."""
import copy

import torch
from torch import nn

from diffreg_hip import lib
from models.matching import Matching
from models.position_encoding import VolumetricPositionEncoding as VolPE, half_tables
from models.procrustes import SoftProcrustesLayer


class GeometryAttentionLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        d_model, nhead = config["feature_dim"], config["n_head"]
        self.dim, self.nhead, self.pe_type = d_model // nhead, nhead, config["pe_type"]
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(d_model * 2, d_model * 2, bias=False), nn.ReLU(True),
                                 nn.Linear(d_model * 2, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def weight_tensors(self):
        sd = self.state_dict()
        return [sd[k].detach().float().contiguous() for k in lib._LAYER_KEYS]

    @torch.no_grad()
    def forward(self, x, source, x_pe, source_pe, x_mask=None, source_mask=None):
        C = self.dim * self.nhead
        if self.pe_type == "rotary":
            if x_pe is None:                                   # entangled: the code is already in the features (transformero.py:65)
                return lib.attention_layer(self.weight_tensors(), C, self.nhead, x, source, x_mask=x_mask, y_mask=source_mask)
            cx, sx = half_tables(x_pe)
            cy, sy = half_tables(source_pe)
            return lib.attention_layer(self.weight_tensors(), C, self.nhead, x, source, cx, sx, cy, sy, x_mask, source_mask)
        if self.pe_type == "sinusoidal":                       # w (x + p): q and k see the code, v and the residual do not (transformero.py:50-57)
            if x_pe is None:
                return lib.attention_layer(self.weight_tensors(), C, self.nhead, x, source, x_mask=x_mask, y_mask=source_mask)
            return lib.attention_layer(self.weight_tensors(), C, self.nhead, x, source, x_mask=x_mask, y_mask=source_mask,
                                       xq=x + x_pe, yk=source + source_pe)
        raise KeyError(self.pe_type)


class RepositioningTransformer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.d_model, self.nhead = config["feature_dim"], config["n_head"]
        self.layer_types = config["layer_types"]
        self.positioning_type = config["positioning_type"]
        self.pe_type, self.entangled = config["pe_type"], config["entangled"]
        self.positional_encoding = VolPE(config)
        encoder_layer = GeometryAttentionLayer(config)
        self.layers = nn.ModuleList()
        for l_type in self.layer_types:
            if l_type in ("self", "cross"):
                self.layers.append(copy.deepcopy(encoder_layer))
            elif l_type == "positioning":
                if self.positioning_type == "procrustes":
                    pos = nn.ModuleList()
                    pos.append(Matching(config["feature_matching"]))
                    pos.append(SoftProcrustesLayer(config["procrustes"]))
                    self.layers.append(pos)
                elif self.positioning_type in ("oracle", "randSO3"):
                    self.layers.append(None)
                else:
                    raise KeyError(self.positioning_type + " undefined positional encoding type")
            else:
                raise KeyError(l_type)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    @torch.no_grad()
    def forward(self, src_feat, tgt_feat, s_pcd, t_pcd, src_mask, tgt_mask, data, T=None, timers=None):
        self.timers = timers
        assert self.d_model == src_feat.size(2), "the feature number of src and transformer must be equal"
        if T is not None:
            R, t = T
            src_w = (torch.matmul(R, s_pcd.transpose(1, 2)) + t).transpose(1, 2)
        else:
            src_w = s_pcd
        src_pe, tgt_pe = self.positional_encoding(src_w), self.positional_encoding(t_pcd)
        position_layer = 0
        data.update({"position_layers": {}})
        if self.entangled:
            # transformero.py:234-254: the position code enters the features once, the layers run without one, positioning layers are skipped
            src_feat = VolPE.embed_pos(self.pe_type, src_feat, src_pe)
            tgt_feat = VolPE.embed_pos(self.pe_type, tgt_feat, tgt_pe)
            for layer, name in zip(self.layers, self.layer_types):
                if name == "self":
                    src_feat = layer(src_feat, src_feat, None, None, src_mask, src_mask)
                    tgt_feat = layer(tgt_feat, tgt_feat, None, None, tgt_mask, tgt_mask)
                elif name == "cross":
                    src_feat = layer(src_feat, tgt_feat, None, None, src_mask, tgt_mask)
                    tgt_feat = layer(tgt_feat, src_feat, None, None, tgt_mask, src_mask)
            return src_feat, tgt_feat, src_pe, tgt_pe
        for layer, name in zip(self.layers, self.layer_types):
            if name == "self":
                src_feat = layer(src_feat, src_feat, src_pe, src_pe, src_mask, src_mask)
                tgt_feat = layer(tgt_feat, tgt_feat, tgt_pe, tgt_pe, tgt_mask, tgt_mask)
            elif name == "cross":
                src_feat = layer(src_feat, tgt_feat, src_pe, tgt_pe, src_mask, tgt_mask)
                tgt_feat = layer(tgt_feat, src_feat, tgt_pe, src_pe, tgt_mask, src_mask)     # updated src (Q11)
            elif name == "positioning":
                if self.positioning_type == "procrustes":
                    conf, match_pred = layer[0](src_feat, tgt_feat, src_pe, tgt_pe, src_mask, tgt_mask, data, pe_type=self.pe_type)
                    position_layer += 1
                    data["position_layers"][position_layer] = {"conf_matrix": conf, "match_pred": match_pred}
                    R, t, R_forwd, t_forwd, condition, solution_mask = layer[1](conf, s_pcd, t_pcd, src_mask, tgt_mask)
                    data["position_layers"][position_layer].update(
                        {"R_s2t_pred": R, "t_s2t_pred": t, "solution_mask": solution_mask, "condition": condition})
                    src_w = (torch.matmul(R_forwd, s_pcd.transpose(1, 2)) + t_forwd).transpose(1, 2)
                elif self.positioning_type == "randSO3":          # transformero.py:202-206
                    src_w = self.rand_rot_pcd(s_pcd, src_mask)
                elif self.positioning_type == "oracle":           # transformero.py:209-216: the ground-truth pose re-poses the source
                    src_w = (torch.matmul(data["batched_rot"], s_pcd.transpose(1, 2)) + data["batched_trn"]).transpose(1, 2)
                else:
                    raise KeyError(self.positioning_type + " undefined positional encoding type")
                src_pe, tgt_pe = self.positional_encoding(src_w), self.positional_encoding(t_pcd)
            else:
                raise KeyError(name)
        return src_feat, tgt_feat, src_pe, tgt_pe

    def rand_rot_pcd(self, pcd, mask):
        """transformero.py:261-280: a random rotation (numpy's global generator, Euler zyx) about the masked centroid; like the reference
        it zeroes the padded rows of `pcd` IN PLACE"""
        import numpy as np
        from scipy.spatial.transform import Rotation
        pcd[~mask] = 0.
        n = mask.shape[1]
        count = mask.sum(dim=1, keepdim=True).view(-1, 1, 1)
        angles = np.random.rand(pcd.shape[0], 3) * np.pi * 2
        rot = torch.from_numpy(Rotation.from_euler("zyx", angles).as_matrix()).to(pcd)
        centre = pcd.mean(dim=1, keepdim=True) * n / count
        return torch.matmul(rot, (pcd - centre).transpose(1, 2)).transpose(1, 2) + centre

"""VolumetricPositionEncoding with the reference's interface (3D/models/position_encoding.py:5-87);
the sinusoid bank is computed by dr_vol_pe_f32."""
import torch
from torch import nn

from diffreg_hip import lib


def _cfg(config, key):
    return config[key] if isinstance(config, dict) else getattr(config, key)


class VolumetricPositionEncoding(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.feature_dim = _cfg(config, "feature_dim")
        self.vol_bnds = _cfg(config, "vol_bnds")
        self.voxel_size = _cfg(config, "voxel_size")
        self.vol_origin = self.vol_bnds[0]
        self.pe_type = _cfg(config, "pe_type")

    def voxelize(self, xyz):
        origin = torch.as_tensor(self.vol_origin, dtype=torch.float32, device=xyz.device).view(1, 1, -1)
        return (xyz - origin) / self.voxel_size

    @staticmethod
    def embed_rotary(x, cos, sin):
        """x*cos + swap(x)*sin with swap(x)[2k] = -x[2k+1], swap(x)[2k+1] = x[2k]."""
        x2 = torch.stack([-x[..., 1::2], x[..., ::2]], dim=-1).reshape_as(x).contiguous()
        return x * cos + x2 * sin

    @staticmethod
    def embed_pos(pe_type, x, pe):
        if pe_type == "rotary":
            return VolumetricPositionEncoding.embed_rotary(x, pe[..., 0], pe[..., 1])
        if pe_type == "sinusoidal":
            return x + pe
        raise KeyError(pe_type)

    def tables(self, XYZ):
        """[B,N,3] -> un-duplicated (cos, sin) tables [B*N, C/2] (the layout the HIP kernels consume)."""
        B, N, _ = XYZ.shape
        return lib.vol_pe(XYZ.reshape(B * N, 3), self.feature_dim, self.vol_origin, self.voxel_size)

    def forward(self, XYZ):
        """[B,N,3] -> position code [B,N,C,2] (cos, sin; each angle on two adjacent channels)."""
        B, N, _ = XYZ.shape
        cos, sin = self.tables(XYZ)
        C = self.feature_dim
        if self.pe_type == "sinusoidal":
            # position_encoding.py:68-69: cat[sin x, cos x, sin y, cos y, sin z, cos z], each C / 6 wide -- the same sinusoid bank
            # dr_vol_pe_f32 computes for the rotary form (tables = [x | y | z] thirds of C / 6 angles), re-ordered
            k = C // 6
            c3, s3 = cos.view(B, N, 3, k), sin.view(B, N, 3, k)
            return torch.stack([s3, c3], dim=3).reshape(B, N, 6 * k)
        if self.pe_type != "rotary":
            raise KeyError(self.pe_type)
        cos = cos.view(B, N, C // 2).repeat_interleave(2, dim=-1)
        sin = sin.view(B, N, C // 2).repeat_interleave(2, dim=-1)
        return torch.stack([cos, sin], dim=-1)


def half_tables(pe):
    """position code [B,N,C,2] -> (cos, sin) [B*N, C/2]."""
    B, N, C, _ = pe.shape
    return (pe[..., 0::2, 0].reshape(B * N, C // 2).contiguous(), pe[..., 0::2, 1].reshape(B * N, C // 2).contiguous())

"""Shared builders for the parity tests: synthetic pairs/weights as torch tensors."""
import numpy as np
import torch

from diffreg_hip import synth

HEAD_GAIN = 24.0   # must match oracle/make_golden.py


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def weights(variant):
    C = synth.VARIANTS[variant]["C"]
    return {k: T(v) for k, v in synth.make_weights(C, seed=7, head_gain=HEAD_GAIN).items()}


def pair(variant, N, M, seed):
    C = synth.VARIANTS[variant]["C"]
    p = synth.make_pair(N, M, C, seed=seed)
    return p, dict(f_s=T(p["src_feats"])[None], f_t=T(p["tgt_feats"])[None], p_s=T(p["s_pcd"])[None],
                   p_t=T(p["t_pcd"])[None], x_T=T(p["x_T"])[None])


def masks(N, M, nv=None, mv=None):
    nv = N if nv is None else nv
    mv = M if mv is None else mv
    return torch.arange(N)[None] < nv, torch.arange(M)[None] < mv


def sinkhorn_case(N, M, nv, mv, dtype):
    sc = T(3.0 * synth.hash_normal(1, N * 1000 + M, (1, N, M))).to(dtype)
    sm, tm = masks(N, M, nv, mv)
    return sc.masked_fill(~(sm[:, :, None] & tm[:, None, :]), float("-inf")), sm, tm

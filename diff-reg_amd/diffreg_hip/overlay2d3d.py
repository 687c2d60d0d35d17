"""Drop-in acceleration of the 2D-3D reverse sampling for an UNMODIFIED Diff-Reg-2d3d checkout.

The loop of MATR2D3D.forward (EXP/model.py:637-694) is inline code: per step it calls, in this order,

    self.get_warped_from_noising_matching3D3D(s_pcd, t_pcd_da, src_mask, tgt_mask_da, x)   -> warped points (+ R, t)      :655-656
    self.denoising_transformer(img_feats, img_dino, img_pixels, pcd_feats, warped)         -> fused img / pcd features   :658-664
    self.denoising_coarse_matching(pcd_feats, img_feats, src_mask, tgt_mask, True)         -> x_start (+ 3 unused)       :665-666

and then updates x with a few element-wise torch ops (:668-680).  `accelerate(model)` re-binds these three call sites ON THE
INSTANCE (the module's weights, state_dict and training branch are untouched): during the first step of an eval forward the
three calls record their arguments, the third one runs the WHOLE loop once in libdiffreg_hip (dr_denoise_loop_2d3d, through
DenoiseEngine2D3D, started from the very x the reference drew) and every step's matching call then hands back the x_start of
that step from the device trace.  The reference's own DDIM arithmetic, final Sinkhorn and read-out (:682-694) run unchanged on
those x_start, so `conf_matrix_pred` and everything behind it are the reference's code on the accelerated loop's values.

    from diffreg_hip.overlay2d3d import accelerate
    model = create_model(cfg).cuda().eval();  accelerate(model)          # EXP/eval.py, after loading the checkpoint

Under model.train() the three calls fall through to the original code.
"""
import types

import torch

from .engine import DenoiseEngine2D3D


class LoopOverlay2D3D:
    def __init__(self, model, n_head=4, engine_kwargs=None):
        self.model = model
        self.n_head = n_head
        self.engine_kwargs = dict(engine_kwargs or {})
        self.engine = None
        self._k = 0                 # step inside the current eval forward
        self._rec = {}
        self._out = None
        t, m = model.denoising_transformer, model.denoising_coarse_matching
        self._orig = dict(warp=model.get_warped_from_noising_matching3D3D, transformer=t.forward, matching=m.forward)
        model.get_warped_from_noising_matching3D3D = self._warp
        t.forward = self._transformer
        m.forward = self._matching
        model._dr_overlay = self

    def remove(self):
        """restore the three call sites"""
        m = self.model
        for obj, name in ((m, "get_warped_from_noising_matching3D3D"), (m.denoising_transformer, "forward"), (m.denoising_coarse_matching, "forward")):
            if name in obj.__dict__:
                del obj.__dict__[name]
        m.__dict__.pop("_dr_overlay", None)

    def refresh(self):
        """call after the weights changed (load_state_dict, .to()): the engine holds its own device copy"""
        self.engine = None

    # ---- engine from the model's own weights and hyper-parameters ------------------------------------------------------------
    def _build(self, device):
        sd = {k: v for k, v in self.model.state_dict().items() if k.startswith("denoising_transformer.") or k.startswith("denoising_coarse_matching.")}
        p = "denoising_transformer."
        n_layers = 1 + max(int(k[len(p + "transformer."):].split(".")[0]) for k in sd if k.startswith(p + "transformer."))
        proc, match = self.model.denoising_soft_procrustes, self.model.denoising_coarse_matching
        try:                                 # vision3d TransformerLayer -> AttentionLayer -> MultiHeadAttention.num_heads (vision3d/layers/transformer.py:29)
            self.n_head = int(self.model.denoising_transformer.transformer[0].attention.attention.num_heads)
        except (AttributeError, IndexError, TypeError):
            pass
        kw = dict(C=sd[p + "out_proj.weight"].shape[0], H=self.n_head, n_layers=n_layers, img_dim=sd[p + "img_in_proj.weight"].shape[1],
                  dino_dim=sd[p + "img_in_proj_dino.weight"].shape[1], pcd_dim=sd[p + "pcd_in_proj.weight"].shape[1],
                  steps=int(self.model.sampling_timesteps), sk_iters=int(match.skh_iters), sample_rate=float(proc.sample_rate),
                  max_condition_num=float(proc.max_condition_num), device=device)
        kw.update(self.engine_kwargs)
        self.engine = DenoiseEngine2D3D(sd, **kw)

    # ---- the three call sites ------------------------------------------------------------------------------------------------
    def _warp(self, s_pcd, t_pcd, src_mask, tgt_mask, matrix):
        if self.model.training:
            return self._orig["warp"](s_pcd, t_pcd, src_mask, tgt_mask, matrix)
        if self._k == 0:
            self._rec = dict(s_pcd=s_pcd, t_pcd_da=t_pcd, src_mask=src_mask, tgt_mask_da=tgt_mask, x_T=matrix.detach().clone())
            self._out = None
        if src_mask is not None:            # the reference masks x IN PLACE here (:832-834) and the -inf entries persist in its x
            matrix.masked_fill_(~(src_mask[..., None] * tgt_mask[:, None]).bool(), float("-inf"))
        if self._out is not None:           # steps >= 1: this step's warp from the device trace
            Rf, tf = self._out["R_forwd"][self._k], self._out["t_forwd"][self._k]
            return (torch.matmul(Rf, s_pcd.transpose(1, 2)) + tf).transpose(1, 2), t_pcd.type(torch.float32), Rf, tf
        # step 0: the loop has not run yet (its remaining inputs arrive with the next two calls); nothing on this path reads these
        eye = torch.eye(3, device=s_pcd.device)[None].expand(s_pcd.shape[0], 3, 3)
        return s_pcd.type(torch.float32), t_pcd.type(torch.float32), eye, torch.zeros(s_pcd.shape[0], 3, 1, device=s_pcd.device)

    def _transformer(self, img_feats, img_dino, img_pixels, pcd_feats, pcd_points):
        if self.model.training:
            return self._orig["transformer"](img_feats, img_dino, img_pixels, pcd_feats, pcd_points)
        if self._k == 0:
            self._rec.update(img_feats=img_feats, img_dino=img_dino, img_pixels=img_pixels, pcd_feats=pcd_feats)
        if self._out is not None:           # the fused features of the LAST step (the loop keeps no others)
            return self._out["img_feats"], self._out["pcd_feats"]
        return img_feats, pcd_feats         # step 0 placeholders, consumed by the matching call below only

    def _matching(self, src_feats, tgt_feats, src_mask, tgt_mask, *args, **kwargs):
        if self.model.training:
            return self._orig["matching"](src_feats, tgt_feats, src_mask, tgt_mask, *args, **kwargs)
        if self._k == 0:
            r = self._rec
            dev = r["s_pcd"].device
            if self.engine is None or self.engine.device != dev:
                self._build(dev)
            masks = None if src_mask is None else (src_mask, tgt_mask, r["tgt_mask_da"])
            f = lambda t_: t_.detach().float()
            self._out = self.engine.run(f(r["img_feats"]), f(r["img_dino"]), f(r["img_pixels"]), f(r["pcd_feats"]), f(r["s_pcd"]), f(r["t_pcd_da"]),
                                        f(r["x_T"]), masks, trace=True)
        x_start = self._out["x0"][self._k]
        self._k += 1
        if self._k >= self.engine.steps:
            self._k = 0
            self._rec = {}
        return x_start, None, None, None


def accelerate(model, n_head=4, **engine_kwargs):
    """install the overlay on a MATR2D3D instance (see the module docstring); returns the LoopOverlay2D3D (`.remove()` undoes it)"""
    return LoopOverlay2D3D(model, n_head=n_head, engine_kwargs=engine_kwargs)

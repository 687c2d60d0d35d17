"""Host side of the reverse-diffusion matching engine: owns the device copies of the denoiser /
matching-head weights, the diffusion schedule, the workspace and (optionally) a captured HIP graph of
the whole loop, and calls dr_denoise_loop through the C ABI.

Mirrors the eval branch of Pipeline.forward (3D/models/pipeline.py:221-283, 4D/models/pipeline.py:156-197)
for P independent scene pairs (the reference runs B = 1; P pairs are P independent B = 1 problems)."""
import ctypes
import math

import numpy as np
import torch

from . import lib

VARIANTS = {"3dmatch": 0, "4dmatch": 1}


def cosine_alphas_cumprod(timesteps=1000, s=0.008):
    """float64 alphas_cumprod of cosine_beta_schedule (3D/models/pipeline.py:83-93,151-156)."""
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    return torch.cumprod(1.0 - betas, dim=0)


def sampling_times(steps, timesteps=1000):
    """reversed int(linspace(0, T-1, steps+1)) (pipeline.py:229-231, quirk Q20)."""
    t = torch.linspace(0, timesteps - 1, steps=steps + 1)
    return list(reversed(t.int().tolist()))


class DenoiseEngine:
    def __init__(self, state, *, variant, C, H, voxel, origin, steps, sk_iters=3, sample_rate=1.0, max_condition_num=0.0,
                 n_layers=6, device="cuda:0", strict_f64=False, prefix_t="denoising_transformer.",
                 prefix_m="denoising_coarse_matching."):
        """state: mapping name -> tensor in the reference state-dict layout."""
        lib.ensure_init()
        self.device = torch.device(device)
        self.variant, self.C, self.H, self.n_layers = variant, C, H, n_layers
        self.steps = steps
        dev = self.device
        f = lambda k: state[k].detach().to(device=dev, dtype=torch.float32).contiguous()
        self._tensors = []
        self._layers = (lib.LayerWeights * n_layers)()
        for l in range(n_layers):
            ts = [f(prefix_t + "layers.%d.%s" % (l, k)) for k in lib._LAYER_KEYS]
            self._tensors.append(ts)
            self._layers[l] = lib.layer_weights(ts)
        self.src_proj = f(prefix_m + "src_proj.weight")
        self.bin_score = f(prefix_m + "bin_score").reshape(1)
        self.freq = lib.pe_freq(C, dev)
        self.w = lib.LoopWeights()
        self.w.layers = ctypes.cast(self._layers, ctypes.POINTER(lib.LayerWeights))
        self.w.src_proj = self.src_proj.data_ptr()
        self.w.bin_score = self.bin_score.data_ptr()
        self.w.pe_freq = self.freq.data_ptr()
        self._ac = np.ascontiguousarray(cosine_alphas_cumprod().numpy())
        self._times = np.ascontiguousarray(np.asarray(sampling_times(steps), dtype=np.int32))
        cfg = lib.LoopConfig()
        cfg.variant = VARIANTS[variant]
        cfg.C, cfg.H, cfg.n_layers, cfg.steps, cfg.sk_iters = C, H, n_layers, steps, sk_iters
        cfg.voxel = voxel
        cfg.origin[0], cfg.origin[1], cfg.origin[2] = origin
        cfg.sample_rate, cfg.max_condition_num = sample_rate, max_condition_num
        cfg.flags = 1 if strict_f64 else 0
        cfg.h_alphas_cumprod = self._ac.ctypes.data
        cfg.h_times = self._times.ctypes.data
        self.cfg = cfg
        self._ws = None
        self._graphs = {}

    # ------------------------------------------------------------------------------------------
    def _workspace(self, P, N, M):
        need = lib.raw().dr_denoise_loop_workspace_bytes(ctypes.byref(self.cfg), P, N, M)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    def denoise_match(self, src_feats, tgt_feats, s_pcd_warped, t_pcd, src_mask=None, tgt_mask=None):
        """one denoiser + matching-head evaluation: -> (src_out, tgt_out, conf)"""
        P, N, C = src_feats.shape
        M = tgt_feats.shape[1]
        ws, need = self._workspace(P, N, M)
        so, to = torch.empty_like(src_feats), torch.empty_like(tgt_feats)
        conf = torch.empty(P, N, M, device=self.device)
        sm, tm = lib.mask_u8(src_mask), lib.mask_u8(tgt_mask)
        lib.check(lib.raw().dr_denoiser_match_f32(
            ctypes.byref(self.cfg), ctypes.byref(self.w), P, N, M, lib.ptr(src_feats.contiguous()), lib.ptr(tgt_feats.contiguous()),
            lib.ptr(s_pcd_warped.contiguous()), lib.ptr(t_pcd.contiguous()), lib.ptr(sm), lib.ptr(tm), lib.ptr(so), lib.ptr(to),
            lib.ptr(conf), lib.ptr(ws), need, lib.stream_of(src_feats)))
        return so, to, conf

    def _enqueue(self, bufs):
        b = bufs
        tr = None
        if b.get("trace"):
            tr = lib.LoopTrace()
            tr.x0, tr.R_forwd, tr.t_forwd, tr.cond = (b["tr_x0"].data_ptr(), b["tr_R"].data_ptr(), b["tr_t"].data_ptr(),
                                                      b["tr_cond"].data_ptr())
        lib.check(lib.raw().dr_denoise_loop(
            ctypes.byref(self.cfg), ctypes.byref(self.w), b["P"], b["N"], b["M"], lib.ptr(b["src_feats"]), lib.ptr(b["tgt_feats"]),
            lib.ptr(b["s_pcd"]), lib.ptr(b["t_pcd"]), lib.ptr(b["src_mask"]), lib.ptr(b["tgt_mask"]), lib.ptr(b["x_T"]),
            lib.ptr(b["noise"]), lib.ptr(b["conf"]), lib.ptr(b["x_final"]), lib.ptr(b["matches"]), lib.ptr(b["match_count"]),
            lib.ptr(b["R_final"]), lib.ptr(b["t_final"]), ctypes.byref(tr) if tr is not None else None, lib.ptr(b["ws"]),
            b["ws_bytes"], lib.stream_of(b["conf"])))

    def make_buffers(self, P, N, M, masked=False, trace=False, private_ws=False):
        dev, C, S = self.device, self.C, self.steps
        ws, need = self._workspace(P, N, M)
        if private_ws:          # concurrent batches must not share scratch memory
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        b = dict(P=P, N=N, M=M, ws=ws, ws_bytes=need, trace=trace,
                 src_feats=torch.zeros(P, N, C, device=dev), tgt_feats=torch.zeros(P, M, C, device=dev),
                 s_pcd=torch.zeros(P, N, 3, device=dev), t_pcd=torch.zeros(P, M, 3, device=dev),
                 src_mask=torch.ones(P, N, dtype=torch.uint8, device=dev) if masked else None,
                 tgt_mask=torch.ones(P, M, dtype=torch.uint8, device=dev) if masked else None,
                 x_T=torch.zeros(P, N, M, device=dev),
                 noise=torch.zeros(S, P, N, M, device=dev) if self.variant == "4dmatch" else None,
                 conf=torch.empty(P, N, M, dtype=torch.float64, device=dev),
                 x_final=torch.empty(P, N, M, dtype=torch.float64, device=dev),
                 matches=torch.zeros(P, N + M, 3, dtype=torch.int64, device=dev) if self.variant == "3dmatch" else None,
                 match_count=torch.zeros(P, dtype=torch.int32, device=dev) if self.variant == "3dmatch" else None,
                 R_final=torch.empty(P, 3, 3, device=dev), t_final=torch.empty(P, 3, 1, device=dev))
        if trace:
            b.update(tr_x0=torch.empty(S, P, N, M, device=dev), tr_R=torch.empty(S, P, 3, 3, device=dev),
                     tr_t=torch.empty(S, P, 3, 1, device=dev), tr_cond=torch.empty(S, P, dtype=torch.float64, device=dev))
        return b

    def run(self, src_feats, tgt_feats, s_pcd, t_pcd, x_T, src_mask=None, tgt_mask=None, noise=None, trace=False,
            graph=False, _slot=0):
        """Run the loop for P pairs.  Returns a dict of device tensors (conf float64, x_final, matches list (3D),
        R_final, t_final, and the per-step trace when asked)."""
        P, N, C = src_feats.shape
        M = tgt_feats.shape[1]
        masked = src_mask is not None
        key = (P, N, M, masked, trace, bool(graph), _slot)
        ent = self._graphs.get(key)
        if ent is None:
            b = self.make_buffers(P, N, M, masked=masked, trace=trace, private_ws=_slot > 0)
            g = None
            if graph:
                self._fill(b, src_feats, tgt_feats, s_pcd, t_pcd, x_T, src_mask, tgt_mask, noise)
                s = torch.cuda.Stream(device=self.device)
                s.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(s):
                    self._enqueue(b)                      # warm-up outside the capture
                torch.cuda.current_stream(self.device).wait_stream(s)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._enqueue(b)
            ent = (b, g)
            self._graphs[key] = ent
        b, g = ent
        self._fill(b, src_feats, tgt_feats, s_pcd, t_pcd, x_T, src_mask, tgt_mask, noise)
        if g is not None:
            g.replay()
        else:
            self._enqueue(b)
        return self._collect(b)

    def _fill(self, b, src_feats, tgt_feats, s_pcd, t_pcd, x_T, src_mask, tgt_mask, noise):
        b["src_feats"].copy_(src_feats); b["tgt_feats"].copy_(tgt_feats)
        b["s_pcd"].copy_(s_pcd); b["t_pcd"].copy_(t_pcd); b["x_T"].copy_(x_T)
        if b["src_mask"] is not None:
            b["src_mask"].copy_(lib.mask_u8(src_mask)); b["tgt_mask"].copy_(lib.mask_u8(tgt_mask))
        if b["noise"] is not None:
            if noise is None:
                raise RuntimeError("the 4dmatch variant adds sigma*noise every step: pass noise [steps,P,N,M]")
            b["noise"].copy_(noise)

    def _collect(self, b):
        out = dict(conf_matrix_pred=b["conf"], x_final=b["x_final"], R_final=b["R_final"], t_final=b["t_final"])
        if b["matches"] is not None:
            out["matches_padded"], out["match_count"] = b["matches"], b["match_count"]
        if b["trace"]:
            out.update(x0=b["tr_x0"], R_forwd=b["tr_R"], t_forwd=b["tr_t"], cond=b["tr_cond"])
        return out

    # ------------------------------------------------------------------------------------------
    def run_streams(self, groups, n_streams=2):
        """Run several independent batches of pairs concurrently, one captured graph per batch, replayed on
        `n_streams` HIP streams so that the tails / small launches of one batch overlap the big launches of another.
        groups: list of dicts with the keyword arguments of run() (src_feats, tgt_feats, s_pcd, t_pcd, x_T, ...).
        Returns the list of result dicts (buffers are per group and stay valid until the group is run again)."""
        cur = torch.cuda.current_stream(self.device)
        if not hasattr(self, "_streams") or len(self._streams) < n_streams:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]
        outs = []
        for gi, kw in enumerate(groups):
            st = self._streams[gi % n_streams]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(self.run(graph=True, _slot=gi, **kw))
        for st in self._streams[:n_streams]:
            cur.wait_stream(st)
        return outs

    @staticmethod
    def match_list(out):
        """[K_p,3] int64 tensors (one host sync)."""
        cnt = out["match_count"].cpu().tolist()
        return [out["matches_padded"][p, :cnt[p]] for p in range(len(cnt))]

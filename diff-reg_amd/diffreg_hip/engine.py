"""Host side of the reverse-diffusion matching engine: owns the device copies of the denoiser /
matching-head weights, the diffusion schedule, the workspace and (optionally) a captured HIP graph of
the whole loop, and calls dr_denoise_loop through the C ABI.

Mirrors the eval branch of Pipeline.forward (3D/models/pipeline.py:221-283, 4D/models/pipeline.py:156-197)
for P independent scene pairs (the reference runs B = 1; P pairs are P independent B = 1 problems)."""
import collections
import ctypes
import math

import numpy as np
import os
import threading
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



_STREAM_POOL = {}
_STREAM_POOL_LOCK = threading.Lock()


class _CallStatus:
    """Handle on the status word of the workspace a loop call ran on (dr_denoise_loop_status).  Travels in the result dict as `_status`;
    .check() waits for the current stream and raises if THAT call's co-resident Sinkhorn timed out -- per workspace, so concurrent batches /
    engines are told apart (the process-wide dr_device_status flag cannot).  clone() returns the handle itself (result dicts are cloned key by key)."""

    def __init__(self, ws):
        self.ws = ws

    def check(self, clear=True):
        lib.loop_status(self.ws, clear)

    def clone(self):
        return self


def _side_streams(device, n):
    """The side streams of run_streams, shared by every engine of the process: the runtime maps streams onto a handful of hardware queues in
    creation order, so engines that each created their own would end up with streams that share a queue (measured: three concurrent cfg3 calls
    163 pairs/s behind other engines' streams against 240 with the first three streams of a process).
    The pool is per DEVICE INDEX (a bare "cuda" means the current device) and guarded by a lock; the streams themselves are shared: run_streams
    calls issued from different host threads at the same time interleave their work on the same side streams (each call still waits for and is
    waited on by its own caller's stream, so results are correct; the overlap is then no longer per engine)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    with _STREAM_POOL_LOCK:
        pool = _STREAM_POOL.setdefault(key, [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(device=torch.device("cuda", key)))
        return pool[:n]


class DenoiseEngine:
    def __init__(self, state, *, variant, C, H, voxel, origin, steps, sk_iters=3, sample_rate=1.0, max_condition_num=0.0,
                 n_layers=6, device="cuda:0", strict_f64=False, prefix_t="denoising_transformer.",
                 prefix_m="denoising_coarse_matching.", prepack=True, planes=None, cache_entries=4, attn_f16=False):
        """state: mapping name -> tensor in the reference state-dict layout.
        cache_entries: run() keeps static buffers (and, with graph=True, a captured HIP graph) per call shape; at most this many
        shapes stay cached, least recently used first out (its buffers and graph are freed)."""
        lib.ensure_init()
        self.device = torch.device(device)
        self.variant, self.C, self.H, self.n_layers = variant, C, H, n_layers
        self.steps = steps
        dev = self.device
        # A private snapshot of every weight (copy=True: never an alias of a live nn.Parameter): the f32 kernels read these tensors
        # and the plane path reads images packed from them below, so both GEMM paths always see the same values.  A caller whose
        # parameters change (optimizer.step(), in-place edits) builds a new engine -- models.pipeline.Pipeline does so by itself
        # from the parameters' version counters.
        f = lambda k: state[k].detach().to(device=dev, dtype=torch.float32, copy=True).contiguous()
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
        # planes: None = the size rule picks the GEMM path; True / False = DR_LOOP_PLANES_FORCE / DR_LOOP_PLANES_OFF
        # attn_f16: the OPT-IN reduced-precision attention of the plane path (DR_LOOP_ATTN_F16: one fp16 product per contraction)
        cfg.flags = (1 if strict_f64 else 0) | (4 if planes is True else 0) | (8 if planes is False else 0) | (16 if attn_f16 else 0)
        cfg.h_alphas_cumprod = self._ac.ctypes.data
        cfg.h_times = self._times.ctypes.data
        self.cfg = cfg
        self._ws = None
        self._graphs = collections.OrderedDict()
        self._cache_entries = max(1, int(cache_entries))
        # the snapshot above is immutable for the life of the engine: its plane images are packed once (dr_loop_prepack)
        self._packed = None
        nb = lib.raw().dr_loop_prepack_bytes(ctypes.byref(cfg))
        if nb and prepack:
            self._packed = torch.empty(nb, dtype=torch.uint8, device=dev)
            lib.check(lib.raw().dr_loop_prepack(ctypes.byref(cfg), ctypes.byref(self.w), self._packed.data_ptr(), nb,
                                                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            self.w.prepacked = self._packed.data_ptr()
            # the pack kernels were enqueued on the stream current at construction; runs may come on other streams (run_streams): order them
            torch.cuda.current_stream(dev).synchronize()

    # ------------------------------------------------------------------------------------------
    GUARD = 64 * 1024          # bytes of 0xA5 on either side of every buffer when guard bands are on (tests)

    def enable_guards(self, on=True):
        """Tests: allocate every static buffer (inputs, outputs, workspace) between two 64 KiB bands of 0xA5 and let
        check_guards() verify them -- an out-of-bounds write of any kernel of the loop shows up as a damaged band instead of
        as a wrong value somewhere else.  Affects buffers allocated from now on."""
        self._guard = bool(on)
        self._guarded = []

    def _alloc(self, shape, dtype=torch.float32, fill=None):
        n = 1
        for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
            n *= int(d)
        if not getattr(self, "_guard", False):
            t = torch.empty(shape, dtype=dtype, device=self.device)
            if fill is not None:
                t.fill_(fill)
            return t
        esz = torch.empty((), dtype=dtype).element_size()
        nbytes = (n * esz + 255) // 256 * 256
        raw = torch.full((self.GUARD + nbytes + self.GUARD,), 0xA5, dtype=torch.uint8, device=self.device)
        t = raw[self.GUARD:self.GUARD + n * esz].view(dtype).view(shape)
        if fill is not None:
            t.fill_(fill)
        self._guarded.append((raw, n * esz))
        return t

    def check_guards(self):
        torch.cuda.synchronize(self.device)
        for raw, nbytes in getattr(self, "_guarded", []):
            lo, hi = raw[:self.GUARD], raw[self.GUARD + (nbytes + 255) // 256 * 256:]
            if not (bool((lo == 0xA5).all()) and bool((hi == 0xA5).all())):
                raise AssertionError("guard band damaged around a buffer of %d bytes" % nbytes)
        return len(getattr(self, "_guarded", []))

    def _workspace(self, P, N, M):
        need = lib.raw().dr_denoise_loop_workspace_bytes(ctypes.byref(self.cfg), P, N, M)
        if self._ws is None or self._ws.numel() < need:
            self._ws = self._alloc(need, torch.uint8)
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
        if b.get("trace") or "feats_nopos" in b or b.get("force"):
            tr = lib.LoopTrace()
        if b.get("trace"):
            tr.x0, tr.R_forwd, tr.t_forwd, tr.cond = (b["tr_x0"].data_ptr(), b["tr_R"].data_ptr(), b["tr_t"].data_ptr(),
                                                      b["tr_cond"].data_ptr())
        if b.get("trace") == "full":
            tr.x_next, tr.topk_idx, tr.wconf = b["tr_xn"].data_ptr(), b["tr_topk"].data_ptr(), b["tr_wconf"].data_ptr()
        f = b.get("force")
        if f:
            tr.force_x = f["x"].data_ptr() if f.get("x") is not None else None
            if f.get("R") is not None:
                tr.force_R, tr.force_t = f["R"].data_ptr(), f["t"].data_ptr()
        if "feats_nopos" in b:
            tr.feats_nopos, tr.feats_pos = b["feats_nopos"].data_ptr(), b["feats_pos"].data_ptr()
        lib.check(lib.raw().dr_denoise_loop(
            ctypes.byref(self.cfg), ctypes.byref(self.w), b["P"], b["N"], b["M"], lib.ptr(b["src_feats"]), lib.ptr(b["tgt_feats"]),
            lib.ptr(b["s_pcd"]), lib.ptr(b["t_pcd"]), lib.ptr(b["src_mask"]), lib.ptr(b["tgt_mask"]), lib.ptr(b["x_T"]),
            lib.ptr(b["noise"]), lib.ptr(b["conf"]), lib.ptr(b["x_final"]), lib.ptr(b["matches"]), lib.ptr(b["match_count"]),
            lib.ptr(b["R_final"]), lib.ptr(b["t_final"]), ctypes.byref(tr) if tr is not None else None, lib.ptr(b["ws"]),
            b["ws_bytes"], lib.stream_of(b["conf"])))

    def make_buffers(self, P, N, M, masked=False, trace=False, private_ws=False, side_outputs=False):
        dev, C, S = self.device, self.C, self.steps
        ws, need = self._workspace(P, N, M)
        if private_ws:          # concurrent batches must not share scratch memory
            ws = self._alloc(need, torch.uint8)
        A = self._alloc
        b = dict(P=P, N=N, M=M, ws=ws, ws_bytes=need, trace=trace,
                 src_feats=A((P, N, C), fill=0), tgt_feats=A((P, M, C), fill=0),
                 s_pcd=A((P, N, 3), fill=0), t_pcd=A((P, M, 3), fill=0),
                 src_mask=A((P, N), torch.uint8, fill=1) if masked else None,
                 tgt_mask=A((P, M), torch.uint8, fill=1) if masked else None,
                 x_T=A((P, N, M), fill=0),
                 noise=A((S, P, N, M), fill=0) if self.variant == "4dmatch" else None,
                 conf=A((P, N, M), torch.float64),
                 x_final=A((P, N, M), torch.float64),
                 matches=A((P, N + M, 3), torch.int64, fill=0) if self.variant == "3dmatch" else None,
                 match_count=A((P,), torch.int32, fill=0) if self.variant == "3dmatch" else None,
                 R_final=A((P, 3, 3)), t_final=A((P, 3, 1)))
        if trace:
            b.update(tr_x0=A((S, P, N, M)), tr_R=A((S, P, 3, 3)), tr_t=A((S, P, 3, 1)), tr_cond=A((S, P), torch.float64))
        if trace == "full":
            K = int(float(torch.tensor(float(max(N, M)), dtype=torch.float32) * self.cfg.sample_rate))
            b.update(tr_xn=A((S, P, N, M), torch.float64), tr_topk=A((S, P, K), torch.int32), tr_wconf=A((S, P, N, M)))
        if side_outputs:
            b.update(feats_nopos=A((P * (N + M), C)), feats_pos=A((P * (N + M), C)))
        return b

    def run(self, src_feats, tgt_feats, s_pcd, t_pcd, x_T, src_mask=None, tgt_mask=None, noise=None, trace=False,
            graph=False, _slot=0, ragged=False, side_outputs=False, borrow=False, force=None):
        """Run the loop for P pairs.  Returns a dict of device tensors (conf float64, x_final, matches list (3D),
        R_final, t_final, and the per-step trace when asked).  ragged=True (with masks): the masks are the true extents
        of pairs padded to (N, M) and every pair gets the result of its own unpadded run (DR_LOOP_RAGGED).
        side_outputs=True adds what Matching.forward leaves in `data` at the last step (src/tgt_feats, *_nopos).

        Shapes are cached (static buffers + captured graph), at most `cache_entries` of them, least recently used first out.
        With graph=True a shape runs eagerly the FIRST time it is seen and is captured when it comes again, so a stream of
        ever-changing shapes (the reference tester: B = 1, N and M differ from pair to pair) never pays warm-up + capture for
        a graph it will not replay.  The returned tensors are COPIES unless borrow=True (then they are the cached static
        buffers, overwritten by the next run() of the same shape)."""
        P, N, C = src_feats.shape
        M = tgt_feats.shape[1]
        masked = src_mask is not None
        self.cfg.flags = (self.cfg.flags & ~2) | (2 if (ragged and masked) else 0)
        key = (P, N, M, masked, trace, _slot, bool(ragged and masked), bool(side_outputs))
        ent = self._graphs.get(key)
        if ent is None:
            ent = dict(b=self.make_buffers(P, N, M, masked=masked, trace=trace, private_ws=_slot > 0, side_outputs=side_outputs), g=None, uses=0)
            self._graphs[key] = ent
            while len(self._graphs) > self._cache_entries:
                _, old = self._graphs.popitem(last=False)      # frees the graph and the buffers of the least recently used shape
                old.clear()
        else:
            self._graphs.move_to_end(key)
        b = ent["b"]
        self._fill(b, src_feats, tgt_feats, s_pcd, t_pcd, x_T, src_mask, tgt_mask, noise)
        # teacher forcing (parity tests; dr_loop_trace.force_*): x [steps,P,N,M] float64 = the state entering every step, optional
        # R [steps,P,3,3] / t [steps,P,3,1] float32 = the pose every step warps with.  trace="full" adds x_next / topk_idx / wconf.
        b["force"] = None
        if force is not None:
            if graph:
                raise RuntimeError("teacher forcing is a test facility of the eager path")
            S = self.steps
            fx = force.get("x")
            b["force"] = dict(x=None if fx is None else fx.to(self.device, torch.float64).reshape(S, P, N, M).contiguous(),
                              R=None if force.get("R") is None else force["R"].to(self.device, torch.float32).reshape(S, P, 9).contiguous(),
                              t=None if force.get("R") is None else force["t"].to(self.device, torch.float32).reshape(S, P, 3).contiguous())
        if graph and ent["g"] is None and ent["uses"] >= 1:
            # second visit of this shape: capture (the earlier eager run was the warm-up)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._enqueue(b)
            ent["g"] = g
        if graph and ent["g"] is not None:
            ent["g"].replay()
        else:
            self._enqueue(b)
        ent["uses"] += 1
        out = self._collect(b)
        return out if borrow else {k: v.clone() for k, v in out.items()}

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
        out = dict(conf_matrix_pred=b["conf"], x_final=b["x_final"], R_final=b["R_final"], t_final=b["t_final"], _status=_CallStatus(b["ws"]))
        if b["matches"] is not None:
            out["matches_padded"], out["match_count"] = b["matches"], b["match_count"]
        if b["trace"]:
            out.update(x0=b["tr_x0"], R_forwd=b["tr_R"], t_forwd=b["tr_t"], cond=b["tr_cond"])
        if b["trace"] == "full":
            out.update(x_next=b["tr_xn"], topk_idx=b["tr_topk"], wconf=b["tr_wconf"])
        if "feats_nopos" in b:
            P, N, M, C = b["P"], b["N"], b["M"], self.C
            for name, t_ in (("nopos", b["feats_nopos"]), ("pos", b["feats_pos"])):
                sfx = "_nopos" if name == "nopos" else ""
                out["src_feats" + sfx] = t_[:P * N].view(P, N, C)
                out["tgt_feats" + sfx] = t_[P * N:].view(P, M, C)
        return out

    # ------------------------------------------------------------------------------------------
    def run_streams(self, groups, n_streams=2):
        """Run several independent batches of pairs concurrently, one captured graph per batch, replayed on
        `n_streams` HIP streams so that the tails / small launches of one batch overlap the big launches of another.
        groups: list of dicts with the keyword arguments of run() (src_feats, tgt_feats, s_pcd, t_pcd, x_T, ...).
        Returns the list of result dicts (borrowed: the static buffers of each group, valid until the group is run again).
        The first pass of a group runs eagerly, the second captures its graph, later ones replay it."""
        cur = torch.cuda.current_stream(self.device)
        self._streams = _side_streams(self.device, n_streams)
        # (experiment knob, read only under DR_DIAGNOSTICS=1 like the library's own: a start skew between the streams, in microseconds)
        self._skew_cycles = int(float(os.environ.get("DR_STREAM_SKEW_US", "0")) * 2000) if os.environ.get("DR_DIAGNOSTICS") == "1" else 0
        self._cache_entries = max(self._cache_entries, len(groups) + 2)        # every group keeps its own slot
        outs = []
        for gi, kw in enumerate(groups):
            st = self._streams[gi % n_streams]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                if self._skew_cycles and gi % n_streams:
                    torch.cuda._sleep(self._skew_cycles * (gi % n_streams))
                outs.append(self.run(graph=True, _slot=gi, borrow=True, **kw))
        for st in self._streams[:n_streams]:
            cur.wait_stream(st)
        return outs

    def run_ragged(self, pairs, noise=None, graph=False):
        """Pairs of DIFFERENT sizes in one call: `pairs` is a list of dicts with src_feats [N_i,C], tgt_feats
        [M_i,C], s_pcd [N_i,3], t_pcd [M_i,3], x_T [N_i,M_i] (device tensors).  They are padded to the largest extents,
        run as one batch with DR_LOOP_RAGGED and cropped again: each entry of the returned list equals the pair's own
        B = 1 run (the reference's pad-and-mask batching does not, quirk Q19)."""
        P = len(pairs)
        Ns = [int(q["src_feats"].shape[0]) for q in pairs]
        Ms = [int(q["tgt_feats"].shape[0]) for q in pairs]
        N, M, dev, C = max(Ns), max(Ms), self.device, self.C
        fs, ft = torch.zeros(P, N, C, device=dev), torch.zeros(P, M, C, device=dev)
        ps, pt = torch.zeros(P, N, 3, device=dev), torch.zeros(P, M, 3, device=dev)
        xT = torch.zeros(P, N, M, device=dev)
        sm, tm = torch.zeros(P, N, dtype=torch.bool, device=dev), torch.zeros(P, M, dtype=torch.bool, device=dev)
        for i, q in enumerate(pairs):
            n, m = Ns[i], Ms[i]
            fs[i, :n] = q["src_feats"]; ft[i, :m] = q["tgt_feats"]; ps[i, :n] = q["s_pcd"]; pt[i, :m] = q["t_pcd"]
            xT[i, :n, :m] = q["x_T"]; sm[i, :n] = True; tm[i, :m] = True
        nz = None
        if noise is not None:
            nz = torch.zeros(self.steps, P, N, M, device=dev)
            for i, z in enumerate(noise):
                nz[:, i, :Ns[i], :Ms[i]] = z
        out = self.run(fs, ft, ps, pt, xT, sm, tm, noise=nz, graph=graph, ragged=True, borrow=True)
        res = []
        cnt = out["match_count"].cpu().tolist() if "match_count" in out else None
        out["_status"].check()                  # this call's own status word ...
        lib.device_status(self.device)          # ... and the process-wide flag (the count read above synchronised already)
        for i in range(P):
            r = dict(conf_matrix_pred=out["conf_matrix_pred"][i, :Ns[i], :Ms[i]].clone(), R_final=out["R_final"][i].clone(),
                     t_final=out["t_final"][i].clone())
            if cnt is not None:
                r["match_pred"] = out["matches_padded"][i, :cnt[i]].clone()
            res.append(r)
        return res

    @staticmethod
    def match_list(out):
        """[K_p,3] int64 tensors (one host sync; raises if a kernel of the run reported a device-side failure)."""
        cnt = out["match_count"].cpu().tolist()
        if "_status" in out:
            out["_status"].check()              # the status word of the workspace THIS call ran on
        lib.device_status(out["match_count"].device)
        return [out["matches_padded"][p, :cnt[p]] for p in range(len(cnt))]


class DenoiseEngine2D3D:
    """2D-3D variant: reverse sampling of MATR2D3D.forward (EXP/model.py:637-694) with the
    CrossModalFusionModule denoiser, through dr_denoise_loop_2d3d.  `state` uses the reference's names
    (`denoising_transformer.*` = CrossModalFusionModule, `denoising_coarse_matching.*` = Matching)."""
    _LAYER = (("q_w", "attention.attention.q_token_layer.weight"), ("q_b", "attention.attention.q_token_layer.bias"),
              ("k_w", "attention.attention.k_token_layer.weight"), ("k_b", "attention.attention.k_token_layer.bias"),
              ("v_w", "attention.attention.v_token_layer.weight"), ("v_b", "attention.attention.v_token_layer.bias"),
              ("lin_w", "attention.linear.weight"), ("lin_b", "attention.linear.bias"),
              ("norm1_w", "attention.norm.weight"), ("norm1_b", "attention.norm.bias"),
              ("expand_w", "output.expand.weight"), ("expand_b", "output.expand.bias"),
              ("squeeze_w", "output.squeeze.weight"), ("squeeze_b", "output.squeeze.bias"),
              ("norm2_w", "output.norm.weight"), ("norm2_b", "output.norm.bias"))
    _TOP = (("img_emb_b", "img_emb_proj.bias"), ("pcd_emb_b", "pcd_emb_proj.bias"), ("img_in_w", "img_in_proj.weight"),
            ("img_in_b", "img_in_proj.bias"), ("dino_w", "img_in_proj_dino.weight"), ("dino_b", "img_in_proj_dino.bias"),
            ("all_w", "img_in_proj_all.weight"), ("all_b", "img_in_proj_all.bias"), ("pcd_in_w", "pcd_in_proj.weight"),
            ("pcd_in_b", "pcd_in_proj.bias"), ("out_w", "out_proj.weight"), ("out_b", "out_proj.bias"))

    def __init__(self, state, *, C=256, H=4, n_layers=6, img_dim=512, dino_dim=1024, pcd_dim=512, steps=10, sk_iters=3,
                 sample_rate=1.0, max_condition_num=200.0, device="cuda:0", strict_f64=False,
                 prefix_t="denoising_transformer.", prefix_m="denoising_coarse_matching.", planes=None, prepack=True, attn_f16=False):
        """planes: None = the size rule picks the GEMM / attention path (plane images from 4096 token rows on: two cfg5 pairs per call);
        True / False = DR_LOOP_PLANES_FORCE / DR_LOOP_PLANES_OFF.  The weights are snapshotted (copy=True) and, for the plane path,
        packed once (dr_loop2d3d_prepack): build a new engine when the parameters change."""
        lib.ensure_init()
        self.device = torch.device(device)
        self.C, self.steps = C, steps
        f = lambda k: state[k].detach().to(device=self.device, dtype=torch.float32, copy=True).contiguous()
        self._keep = []
        self._layers = (lib.FusionLayerWeights * n_layers)()
        for l in range(n_layers):
            for field, name in self._LAYER:
                tns = f(prefix_t + "transformer.%d.%s" % (l, name))
                self._keep.append(tns)
                setattr(self._layers[l], field, tns.data_ptr())
        w = lib.FusionWeights()
        w.layers = ctypes.cast(self._layers, ctypes.POINTER(lib.FusionLayerWeights))
        for field, name in self._TOP:
            tns = f(prefix_t + name)
            self._keep.append(tns)
            setattr(w, field, tns.data_ptr())
        # K = 42 / 63 are not multiples of 4: zero-pad the embedding projections to 44 / 64 input columns
        ie = torch.nn.functional.pad(f(prefix_t + "img_emb_proj.weight"), (0, 2)).contiguous()
        pe = torch.nn.functional.pad(f(prefix_t + "pcd_emb_proj.weight"), (0, 1)).contiguous()
        sp = f(prefix_m + "src_proj.weight")
        bs = f(prefix_m + "bin_score").reshape(1)
        self._keep += [ie, pe, sp, bs]
        w.img_emb_w, w.pcd_emb_w, w.src_proj, w.bin_score = ie.data_ptr(), pe.data_ptr(), sp.data_ptr(), bs.data_ptr()
        self.w = w
        self._ac = np.ascontiguousarray(cosine_alphas_cumprod().numpy())
        self._cfgs = {}
        self._base = dict(C=C, H=H, n_layers=n_layers, img_dim=img_dim, dino_dim=dino_dim, pcd_dim=pcd_dim, sk_iters=sk_iters,
                          sample_rate=sample_rate, max_condition_num=max_condition_num,
                          flags=(1 if strict_f64 else 0) | (4 if planes is True else 0) | (8 if planes is False else 0) | (16 if attn_f16 else 0))
        self._ws = None
        self._packed = None
        cfg0 = self._cfg(steps)
        nb = lib.raw().dr_loop2d3d_prepack_bytes(ctypes.byref(cfg0))
        if nb and prepack and planes is not False:
            self._packed = torch.empty(nb, dtype=torch.uint8, device=self.device)
            lib.check(lib.raw().dr_loop2d3d_prepack(ctypes.byref(cfg0), ctypes.byref(self.w), self._packed.data_ptr(), nb,
                                                    ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            self.w.prepacked = self._packed.data_ptr()
            torch.cuda.current_stream(self.device).synchronize()       # (later runs may come on other streams: see DenoiseEngine)

    def _cfg(self, steps):
        if steps not in self._cfgs:
            cfg = lib.Loop2D3DConfig()
            for k, v in self._base.items():
                setattr(cfg, k, v)
            cfg.steps = steps
            times = np.ascontiguousarray(np.asarray(sampling_times(max(steps, 1)), dtype=np.int32))
            cfg.h_alphas_cumprod = self._ac.ctypes.data
            cfg.h_times = times.ctypes.data
            self._cfgs[steps] = (cfg, times)
        return self._cfgs[steps][0]

    def _call(self, steps, img_feats, img_dino, img_pixels, pcd_feats, s_pcd, t_pcd_da, masks, x_T, trace, force=None):
        P, M, _ = img_feats.shape
        N = pcd_feats.shape[1]
        cfg = self._cfg(steps)
        need = lib.raw().dr_denoise_loop_2d3d_workspace_bytes(ctypes.byref(cfg), P, N, M)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        dev = self.device
        c = lambda t_: None if t_ is None else t_.contiguous()
        conf = torch.empty(P, N, M, dtype=torch.float64, device=dev)
        xf = torch.empty(P, N, M, dtype=torch.float64, device=dev)
        matches = torch.zeros(P, N + M, 3, dtype=torch.int64, device=dev)
        cnt = torch.zeros(P, dtype=torch.int32, device=dev)
        img_out = torch.empty(P, M, self.C, device=dev)
        pcd_out = torch.empty(P, N, self.C, device=dev)
        sm, tm, tmd = (lib.mask_u8(m) for m in masks) if masks is not None else (None, None, None)
        tr, trb = None, {}
        if trace and steps > 0:
            trb = dict(x0=torch.empty(steps, P, N, M, device=dev), R_forwd=torch.empty(steps, P, 3, 3, device=dev),
                       t_forwd=torch.empty(steps, P, 3, 1, device=dev), cond=torch.empty(steps, P, dtype=torch.float64, device=dev))
            tr = lib.LoopTrace()
            tr.x0, tr.R_forwd, tr.t_forwd, tr.cond = (trb[k].data_ptr() for k in ("x0", "R_forwd", "t_forwd", "cond"))
            if trace == "full":
                K = int(float(torch.tensor(float(max(N, M)), dtype=torch.float32) * self._base["sample_rate"]))
                trb.update(x_next=torch.empty(steps, P, N, M, dtype=torch.float64, device=dev),
                           topk_idx=torch.empty(steps, P, K, dtype=torch.int32, device=dev), wconf=torch.empty(steps, P, N, M, device=dev))
                tr.x_next, tr.topk_idx, tr.wconf = (trb[k].data_ptr() for k in ("x_next", "topk_idx", "wconf"))
        keep = []
        if force is not None and steps > 0:            # teacher forcing (parity tests; dr_loop_trace.force_*)
            tr = tr or lib.LoopTrace()
            if force.get("x") is not None:
                keep.append(force["x"].to(dev, torch.float64).reshape(steps, P, N, M).contiguous())
                tr.force_x = keep[-1].data_ptr()
            if force.get("R") is not None:
                keep.append(force["R"].to(dev, torch.float32).reshape(steps, P, 9).contiguous())
                keep.append(force["t"].to(dev, torch.float32).reshape(steps, P, 3).contiguous())
                tr.force_R, tr.force_t = keep[-2].data_ptr(), keep[-1].data_ptr()
        lib.check(lib.raw().dr_denoise_loop_2d3d(
            ctypes.byref(cfg), ctypes.byref(self.w), P, N, M, lib.ptr(c(img_feats)), lib.ptr(c(img_dino)), lib.ptr(c(img_pixels)),
            lib.ptr(c(pcd_feats)), lib.ptr(c(s_pcd)), lib.ptr(c(t_pcd_da)), lib.ptr(sm), lib.ptr(tm), lib.ptr(tmd), lib.ptr(c(x_T)),
            lib.ptr(conf), lib.ptr(xf), lib.ptr(matches), lib.ptr(cnt), lib.ptr(img_out), lib.ptr(pcd_out),
            ctypes.byref(tr) if tr is not None else None, lib.ptr(self._ws), need, lib.stream_of(img_feats)))
        out = dict(conf_matrix_pred=conf, x_final=xf, matches_padded=matches, match_count=cnt, img_feats=img_out, pcd_feats=pcd_out,
                   _status=_CallStatus(self._ws))
        out.update(trb)
        if keep:
            out["_forced_inputs"] = keep          # (alive until the caller drops the result: the call is asynchronous)
        return out

    def fuse_and_match(self, img_feats, img_dino, img_pixels, pcd_feats, pcd_points, masks=None):
        """CrossModalFusionModule.forward + Matching.forward once: -> img feats, pcd feats, conf (x_start) float32."""
        o = self._call(0, img_feats, img_dino, img_pixels, pcd_feats, pcd_points, None, masks, None, False)
        return o["img_feats"], o["pcd_feats"], o["conf_matrix_pred"].float()

    def run(self, img_feats, img_dino, img_pixels, pcd_feats, s_pcd, t_pcd_da, x_T, masks=None, trace=False, force=None):
        """trace=True: per-step x0 / R_forwd / t_forwd / cond; trace="full" adds x_next / topk_idx / wconf; force = dict(x=[steps,P,N,M],
        R=[steps,P,3,3], t=[steps,P,3,1]): teacher forcing for the parity tests (dr_loop_trace.force_*)."""
        return self._call(self.steps, img_feats, img_dino, img_pixels, pcd_feats, s_pcd, t_pcd_da, masks, x_T, trace, force)

    # ------------------------------------------------------------------------------------------
    _ARGS = ("img_feats", "img_dino", "img_pixels", "pcd_feats", "s_pcd", "t_pcd_da", "x_T")

    def _slot(self, slot, kw):
        """static buffers (inputs, outputs, a PRIVATE workspace) + captured graph of one concurrent batch"""
        shapes = tuple(tuple(kw[k].shape) for k in self._ARGS)
        masked = kw.get("masks") is not None
        if not hasattr(self, "_slots"):
            self._slots = {}
        ent = self._slots.get(slot)
        if ent is None or ent["shapes"] != shapes or ent["masked"] != masked:
            P, M, _ = kw["img_feats"].shape
            N = kw["pcd_feats"].shape[1]
            dev = self.device
            cfg = self._cfg(self.steps)
            need = lib.raw().dr_denoise_loop_2d3d_workspace_bytes(ctypes.byref(cfg), P, N, M)
            ent = dict(shapes=shapes, masked=masked, P=P, N=N, M=M, g=None, uses=0, need=need,
                       ws=torch.empty(need, dtype=torch.uint8, device=dev),
                       inp={k: torch.empty(kw[k].shape, dtype=torch.float32, device=dev) for k in self._ARGS},
                       masks=tuple(torch.empty(m.shape, dtype=torch.uint8, device=dev) for m in kw["masks"]) if masked else None,
                       conf=torch.empty(P, N, M, dtype=torch.float64, device=dev), xf=torch.empty(P, N, M, dtype=torch.float64, device=dev),
                       matches=torch.zeros(P, N + M, 3, dtype=torch.int64, device=dev), cnt=torch.zeros(P, dtype=torch.int32, device=dev),
                       img_out=torch.empty(P, M, self.C, device=dev), pcd_out=torch.empty(P, N, self.C, device=dev))
            self._slots[slot] = ent
        return ent

    def _enqueue_slot(self, e):
        cfg = self._cfg(self.steps)
        i, m = e["inp"], e["masks"] or (None, None, None)
        lib.check(lib.raw().dr_denoise_loop_2d3d(
            ctypes.byref(cfg), ctypes.byref(self.w), e["P"], e["N"], e["M"], lib.ptr(i["img_feats"]), lib.ptr(i["img_dino"]), lib.ptr(i["img_pixels"]),
            lib.ptr(i["pcd_feats"]), lib.ptr(i["s_pcd"]), lib.ptr(i["t_pcd_da"]), lib.ptr(m[0]), lib.ptr(m[1]), lib.ptr(m[2]), lib.ptr(i["x_T"]),
            lib.ptr(e["conf"]), lib.ptr(e["xf"]), lib.ptr(e["matches"]), lib.ptr(e["cnt"]), lib.ptr(e["img_out"]), lib.ptr(e["pcd_out"]),
            None, lib.ptr(e["ws"]), e["need"], lib.stream_of(e["conf"])))

    def run_static(self, slot=0, graph=True, **kw):
        """run() on static buffers with the whole loop captured in a HIP graph on the SECOND pass of a slot (the first runs eagerly:
        warm-up) and replayed afterwards.  Returns the slot's static output tensors (overwritten by its next pass)."""
        e = self._slot(slot, kw)
        for k in self._ARGS:
            e["inp"][k].copy_(kw[k])
        if e["masked"]:
            for dst, src in zip(e["masks"], kw["masks"]):
                dst.copy_(lib.mask_u8(src))
        if graph and e["g"] is None and e["uses"] >= 1:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._enqueue_slot(e)
            e["g"] = g
        if graph and e["g"] is not None:
            e["g"].replay()
        else:
            self._enqueue_slot(e)
        e["uses"] += 1
        return dict(conf_matrix_pred=e["conf"], x_final=e["xf"], matches_padded=e["matches"], match_count=e["cnt"], _status=_CallStatus(e["ws"]))

    def run_streams(self, groups, n_streams=2):
        """Several independent batches of pairs concurrently (DenoiseEngine.run_streams for the 2D-3D loop): one captured graph per batch,
        replayed on `n_streams` HIP streams, each batch with its own workspace.  groups: list of dicts with run()'s tensor arguments."""
        cur = torch.cuda.current_stream(self.device)
        self._streams = _side_streams(self.device, n_streams)
        outs = []
        for gi, kw in enumerate(groups):
            st = self._streams[gi % n_streams]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(self.run_static(slot=gi, graph=True, **kw))
        for st in self._streams[:n_streams]:
            cur.wait_stream(st)
        return outs

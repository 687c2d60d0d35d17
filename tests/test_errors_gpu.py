"""The C ABI refuses bad calls with its status codes (include/diffreg_hip.h:27-31) before anything is launched: NULL pointers, sizes the
build does not take, missing or short workspaces.  Through the raw ctypes table, as a binding in another language would see them."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diff-reg_amd"))
from tests.helpers import pair, weights
from tests.test_loop_gpu import engine

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OK, EINVAL, ELAUNCH, ENOSUP, EWORKSPACE, ETIMEOUT = 0, -1, -2, -3, -4, -5


def test_status_strings():
    from diffreg_hip import lib
    r = lib.raw()
    assert r.dr_strerror(OK).decode() == "ok"
    for code in (EINVAL, ELAUNCH, ENOSUP, EWORKSPACE, ETIMEOUT):
        assert r.dr_strerror(code).decode() not in ("ok", "unknown error")
    assert r.dr_strerror(-99).decode() == "unknown error"


def test_sinkhorn_refuses_bad_calls():
    from diffreg_hip import lib
    r = lib.raw()
    s = torch.randn(2, 64, 64, device=DEV)
    out = torch.empty_like(s)
    b = torch.ones(1, device=DEV)
    st = lib.stream_of(s)
    assert r.dr_sinkhorn_f32(2, 64, 64, None, None, None, lib.ptr(b), 3, 0, lib.ptr(out), None, 0, st) == EINVAL
    assert r.dr_sinkhorn_f32(2, 64, 64, lib.ptr(s), None, None, lib.ptr(b), 3, 0, None, None, 0, st) == EINVAL
    assert r.dr_sinkhorn_f32(2, 0, 64, lib.ptr(s), None, None, lib.ptr(b), 3, 0, lib.ptr(out), None, 0, st) == EINVAL
    # a tile beyond the register-resident size needs the workspace dr_sinkhorn_workspace_bytes reports
    big = torch.randn(1, 600, 600, device=DEV)
    need = r.dr_sinkhorn_workspace_bytes(1, 600, 600, 4, 0)
    assert need > 0
    assert r.dr_sinkhorn_f32(1, 600, 600, lib.ptr(big), None, None, lib.ptr(b), 3, 0, lib.ptr(torch.empty_like(big)), None, 0, st) == EWORKSPACE
    ws = torch.empty(need - 256, dtype=torch.uint8, device=DEV)
    assert r.dr_sinkhorn_f32(1, 600, 600, lib.ptr(big), None, None, lib.ptr(b), 3, 0, lib.ptr(torch.empty_like(big)), lib.ptr(ws), need - 256, st) == EWORKSPACE
    torch.cuda.synchronize()


def test_linear_and_fine_matching_refuse_unsupported_shapes():
    from diffreg_hip import lib
    r = lib.raw()
    x = torch.randn(16, 6, device=DEV); W = torch.randn(8, 6, device=DEV); out = torch.empty(16, 8, device=DEV)
    st = lib.stream_of(x)
    assert r.dr_linear_f32(16, 8, 6, lib.ptr(x), lib.ptr(W), lib.ptr(out), 0, None, None, 0, 1.0, st) == ENOSUP      # K % 4 != 0
    assert r.dr_linear_f32(16, 8, 0, lib.ptr(x), lib.ptr(W), lib.ptr(out), 0, None, None, 0, 1.0, st) == EINVAL
    assert r.dr_linear_f32(16, 8, 8, lib.ptr(x), lib.ptr(W), lib.ptr(out), 2, None, None, 8, 1.0, st) == EINVAL      # rotary (epilogue flag 2) without tables
    f = torch.randn(10, 8, device=DEV); idx = torch.zeros(1, 200, dtype=torch.int64, device=DEV)
    o = torch.empty(1, 4, 200, device=DEV)
    assert r.dr_patch_similarity_f32(1, 4, 200, 8, lib.ptr(f), lib.ptr(idx), lib.ptr(f), lib.ptr(idx), 10, lib.ptr(o), st) == ENOSUP   # > 128 patch points
    a = torch.arange(100, dtype=torch.int64, device=DEV)
    keys = torch.empty(100, dtype=torch.int64, device=DEV); cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(64, dtype=torch.uint8, device=DEV)
    assert r.dr_unique_pairs_i64(100, lib.ptr(a), lib.ptr(a), 1000, lib.ptr(keys), lib.ptr(cnt), lib.ptr(ws), 64, st) == EWORKSPACE
    assert r.dr_unique_pairs_i64(100, None, lib.ptr(a), 1000, lib.ptr(keys), lib.ptr(cnt), lib.ptr(ws), 64, st) == EINVAL
    torch.cuda.synchronize()


def test_loop_refuses_bad_calls():
    from diffreg_hip import lib
    variant, N, M = "3dmatch", 32, 32
    eng = engine(variant, 2, 200)
    b = eng.make_buffers(1, N, M)
    _, p = pair(variant, N, M, 3)
    for k, src in (("src_feats", "f_s"), ("tgt_feats", "f_t"), ("s_pcd", "p_s"), ("t_pcd", "p_t"), ("x_T", "x_T")):
        b[k].copy_(p[src].to(DEV))
    r = lib.raw()

    def call(**over):
        a = dict(b, **over)
        return r.dr_denoise_loop(
            ctypes.byref(eng.cfg), ctypes.byref(eng.w), a["P"], a["N"], a["M"], lib.ptr(a["src_feats"]), lib.ptr(a["tgt_feats"]),
            lib.ptr(a["s_pcd"]), lib.ptr(a["t_pcd"]), lib.ptr(a["src_mask"]), lib.ptr(a["tgt_mask"]), lib.ptr(a["x_T"]),
            lib.ptr(a["noise"]), lib.ptr(a["conf"]), lib.ptr(a["x_final"]), lib.ptr(a["matches"]), lib.ptr(a["match_count"]),
            lib.ptr(a["R_final"]), lib.ptr(a["t_final"]), None, lib.ptr(a["ws"]), a["ws_bytes"], lib.stream_of(b["conf"]))

    assert call() == OK
    assert call(src_feats=None) == EINVAL
    assert call(conf=None) == EINVAL
    assert call(ws=None) == EWORKSPACE
    assert call(ws_bytes=b["ws_bytes"] // 2) == EWORKSPACE
    assert call(src_mask=torch.ones(1, N, dtype=torch.uint8, device=DEV)) == EINVAL          # one mask without the other
    assert call(matches=None) == EINVAL                                                        # matches without match_count
    assert call(R_final=None) == EINVAL                                                        # R without t
    assert call() == OK                                                                        # and the engine is still usable
    torch.cuda.synchronize()


def test_mutual_topk_refuses_k_beyond_the_matrix():
    """torch.topk in the reference raises when k exceeds a dimension (mutual_topk_select.py:27-28): same message from the wrapper,
    DR_EINVAL from the C entry"""
    from diffreg_hip import lib
    sc = torch.rand(2, 4, 3, device=DEV)
    with pytest.raises(RuntimeError, match="selected index k out of range"):
        lib.batch_mutual_topk_select(sc, 10)
    r = lib.raw()
    idx = torch.empty(64, 3, dtype=torch.int64, device=DEV); s = torch.empty(64, device=DEV); tot = torch.zeros(1, dtype=torch.int32, device=DEV)
    wsb = r.dr_mutual_topk_workspace_bytes(2, 4, 3)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    assert r.dr_mutual_topk_select_f32(2, 4, 3, lib.ptr(sc), 4, 1, 0, 0.0, 1, None, None, lib.ptr(idx), lib.ptr(s), 64, lib.ptr(tot), lib.ptr(ws), wsb,
                                       lib.stream_of(sc)) == EINVAL
    assert r.dr_mutual_topk_select_f32(2, 4, 3, lib.ptr(sc), 3, 1, 0, 0.0, 1, None, None, lib.ptr(idx), lib.ptr(s), 64, lib.ptr(tot), lib.ptr(ws), wsb,
                                       lib.stream_of(sc)) == OK
    torch.cuda.synchronize()


def test_coresident_sinkhorn_fails_loudly():
    """The single-launch Sinkhorn of tiles beyond 256 x 256 waits for column sums of other workgroups with bounded spins.  With the spin bound
    at ONE poll (debug header) the first failed poll already gives up: the workgroup must then poison its outputs (NaN) and raise the sticky
    device flag that dr_device_status turns into DR_ETIMEOUT -- never a silent wrong matrix with DR_OK.  Afterwards the flag is clear and the
    same call gives the oracle's result again."""
    from diffreg_hip import lib
    from oracle import diffreg_oracle as orc
    r = lib.raw()
    lib.ensure_init()
    x = torch.randn(1, 1024, 2048, device=DEV) * 2
    a = torch.tensor(1.0, device=DEV)
    good = lib.sinkhorn(x, a, 3)
    assert r.dr_device_status(lib.stream_of(x), 1) == OK and torch.isfinite(good).all()
    r.dr_debug_sinkhorn_spin_limit(1)
    try:
        timed_out = False
        for _ in range(20):                          # 128 workgroups polling once each: some poll fails in practically every launch
            bad = lib.sinkhorn(x, a, 3)
            rc = r.dr_device_status(lib.stream_of(x), 0)        # (not cleared: the flag is sticky)
            if rc != OK:
                assert rc == ETIMEOUT and r.dr_strerror(rc).decode() not in ("ok", "unknown error")
                assert torch.isnan(bad).any()        # the failure is in the data too
                with pytest.raises(RuntimeError, match="gave up waiting"):
                    lib.device_status(x.device)      # the host mirror raises, and clears
                assert r.dr_device_status(lib.stream_of(x), 0) == OK
                timed_out = True
                break
            assert torch.equal(bad, good)            # no timeout -> the ordinary, bit-reproducible result
        assert timed_out
    finally:
        r.dr_debug_sinkhorn_spin_limit(0)
        r.dr_device_status(lib.stream_of(x), 1)
    again = lib.sinkhorn(x, a, 3)
    assert r.dr_device_status(lib.stream_of(x), 1) == OK
    assert torch.equal(again, good)
    sm, tm = torch.ones(1, 1024, dtype=torch.bool), torch.ones(1, 2048, dtype=torch.bool)
    ref = orc.sinkhorn_conf(x.cpu(), torch.tensor(1.0), 3, sm, tm)
    assert ((again.cpu() - ref).abs() / ref.clamp_min(1e-30))[ref > 1e-12 * ref.max()].max().item() < 2e-5


def test_procrustes_and_readout_take_caller_workspaces():
    """dr_procrustes_f32 / dr_top1_union_* allocate nothing: tiles beyond 256 x 256 (read-out: from 32 rows on) need the caller's
    workspace of dr_*_workspace_bytes, DR_EWORKSPACE without it; small tiles need none."""
    from diffreg_hip import lib
    r = lib.raw()
    lib.ensure_init()
    P, N, M = 2, 300, 400
    conf = torch.rand(P, N, M, device=DEV)
    sp, tp = torch.randn(P, N, 3, device=DEV), torch.randn(P, M, 3, device=DEV)
    R = torch.empty(P, 9, device=DEV); t = torch.empty(P, 3, device=DEV); Rf = torch.empty_like(R); tf = torch.empty_like(t)
    cond = torch.empty(P, dtype=torch.float64, device=DEV); ok = torch.empty(P, dtype=torch.int32, device=DEV)
    st = lib.stream_of(conf)
    need = r.dr_procrustes_workspace_bytes(P, N, M)
    assert need > 0 and r.dr_procrustes_workspace_bytes(P, 256, 256) == 0

    def fit(ws, nbytes, n=N, m=M, c=conf):
        return r.dr_procrustes_f32(P, n, m, lib.ptr(c), lib.ptr(sp), lib.ptr(tp), None, None, 0, 1.0, 200.0, lib.ptr(R), lib.ptr(t), lib.ptr(Rf),
                                   lib.ptr(tf), lib.ptr(cond), lib.ptr(ok), None, lib.ptr(ws), nbytes, st)
    assert fit(None, 0) == EWORKSPACE
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    assert fit(ws, need - 1) == EWORKSPACE
    assert fit(ws, need) == OK
    assert fit(None, 0, 200, 256, torch.rand(P, 200, 256, device=DEV)) == OK          # register-resident tiles need none
    out = torch.empty(P, N + M, 3, dtype=torch.int64, device=DEV); cnt = torch.empty(P, dtype=torch.int32, device=DEV)
    need1 = r.dr_top1_union_workspace_bytes(P, N, M, 4)
    assert need1 > 0 and r.dr_top1_union_workspace_bytes(P, 16, M, 4) == 0 and r.dr_top1_union_workspace_bytes(P, N, M, 3) == 0
    assert r.dr_top1_union_f32(P, N, M, lib.ptr(conf), lib.ptr(out), lib.ptr(cnt), None, 0, st) == EWORKSPACE
    ws1 = torch.empty(need1, dtype=torch.uint8, device=DEV)
    assert r.dr_top1_union_f32(P, N, M, lib.ptr(conf), lib.ptr(out), lib.ptr(cnt), lib.ptr(ws1), need1, st) == OK
    torch.cuda.synchronize()
    rows = conf.argmax(2)
    got = lib.top1_union(conf)
    for p in range(P):
        have = set(map(tuple, got[p][:, 1:].cpu().tolist()))
        assert all((i, int(rows[p, i])) in have for i in range(N))


def test_loop_timeouts_are_attributed_to_the_call_that_had_them():
    """dr_denoise_loop_status: every loop call zeroes, and a timed-out co-resident Sinkhorn of that call sets, a status word in the call's OWN
    workspace.  Two engines: A's loop runs with the spin bound at one poll (times out), B's afterwards with the normal bound -- A's handle
    raises, B's does not (the process-wide dr_device_status flag is set by A and cannot tell them apart), and a clean re-run of A clears A's word."""
    from bench import make_engine, make_inputs
    from diffreg_hip import lib
    r = lib.raw()
    N = M = 384                                      # beyond the register-resident tiles: the loop's Sinkhorn calls take the co-resident form
    _, eng_a = make_engine("3dmatch", 2, 200.0, DEV)
    _, eng_b = make_engine("3dmatch", 2, 200.0, DEV)
    _, inp = make_inputs("3dmatch", 1, N, M, 8100, DEV)
    args = (inp["f_s"], inp["f_t"], inp["p_s"], inp["p_t"], inp["x_T"])
    r.dr_device_status(lib.stream_of(inp["x_T"]), 1)
    good = eng_b.run(*args, graph=False)
    good["_status"].check()
    try:
        timed_out = False
        for _ in range(20):
            r.dr_debug_sinkhorn_spin_limit(1)
            out_a = eng_a.run(*args, graph=False, borrow=True)
            torch.cuda.synchronize()
            r.dr_debug_sinkhorn_spin_limit(0)
            out_b = eng_b.run(*args, graph=False, borrow=True)
            torch.cuda.synchronize()
            if r.dr_denoise_loop_status(lib.ptr(out_a["_status"].ws), lib.stream_of(inp["x_T"]), 0) == ETIMEOUT:
                assert r.dr_device_status(lib.stream_of(inp["x_T"]), 0) == ETIMEOUT            # the process-wide flag: set, no owner
                out_b["_status"].check()                                                       # B's call was clean ...
                assert torch.equal(out_b["conf_matrix_pred"], good["conf_matrix_pred"])
                with pytest.raises(RuntimeError, match="gave up waiting"):
                    out_a["_status"].check()                                                   # ... A's was not; the check clears A's word
                assert r.dr_denoise_loop_status(lib.ptr(out_a["_status"].ws), lib.stream_of(inp["x_T"]), 0) == OK
                timed_out = True
                break
        assert timed_out
    finally:
        r.dr_debug_sinkhorn_spin_limit(0)
        r.dr_device_status(lib.stream_of(inp["x_T"]), 1)
    again = eng_a.run(*args, graph=False)
    again["_status"].check()
    assert torch.equal(again["conf_matrix_pred"], good["conf_matrix_pred"])


def test_round_6_entries_refuse_bad_calls():
    """the entries ABI 0.2.1 added (KPConv influence modes, the dual-softmax backward, the wide weight layout, the per-call status word) and the
    double-width Sinkhorn backward workspace: bad arguments and short workspaces come back as status codes, nothing is launched"""
    from diffreg_hip import lib
    r = lib.raw()
    lib.ensure_init()
    q = torch.rand(8, 3, device=DEV); idx = torch.zeros(8, 4, dtype=torch.int64, device=DEV); x = torch.rand(8, 4, device=DEV)
    kp = torch.rand(15, 3, device=DEV); out = torch.empty(8, 60, device=DEV)
    st = lib.stream_of(q)
    args = lambda infl, closest: (8, 8, 4, 4, 15, lib.ptr(q), lib.ptr(q), lib.ptr(idx), lib.ptr(x), lib.ptr(kp), 0.05, infl, closest, lib.ptr(out), 60, st)
    assert r.dr_kpconv_gather_mode_f32(*args(3, 0)) == EINVAL and r.dr_kpconv_gather_mode_f32(*args(-1, 0)) == EINVAL       # KP_influence outside 0..2
    assert r.dr_kpconv_gather_mode_f32(*args(lib.KP_INFLUENCE["gaussian"], 1)) == OK
    # dual-softmax backward: temperature, one mask without the other, a short workspace
    sim = torch.randn(2, 16, 12, device=DEV); g = torch.randn_like(sim); gs = torch.empty_like(sim)
    need = r.dr_dual_softmax_backward_workspace_bytes(2, 16, 12)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    m = torch.ones(2, 16, dtype=torch.uint8, device=DEV)
    call = lambda T, sm, tm, w, nb: r.dr_dual_softmax_backward_f32(2, 16, 12, lib.ptr(sim), T, sm, tm, lib.ptr(g), lib.ptr(gs), w, nb, st)
    assert call(0.0, None, None, lib.ptr(ws), need) == EINVAL and call(0.1, lib.ptr(m), None, lib.ptr(ws), need) == EINVAL
    assert call(0.1, None, None, lib.ptr(ws), need - 4) == EWORKSPACE and call(0.1, None, None, None, 0) == EWORKSPACE
    assert call(0.1, None, None, lib.ptr(ws), need) == OK
    # the Sinkhorn backward's vectors are doubles since round 6: the reported size is what the call insists on
    sc = torch.randn(1, 16, 12, device=DEV); one = torch.ones(1, device=DEV); ga = torch.empty(1, device=DEV)
    need = r.dr_sinkhorn_backward_workspace_bytes(1, 16, 12, 3)
    assert need == (8 + 2 * 3 * 17 + (2 * 3 + 1) * 13 + 17) * 8          # header, u / ub [T][N+1], v / vb [T+1 | T][M+1], the rows' dustbin partials -- doubles
    w2 = torch.empty(need, dtype=torch.uint8, device=DEV)
    skb = lambda nb: r.dr_sinkhorn_backward_f32(1, 16, 12, lib.ptr(sc), None, None, lib.ptr(one), 3, lib.ptr(g[:1].contiguous()), lib.ptr(gs[:1].contiguous()),
                                                 lib.ptr(ga), lib.ptr(w2), nb, st)
    assert skb(need // 2) == EWORKSPACE and skb(need) == OK
    # the wide weight layout: C beyond one logical block of 576 columns, a misaligned destination
    W = torch.randn(600, 64, device=DEV)
    assert r.dr_plane_weight_bytes_wide(1, 600, 64, 64, 64) == 0                   # (no size for a shape the layout does not have)
    buf = torch.empty(int(r.dr_plane_weight_bytes_wide(1, 528, 64, 64, 64)) + 64, dtype=torch.uint8, device=DEV)
    assert r.dr_pack_weight_planes_wide_f32(1, 600, 64, 64, 64, lib.ptr(W), lib.ptr(buf), st) == EINVAL
    assert r.dr_pack_weight_planes_wide_f32(1, 528, 64, 64, 64, lib.ptr(W), buf.data_ptr() + 4, st) == EINVAL
    assert r.dr_pack_weight_planes_wide_f32(0, 528, 64, 64, 64, lib.ptr(W), lib.ptr(buf), st) == EINVAL
    # the status word of a loop workspace
    assert r.dr_denoise_loop_status(None, st, 1) == EINVAL
    torch.cuda.synchronize()

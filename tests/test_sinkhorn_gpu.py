"""Parity of dr_sinkhorn_* (HIP, through the C ABI) with the oracle.  Needs a GPU."""
import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import diffreg_oracle as orc
from tests.helpers import T, masks, sinkhorn_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [(128, 128, 128, 128, 1.0), (200, 256, 200, 256, 0.37), (96, 80, 70, 61, 1.0), (256, 256, 256, 256, 1.0),
         (5, 7, 5, 7, 1.0), (1, 1, 1, 1, 1.0), (255, 253, 201, 77, 2.5), (130, 60, 130, 60, 1.0),
         (64, 250, 64, 250, 1.0), (300, 200, 300, 180, 1.0), (512, 512, 512, 512, 1.0), (257, 33, 257, 33, 1.0)]


def rel_err(got, ref):
    """(max abs error, max relative error over entries that are not negligible)."""
    got, ref = got.double().cpu(), ref.double()
    ae = (got - ref).abs().max().item()
    sel = ref.abs() > 1e-12 * ref.abs().max()
    re_ = ((got - ref).abs() / ref.abs().clamp_min(1e-30))[sel].max().item()
    return ae, re_


@pytest.mark.parametrize("N,M,nv,mv,alpha", CASES)
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_conf_matches_oracle(N, M, nv, mv, alpha, dt):
    from diffreg_hip import lib
    tdt = torch.float32 if dt == "f32" else torch.float64
    sc, sm, tm = sinkhorn_case(N, M, nv, mv, tdt)
    a = torch.tensor(alpha)
    ref = orc.sinkhorn_log(sc, a, 3, sm, tm).exp()[:, :-1, :-1]
    raw = T(3.0 * synth.hash_normal(1, N * 1000 + M, (1, N, M))).to(tdt)      # unmasked buffer
    got = lib.sinkhorn(raw.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV), apply_mask=True)
    assert got.dtype == tdt and got.shape == (1, N, M)
    ae, re_ = rel_err(got, ref)
    assert re_ < 2e-5 and ae < 1e-6, (ae, re_)
    if dt == "f64":
        g32 = lib.sinkhorn(raw.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV), apply_mask=True, out_f32=True)
        assert g32.dtype == torch.float32
        e32 = rel_err(g32, ref)
        assert e32[1] < 2e-5, e32
        gs = lib.sinkhorn(raw.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV), apply_mask=True, strict=True)
        es = rel_err(gs, ref)
        # float64 arithmetic end to end; the marginals are float32 logs in the reference (Q22), so the
        # device logf vs host log ulp shows up at the 1e-7 level
        assert es[1] < 2e-6, es


@pytest.mark.parametrize("N,M,nv,mv", [(128, 128, 128, 128), (96, 80, 70, 61), (256, 256, 256, 256), (40, 300, 40, 300)])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_log_output_matches_oracle(N, M, nv, mv, dt):
    from diffreg_hip import lib
    tdt = torch.float32 if dt == "f32" else torch.float64
    sc, sm, tm = sinkhorn_case(N, M, nv, mv, tdt)
    a = torch.tensor(1.0)
    ref = orc.sinkhorn_log(sc, a, 3, sm, tm)
    got = lib.sinkhorn(sc.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV), log_output=True, strict=(dt == "f64")).cpu()
    assert got.shape == (1, N + 1, M + 1)
    fin = torch.isfinite(ref)
    assert torch.equal(fin, torch.isfinite(got))
    tol = 1e-4 if dt == "f32" else 5e-6         # 1e-4 abs on the log assignment
    err = (got[fin] - ref[fin]).abs().max().item()
    assert err < tol, err


@pytest.mark.parametrize("B,N,M,nv,mv", [(2, 564, 629, 564, 629), (4, 564, 629, 564, 629), (6, 648, 655, 601, 540), (8, 512, 512, 470, 512),
                                         (3, 300, 1000, 300, 1000)])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_batched_large_tiles_coresident(B, N, M, nv, mv, dt):
    """Tiles beyond 256 x 256 run in ONE co-resident launch while all their workgroups fit the chip beside a second launch: one row
    per wave for small batches (B ceil(N / 8) <= 256 workgroups), two rows per wave up to 768 columns beyond that (4 real-size pairs,
    cfg3's 8 x 512 x 512), the multi-launch grid form otherwise.  Every tile of the batch is held to the oracle on its own scores."""
    from diffreg_hip import lib
    tdt = torch.float32 if dt == "f32" else torch.float64
    a = torch.tensor(1.0)
    raw = torch.cat([T(3.0 * synth.hash_normal(7 + b, N * 1000 + M, (1, N, M))) for b in range(B)]).to(tdt)
    sm = torch.arange(N)[None].expand(B, N) < torch.tensor([nv - 3 * b for b in range(B)])[:, None]
    tm = torch.arange(M)[None].expand(B, M) < torch.tensor([mv - 5 * b for b in range(B)])[:, None]
    got = lib.sinkhorn(raw.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV), apply_mask=True)
    assert got.dtype == tdt and got.shape == (B, N, M)
    for b in range(B):
        sc = raw[b:b + 1].masked_fill(~(sm[b][None, :, None] & tm[b][None, None, :]), float("-inf"))
        ref = orc.sinkhorn_log(sc, a, 3, sm[b:b + 1], tm[b:b + 1]).exp()[:, :-1, :-1]
        ae, re_ = rel_err(got[b:b + 1], ref)
        assert re_ < 2e-5 and ae < 1e-6, (b, ae, re_)


@pytest.mark.parametrize("B,N,M,nv,mv", [(8, 1024, 2048, 1000, 2000), (5, 1000, 1530, 1000, 1530), (7, 1024, 1000, 1024, 990), (3, 997, 2047, 900, 2047),
                                         (8, 1024, 1280, 1024, 1280), (5, 1024, 1532, 1024, 1532), (16, 512, 800, 512, 800)])
@pytest.mark.parametrize("dt,out_f32", [("f32", False), ("f64", True), ("f64", False)])
def test_batch_form_keeps_the_whole_batch_in_registers(B, N, M, nv, mv, dt, out_f32):
    """The BATCH form of the co-resident Sinkhorn (round 5; cfg5's 8 x 1024 x 2048 per call, real 2D-3D sizes 1 000 x 1 530, odd extents): a wave
    keeps eight rows of exponentials in registers, a workgroup = 32 rows = one CU, a tile's column sums cross its <= 32 workgroups once per
    iteration; one launch.  Every tile is held to the oracle on its own scores (different masks per tile), the three type pairs the loops use
    (head: f32 -> f32; state: f64 -> f32 warp confidences; read-out: f64 -> f64), and two launches give the same bits."""
    from diffreg_hip import lib
    tdt = torch.float32 if dt == "f32" else torch.float64
    a = torch.tensor(1.0)
    raw = torch.cat([T(3.0 * synth.hash_normal(17 + b, N * 1000 + M, (1, N, M))) for b in range(B)]).to(tdt)
    sm = torch.arange(N)[None].expand(B, N) < torch.tensor([nv - 3 * b for b in range(B)])[:, None]
    tm = torch.arange(M)[None].expand(B, M) < torch.tensor([mv - 5 * b for b in range(B)])[:, None]
    x = raw.to(DEV)
    from tests.helpers import guarded
    odt = torch.float32 if (dt == "f32" or out_f32) else torch.float64
    gout, check = guarded((B, N, M), odt, DEV)                    # the output between two 64 KiB guard bands
    got = lib.sinkhorn(x, a.to(DEV), 3, sm.to(DEV), tm.to(DEV), apply_mask=True, out_f32=out_f32, out=gout).clone()
    check()
    assert got.dtype == odt and got.shape == (B, N, M)
    again = lib.sinkhorn(x, a.to(DEV), 3, sm.to(DEV), tm.to(DEV), apply_mask=True, out_f32=out_f32)
    assert torch.equal(got, again)
    lib.device_status(DEV)
    for b in sorted(set((0, B // 2, B - 1))):
        sc = raw[b:b + 1].masked_fill(~(sm[b][None, :, None] & tm[b][None, None, :]), float("-inf"))
        ref = orc.sinkhorn_log(sc, a, 3, sm[b:b + 1], tm[b:b + 1]).exp()[:, :-1, :-1]
        ae, re_ = rel_err(got[b:b + 1], ref)
        assert re_ < 2e-5 and ae < 1e-6, (b, ae, re_)
    # the multi-launch grid form (the path of batches that do not fit the chip) agrees with it to rounding
    lib.raw().dr_debug_enable_env(1)
    import os
    os.environ["DR_SK_BATCH"] = "0"
    try:
        grid = lib.sinkhorn(x, a.to(DEV), 3, sm.to(DEV), tm.to(DEV), apply_mask=True, out_f32=out_f32)
    finally:
        os.environ.pop("DR_SK_BATCH")
        lib.raw().dr_debug_enable_env(1 if os.environ.get("DR_DIAGNOSTICS") == "1" else 0)
    big = grid.double() > 1e-12
    assert ((got.double() - grid.double()).abs() / grid.double().clamp_min(1e-30))[big].max().item() < 2e-5


def test_minshift_and_batch_of_different_masks():
    from diffreg_hip import lib
    B, N, M = 5, 256, 256
    x = T(synth.hash_normal(9, 1, (B, N, M))).float() * 2.0 + 3.0
    a = torch.tensor(1.0)
    nvs, mvs = [256, 200, 256, 17, 255], [256, 256, 131, 250, 1]
    sm = torch.stack([torch.arange(N) < n for n in nvs])
    tm = torch.stack([torch.arange(M) < m for m in mvs])
    ref = []
    for b in range(B):
        s = x[b:b + 1] - x[b].min()
        ref.append(orc.sinkhorn_conf(s, a, 3, sm[b:b + 1], tm[b:b + 1]))
    ref = torch.cat(ref)
    got = lib.sinkhorn(x.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV), minshift=True, apply_mask=True)
    assert rel_err(got, ref)[1] < 2e-5
    xd = x.double()
    got64 = lib.sinkhorn(xd.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV), minshift=True, apply_mask=True)
    assert got64.dtype == torch.float64 and rel_err(got64, ref)[1] < 2e-5


@pytest.mark.parametrize("N,M,nv,mv,minshift", [(300, 400, 300, 400, True), (300, 400, 257, 333, True), (40, 2304, 40, 2304, False),
                                                (24, 2100, 20, 2011, True)])
@pytest.mark.parametrize("dt,strict", [("f32", False), ("f64", False), ("f64", True)])
def test_streaming_kernel_matches_oracle(N, M, nv, mv, minshift, dt, strict):
    """The one-workgroup-per-tile streaming kernel (sk_stream_kernel) serves the min-shift inside the call on tiles beyond
    256 x 256 and every tile wider than 2048 columns -- shapes no other test reaches (round 3 shipped it with the first row
    pass not accumulating its sum; nothing noticed)."""
    from diffreg_hip import lib
    tdt = torch.float32 if dt == "f32" else torch.float64
    raw = (T(3.0 * synth.hash_normal(11, N * 1000 + M, (2, N, M))) + 1.5).to(tdt)
    sm = (torch.arange(N)[None] < torch.tensor([nv, max(1, nv - 7)])[:, None])
    tm = (torch.arange(M)[None] < torch.tensor([mv, max(1, mv - 11)])[:, None])
    a = torch.tensor(0.8)
    got = lib.sinkhorn(raw.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV), apply_mask=True, minshift=minshift, strict=strict)
    for b in range(2):
        s = raw[b:b + 1] - (raw[b].min() if minshift else 0)
        s = s.masked_fill(~(sm[b][None, :, None] & tm[b][None, None, :]), float("-inf"))
        ref = orc.sinkhorn_log(s, a, 3, sm[b:b + 1], tm[b:b + 1]).exp()[:, :-1, :-1]
        ae, re_ = rel_err(got[b:b + 1], ref)
        assert re_ < 2e-5 and ae < 1e-6, (b, ae, re_)


@pytest.mark.parametrize("N,M,nv,mv", [(128, 128, 100, 77), (256, 256, 201, 256), (96, 80, 33, 80), (300, 400, 257, 333),
                                       (512, 512, 400, 511), (64, 1200, 50, 1111)])
@pytest.mark.parametrize("dt,strict", [("f32", False), ("f64", False), ("f64", True)])
def test_ragged_padded_tile_equals_unpadded_problem(N, M, nv, mv, dt, strict):
    """DR_SK_RAGGED: a tile padded to (N, M) with masks gives exactly the Sinkhorn of its own (nv x mv) problem --
    unlike the reference's pad-and-mask call, where padded rows / columns keep marginal mass (quirk Q19).  Every
    kernel path: register-resident, multi-workgroup, strict fp64; conf and log output."""
    from diffreg_hip import lib
    dtype = torch.float32 if dt == "f32" else torch.float64
    g = torch.Generator().manual_seed(N * 7 + M)
    small = (torch.randn(2, nv, mv, generator=g) * 3).to(dtype)
    big = torch.full((2, N, M), 5.0, dtype=dtype)            # garbage in the padding: must not matter
    big[:, :nv, :mv] = small
    sm = (torch.arange(N)[None] < nv).expand(2, N).contiguous()
    tm = (torch.arange(M)[None] < mv).expand(2, M).contiguous()
    a = torch.tensor(0.7)
    ones_s, ones_t = torch.ones(2, nv, dtype=torch.bool), torch.ones(2, mv, dtype=torch.bool)
    for log_output in (False, True):
        ref = lib.sinkhorn(small.to(DEV), a, 3, ones_s.to(DEV), ones_t.to(DEV), apply_mask=True, strict=strict, log_output=log_output)
        got = lib.sinkhorn(big.to(DEV), a, 3, sm.to(DEV), tm.to(DEV), apply_mask=True, strict=strict, ragged=True, log_output=log_output)
        if log_output:
            # valid block + dustbin row / column of the (nv+1) x (mv+1) problem sit at [:nv, :mv], [N, :mv], [:nv, M], [N, M]
            gv = torch.cat([torch.cat([got[:, :nv, :mv], got[:, :nv, M:M + 1]], 2),
                            torch.cat([got[:, N:N + 1, :mv], got[:, N:N + 1, M:M + 1]], 2)], 1)
            assert (gv - ref).abs().max().item() < 2e-5
        else:
            assert (got[:, :nv, :mv] - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item() * 1e3)
            assert got[:, nv:, :].abs().max().item() == 0 if nv < N else True
            assert got[:, :, mv:].abs().max().item() == 0 if mv < M else True


def test_extreme_scores_do_not_overflow():
    """the scaling form must survive what the log-domain reference survives (dustbins keep sums > 0)."""
    from diffreg_hip import lib
    N, M = 64, 96
    x = T(synth.hash_normal(4, 2, (1, N, M))).float() * 60.0         # spread of ~ +-200
    x[0, 3, :] = -300.0
    x[0, :, 5] = 250.0
    sm, tm = masks(N, M)
    a = torch.tensor(1.0)
    ref = orc.sinkhorn_conf(x, a, 3, sm, tm)
    got = lib.sinkhorn(x.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV)).cpu()
    assert torch.isfinite(got).all()
    assert (got.double() - ref.double()).abs().max().item() < 1e-6
    big = ref > 1e-8
    assert ((got.double() - ref.double()).abs() / ref.double().clamp_min(1e-30))[big].max().item() < 1e-4


def test_full_size_batch_properties():
    """BASELINE-size batch (1024 tiles of 256x256): size-independent properties of the output."""
    from diffreg_hip import lib
    B, N, M = 1024, 256, 256
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(B, N, M, generator=g) * 2).to(DEV)
    a = torch.tensor(1.0, device=DEV)
    conf = lib.sinkhorn(x, a, 3)
    cs = conf.sum(1)
    assert torch.isfinite(conf).all() and (conf >= 0).all()
    assert (cs <= 1.0 + 1e-5).all()
    again = lib.sinkhorn(x, a, 3)
    assert torch.equal(conf, again)                       # deterministic
    perm = torch.randperm(B, generator=g).to(DEV)
    assert torch.equal(lib.sinkhorn(x[perm].contiguous(), a, 3), conf[perm])   # tiles are independent
    for b in (0, 511, 1023):
        sm, tm = masks(N, M)
        ref = orc.sinkhorn_conf(x[b:b + 1].cpu(), torch.tensor(1.0), 3, sm, tm)
        assert rel_err(conf[b:b + 1], ref)[1] < 2e-5


@pytest.mark.parametrize("B", [512, 513, 777, 1300])
def test_persistent_kernel_equals_per_tile_kernel(B):
    """launches of >= 512 float tiles of 256 x 256 run the persistent LDS-prefetching kernel (one workgroup per CU walking
    over tiles, non-temporal loads/stores); it must give the bits of the one-tile-per-workgroup kernel that smaller launches use"""
    from diffreg_hip import lib
    g = torch.Generator(device="cpu").manual_seed(B)
    x = (torch.randn(B, 256, 256, generator=g) * 3).to(DEV)
    a = torch.tensor(0.37, device=DEV)
    big = lib.sinkhorn(x, a, 3)
    small = torch.cat([lib.sinkhorn(x[i:i + 200].contiguous(), a, 3) for i in range(0, B, 200)])
    assert torch.equal(big, small)
    for b in (0, B - 1):
        sm, tm = masks(256, 256)
        ref = orc.sinkhorn_conf(x[b:b + 1].cpu(), torch.tensor(0.37), 3, sm, tm)
        assert rel_err(big[b:b + 1], ref)[1] < 2e-5


@pytest.mark.parametrize("N,M,nv,mv", [(256, 256, 256, 256), (128, 128, 100, 77), (96, 80, 70, 61), (255, 253, 201, 77), (5, 7, 5, 7)])
def test_f16_entry_matches_oracle_on_f16_scores(N, M, nv, mv):
    """dr_sinkhorn_f16 (opt-in, SURVEY 8b's reduced-precision entry): fp16 score tiles in, fp16 confidences out, the iteration in fp32 -- the
    oracle on the SAME fp16-rounded scores, to fp16's own rounding of the result (2^-11 relative, 6e-8 absolute in the subnormal range);
    tiles beyond 256 x 256 are refused."""
    from diffreg_hip import lib
    sc, sm, tm = sinkhorn_case(N, M, nv, mv, torch.float32)
    a = torch.tensor(1.0)
    raw = T(3.0 * synth.hash_normal(1, N * 1000 + M, (1, N, M))).half()
    scm = raw.float().masked_fill(~(sm[:, :, None] & tm[:, None, :]), float("-inf"))
    ref = orc.sinkhorn_log(scm, a, 3, sm, tm).exp()[:, :-1, :-1].double()
    got = lib.sinkhorn_f16(raw.to(DEV), a.to(DEV), 3, sm.to(DEV), tm.to(DEV), apply_mask=True)
    assert got.dtype == torch.float16 and got.shape == (1, N, M)
    err = (got.double().cpu() - ref).abs()
    assert bool((err <= 6e-4 * ref + 7e-8).all()), float((err / (ref + 1e-7)).max())
    if nv == N and mv == M:                                     # unmasked: the plain-tile kernels (256 x 256 and 128 x 128 take the fast form)
        ref2 = orc.sinkhorn_log(raw.float(), a, 3, sm, tm).exp()[:, :-1, :-1].double()
        got2 = lib.sinkhorn_f16(raw.to(DEV).repeat(3, 1, 1), a.to(DEV), 3)
        err2 = (got2.double().cpu() - ref2).abs()
        assert bool((err2 <= 6e-4 * ref2 + 7e-8).all())
    with pytest.raises(RuntimeError):
        lib.sinkhorn_f16(torch.zeros(1, 300, 300, dtype=torch.float16, device=DEV), a.to(DEV), 3)


def test_bad_arguments():
    from diffreg_hip import lib
    with pytest.raises(RuntimeError):
        lib.sinkhorn(torch.zeros(1, 4, 4, device=DEV), torch.tensor(1.0), 0)


def test_two_concurrent_batch_form_launches_make_progress():
    """Two batch-form launches on two streams (the engines' concurrent 8-pair calls): each wants every CU.  A tile needs only ITS 32 workgroups
    co-resident and workgroup ids are tile-major, so whichever tiles are resident finish and free their CUs -- no launch may hold a part of every
    tile.  (A round-5 variant dealt ids as (id % B, id / B) for XCD locality: alone 0-3 % faster, two concurrent launches spun into their
    time-outs, seconds per call.)  Checked: same bits as the launch alone, no time-out flag, and a wall time far below one time-out."""
    import time
    from diffreg_hip import lib
    B, N, M = 8, 1024, 2048
    a = torch.tensor(1.0, device=DEV)
    xs = [T(3.0 * synth.hash_normal(31 + s, 5, (B, N, M))).to(DEV) for s in range(2)]
    alone = [lib.sinkhorn(x, a, 3) for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    outs = [[], []]
    t0 = time.perf_counter()
    for rep in range(6):
        for s in range(2):
            with torch.cuda.stream(streams[s]):
                outs[s].append(lib.sinkhorn(xs[s], a, 3))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.device_status(DEV)                                         # raises on DR_ETIMEOUT
    for s in range(2):
        for o in outs[s]:
            assert torch.equal(o, alone[s])
    assert dt < 0.5, dt                                            # (12 launches of ~0.1 ms; one time-out alone is seconds)

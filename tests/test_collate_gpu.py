"""GPU parity of the collate-time ops through the C ABI: dr_grid_subsample_f32, dr_radius_neighbors_f32 against
oracle/collate_oracle.py and against the reference's own C++ (oracle/_ref, when built), and the device level loop against the
index arrays the KPFCN backbone is tested with."""
import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import collate_oracle as co
from tests.test_collate_oracle import scene, rows_equal_up_to_ties

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n_src,n_tgt,seed,dl", [(1400, 1200, 0, 0.05), (3000, 2500, 1, 0.05), (3000, 2500, 1, 0.1), (700, 900, 2, 0.2),
                                                 (20000, 17000, 4, 0.03), (5, 3, 5, 0.05)])
def test_grid_subsample_bit_exact(n_src, n_tgt, seed, dl):
    from diffreg_hip.collate import batch_grid_subsampling_kpconv
    P, L = scene(n_src, n_tgt, seed)
    sp, sl = batch_grid_subsampling_kpconv(T(P), T(L), sampleDl=dl)
    op, ol = co.grid_subsample_batch(P, L, dl)
    assert sl.dtype == torch.int32 and np.array_equal(sl.cpu().numpy(), ol)
    assert np.array_equal(sp.cpu().numpy(), op)               # same points, same order, same float32 bits
    if co.ref_lib() is not None and n_src <= 3000:
        rp, rl = co.ref_subsample_batch(P, L, dl)
        assert np.array_equal(rp[co.canonical_order(rp, rl, P, L, dl)], sp.cpu().numpy())


def test_grid_subsample_four_clouds_and_empty():
    from diffreg_hip import lib
    Pa, La = scene(900, 800, 6)
    Pb, Lb = scene(500, 700, 7)
    P, L = np.concatenate([Pa, Pb]), np.concatenate([La, [0], Lb]).astype(np.int32)      # an empty cloud in the middle
    out, ol, tot, status = lib.grid_subsample(T(P), T(L), 0.06)
    op, ol_ref = co.grid_subsample_batch(P, L, 0.06)
    assert int(status) == 0 and int(tot) == len(op) and np.array_equal(ol.cpu().numpy(), ol_ref)
    assert np.array_equal(out[:len(op)].cpu().numpy(), op)
    # extent / cell beyond 65533 cells per axis is reported, not silently wrong
    far = P.copy(); far[0, 0] += 1e4
    _, _, _, st2 = lib.grid_subsample(T(far), T(L), 0.06)
    assert int(st2) == 1


@pytest.mark.parametrize("n_src,n_tgt,seed,dl,limit", [(1400, 1200, 0, 0.05, 40), (3000, 2500, 1, 0.1, 35), (20000, 17000, 4, 0.03, 38)])
def test_radius_neighbors_vs_oracle_and_reference(n_src, n_tgt, seed, dl, limit):
    from diffreg_hip.collate import batch_neighbors_kpconv
    P, L = scene(n_src, n_tgt, seed)
    sp, sl = co.grid_subsample_batch(P, L, dl)
    for (Q, ql, S, sl_, r) in ((P, L, P, L, 1.25 * dl), (sp, sl, P, L, 1.25 * dl), (P, L, sp, sl, 2.5 * dl)):
        got = batch_neighbors_kpconv(T(Q), T(S), T(ql), T(sl_), r, limit).cpu().numpy()
        if len(Q) <= 6000:
            want = co.radius_neighbors_batch(Q, S, ql, sl_, r)[:, :limit]
            assert got.dtype == np.int64 and np.array_equal(got, want)          # the oracle breaks ties like the kernel
        if co.ref_lib() is not None:
            ref = co.ref_batch_query(Q, S, ql, sl_, r)[:, :limit]                # the reference's own kd-tree search
            assert rows_equal_up_to_ties(ref.astype(np.int64), got, Q, S)


def test_radius_neighbors_edge_cases():
    from diffreg_hip import lib
    P, L = scene(600, 500, 8)
    # queries far outside the support cloud find nothing; the row is all padding
    n0 = int(L[0])
    Q = np.concatenate([P[:10] + 50.0, P[n0:n0 + 10]]).astype(np.float32)
    out, mc, st = lib.radius_neighbors(T(Q), T(P), T(np.array([10, 10], np.int32)), T(L), 0.08, 16)
    o = out.cpu().numpy()
    assert (o[:10] == len(P)).all() and (o[10:, 0] == np.arange(n0, n0 + 10)).all()   # a support point is its own nearest neighbour
    # a query never sees the other cloud of the pair
    assert (o[10:] >= n0).all()
    # untruncated width and truncation agree on the common columns
    from diffreg_hip.collate import batch_neighbors_kpconv
    a = batch_neighbors_kpconv(T(P), T(P), T(L), T(L), 0.08, 64).cpu().numpy()
    b = batch_neighbors_kpconv(T(P), T(P), T(L), T(L), 0.08, 7).cpu().numpy()
    assert b.shape[1] == 7 and np.array_equal(a[:, :7], b)
    with pytest.raises(RuntimeError):
        lib.radius_neighbors(torch.from_numpy(Q), torch.from_numpy(P), torch.tensor([10, 10]), torch.from_numpy(L), 0.08, 16)   # CPU tensors


def test_level_loop_matches_cpu_collate_and_feeds_the_backbone():
    """build_kpfcn_inputs (dataloader.py:120-211 on device) against the same loop run with the oracle, level by level"""
    from diffreg_hip.collate import build_kpfcn_inputs
    P, L = scene(1400, 1200, 0)
    cfg = dict(architecture=synth.KPFCN_ARCH, first_subsampling_dl=0.05, conv_radius=2.5, deform_radius=5.0)
    limits = [28, 28, 30, 32]
    got = build_kpfcn_inputs(T(P), T(L), cfg, limits)
    pts, lens, r_n = P, L, 0.05 * 2.5
    for lvl in range(4):
        assert np.array_equal(got["points"][lvl].cpu().numpy(), pts) and np.array_equal(got["stack_lengths"][lvl].cpu().numpy(), lens)
        conv = co.radius_neighbors_batch(pts, pts, lens, lens, r_n)[:, :limits[lvl]]
        assert np.array_equal(got["neighbors"][lvl].cpu().numpy(), conv)
        if lvl == 3:
            assert got["pools"][lvl].shape == (0, 1) and got["upsamples"][lvl].shape == (0, 1)
            break
        pp, pl = co.grid_subsample_batch(pts, lens, 2 * r_n / 2.5)
        assert np.array_equal(got["pools"][lvl].cpu().numpy(), co.radius_neighbors_batch(pp, pts, pl, lens, r_n)[:, :limits[lvl]])
        assert np.array_equal(got["upsamples"][lvl].cpu().numpy(), co.radius_neighbors_batch(pts, pp, lens, pl, 2 * r_n)[:, :limits[lvl]])
        pts, lens, r_n = pp, pl, 2 * r_n


def test_device_collate_feeds_the_backbone(golden):
    """end of row f4 -> row f1: the index arrays built on the device drive the KPFCN coarse phase; same features as the
    oracle backbone on the same arrays"""
    from diffreg_hip.backbone import KPFCNEngine
    from diffreg_hip.collate import build_kpfcn_inputs
    from oracle import kpfcn_oracle as ko
    from tests.test_oracle_golden import kpfcn_inputs
    g, sd, tb = kpfcn_inputs(golden)
    cfg = dict(architecture=synth.KPFCN_ARCH, first_subsampling_dl=synth.KPFCN_CFG["first_subsampling_dl"], conv_radius=2.5, deform_radius=5.0)
    lengths = torch.as_tensor(np.asarray(synth.make_kpfcn_batch()["stack_lengths"][0]), dtype=torch.int32)
    batch = build_kpfcn_inputs(tb["points"][0].cuda(), lengths.cuda(), cfg, [28, 28, 30, 32])
    batch["features"] = torch.ones(len(tb["points"][0]), 1, device="cuda")
    out = KPFCNEngine(sd, device="cuda:0").forward(batch).cpu()
    cpu = {k: [t.cpu() for t in v] if isinstance(v, list) else v.cpu() for k, v in batch.items()}
    ref = ko.kpfcn_coarse(sd, cpu)
    assert out.shape == ref.shape and out.shape[0] == int(batch["stack_lengths"][-2].sum())   # coarse_level = -2
    assert (out - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_raw_clouds_to_conf_matrix_on_device(golden):
    """rows f4 -> f1 -> a1 in one go: two raw clouds -> collate_fn_device -> Pipeline (overlay backbone + denoising loop) ->
    conf_matrix_pred, against oracle backbone + oracle loop on the arrays the device collate produced (themselves held to the
    collate oracle above)."""
    from diffreg_hip.collate import collate_fn_device
    from models.pipeline import Pipeline
    from oracle import kpfcn_oracle as ko, diffreg_oracle as orc
    from tests.test_oracle_golden import kpfcn_inputs
    from tests.test_models_api_gpu import ref_like_config, to_attr
    from tests.helpers import weights
    variant, steps, mc = "3dmatch", 2, 200
    g, bsd, tb = kpfcn_inputs(golden)
    cfg = ref_like_config(variant, steps, mc)
    kc = dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), KP_influence="linear", aggregation_mode="sum", deformable=False,
              use_batch_norm=True, fine_feature_dim=264, coarse_level=-2, deform_radius=5.0)
    cfg.kpfcn_config = to_attr(kc)
    model = Pipeline(cfg)
    sd = model.state_dict()
    W = weights(variant)
    sd.update(W)
    sd.update({"backbone." + k: v for k, v in bsd.items()})
    model.load_state_dict(sd)
    model = model.to("cuda:0").eval()
    L = synth.make_kpfcn_batch()["stack_lengths"][0]
    P0 = tb["points"][0]
    data = collate_fn_device([(P0[:L[0]].cuda(), P0[L[0]:].cuda())], kc, [28, 28, 30, 32])
    ns, nt = [int(v) for v in data["stack_lengths"][-2]]
    assert data["src_mask"].shape == (1, ns) and data["tgt_mask"].all() and torch.equal(data["tgt_ind_coarse"].cpu(), torch.arange(ns, ns + nt))
    x_T = torch.from_numpy(synth.hash_normal(78, 5, (1, ns, nt))).float()
    data["x_T"] = x_T.cuda()
    cpu = {k: [t.cpu() for t in v] if isinstance(v, list) else v.cpu() for k, v in data.items()}
    out = model(data)
    conf = out["conf_matrix_pred"][0].cpu()
    feats = ko.kpfcn_coarse(bsd, cpu)
    pts = cpu["points"][2]
    v = synth.VARIANTS[variant]
    ms, mt = torch.ones(1, ns, dtype=torch.bool), torch.ones(1, nt, dtype=torch.bool)
    ref = orc.denoise_loop(W, v, feats[None, :ns], feats[None, ns:], pts[None, :ns], pts[None, ns:], ms, mt, x_T, steps, mc, variant=variant)
    d = (conf - ref["conf_matrix_pred"][0]).abs()
    assert (d > 1e-4).double().mean().item() <= 1e-3, d.max().item()


def test_radius_neighbors_more_than_cap_points_in_the_ball():
    """a dense clump: 300 supports inside one ball, more than the 64 candidates a lane keeps before ranking -- the kept set must
    still be the `limit` nearest (the farthest kept candidate makes room), in order"""
    from diffreg_hip import lib
    clump = (0.01 * synth.hash_normal(3, 5, (300, 3))).astype(np.float32) + np.array([0.5, 0.5, 0.5], np.float32)
    bg, _ = scene(500, 400, 9)
    S = np.concatenate([clump, bg[:400]]).astype(np.float32)
    sl = np.array([len(S)], np.int32)
    Q = np.concatenate([clump[:20], bg[400:420]]).astype(np.float32)
    ql = np.array([len(Q)], np.int32)
    for limit in (64, 40):
        out, mc, st = lib.radius_neighbors(T(Q), T(S), T(ql), T(sl), 0.06, limit)
        want = co.radius_neighbors_batch(Q, S, ql, sl, 0.06)
        assert int(mc) == want.shape[1] and want.shape[1] > 64
        assert np.array_equal(out.cpu().numpy(), want[:, :limit])


def test_grid_subsample_large_cloud_and_negative_coordinates():
    """131 072-pair sort (six global bitonic levels), coordinates on both sides of the origin, a cell size that does not divide
    the extent; against the reference's own C++ when it is built, else the oracle"""
    from diffreg_hip.collate import batch_grid_subsampling_kpconv
    u = synth.hash_uniform(11, 3, (90000, 3), -1.0, 1.0)
    P = np.stack([3.1 * u[:, 0], 2.3 * u[:, 1], 0.4 * np.sin(2 * u[:, 0]) + 0.05 * u[:, 2]], 1).astype(np.float32)
    L = np.array([50000, 40000], np.int32)
    sp, sl = batch_grid_subsampling_kpconv(T(P), T(L), sampleDl=0.037)
    if co.ref_lib() is not None:
        rp, rl = co.ref_subsample_batch(P, L, 0.037)
        assert np.array_equal(rl, sl.cpu().numpy())
        assert np.array_equal(rp[co.canonical_order(rp, rl, P, L, 0.037)], sp.cpu().numpy())
    else:
        op, ol = co.grid_subsample_batch(P, L, 0.037)
        assert np.array_equal(op, sp.cpu().numpy())

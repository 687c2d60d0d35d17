"""Plane-image ops (include/diffreg_hip.h "Plane images"; csrc/pgemm.hip): the layer nn.Linears of the loop with both operands
as fp16 hi / lo plane images, LayerNorm (+ residual) in the GEMM epilogue.  Against float64 products / torch LayerNorm; every
output between guard bands.  Needs a GPU."""
import numpy as np
import pytest
import torch

from diffreg_hip import lib
from tests.helpers import guarded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def image_like(rows, K):
    return guarded((lib.raw().dr_plane_image_bytes(rows, K),), torch.uint8, DEV, fill=0)


@pytest.mark.parametrize("rows", [1, 127, 1000, 4096 + 33])
@pytest.mark.parametrize("C", [432, 528, 256])
def test_image_round_trip_and_bounds(rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * torch.rand(rows, 1, generator=g) * 7).to(DEV)
    img, bnd = lib.planes_from_f32(x)
    back = lib.planes_to_f32(img, bnd, rows, C)
    assert rel(back, x.double()) < 3e-7                        # 22 significand bits relative to the row bound
    assert torch.equal(bnd, x.abs().amax(1))


@pytest.fixture
def row_block(request):
    """force the plane GEMM's workgroup height through the debug knob (include/diffreg_hip_debug.h): 0 = 128 rows, 2 = 64;
    None = the launcher's own rule (64 rows while the launch has fewer 128-row blocks than CUs)"""
    import os
    mode = request.param
    if mode is not None:
        lib.raw().dr_debug_enable_env(1)
        os.environ["DR_PG_HALF"] = str(mode)
    yield mode
    if mode is not None:
        os.environ.pop("DR_PG_HALF", None)
        lib.raw().dr_debug_enable_env(0)


@pytest.mark.parametrize("row_block", [None, 0, 2], indirect=True)
@pytest.mark.parametrize("rows,C", [(1000, 432), (300, 528), (129, 256)])
def test_layer_chain_against_float64(rows, C, row_block):
    """q|k|v with rotary (DR_PL_F32), merge + LayerNorm (DR_PL_LN), mlp0 on [x | msg] + ReLU (DR_PL_PLANES), mlp2 + LayerNorm +
    residual (DR_PL_LN): one GeometryAttentionLayer's GEMMs (transformero.py:60-96) through the plane ops, in each workgroup height."""
    torch.manual_seed(rows)
    x = torch.randn(rows, C, device=DEV) * (torch.rand(rows, 1, device=DEV) * 5 + 0.01)
    img, bnd = lib.planes_from_f32(x)
    # ---- q | k | v, rotary on q and k, three [rows, C] matrices
    W = torch.randn(3 * C, C, device=DEV) / C ** 0.5
    pk = lib.pack_weight_planes(W, 3, C)
    ang = torch.rand(rows, C // 2, device=DEV) * 6.28
    cosT, sinT = ang.cos().contiguous(), ang.sin().contiguous()
    out, chk = guarded((3, rows, C), torch.float32, DEV, fill=float("nan"))
    lib.linear_planes(rows, C, 3, img, bnd, C, pk, lib.PL_F32, out=out, ldo=C, blk_stride=rows * C, cos_t=cosT, sin_t=sinT, rot_mask=3,
                      rot_C=C, scale=0.5)
    chk()
    ref = x.double() @ W.double().t()

    def rot(z):
        e, o = z[:, 0::2], z[:, 1::2]
        return torch.stack([e * cosT.double() - o * sinT.double(), o * cosT.double() + e * sinT.double()], -1).reshape(z.shape)
    want = torch.stack([rot(ref[:, :C]), rot(ref[:, C:2 * C]), ref[:, 2 * C:]]) * 0.5
    assert not torch.isnan(out).any() and rel(out, want) < 2e-6
    # ---- merge + norm1 -> image only
    g1, b1 = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    Wm = torch.randn(C, C, device=DEV) / C ** 0.5
    lnb = lib.ln_bound(g1, b1)
    msg_img, chk_i = image_like(rows, C)
    msg_b, chk_b = guarded((rows,), torch.float32, DEV, fill=0)
    lib.linear_planes(rows, C, 1, img, bnd, C, lib.pack_weight_planes(Wm, 1, C), lib.PL_LN, out_image=msg_img, out_image_k=C, out_bound=msg_b,
                      gamma=g1, beta=b1, lnb=lnb)
    chk_i(); chk_b()
    refm = torch.nn.functional.layer_norm(x.double() @ Wm.double().t(), (C,), g1.double(), b1.double())
    assert rel(lib.planes_to_f32(msg_img, msg_b, rows, C), refm) < 3e-6
    assert bool((msg_b >= refm.abs().amax(1).float()).all())           # the analytic bound holds
    # ---- mlp0 on [x | msg], ReLU -> image of 2C columns
    W1 = torch.randn(2 * C, 2 * C, device=DEV) / (2 * C) ** 0.5
    hid_img, chk_h = image_like(rows, 2 * C)
    hid_b, chk_hb = guarded((rows,), torch.float32, DEV, fill=0)
    lib.linear_planes(rows, C, 2, img, bnd, C, lib.pack_weight_planes(W1, 2, C), lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=C, out_image=hid_img,
                      out_image_k=2 * C, out_bound=hid_b, relu=True)
    chk_h(); chk_hb()
    refh = torch.relu(torch.cat([x.double(), refm], 1) @ W1.double().t())
    assert rel(lib.planes_to_f32(hid_img, hid_b, rows, 2 * C), refh) < 4e-6 and bool((hid_b >= refh.abs().amax(1).float()).all())
    # ---- mlp2 + norm2 + residual -> fp32 rows and image
    W2 = torch.randn(C, 2 * C, device=DEV) / (2 * C) ** 0.5
    o32, chk_o = guarded((rows, C), torch.float32, DEV, fill=float("nan"))
    o_img, chk_oi = image_like(rows, C)
    o_b, _ = guarded((rows,), torch.float32, DEV, fill=0)
    lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, lib.pack_weight_planes(W2, 1, C), lib.PL_LN, out=o32, ldo=C, out_image=o_img,
                      out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=x, ldr=C, bound_resid=bnd, lnb=lnb)
    chk_o(); chk_oi()
    refo = x.double() + torch.nn.functional.layer_norm(refh @ W2.double().t(), (C,), g1.double(), b1.double())
    assert rel(o32, refo) < 2e-6 and rel(lib.planes_to_f32(o_img, o_b, rows, C), refo) < 2e-6
    assert bool((o_b >= refo.abs().amax(1).float()).all())


def test_head_padded_k_order():
    """the merge projection behind the attention kernel's image: head h at k = 112 h (d = 108 padded to 112, zeros in the pad)"""
    rows, C = 700, 432
    torch.manual_seed(0)
    att = torch.randn(rows, C, device=DEV)
    attp = torch.zeros(rows, 448, device=DEV)
    for hh in range(4):
        attp[:, 112 * hh:112 * hh + 108] = att[:, 108 * hh:108 * (hh + 1)]
    aimg, ab = lib.planes_from_f32(attp)
    Wm = torch.randn(C, C, device=DEV) / C ** 0.5
    out, chk = guarded((rows, C), torch.float32, DEV, fill=float("nan"))
    lib.linear_planes(rows, C, 1, aimg, ab, 448, lib.pack_weight_planes(Wm, 1, C, piece_len=108, piece_pad=112), lib.PL_F32, out=out, ldo=C)
    chk()
    assert rel(out, att.double() @ Wm.double().t()) < 2e-6


def test_degenerate_rows_stay_in_their_rows():
    """zero rows, rows 1e-6 and 1e8 times the others, and a row with inf / nan: every other row keeps fp32-level accuracy (the
    per-row power-of-two scaling isolates rows; nothing is shared across rows in the GEMM or the LayerNorm epilogue)"""
    rows, C = 384, 432
    torch.manual_seed(1)
    x = torch.randn(rows, C, device=DEV)
    x[3] = 0.0
    x[10] *= 1e-6
    x[11] *= 1e8
    x[200, 5] = float("inf")
    x[201, 7] = float("nan")
    img, bnd = lib.planes_from_f32(x)
    W = torch.randn(C, C, device=DEV) / C ** 0.5
    out = torch.empty(rows, C, device=DEV)
    lib.linear_planes(rows, C, 1, img, bnd, C, lib.pack_weight_planes(W, 1, C), lib.PL_F32, out=out, ldo=C)
    ref = x.double() @ W.double().t()
    good = torch.ones(rows, dtype=torch.bool, device=DEV)
    good[200] = good[201] = False
    err = (out.double() - ref).abs()[good] / ref.abs().amax(1, keepdim=True).clamp_min(1e-30)[good]
    assert float(err.max()) < 3e-6
    assert torch.equal(out[3], torch.zeros(C, device=DEV))
    assert not torch.isfinite(out[200]).all() and not torch.isfinite(out[201]).all()


@pytest.mark.parametrize("P,Lq,Lk,H,d", [(1, 32, 32, 1, 108), (2, 128, 64, 4, 108), (3, 200, 256, 4, 108), (2, 96, 80, 4, 132), (2, 64, 160, 4, 64)])
@pytest.mark.parametrize("masked", [False, True])
def test_attention_on_plane_images(P, Lq, Lk, H, d, masked):
    """dr_attention_planes: q | k | v as head-padded plane images with bounds (what the q|k|v GEMM's epilogue writes), softmax(q k^T /
    sqrt(d)) v per head as a plane image (what the merge GEMM reads) -- against float64 attention (transformero.py:73-84: keys
    outside k_mask get -inf)."""
    torch.manual_seed(P * 1000 + Lq + d)
    C = H * d
    q = torch.randn(P, Lq, C, device=DEV)
    k = torch.randn(P, Lk, C, device=DEV)
    v = torch.randn(P, Lk, C, device=DEV) * 3
    qm = km = None
    if masked:
        qm = torch.ones(P, Lq, dtype=torch.bool, device=DEV)
        km = torch.ones(P, Lk, dtype=torch.bool, device=DEV)
        qm[:, Lq - 5:] = False
        km[:, Lk - 7:] = False
        km[0, :3] = False
    o = lib.attention_planes(q, k, v, H, qm, km)
    qh, kh, vh = (z.double().view(P, -1, H, d).transpose(1, 2) for z in (q, k, v))
    logit = qh @ kh.transpose(-1, -2) / d ** 0.5
    if masked:
        logit = logit.masked_fill(~km[:, None, None, :], float("-inf"))
    ref = (torch.softmax(logit, -1) @ vh).transpose(1, 2).reshape(P, Lq, C)
    keep = qm if masked else torch.ones(P, Lq, dtype=torch.bool, device=DEV)
    assert not torch.isnan(o[keep]).any()
    assert float((o.double() - ref)[keep].abs().max()) < 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("P,Lq,Lk,H,d", [(2, 128, 64, 4, 108), (2, 96, 80, 4, 132), (2, 64, 160, 4, 64)])
def test_attention_on_plane_images_f16_entry(P, Lq, Lk, H, d):
    """dr_attention_planes_f16 (opt-in, SURVEY 8b's reduced-precision entry): ONE fp16 product per contraction.  Not the three-product result
    (the mode is on) and within the 11-bit operands' error of float64 attention: 3e-3 of the largest output."""
    torch.manual_seed(P + Lq + d)
    C = H * d
    q, k, v = torch.randn(P, Lq, C, device=DEV), torch.randn(P, Lk, C, device=DEV), torch.randn(P, Lk, C, device=DEV) * 3
    o3, o1 = lib.attention_planes(q, k, v, H), lib.attention_planes(q, k, v, H, f16=True)
    qh, kh, vh = (z.double().view(P, -1, H, d).transpose(1, 2) for z in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / d ** 0.5, -1) @ vh).transpose(1, 2).reshape(P, Lq, C)
    e3, e1 = float((o3.double() - ref).abs().max()), float((o1.double() - ref).abs().max())
    assert e3 < 1e-5 * float(ref.abs().max()) and 1e-5 * float(ref.abs().max()) < e1 < 3e-3 * float(ref.abs().max()), (e3, e1)


@pytest.mark.parametrize("gain1", [1.0, 1.37, 0.6])
def test_column_blocks_sharing_one_image_share_one_scale(gain1):
    """ADVICE (round 2, high): mlp0's two column blocks write ONE hidden image with ONE bound per row.  The scale of a row must
    be the same in both blocks whatever their weight norms: rows whose bound sweeps across powers of two (bound x wnorm[0] and
    bound x wnorm[1] on opposite sides of one for many of them), block 1 with a larger / smaller norm than block 0."""
    rows, C = 2048, 432
    torch.manual_seed(11)
    scale = 2.0 ** (torch.linspace(-3, 9, rows, device=DEV))[:, None]           # row bounds from 2^-3 to 2^9, 170 rows per octave
    x = torch.randn(rows, C, device=DEV) * scale
    m = torch.randn(rows, C, device=DEV) * scale * 0.3
    ximg, xb = lib.planes_from_f32(x)
    mimg, mb = lib.planes_from_f32(m)
    W1 = torch.randn(2 * C, 2 * C, device=DEV) / (2 * C) ** 0.5
    W1[C:] *= gain1
    hid_img, chk_h = image_like(rows, 2 * C)
    hid_b, chk_hb = guarded((rows,), torch.float32, DEV, fill=0)
    lib.linear_planes(rows, C, 2, ximg, xb, C, lib.pack_weight_planes(W1, 2, C), lib.PL_PLANES, a1=mimg, b1=mb, k1=C, out_image=hid_img,
                      out_image_k=2 * C, out_bound=hid_b, relu=True)
    chk_h(); chk_hb()
    refh = torch.relu(torch.cat([x.double(), m.double()], 1) @ W1.double().t())
    got = lib.planes_to_f32(hid_img, hid_b, rows, 2 * C).double()
    row_err = (got - refh).abs().amax(1) / refh.abs().amax(1)                  # per ROW: a wrong power of two in one block is O(1) here
    assert float(row_err.max()) < 4e-6, (float(row_err.max()), int(row_err.argmax()))
    assert bool((hid_b.double() >= refh.abs().amax(1)).all())                   # the stored bound IS an upper bound for both blocks
    # and the consumer (mlp2 + LayerNorm) reads every row with the right scale
    W2 = torch.randn(C, 2 * C, device=DEV) / (2 * C) ** 0.5
    g1, b1 = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    o32, chk_o = guarded((rows, C), torch.float32, DEV, fill=float("nan"))
    lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, lib.pack_weight_planes(W2, 1, C), lib.PL_LN, out=o32, ldo=C, gamma=g1, beta=b1,
                      lnb=lib.ln_bound(g1, b1))
    chk_o()
    refo = torch.nn.functional.layer_norm(refh @ W2.double().t(), (C,), g1.double(), b1.double())
    assert float(((o32.double() - refo).abs().amax(1) / refo.abs().amax(1)).max()) < 1e-5


@pytest.mark.parametrize("row_block", [None, 0, 2], indirect=True)
@pytest.mark.parametrize("rows,C", [(1000, 256), (129, 256), (4500, 256), (300, 128), (700, 432)])
def test_vision3d_layer_chain_against_float64(rows, C, row_block):
    """The GEMMs of one vision3d TransformerLayer (Diff-Reg-2d3d/vision3d/layers/transformer.py:58-158, 188-301) through the plane ops'
    round-4 epilogue modes: q | k | v with BIASES and no rotary (DR_PL_PLANES into three images, plus an fp32 copy), the output projection
    with bias and z = LayerNorm(linear(h) + x) (DR_PL_LN, ln_postadd), expand with bias + ReLU (two column blocks), squeeze with bias and
    out = LayerNorm(z + squeeze(...)).  C = 256 / 128 run the 256-column geometry (pgemm_kernel<4,4>), C = 432 the 448-column one."""
    torch.manual_seed(rows + C)
    x = torch.randn(rows, C, device=DEV) * (torch.rand(rows, 1, device=DEV) * 5 + 0.01)
    img, bnd = lib.planes_from_f32(x)
    # ---- q | k | v with biases -> three images (each its own bound array) + fp32 copy
    W = torch.randn(3 * C, C, device=DEV) / C ** 0.5
    b = torch.randn(3 * C, device=DEV) * 0.3
    big, chk_big = image_like(rows, 3 * C)
    bb, chk_bb = guarded((1, rows), torch.float32, DEV, fill=0)
    o32, chk_o = guarded((3, rows, C), torch.float32, DEV, fill=float("nan"))
    pk = lib.pack_weight_planes(W, 3, C)
    bmax = torch.empty(3, device=DEV)
    lib.check(lib.raw().dr_bias_max_f32(3, C, lib.ptr(b), lib.ptr(bmax), lib.stream_of(b)))
    assert torch.allclose(bmax.cpu(), b.view(3, C).abs().amax(1).cpu() * 1.0001, rtol=1e-6)
    lib.linear_planes(rows, C, 3, img, bnd, C, pk, lib.PL_PLANES, out=o32, ldo=C, blk_stride=rows * C, out_image=big, out_image_k=3 * C,
                      out_bound=bb[0], bias=b)
    # (one image of 3 C columns here: blocks side by side; the loop's per-block images are covered by the loop tests)
    chk_big(); chk_bb(); chk_o()
    ref = x.double() @ W.double().t() + b.double()
    assert not torch.isnan(o32).any()
    assert rel(o32.permute(1, 0, 2).reshape(rows, 3 * C), ref) < 2e-6
    back = lib.planes_to_f32(big, bb[0], rows, 3 * C)
    assert rel(back, ref) < 3e-6 and bool((bb[0].double() >= ref.abs().amax(1)).all())
    # ---- z = LayerNorm(h W_lin^T + b_lin + x)
    h = torch.randn(rows, C, device=DEV) * 2
    himg, hb = lib.planes_from_f32(h)
    Wl = torch.randn(C, C, device=DEV) / C ** 0.5
    bl = torch.randn(C, device=DEV) * 0.2
    g1, b1 = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    lnb = lib.ln_bound(g1, b1)
    z32, chk_z = guarded((rows, C), torch.float32, DEV, fill=float("nan"))
    z_img, chk_zi = image_like(rows, C)
    z_b, chk_zb = guarded((rows,), torch.float32, DEV, fill=0)
    lib.linear_planes(rows, C, 1, himg, hb, C, lib.pack_weight_planes(Wl, 1, C), lib.PL_LN, out=z32, ldo=C, out_image=z_img, out_image_k=C,
                      out_bound=z_b, gamma=g1, beta=b1, resid=x, ldr=C, bound_resid=bnd, lnb=lnb, bias=bl, ln_postadd=True)
    chk_z(); chk_zi(); chk_zb()
    refz = torch.nn.functional.layer_norm(h.double() @ Wl.double().t() + bl.double() + x.double(), (C,), g1.double(), b1.double())
    assert rel(z32, refz) < 3e-6 and rel(lib.planes_to_f32(z_img, z_b, rows, C), refz) < 3e-6
    assert bool((z_b.double() >= refz.abs().amax(1)).all())
    # ---- hidden = relu(expand(z)): two column blocks of C, ONE image of 2 C columns
    We = torch.randn(2 * C, C, device=DEV) / C ** 0.5
    be = torch.randn(2 * C, device=DEV) * 0.2
    hid_img, chk_h = image_like(rows, 2 * C)
    hid_b, chk_hb = guarded((rows,), torch.float32, DEV, fill=0)
    lib.linear_planes(rows, C, 2, z_img, z_b, C, lib.pack_weight_planes(We, 2, C), lib.PL_PLANES, out_image=hid_img, out_image_k=2 * C,
                      out_bound=hid_b, relu=True, bias=be)
    chk_h(); chk_hb()
    refh = torch.relu(refz @ We.double().t() + be.double())
    assert rel(lib.planes_to_f32(hid_img, hid_b, rows, 2 * C), refh) < 4e-6 and bool((hid_b.double() >= refh.abs().amax(1)).all())
    # ---- out = LayerNorm(z + squeeze(hidden))
    Ws = torch.randn(C, 2 * C, device=DEV) / (2 * C) ** 0.5
    bs = torch.randn(C, device=DEV) * 0.2
    o2, chk_o2 = guarded((rows, C), torch.float32, DEV, fill=float("nan"))
    o_img, chk_oi = image_like(rows, C)
    o_b, _ = guarded((rows,), torch.float32, DEV, fill=0)
    lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, lib.pack_weight_planes(Ws, 1, C), lib.PL_LN, out=o2, ldo=C, out_image=o_img,
                      out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=z32, ldr=C, bound_resid=z_b, lnb=lnb, bias=bs, ln_postadd=True)
    chk_o2(); chk_oi()
    refo = torch.nn.functional.layer_norm(refz + refh @ Ws.double().t() + bs.double(), (C,), g1.double(), b1.double())
    assert rel(o2, refo) < 4e-6 and rel(lib.planes_to_f32(o_img, o_b, rows, C), refo) < 4e-6
    assert bool((o_b.double() >= refo.abs().amax(1)).all())


@pytest.mark.parametrize("row_block", [0], indirect=True)
@pytest.mark.parametrize("m16", ["1", "0"])
@pytest.mark.parametrize("rows", [300, 1, 129])
@pytest.mark.parametrize("k0,k1", [(432, 0), (448, 0), (432, 432), (448, 432), (432, 448), (864, 0), (80, 432), (64, 0)])
def test_chunk_pairs_and_virtual_chunks(k0, k1, m16, row_block, rows):
    """The 128-row plane GEMM contracts k-chunks in PAIRS (v_mfma_f32_16x16x32_f16, DR_PG_M16=1: the default): segments with an even and an odd
    number of 16-deep chunks, alone and as [A0 | A1] with different row scales (the odd ones end in a virtual zero chunk; the accumulators are
    rescaled at the segment boundary), a segment shorter than five chunks (falls back to the 32x32x16 loop), both loops against float64 --
    through all three epilogues (fp32 rows, plane image + ReLU, LayerNorm + residual)."""
    import os
    os.environ["DR_PG_M16"] = m16
    try:
        C = 432
        torch.manual_seed(k0 + 7 * k1 + rows)
        x0 = torch.randn(rows, k0, device=DEV) * (torch.rand(rows, 1, device=DEV) * 5 + 0.01)
        x1 = torch.randn(rows, k1, device=DEV) * (torch.rand(rows, 1, device=DEV) * 300 + 1e-3) if k1 else None
        i0, b0 = lib.planes_from_f32(x0)
        i1, b1 = lib.planes_from_f32(x1) if k1 else (None, None)
        X = torch.cat([x0, x1], 1).double() if k1 else x0.double()
        K = k0 + k1
        W = torch.randn(C, K, device=DEV) / K ** 0.5
        pk = lib.pack_weight_planes(W, 1, C)
        ref = X @ W.double().t()
        out, chk = guarded((rows, C), torch.float32, DEV, fill=float("nan"))
        lib.linear_planes(rows, C, 1, i0, b0, k0, pk, lib.PL_F32, a1=i1, b1=b1, k1=k1, out=out, ldo=C)
        chk()
        assert not torch.isnan(out).any() and rel(out, ref) < 2e-6
        img, chk_i = image_like(rows, C)
        bnd, chk_b = guarded((rows,), torch.float32, DEV, fill=0)
        lib.linear_planes(rows, C, 1, i0, b0, k0, pk, lib.PL_PLANES, a1=i1, b1=b1, k1=k1, out_image=img, out_image_k=C, out_bound=bnd, relu=True)
        chk_i(); chk_b()
        assert rel(lib.planes_to_f32(img, bnd, rows, C), torch.relu(ref)) < 3e-6
        g1, be = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
        res = torch.randn(rows, C, device=DEV)
        rb = res.abs().amax(1)
        o32, chk_o = guarded((rows, C), torch.float32, DEV, fill=float("nan"))
        lib.linear_planes(rows, C, 1, i0, b0, k0, pk, lib.PL_LN, a1=i1, b1=b1, k1=k1, out=o32, ldo=C, out_image=img, out_image_k=C, out_bound=bnd,
                          gamma=g1, beta=be, resid=res, ldr=C, bound_resid=rb, lnb=lib.ln_bound(g1, be))
        chk_o(); chk_i()
        want = res.double() + torch.nn.functional.layer_norm(ref, (C,), g1.double(), be.double())
        assert rel(o32, want) < 3e-6 and rel(lib.planes_to_f32(img, bnd, rows, C), want) < 3e-6
    finally:
        os.environ.pop("DR_PG_M16", None)


@pytest.mark.parametrize("rows", [1, 129, 300, 1000])
@pytest.mark.parametrize("C", [528, 576, 320, 464])
def test_wide_wave_kernel_against_float64_and_the_block_layout(rows, C):
    """The 128 x 288 wide-wave geometry (pgemm16w_kernel; weights packed by dr_pack_weight_planes_wide_f32, dr_planes_linear.weight_layout = WIDE) on
    the launches it serves: q | k | v with the rotary code into three fp32 blocks (DR_PL_F32; the per-block head-padded images of the loop are held
    to the reference by the 4DMatch loop fixtures), mlp0 on a two-segment operand [x | msg] + bias + ReLU into one image shared by two column blocks.  Against float64 products,
    and against the block-layout kernel's images entry for entry (the same arithmetic up to the fp32 summation order of a k-chunk pair)."""
    torch.manual_seed(rows + C)
    x = torch.randn(rows, C, device=DEV) * (torch.rand(rows, 1, device=DEV) * 5 + 0.01)
    img, bnd = lib.planes_from_f32(x)
    W = torch.randn(3 * C, C, device=DEV) / C ** 0.5
    ang = torch.rand(rows, C // 2, device=DEV) * 6.28
    cosT, sinT = ang.cos().contiguous(), ang.sin().contiguous()
    ref = x.double() @ W.double().t()

    def rot(z):
        e, o = z[:, 0::2], z[:, 1::2]
        return torch.stack([e * cosT.double() - o * sinT.double(), o * cosT.double() + e * sinT.double()], -1).reshape(z.shape)
    want = torch.stack([rot(ref[:, :C]), rot(ref[:, C:2 * C]), ref[:, 2 * C:]]) * 0.5
    outs = {}
    for wide in (False, True):
        pk = lib.pack_weight_planes(W, 3, C, wide=wide)
        out, chk = guarded((3, rows, C), torch.float32, DEV, fill=float("nan"))
        lib.linear_planes(rows, C, 3, img, bnd, C, pk, lib.PL_F32, out=out, ldo=C, blk_stride=rows * C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C, scale=0.5,
                          wide=wide)
        chk()
        assert not torch.isnan(out).any() and rel(out, want) < 2e-6, wide
        outs[wide] = out
    assert rel(outs[True], outs[False].double()) < 1e-6
    # ---- mlp0: [x | msg] (two segments with different row scales) + bias + ReLU -> ONE image of 2 C columns written by two column blocks
    msg = torch.randn(rows, C, device=DEV) * 0.03
    mimg, mb = lib.planes_from_f32(msg)
    W0 = torch.randn(2 * C, 2 * C, device=DEV) / (2 * C) ** 0.5
    bias = torch.randn(2 * C, device=DEV) * 0.2
    want0 = torch.relu(torch.cat([x, msg], 1).double() @ W0.double().t() + bias.double())
    hs = {}
    for wide in (False, True):
        h_img, chk_i = image_like(rows, 2 * C)
        h_b, chk_b = guarded((rows,), torch.float32, DEV, fill=0)
        o32, chk_o = guarded((rows, 2 * C), torch.float32, DEV, fill=float("nan"))
        lib.linear_planes(rows, C, 2, img, bnd, C, lib.pack_weight_planes(W0, 2, C, wide=wide), lib.PL_PLANES, a1=mimg, b1=mb, k1=C, out_image=h_img,
                          out_image_k=2 * C, out_bound=h_b, relu=True, bias=bias, out=o32, ldo=2 * C, blk_stride=C, wide=wide)
        chk_i(); chk_b(); chk_o()
        back = lib.planes_to_f32(h_img, h_b, rows, 2 * C)
        assert rel(back, want0) < 2e-6 and rel(o32, want0) < 2e-6, wide
        hs[wide] = (back, h_b.clone())
    assert torch.equal(hs[True][1], hs[False][1])               # the same bounds: both kernels scale a row of one image alike
    assert rel(hs[True][0], hs[False][0].double()) < 1e-6


@pytest.mark.parametrize("rows", [4096, 4000, 130, 8192])
@pytest.mark.parametrize("C,K", [(432, 432), (432, 864), (528, 528), (528, 1024), (256, 512)])      # (dr_planes_from_f32 takes K <= 1 024)
def test_k_split_layernorm_launch_against_float64_and_the_unsplit_launch(rows, C, K):
    """DR_PL_LN with a split workspace (ABI 0.2.2): a launch of 64-row workgroups that fills at most half the chip gives every row block to TWO
    workgroups, each half of k; they swap half of their partial sums and each finish half of the LayerNorm rows.  Row counts off the 64- and
    128-row grid, merge-like (K = C) and mlp2-like (K = 2 C) shapes of all three geometries: float64 accuracy as the unsplit launch, results
    close to it but not bit-equal where the split ran (another summation order), the status word clean, nothing written outside the outputs."""
    torch.manual_seed(rows + C + K)
    h = torch.randn(rows, K, device=DEV) * (torch.rand(rows, 1, device=DEV) * 3 + 0.05)
    x = torch.randn(rows, C, device=DEV)
    himg, hb = lib.planes_from_f32(h)
    _, xb = lib.planes_from_f32(x)
    W = torch.randn(C, K, device=DEV) / K ** 0.5
    g1, b1 = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    lnb, pk = lib.ln_bound(g1, b1), lib.pack_weight_planes(W, 1, C)
    ws = lib.plane_split_workspace(C, DEV)
    res = {}
    for tag, w in (("unsplit", None), ("split", ws)):
        o32, chk_o = guarded((rows, C), torch.float32, DEV, fill=float("nan"))
        o_img, chk_oi = image_like(rows, C)
        o_b, _ = guarded((rows,), torch.float32, DEV, fill=0)
        lib.linear_planes(rows, C, 1, himg, hb, K, pk, lib.PL_LN, out=o32, ldo=C, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1,
                          resid=x, ldr=C, bound_resid=xb, lnb=lnb, split_ws=w)
        chk_o(); chk_oi()
        res[tag] = (o32.clone(), lib.planes_to_f32(o_img, o_b, rows, C))
    lib.plane_split_status(ws)
    ref = x.double() + torch.nn.functional.layer_norm(h.double() @ W.double().t(), (C,), g1.double(), b1.double())
    for tag in res:
        assert rel(res[tag][0], ref) < 2e-6 and rel(res[tag][1], ref) < 2e-6, tag
    split_ran = 2 * ((rows + 63) // 64) <= 256 and K // 16 >= 8
    same = torch.equal(res["split"][0], res["unsplit"][0])
    assert same != split_ran or K // 16 < 12, (same, split_ran)          # (the launcher's own minimum of chunks per half decides the short shapes)


@pytest.mark.parametrize("rows", [4096, 3900, 200])
@pytest.mark.parametrize("two_segments", [True, False])
def test_k_split_wide_wave_launch_against_float64_and_the_unsplit_launch(rows, two_segments):
    """the wide-wave kernel's split (launches of at most half a chip of 128 x 288 tiles): a two-segment operand [x | msg] split BY SEGMENT with the
    first half rescaled to the second's row scale, a one-segment operand split at an even chunk; DR_PL_PLANES with ReLU into an image (mlp0) and
    DR_PL_F32 with rotary (the head / a lone q projection)."""
    C = 528
    torch.manual_seed(rows + (7 if two_segments else 0))
    x = torch.randn(rows, C, device=DEV) * (torch.rand(rows, 1, device=DEV) * 4 + 0.02)
    m = torch.randn(rows, C, device=DEV) * (torch.rand(rows, 1, device=DEV) * 0.3 + 0.01)          # (another scale per row than x's)
    ximg, xb = lib.planes_from_f32(x)
    mimg, mb = lib.planes_from_f32(m)
    ws = lib.plane_split_workspace(C, DEV)
    if two_segments:
        W = torch.randn(2 * C, 2 * C, device=DEV) / (2 * C) ** 0.5
        pk = lib.pack_weight_planes(W, 2, C, wide=True)
        ref = torch.relu(torch.cat([x, m], 1).double() @ W.double().t())
        out = {}
        for tag, w in (("unsplit", None), ("split", ws)):
            img, chk = image_like(rows, 2 * C)
            bnd, _ = guarded((rows,), torch.float32, DEV, fill=0)
            lib.linear_planes(rows, C, 2, ximg, xb, C, pk, lib.PL_PLANES, a1=mimg, b1=mb, k1=C, out_image=img, out_image_k=2 * C, out_bound=bnd, relu=True,
                              wide=True, split_ws=w)
            chk()
            out[tag] = lib.planes_to_f32(img, bnd, rows, 2 * C)
            assert bool((bnd.double() >= ref.abs().amax(1)).all())
    else:
        W = torch.randn(C, C, device=DEV) / C ** 0.5
        pk = lib.pack_weight_planes(W, 1, C, wide=True)
        ang = torch.rand(rows, C // 2, device=DEV) * 6.28
        cos_t, sin_t = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
        y = x.double() @ W.double().t()
        sw = torch.stack([-y[:, 1::2], y[:, 0::2]], -1).reshape(rows, C)
        ref = (y * torch.repeat_interleave(cos_t.double(), 2, 1) + sw * torch.repeat_interleave(sin_t.double(), 2, 1)) * 0.25
        out = {}
        for tag, w in (("unsplit", None), ("split", ws)):
            o32, chk = guarded((rows, C), torch.float32, DEV, fill=float("nan"))
            lib.linear_planes(rows, C, 1, ximg, xb, C, pk, lib.PL_F32, out=o32, ldo=C, cos_t=cos_t, sin_t=sin_t, rot_mask=1, rot_C=C, scale=0.25,
                              wide=True, split_ws=w)
            chk()
            out[tag] = o32.clone()
    lib.plane_split_status(ws)
    for tag in out:
        assert rel(out[tag], ref) < 3e-6, tag
    assert not torch.equal(out["split"], out["unsplit"])                 # <= 128 tiles: the split ran
    assert (out["split"].double() - out["unsplit"].double()).abs().max().item() <= 4e-6 * float(ref.abs().max())

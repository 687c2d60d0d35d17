"""CPU: the collate oracle (oracle/collate_oracle.py) against the reference's own C++ compiled from the sources where they lie
(oracle/Makefile -> oracle/_ref/libref_collate.so; built by __graft_entry__.build() where /root/reference exists, shipped to the
GPU box with the snapshot)."""
import numpy as np
import pytest

from diffreg_hip import synth
from oracle import collate_oracle as co

needs_ref = pytest.mark.skipif(co.ref_lib() is None, reason="oracle/_ref not built (python __graft_entry__.py build with /root/reference)")


def scene(n_src, n_tgt, seed, jitter=0.004):
    """two stacked surface-like clouds; the jitter breaks the exact distance ties of the generator's lattice"""
    b = synth.make_kpfcn_batch(n_src=n_src, n_tgt=n_tgt, seed=seed)
    P = b["points"][0] + (jitter * synth.hash_normal(seed, 91, b["points"][0].shape)).astype(np.float32)
    return np.ascontiguousarray(P, np.float32), np.array(b["stack_lengths"][0], np.int32)


def rows_equal_up_to_ties(a, b, Q, S):
    """two neighbour matrices agree: same width, per row the same distance sequence and the same index set (the order inside a
    group of EQUAL float32 distances is unspecified by the reference's std::sort)"""
    assert a.shape == b.shape
    Sx = np.vstack([S, np.full((1, 3), 1e6, np.float32)])
    for k in np.nonzero((a != b).any(1))[0]:
        da = ((Q[k] - Sx[a[k]]) ** 2).astype(np.float32)
        db = ((Q[k] - Sx[b[k]]) ** 2).astype(np.float32)
        da = (da[:, 0] + da[:, 1]) + da[:, 2]
        db = (db[:, 0] + db[:, 1]) + db[:, 2]
        assert np.array_equal(da, db) and sorted(a[k]) == sorted(b[k]), k
    return True


@needs_ref
@pytest.mark.parametrize("n_src,n_tgt,seed,dl", [(1400, 1200, 0, 0.05), (3000, 2500, 1, 0.05), (3000, 2500, 1, 0.1), (700, 900, 2, 0.2)])
def test_grid_subsample_oracle_vs_reference_cpp(n_src, n_tgt, seed, dl):
    P, L = scene(n_src, n_tgt, seed)
    rp, rl = co.ref_subsample_batch(P, L, dl)
    op, ol = co.grid_subsample_batch(P, L, dl)
    assert np.array_equal(rl, ol)
    perm = co.canonical_order(rp, rl, P, L, dl)
    assert np.array_equal(rp[perm], op)                       # bit-exact barycentres, up to the unordered_map's order


@needs_ref
@pytest.mark.parametrize("n_src,n_tgt,seed,dl", [(1400, 1200, 0, 0.05), (3000, 2500, 1, 0.1)])
def test_radius_neighbors_oracle_vs_reference_cpp(n_src, n_tgt, seed, dl):
    P, L = scene(n_src, n_tgt, seed)
    sp, sl = co.grid_subsample_batch(P, L, dl)
    for (Q, ql, S, sl_, r) in ((P, L, P, L, 1.25 * dl), (sp, sl, P, L, 1.25 * dl), (P, L, sp, sl, 2.5 * dl)):
        rn = co.ref_batch_query(Q, S, ql, sl_, r)
        on = co.radius_neighbors_batch(Q, S, ql, sl_, r)
        assert rows_equal_up_to_ties(rn, on, Q, S)


def test_subsample_properties():
    P, L = scene(1400, 1200, 3)
    sp, sl = co.grid_subsample_batch(P, L, 0.07)
    assert sl.sum() == len(sp) and (sl > 0).all() and len(sp) < len(P)
    # idempotent up to float rounding: every barycentre lies in its own voxel, one point per voxel
    sp2, sl2 = co.grid_subsample_batch(sp, sl, 1e-4)
    assert np.array_equal(sl2, sl)


def _reference_levels(architecture):
    """The level rule of the reference's collate loop (3D/datasets/dataloader.py:134-211) restated as the descriptors encoder_levels
    returns: walk the blocks, a level is emitted at a density-changing block, or at a convolution that is the last block or is
    followed by an 'upsample'; 'global' / 'upsample' end the walk."""
    out, blocks = [], []
    for i, b in enumerate(architecture):
        if "global" in b or "upsample" in b:
            break
        if not ("pool" in b or "strided" in b):
            blocks.append(b)
            if i < len(architecture) - 1 and "upsample" not in architecture[i + 1]:
                continue
        down = "pool" in b or "strided" in b
        wide = any("deformable" in x for x in blocks[:-1]) if blocks else None
        out.append((wide, down, down and "deformable" in b))
        blocks = []
    return out


def test_encoder_levels_follow_the_reference_rule():
    from diffreg_hip.collate import encoder_levels
    archs = [
        ["simple", "resnetb", "resnetb_strided", "resnetb", "resnetb", "resnetb_strided", "resnetb", "resnetb", "resnetb_strided",
         "resnetb", "resnetb", "nearest_upsample", "unary", "nearest_upsample", "unary", "nearest_upsample", "unary"],   # 3D/configs/models.py:3-21
        ["simple", "resnetb_deformable", "resnetb", "resnetb_deformable_strided", "resnetb", "nearest_upsample", "unary"],
        ["simple", "resnetb_strided", "max_pool", "resnetb", "resnetb"],                      # two density changes in a row; ends on convs
        ["simple", "resnetb", "resnetb_strided", "resnetb", "global_average", "unary"],       # convs in front of 'global' are dropped
        ["simple", "resnetb"],
        ["resnetb_strided"],
    ]
    for a in archs:
        assert encoder_levels(a) == _reference_levels(a), a
    assert len(encoder_levels(archs[0])) == 4 and len(encoder_levels(archs[3])) == 1

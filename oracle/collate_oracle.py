"""CPU restatement of the reference's collate-time native code: batched grid subsampling and batched radius
neighbours.  TEST INFRASTRUCTURE ONLY (tests/, smoke, the cpu_baseline leg of tools/bench_collate.py).

Citations: CW/ = /root/reference/Diff-Reg-3dmatch/cpp_wrappers/.
  grid_subsample_batch   CW/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:4-211, CW/cpp_utils/cloud/cloud.h (float32 PointXYZ)
  radius_neighbors_batch CW/cpp_neighbors/neighbors/neighbors.cpp:210-333 (nanoflann radiusSearch, sorted, L2_Simple_Adaptor)
Pinned against the reference itself: oracle/Makefile compiles those sources where they lie into oracle/_ref/libref_collate.so
(`ref_lib()` below loads it); tests/test_collate_oracle.py holds this restatement to it bit for bit.

Order of the subsampled points: the reference emits them in the iteration order of a std::unordered_map (whatever libstdc++'s
bucket layout gives); every consumer is permutation-equivariant.  The restatement, like the HIP kernels, emits them per cloud in
ascending (iz, iy, ix) voxel order = ascending map key; `canonical_order` brings the reference's output to that order.
"""
import ctypes
import os

import numpy as np

F = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))


def voxel_coords(points, dl):
    """integer voxel coordinates of one cloud, float32 arithmetic of grid_subsampling.cpp:24-27, 52-55"""
    dl = F(dl)
    mn = points.min(0)
    origin = np.floor(mn * (F(1) / dl)) * dl
    return np.floor((points - origin) / dl).astype(np.int64)


def voxel_rank_key(ijk):
    """a key whose ascending order is ascending (iz, iy, ix)"""
    i = ijk - ijk.min(0)
    n = i.max(0) + 1
    return i[:, 0] + n[0] * (i[:, 1] + n[1] * i[:, 2])


def grid_subsample_batch(points, lengths, dl):
    """points [n,3] float32 (stacked clouds), lengths [B] -> (sub_points [m,3] float32, sub_lengths [B] int32): barycentre of
    every occupied voxel, summed in float32 in input order and multiplied by float(1.0 / count) (grid_subsampling.cpp:88)"""
    out, lens, s = [], [], 0
    for L in lengths:
        P = np.ascontiguousarray(points[s:s + L], dtype=F)
        s += L
        if L == 0:                                           # (the reference reads points[0] of an empty vector here)
            lens.append(0)
            continue
        key = voxel_rank_key(voxel_coords(P, dl))
        order = np.argsort(key, kind="stable")               # groups voxels, keeps the input order inside a voxel
        ks = key[order]
        heads = np.nonzero(np.r_[True, ks[1:] != ks[:-1]])[0]
        ends = np.r_[heads[1:], len(ks)]
        sub = np.empty((len(heads), 3), F)
        for v, (a, b) in enumerate(zip(heads, ends)):
            acc = np.zeros(3, F)
            for i in order[a:b]:
                acc = acc + P[i]                               # float32 +=, input order
            sub[v] = acc * F(1.0 / (b - a))
        out.append(sub)
        lens.append(len(heads))
    return np.concatenate(out) if out else np.zeros((0, 3), F), np.array(lens, np.int32)


def canonical_order(sub_points, sub_lengths, points, lengths, dl):
    """permutation that brings a subsampling of (points, lengths) -- e.g. the reference's, in unordered_map order -- to the
    ascending-voxel order of grid_subsample_batch.  A barycentre lies in its own voxel, so its voxel is recomputed from the
    ORIGINAL cloud's origin."""
    perm, s, t = [], 0, 0
    for L, Ls in zip(lengths, sub_lengths):
        P = np.ascontiguousarray(points[s:s + L], dtype=F)
        dlf = F(dl)
        origin = np.floor(P.min(0) * (F(1) / dlf)) * dlf
        ijk = np.floor((sub_points[t:t + Ls] - origin) / dlf).astype(np.int64)
        ref = voxel_coords(P, dl)
        lo, n = ref.min(0), ref.max(0) - ref.min(0) + 1
        i = ijk - lo
        perm.append(t + np.argsort(i[:, 0] + n[0] * (i[:, 1] + n[1] * i[:, 2]), kind="stable"))
        s += L
        t += Ls
    return np.concatenate(perm) if perm else np.zeros(0, np.int64)


def radius_neighbors_batch(queries, supports, q_lengths, s_lengths, radius):
    """-> int32 [nq, max_count]: for every query the supports of ITS cloud with d2 < r2, ascending d2 (ties: ascending index; the
    reference's std::sort leaves ties unspecified), as indices into the stacked supports, padded with len(supports).
    d2 = ((dx*dx + dy*dy) + dz*dz) in float32 (nanoflann L2_Simple_Adaptor), r2 = float32(radius)^2."""
    r2 = F(radius) * F(radius)
    rows, qs, ss = [], 0, 0
    for Lq, Ls in zip(q_lengths, s_lengths):
        Q = np.ascontiguousarray(queries[qs:qs + Lq], dtype=F)
        S = np.ascontiguousarray(supports[ss:ss + Ls], dtype=F)
        for c0 in range(0, Lq, 2048):
            d = Q[c0:c0 + 2048, None, :] - S[None, :, :]
            d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
            for k in range(d2.shape[0]):
                idx = np.nonzero(d2[k] < r2)[0]
                rows.append(ss + idx[np.lexsort((idx, d2[k][idx]))])
        qs += Lq
        ss += Ls
    width = max((len(r) for r in rows), default=0)
    out = np.full((len(rows), width), len(supports), np.int32)
    for k, r in enumerate(rows):
        out[k, :len(r)] = r
    return out


# ------------------------------------------------------------------------------------------
# the reference itself (oracle/_ref, built by oracle/Makefile from the sources under /root/reference)
# ------------------------------------------------------------------------------------------
def ref_lib():
    path = os.path.join(HERE, "_ref", "libref_collate.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.ref_subsample_batch.restype = ci
    lib.ref_subsample_batch.argtypes = [vp, ci, vp, ci, cf, ci, vp, vp]
    lib.ref_batch_query.restype = ci
    lib.ref_batch_query.argtypes = [vp, ci, vp, ci, vp, vp, ci, cf, vp, ctypes.c_long]
    return lib


def ref_subsample_batch(points, lengths, dl, max_p=0):
    lib = ref_lib()
    P = np.ascontiguousarray(points, F)
    L = np.ascontiguousarray(lengths, np.int32)
    out = np.empty_like(P)
    ol = np.empty(len(L), np.int32)
    m = lib.ref_subsample_batch(P.ctypes.data, len(P), L.ctypes.data, len(L), float(dl), int(max_p), out.ctypes.data, ol.ctypes.data)
    return out[:m].copy(), ol


def ref_batch_query(queries, supports, q_lengths, s_lengths, radius):
    lib = ref_lib()
    Q, S = np.ascontiguousarray(queries, F), np.ascontiguousarray(supports, F)
    ql, sl = np.ascontiguousarray(q_lengths, np.int32), np.ascontiguousarray(s_lengths, np.int32)
    w = lib.ref_batch_query(Q.ctypes.data, len(Q), S.ctypes.data, len(S), ql.ctypes.data, sl.ctypes.data, len(ql), float(radius), None, 0)
    out = np.empty((len(Q), w), np.int32)
    lib.ref_batch_query(Q.ctypes.data, len(Q), S.ctypes.data, len(S), ql.ctypes.data, sl.ctypes.data, len(ql), float(radius),
                        out.ctypes.data, out.size)
    return out

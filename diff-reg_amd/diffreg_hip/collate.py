"""Host-side mirror of the reference's collate-time index construction on the HIP kernels.

Same names and argument meaning as the helpers of `3D/datasets/dataloader.py:13-68` -- which call the C++ extensions
cpp_subsampling / cpp_neighbors on the data-loader's CPU workers -- and the level loop of `collate_fn_3dmatch`
(`dataloader.py:120-211`) that builds `points / neighbors / pools / upsamples / stack_lengths` for the KPFCN backbone.
Everything runs on the device through libdiffreg_hip.so (dr_grid_subsample_f32, dr_radius_neighbors_f32); there is no CPU path.
The reference-shaped helpers synchronise once to cut their result to its data-dependent shape, as the reference's return values
require; `build_kpfcn_inputs` does so once per level.
"""
import torch

from . import lib


def _check(status):
    if int(status.item()) != 0:
        raise RuntimeError("collate: a cloud spans more than 65533 cells on an axis (unsupported extent / cell size)")


def batch_grid_subsampling_kpconv(points, batches_len, features=None, labels=None, sampleDl=0.1, max_p=0, verbose=0,
                                  random_grid_orient=True):
    """-> (s_points [m,3] float32, s_len [B] int32); points only (the collate never passes features / labels), max_p = 0.
    Points come per cloud in ascending (iz, iy, ix) voxel order (the reference: std::unordered_map order)."""
    if features is not None or labels is not None or max_p != 0:
        raise NotImplementedError("device grid subsampling: points only, max_p = 0 (what collate_fn_3dmatch uses)")
    out, ol, tot, status = lib.grid_subsample(points, batches_len, sampleDl)
    m = int(tot.item())
    _check(status)
    return out[:m], ol


def batch_neighbors_kpconv(queries, supports, q_batches, s_batches, radius, max_neighbors):
    """-> int64 [nq, min(max_count, max_neighbors)] neighbour indices into the stacked supports, padded with len(supports)"""
    limit = max_neighbors if max_neighbors > 0 else 64
    out, mc, status = lib.radius_neighbors(queries, supports, q_batches, s_batches, radius, limit)
    w = min(int(mc.item()), limit)
    _check(status)
    if max_neighbors <= 0 and int(mc.item()) > limit:
        raise RuntimeError("collate: %d neighbours in one ball, more than the 64 an untruncated query supports" % int(mc.item()))
    return out[:, :w]


def build_kpfcn_inputs(points, lengths, config, neighborhood_limits):
    """The level loop of collate_fn_3dmatch (dataloader.py:120-211) on device: stacked points [n,3] + lengths [2 B] ->
    dict(points, neighbors, pools, upsamples, stack_lengths) with the reference's dtypes (float32 / int64 / int32).
    config: architecture, first_subsampling_dl, conv_radius, deform_radius (attribute or item access)."""
    get = (lambda k: config[k]) if isinstance(config, dict) else (lambda k: getattr(config, k))
    arch = get("architecture")
    r_normal = get("first_subsampling_dl") * get("conv_radius")
    dev = points.device
    batched_points = points.to(torch.float32).contiguous()
    batched_lengths = lengths.to(device=dev, dtype=torch.int32)
    empty_i = lambda: torch.zeros((0, 1), dtype=torch.int64, device=dev)
    out = dict(points=[], neighbors=[], pools=[], upsamples=[], stack_lengths=[])
    layer_blocks, layer = [], 0
    for block_i, block in enumerate(arch):
        if "global" in block or "upsample" in block:
            break
        if not ("pool" in block or "strided" in block):
            layer_blocks += [block]
            if block_i < len(arch) - 1 and not ("upsample" in arch[block_i + 1]):
                continue
        if layer_blocks:
            r = r_normal * get("deform_radius") / get("conv_radius") if any("deformable" in b for b in layer_blocks[:-1]) else r_normal
            conv_i = batch_neighbors_kpconv(batched_points, batched_points, batched_lengths, batched_lengths, r, neighborhood_limits[layer])
        else:
            conv_i = empty_i()
        if "pool" in block or "strided" in block:
            dl = 2 * r_normal / get("conv_radius")
            pool_p, pool_b = batch_grid_subsampling_kpconv(batched_points, batched_lengths, sampleDl=dl)
            r = r_normal * get("deform_radius") / get("conv_radius") if "deformable" in block else r_normal
            pool_i = batch_neighbors_kpconv(pool_p, batched_points, pool_b, batched_lengths, r, neighborhood_limits[layer])
            up_i = batch_neighbors_kpconv(batched_points, pool_p, batched_lengths, pool_b, 2 * r, neighborhood_limits[layer])
        else:
            pool_i, up_i = empty_i(), empty_i()
            pool_p = torch.zeros((0, 3), dtype=torch.float32, device=dev)
            pool_b = torch.zeros((0,), dtype=torch.int32, device=dev)
        out["points"].append(batched_points)
        out["neighbors"].append(conv_i)
        out["pools"].append(pool_i)
        out["upsamples"].append(up_i)
        out["stack_lengths"].append(batched_lengths)
        batched_points, batched_lengths = pool_p, pool_b
        r_normal *= 2
        layer += 1
        layer_blocks = []
    return out


def collate_fn_device(pairs, config, neighborhood_limits, coarse_level=-2):
    """The inference half of collate_fn_3dmatch (dataloader.py:70-325) on device: `pairs` = [(src_pcd [n,3], tgt_pcd [m,3]
    [, rot [3,3], trn [3,1]]), ...] (device tensors) -> the input dict of Pipeline.forward: the KPFCN index arrays
    (build_kpfcn_inputs), `features` = ones, and the coarse-level masks / split indices of dataloader.py:213-262.
    Ground-truth products of the collate (coarse_matches, correspondences, fine-level subsampling) belong to training and
    metrics and are not built here."""
    dev = pairs[0][0].device
    pts, lens = [], []
    for p in pairs:
        pts += [p[0].to(torch.float32), p[1].to(torch.float32)]
        lens += [p[0].shape[0], p[1].shape[0]]
    points = torch.cat(pts)
    lengths = torch.tensor(lens, dtype=torch.int32, device=dev)
    d = build_kpfcn_inputs(points, lengths, config, neighborhood_limits)
    d["features"] = torch.ones(points.shape[0], 1, device=dev)                    # in_feats_dim = 1 (3DMatch: occupancy)
    cnt = d["stack_lengths"][coarse_level].view(-1, 2).to(torch.int64)            # [B, 2] points of src / tgt at the coarse level
    B = cnt.shape[0]
    smax, tmax = int(cnt[:, 0].max()), int(cnt[:, 1].max())                       # one sync: the padded extents are shapes
    ar_s, ar_t = torch.arange(smax, device=dev), torch.arange(tmax, device=dev)
    d["src_mask"] = ar_s[None, :] < cnt[:, :1]
    d["tgt_mask"] = ar_t[None, :] < cnt[:, 1:]
    start = torch.cumsum(cnt.sum(1), 0) - cnt.sum(1)                               # first coarse row of every pair
    rows = torch.arange(B, device=dev)[:, None]
    d["src_ind_coarse_split"] = (ar_s[None, :] + rows * smax)[d["src_mask"]]
    d["tgt_ind_coarse_split"] = (ar_t[None, :] + rows * tmax)[d["tgt_mask"]]
    d["src_ind_coarse"] = (ar_s[None, :] + start[:, None])[d["src_mask"]]
    d["tgt_ind_coarse"] = (ar_t[None, :] + (start + cnt[:, 0])[:, None])[d["tgt_mask"]]
    if len(pairs[0]) >= 4:
        d["batched_rot"] = torch.stack([p[2].to(torch.float32) for p in pairs])
        d["batched_trn"] = torch.stack([p[3].to(torch.float32).reshape(3, 1) for p in pairs])
    return d

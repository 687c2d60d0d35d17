"""Host-side mirror of the reference's collate-time index construction on the HIP kernels.

Same names and argument meaning as the helpers of `3D/datasets/dataloader.py:13-68` -- which call the C++ extensions
cpp_subsampling / cpp_neighbors on the data-loader's CPU workers -- and the level loop of `collate_fn_3dmatch`
(`dataloader.py:120-211`) that builds `points / neighbors / pools / upsamples / stack_lengths` for the KPFCN backbone.
Everything runs on the device through libdiffreg_hip.so (dr_grid_subsample_f32, dr_radius_neighbors_f32); there is no CPU path.
The reference-shaped helpers synchronise once to cut their result to its data-dependent shape, as the reference's return values
require; `build_kpfcn_inputs` does so once per level of `encoder_levels(architecture)`.
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


def encoder_levels(architecture):
    """The encoder half of a KPFCN architecture string list as resolution levels: [(wide_conv, down, wide_down), ...].
    A level is the run of convolution blocks at one point density; it ends at the block that changes the density (a 'pool' or
    'strided' block: `down` = True) or where the decoder starts ('upsample', or the end of the list; convolutions in front of a
    'global' block are dropped, as the reference's loop drops them).
    wide_conv: the level's convolutions search the deformable radius -- the case when a block BEFORE the level's last convolution
    is deformable (the rule of 3D/datasets/dataloader.py:141-146); wide_down: the density-changing block itself is deformable.
    A level without convolution blocks (two density changes in a row) has wide_conv = None: it gets no neighbour matrix."""
    levels, convs = [], []
    for name in architecture:
        if "global" in name:
            # the reference's loop `continue`s over the convolutions in front of a 'global' block (their successor is not an
            # 'upsample') and then breaks on it: that trailing level is never emitted (dataloader.py:137-146)
            convs = []
            break
        if "upsample" in name:
            break
        if "pool" in name or "strided" in name:
            levels.append((any("deformable" in c for c in convs[:-1]) if convs else None, True, "deformable" in name))
            convs = []
        else:
            convs.append(name)
    if convs:
        levels.append((any("deformable" in c for c in convs[:-1]), False, False))
    return levels


def build_kpfcn_inputs(points, lengths, config, neighborhood_limits):
    """KPFCN index arrays on device (what the level loop of collate_fn_3dmatch builds, dataloader.py:120-211): stacked points
    [n,3] + lengths [2 B] -> dict(points, neighbors, pools, upsamples, stack_lengths) with the reference's dtypes (float32 /
    int64 / int32).  Per level L of encoder_levels(): cell size dl0 2^L, ball radius dl0 conv_radius 2^L (deform_radius in its
    place where the level searches wide); neighbours of the level's cloud in itself; when the level ends in a density change, the
    next cloud = grid subsampling at twice the cell size, `pools` = the next cloud's neighbours in this one, `upsamples` = this
    cloud's neighbours in the next one at twice the radius.
    config: architecture, first_subsampling_dl, conv_radius, deform_radius (attribute or item access)."""
    get = (lambda k: config[k]) if isinstance(config, dict) else (lambda k: getattr(config, k))
    dl0, k_conv, k_deform = get("first_subsampling_dl"), get("conv_radius"), get("deform_radius")
    dev = points.device
    cloud = points.to(torch.float32).contiguous()
    lens = lengths.to(device=dev, dtype=torch.int32)
    no_index = lambda: torch.zeros((0, 1), dtype=torch.int64, device=dev)
    out = dict(points=[], neighbors=[], pools=[], upsamples=[], stack_lengths=[])
    for L, (wide_conv, down, wide_down) in enumerate(encoder_levels(get("architecture"))):
        cell = dl0 * 2 ** L                       # (the reference doubles a running radius: the same float64 values)
        radius = lambda wide: cell * k_conv * k_deform / k_conv if wide else cell * k_conv
        limit = neighborhood_limits[L]
        out["points"].append(cloud)
        out["stack_lengths"].append(lens)
        out["neighbors"].append(no_index() if wide_conv is None else batch_neighbors_kpconv(cloud, cloud, lens, lens, radius(wide_conv), limit))
        if not down:
            out["pools"].append(no_index())
            out["upsamples"].append(no_index())
            cloud, lens = torch.zeros((0, 3), dtype=torch.float32, device=dev), torch.zeros((0,), dtype=torch.int32, device=dev)
            continue
        coarse, coarse_lens = batch_grid_subsampling_kpconv(cloud, lens, sampleDl=2 * cell * k_conv / k_conv)
        r = radius(wide_down)
        out["pools"].append(batch_neighbors_kpconv(coarse, cloud, coarse_lens, lens, r, limit))
        out["upsamples"].append(batch_neighbors_kpconv(cloud, coarse, lens, coarse_lens, 2 * r, limit))
        cloud, lens = coarse, coarse_lens
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

"""Sharding of independent scene pairs over ranks (one process per GPU) and the end-of-run metric gather.

Pairs are independent units (reference inference is B = 1, 3D/lib/tester.py:115): rank r takes the pairs
r, r + W, r + 2W, ...; there is no data-path collective.  The only communication is one all_reduce(SUM) of a
small float64 vector [sum_IR, sum_FMR, sum_RR, n_pairs, ...] at the end, mirroring
Diff-Reg-2d3d/vision3d/utils/distributed.py:57-64 -- RCCL over xGMI on GPUs ("nccl" backend), gloo in the CPU tests."""
import torch
import torch.distributed as dist


def shard_pairs(n_pairs, rank, world):
    """indices of the pairs owned by `rank` (round-robin, sizes differ by at most one)."""
    return list(range(rank, n_pairs, world))


def _group_active():
    """a process group exists: the collectives below then always run through it -- also with ONE rank (`bench.py --dist-single-rank`,
    tests/test_rccl_gpu.py: the only way a box with one GPU can push this module through RCCL at all)"""
    return dist.is_available() and dist.is_initialized()


def gather_metrics(local_sums, device=None):
    """all_reduce(SUM) of a 1-D float64 vector of per-rank sums; returns the global sums (on every rank)."""
    v = torch.as_tensor(local_sums, dtype=torch.float64)
    if device is not None:
        v = v.to(device)
    if _group_active():
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return v


def max_over_ranks(x, device=None):
    v = torch.tensor([float(x)], dtype=torch.float64)
    if device is not None:
        v = v.to(device)
    if _group_active():
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
    return float(v.item())


def gather_per_rank(x, device=None):
    """all_gather of one float per rank -> list (every rank gets the whole list)."""
    v = torch.tensor([float(x)], dtype=torch.float64)
    if device is not None:
        v = v.to(device)
    if _group_active():
        out = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(out, v)
        return [float(o.item()) for o in out]
    return [float(v.item())]


METRIC_NAMES = ("sum_inlier_ratio", "sum_fmr", "sum_registration_recall", "n_pairs", "sum_seconds")


def metric_vector(ir, fmr, rr_ok, seconds):
    """[sum IR, sum FMR, sum RR, n_pairs, sum t] of this rank's pairs (3D/lib/tester.py:73-118 accumulates the same
    per-pair quantities on one process): 1-D tensors / sequences of equal length -> float64 vector on the inputs' device."""
    ir = torch.as_tensor(ir, dtype=torch.float64)
    dev = ir.device
    fmr = torch.as_tensor(fmr, dtype=torch.float64, device=dev)
    rr = torch.as_tensor(rr_ok, dtype=torch.float64, device=dev)
    return torch.stack([ir.sum(), fmr.sum(), rr.sum(), torch.tensor(float(ir.numel()), dtype=torch.float64, device=dev),
                        torch.tensor(float(seconds), dtype=torch.float64, device=dev)])


def reduce_metrics(local_vec):
    """the one collective of a run: all_reduce(SUM) of the metric vector (RCCL over xGMI on GPUs, gloo on CPU) -> dict with
    the global means (IR, FMR, RR) and totals."""
    g = gather_metrics(local_vec, local_vec.device if torch.is_tensor(local_vec) else None)
    n = max(float(g[3]), 1.0)
    d = {k: float(v) for k, v in zip(METRIC_NAMES, g.tolist())}
    d.update(mean_inlier_ratio=float(g[0]) / n, fmr=float(g[1]) / n, registration_recall=float(g[2]) / n)
    return d

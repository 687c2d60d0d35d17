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


def gather_metrics(local_sums, device=None):
    """all_reduce(SUM) of a 1-D float64 vector of per-rank sums; returns the global sums (on every rank)."""
    v = torch.as_tensor(local_sums, dtype=torch.float64)
    if device is not None:
        v = v.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return v


def max_over_ranks(x, device=None):
    v = torch.tensor([float(x)], dtype=torch.float64)
    if device is not None:
        v = v.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
    return float(v.item())

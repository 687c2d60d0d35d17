"""RCCL runs once (mirrors Diff-Reg-2d3d/vision3d/utils/distributed.py:11-77): the metric gather of the multi-GPU path --
all_reduce(SUM) of [sum IR, sum FMR, sum RR, n_pairs, sum t], the max over ranks of the bench contract, the per-rank gather -- pushed
through a ONE-rank "nccl" process group on cuda:0, in-process (no exec, no child process), and bench.py's own rank code path with the
nccl branch forced (`--dist-single-rank`).  This is NOT a scaling measurement: no 8-GPU node is available to this pool, the N > 1
collectives are otherwise covered by the world-size-2 gloo tests (tests/test_shard_cpu.py).  Needs a GPU."""
import json
import os
import sys

import pytest
import torch
import torch.distributed as dist

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def one_rank_nccl():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29800 + os.getpid() % 150)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield
    if dist.is_initialized():
        dist.destroy_process_group()


def test_metric_gather_through_one_rank_rccl(one_rank_nccl):
    from diffreg_hip import shard
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    ir = torch.tensor([0.5, 0.25, 0.0, 0.75], dtype=torch.float64, device=DEV)
    vec = shard.metric_vector(ir, (ir > 0.05).double(), torch.tensor([1.0, 0.0, 0.0, 1.0], device=DEV), 2.5)
    assert vec.is_cuda
    red = shard.reduce_metrics(vec)                      # all_reduce(SUM) on the device through RCCL
    assert red["n_pairs"] == 4.0 and red["mean_inlier_ratio"] == pytest.approx(0.375) and red["fmr"] == pytest.approx(0.75)
    assert red["registration_recall"] == pytest.approx(0.5) and red["sum_seconds"] == pytest.approx(2.5)
    assert shard.max_over_ranks(1.25, torch.device(DEV)) == 1.25                 # all_reduce(MAX)
    assert shard.gather_per_rank(7.0, torch.device(DEV)) == [7.0]               # all_gather
    s = shard.gather_metrics([1.0, 2.0, 3.0], torch.device(DEV))
    assert s.is_cuda and s.tolist() == [1.0, 2.0, 3.0]
    # a barrier and a larger all_reduce on device memory, checked against the input (1 rank: the sum is the tensor itself)
    dist.barrier()
    big = torch.arange(1 << 20, dtype=torch.float32, device=DEV)
    ref = big.clone()
    dist.all_reduce(big)
    torch.cuda.synchronize()
    assert torch.equal(big, ref)


def test_bench_rank_path_with_the_nccl_branch_forced(capsys, monkeypatch):
    """bench.py --gpus 1 --dist-single-rank: init_process_group("nccl"), the barriers around the timed region, max over ranks, the
    metric-vector all_reduce and the per-rank gathers all run through RCCL on cuda:0 (a small pass: 8 pairs, one stream)"""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("WORLD_SIZE", "1"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", str(29950 + os.getpid() % 40))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "8", "--streams", "1",
                                      "--dist-single-rank", "--no-cpu-baseline", "--no-other-configs", "--no-single-pair", "--no-breakdown"])
    bench.main()
    assert not dist.is_initialized()                     # main() tore the group down
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["metric_gather"]["backend"] == "nccl (RCCL)" and r["n_gpus"] == 1
    assert r["metric_gather"]["n_pairs"] == 8.0 and r["metric_gather"]["per_rank_pairs"] == [8]
    assert 0.0 <= r["metric_gather"]["mean_inlier_ratio"] <= 1.0 and r["value"] > 0
    # the line is the short summary the driver can read; the full record is in the file it names
    assert len(line) < bench.MAX_LINE_BYTES and r["details"] and os.path.exists(os.path.join(ROOT, r["details"]))
    assert all(k in r for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data", "config"))


def test_bench_default_shape_line_is_short_and_carries_roofline_and_cpu_baseline(capsys, monkeypatch):
    """a small run with EVERY block of the default run switched on except the other configurations (32 pairs, 2 timed passes): the printed line
    stays under the limit and carries `roofline` (HIP-event-timed, with PMC traffic from profiles/) and `cpu_baseline`"""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("WORLD_SIZE", "1"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "32", "--no-other-configs"])
    monkeypatch.setattr(bench, "cpu_baseline", lambda *a, **k: bench.__dict__["_cpu_baseline_orig"](*a, budget_s=4.0, **k))
    bench.main()
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert len(line) < bench.MAX_LINE_BYTES
    rf = r["roofline"]
    assert all(k in rf for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_us_per_launch", "work_per_launch"))
    assert rf["bound"] in ("mfma", "hbm") and 0 < rf["frac"] < 1
    assert all(k in r["cpu_baseline"] for k in ("value", "unit", "cores", "kind", "sample")) and r["cpu_baseline"]["kind"] == "port"
    assert r["sinkhorn_roofline"]["frac"] > 0.5 and r["single_pair"]["ms_per_pair"] > 0
    assert r["parity_ok"] is True and r["ir_fmr_parity"]["pairs_within_1e4"] == r["ir_fmr_parity"]["pairs"]

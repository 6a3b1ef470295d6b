"""Multi-GPU layout of the hot path: signing sessions (and every batched op) are independent units, so a job of
`total` sessions is sharded across ranks with NO data-path collective (SURVEY.md §8e mode A); torch.distributed is
used only to synchronise the timed region and to agree on the slowest rank's time.  Backend "nccl" is RCCL on ROCm;
the same code runs under "gloo" on CPU (tests/test_dist_cpu.py, world_size 2)."""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous, balanced shard [lo, hi) of `total` independent units for `rank` of `world`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(value, device="cpu"):
    """Largest `value` (e.g. elapsed seconds) over all ranks: the job finishes when its slowest rank does."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(count, device="cpu"):
    """Per-rank unit counts gathered on every rank (used to report whole-job throughput)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(count)]
    t = torch.tensor([int(count)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]

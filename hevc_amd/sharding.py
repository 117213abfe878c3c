"""Stream sharding across GPUs: independent streams, one process per GPU, no data-path collective.

Stream i goes to rank i mod world.  The only communication is bookkeeping for the benchmark: a barrier and
the maximum elapsed time over ranks (torch.distributed: RCCL on GPUs, gloo in CPU tests)."""


def streams_for_rank(n_streams, rank, world):
    """Indices of the streams rank `rank` reconstructs."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    return list(range(rank, n_streams, world))


def max_over_ranks(value, dist=None, device="cpu"):
    """Maximum of a Python float over all ranks (identity when torch.distributed is not initialised)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device="cpu"):
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())

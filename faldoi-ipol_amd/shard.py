"""Sharding of independent image pairs over ranks (one process per GPU) and the timing
reduction bench.py reports.  Pairs never exchange data during a solve, so
the only collectives are the barrier around the timed region, a MAX over per-rank wall times and
a SUM over per-rank work -- all on tiny tensors through torch.distributed (RCCL on GPUs, gloo in
the CPU tests)."""


def assign_pairs(npairs, world, rank):
    """Static round-robin: pair i -> rank i % world (reference-free; pairs are independent)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return list(range(rank, npairs, world))


def reduce_timing(local_seconds, local_units, dist=None, device=None):
    """Returns (max seconds over ranks, total units over ranks)."""
    if dist is None or not dist.is_initialized():
        return float(local_seconds), float(local_units)
    import torch
    t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_scalars(local_value, dist=None, device=None):
    """Every rank's scalar, in rank order, on every rank (one all_gather of a 1-element tensor)."""
    if dist is None or not dist.is_initialized():
        return [float(local_value)]
    import torch
    t = torch.tensor([float(local_value)], dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]

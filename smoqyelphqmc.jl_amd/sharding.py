"""Walker-parallel sharding across the GPUs of one node.

The hot path partitions over independent Monte Carlo walkers exactly like the reference's
one-MPI-rank-per-walker model (tutorials/holstein_honeycomb_mpi.jl:60-72): rank r owns walkers
[r*wpg, (r+1)*wpg), no state is ever exchanged, so there is no data-path collective.  The only
communication is the bench harness's barrier and the max-over-ranks of the wall time.
"""
from __future__ import annotations


def walker_range(rank: int, world_size: int, walkers_per_gpu: int) -> range:
    if not (0 <= rank < world_size) or walkers_per_gpu < 1:
        raise ValueError("bad rank / world size / walkers per GPU")
    return range(rank * walkers_per_gpu, (rank + 1) * walkers_per_gpu)


def reduce_max_time(elapsed: float, device=None) -> float:
    """MAX over ranks of a wall time (identity when torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(units_per_rank: float, world_size: int, elapsed_max: float) -> float:
    """Whole-job throughput: units processed by all ranks divided by the slowest rank's time."""
    return units_per_rank * world_size / elapsed_max

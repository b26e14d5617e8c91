"""Multi-GPU = one process per GPU, each owning a contiguous block of env indices.

Envs are independent, so there is no data-path collective: rank r of `world` simply builds the batch for
global env indices [lo, hi) with seeds base_seed + index.  A given env's trajectory therefore does not depend
on the number of GPUs — tests/test_sharding.py checks exactly that with 2 gloo ranks."""


def shard_range(n_envs_total, rank, world):
    """Contiguous, balanced partition of range(n_envs_total)."""
    per, rem = divmod(n_envs_total, world)
    lo = rank * per + min(rank, rem)
    hi = lo + per + (1 if rank < rem else 0)
    return lo, hi


def shard_seeds(base_seed, n_envs_total, rank, world):
    lo, hi = shard_range(n_envs_total, rank, world)
    return [base_seed + i for i in range(lo, hi)]

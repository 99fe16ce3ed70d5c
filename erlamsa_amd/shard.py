"""Case-range sharding for multi-GPU runs.

A case's output is a pure function of (parent seed, case number, its input, config), so ranks
never exchange data on the mutation path.  Rank r of W takes a contiguous range of case numbers,
the same shape as erlamsa_main:get_threading_mode/3 (reference src/erlamsa_main.erl:95-108) gives
its `--workers`, except that results do not depend on W (parity is defined against workers=1).
"""


def case_range(n_cases, rank, world):
    """Contiguous split of cases 0..n-1: the first n % world ranks take one extra case.
    Returns (first, count)."""
    base, rem = divmod(n_cases, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def weak_first_case(step, rank, world, cases_per_step):
    """bench.py's weak-scaling numbering: every (step, rank) pair gets a fresh block of case
    numbers of the same fuzzer/1 run; 1-based first case number."""
    return (step * world + rank) * cases_per_step + 1


def broadcast_corpus(arena, offsets, src=0):
    """RCCL (or gloo) broadcast of the packed seed arena + offset table from `src`."""
    import torch.distributed as dist
    dist.broadcast(arena, src=src)
    dist.broadcast(offsets, src=src)
    return arena, offsets

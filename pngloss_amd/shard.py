"""Image-level sharding of a batch over the GPUs of one node.

The path shards by WHOLE IMAGES only: rows, channels and tiles of one image are coupled through the running symbol
histogram and the winner-of-row dependence, images are independent.  So the data path needs no
collective at all; RCCL (torch.distributed, backend "nccl") is used only to agree on the static split and to gather
the small per-image result records.  The reference has no equivalent (its file loop, pngloss.c:173, is sequential).
"""
from typing import List, Sequence


def lpt_partition(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first greedy split of items (cost ~ width*height) into `world` shards.
    Deterministic: ties broken by item index, shards by rank.  Returns the item indices of each rank, ascending."""
    shards: List[List[int]] = [[] for _ in range(world)]
    load = [0.0] * world
    for i in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += costs[i]
    return [sorted(s) for s in shards]


def contiguous_partition(n: int, world: int) -> List[List[int]]:
    """Equal-size frames (BASELINE.json configs[3]): frame i -> rank i // ceil(n/world)."""
    per = (n + world - 1) // world
    return [list(range(r * per, min(n, (r + 1) * per))) for r in range(world)]


def gather_records(local_records: list, group=None) -> list:
    """all_gather of per-image result records (plain picklable objects) -> list ordered by image index.
    Each record must carry its global image index under key 'index'.  Works on gloo (CPU tests) and nccl/RCCL."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return sorted(local_records, key=lambda r: r["index"])
    world = dist.get_world_size(group)
    out = [None] * world
    dist.all_gather_object(out, local_records, group=group)
    flat = [r for part in out for r in part]
    return sorted(flat, key=lambda r: r["index"])

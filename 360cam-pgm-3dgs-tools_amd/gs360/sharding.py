"""Frame sharding across ranks/devices.  Units are independent (frame, view) jobs with no exchange step
(reference PC:830-836, DF:2776), so sharding is a pure index calculation and needs no collective."""
from typing import List, Sequence


def frames_for_rank(n_frames: int, world: int, rank: int) -> List[int]:
    """Round-robin frame indices of `rank`; every frame lands on exactly one rank."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return list(range(rank, n_frames, world))


def shard_jobs(jobs: Sequence, world: int, rank: int, key=lambda job: job[1]) -> List:
    """Keep the jobs whose SOURCE (key) belongs to `rank`; all views of one source stay on one rank."""
    order = []
    for job in jobs:
        k = key(job)
        if k not in order:
            order.append(k)
    mine = {order[i] for i in frames_for_rank(len(order), world, rank)}
    return [job for job in jobs if key(job) in mine]

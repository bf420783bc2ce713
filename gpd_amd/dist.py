"""Multi-GPU sharding of independent clouds: cloud i -> rank i mod G, no
data-path collective; torch.distributed only carries the timing barrier / reductions and
the host-side gather of per-cloud results."""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def clouds_of_rank(num_clouds, rank, world):
    """Static round-robin assignment used for BASELINE.json configs[4] (256 clouds over 8 GPUs)."""
    return list(range(rank, num_clouds, world))


def reduce_timing(dist, elapsed, units, device="cpu"):
    """(max over ranks of elapsed, sum over ranks of units).  dist=None: single process."""
    if dist is None:
        return float(elapsed), float(units)
    import torch
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t[0]), float(u[0])


def gather_in_cloud_order(dist, num_clouds, rank, world, local_results):
    """Host-side concatenation of per-cloud results in cloud order (the only 'collective' of
    the path).  local_results: list aligned with clouds_of_rank()."""
    if dist is None:
        return list(local_results)
    gathered = [None] * world
    dist.all_gather_object(gathered, list(local_results))
    out = [None] * num_clouds
    for r in range(world):
        for k, c in enumerate(clouds_of_rank(num_clouds, r, world)):
            out[c] = gathered[r][k]
    return out

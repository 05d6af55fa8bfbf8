"""Batch split over the GPUs of one node: every scene is an independent NLLS problem
(reference src/optimizer.cpp:241-381 builds one ceres::Problem per call), so the path shards with NO data-path
collective. One process per GPU; scenes are regenerated per rank from (seed, scene_id). torch.distributed
(RCCL on GPUs, gloo in CPU tests) is used only for the bench's barrier / max-time / a few summary scalars and the
all_gather of the optimised parameters for the result check on rank 0."""
import os
from typing import Tuple


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Static contiguous split: rank g owns scene ids [g*total/world, (g+1)*total/world)."""
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    return lo, hi


def weak_shard(per_gpu: int, rank: int) -> Tuple[int, int]:
    """Weak scaling: every rank owns `per_gpu` scenes; scene ids are globally unique."""
    return rank * per_gpu, (rank + 1) * per_gpu


def reduce_summary(local: dict, device=None) -> dict:
    """All-reduce the bench summary: 'max_*' keys with MAX, everything else with SUM. No-op for world_size 1."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(local)
    out = {}
    for k in sorted(local):
        t = torch.tensor([float(local[k])], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if k.startswith("max_") else dist.ReduceOp.SUM)
        out[k] = float(t.item())
    return out


def gather_params(params):
    """the optimised parameter blocks of every rank on every rank, [world][B][P] (all_gather over RCCL /
    gloo; 8 x 8192 x 6 doubles = 3.1 MB at BASELINE configs[3]). World size 1: the tensor itself with a leading axis."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return params.unsqueeze(0)
    world = dist.get_world_size()
    out = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(out, params.contiguous())
    return torch.stack(out, dim=0)

"""Process-group helpers over torch.distributed (backend "nccl" IS RCCL on ROCm; gloo on CPU).
Mirrors the names of d2z:utils/comm.py that the reference scripts and modules call."""
import functools
import pickle

import torch
import torch.distributed as dist


def _ok():
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if _ok() else 1


def get_rank() -> int:
    return dist.get_rank() if _ok() else 0


def get_local_rank() -> int:
    import os
    return int(os.environ.get("LOCAL_RANK", 0)) if _ok() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def all_gather(data, group=None):
    if get_world_size() == 1:
        return [data]
    out = [None] * get_world_size()
    dist.all_gather_object(out, data, group=group)
    return out


def gather(data, dst=0, group=None):
    if get_world_size() == 1:
        return [data]
    out = [None] * get_world_size() if get_rank() == dst else None
    dist.gather_object(data, out, dst=dst, group=group)
    return out if get_rank() == dst else []


def shared_random_seed():
    import numpy as np
    return all_gather(int(np.random.randint(2 ** 31)))[0]


def reduce_dict(input_dict, average=True):
    ws = get_world_size()
    if ws < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= ws
        return dict(zip(names, values))


# ---- data-parallel eval helpers: images are independent, so ranks share nothing on the data path ----------
def shard_range(n_items: int, rank: int = None, world_size: int = None):
    """Contiguous shard [begin, end) of n_items for this rank (d2z:data/samplers/distributed_sampler.py:191-194, InferenceSampler)."""
    rank = get_rank() if rank is None else rank
    world_size = get_world_size() if world_size is None else world_size
    shard = (n_items - 1) // world_size + 1 if n_items > 0 else 0
    begin = shard * rank
    return min(begin, n_items), min(begin + shard, n_items)


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (elapsed time): RCCL when the group is nccl, gloo on CPU."""
    if get_world_size() == 1:
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if get_world_size() == 1:
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())

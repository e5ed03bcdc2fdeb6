"""Multi-GPU plumbing of the path: one process per GPU, windows sharded by rank, NO data-path
collective (images are sharded by file, CallConsensusInterface.py:138-145; ranks never exchange
data).  The only communication is a barrier and a max-reduce of the elapsed time for reporting."""
import os


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None):
    """Initialise the default process group if WORLD_SIZE > 1; returns torch.distributed or None.
    backend None -> 'nccl' (= RCCL on ROCm) when CUDA is available, else 'gloo'."""
    rank, local_rank, world = env_world()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value, device="cpu"):
    """max of a python float over all ranks (the timing rule of bench.py)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shard_for_rank(items, rank, world):
    """Round-robin shard of `items` for `rank` (same rule as file sharding)."""
    return [it for i, it in enumerate(items) if i % world == rank]

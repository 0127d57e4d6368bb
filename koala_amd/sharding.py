"""
Multi-GPU host logic: streams are independent, so a job of N streams is cut into contiguous
per-rank ranges, one process and one engine per GPU, and nothing crosses GPUs on the data path.  The only
collective is the final reduction of {frames processed (sum), elapsed seconds (max)} for the aggregate rate.
"""

from typing import Optional, Tuple


def shard_range(num_streams: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of the streams owned by `rank`; sizes differ by at most one."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    base, extra = divmod(num_streams, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def aggregate_throughput(frames_local: int, elapsed_local: float, device: Optional[str] = None) -> Tuple[int, float]:
    """
    All-reduce of the per-rank counters: returns (total frames over all ranks, slowest rank's elapsed seconds).
    Uses the default torch.distributed group (backend `nccl` = RCCL over xGMI on GPUs, `gloo` on CPU); a
    run without a process group passes through unchanged.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return int(frames_local), float(elapsed_local)
    # (a one-rank group still goes through the collective: `torchrun --nproc-per-node 1` on a single-GPU box is how the RCCL
    # branch -- communicator creation on the device, all-reduce of device tensors -- gets executed at all)
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    frames = torch.tensor([frames_local], dtype=torch.int64, device=dev)
    elapsed = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
    dist.all_reduce(frames, op=dist.ReduceOp.SUM)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    return int(frames.item()), float(elapsed.item())


__all__ = ["shard_range", "aggregate_throughput"]

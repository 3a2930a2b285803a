"""Frame sharding for batched mode: frames are independent, so N ranks = N independent shards and the
only communication is one end-of-run reduction of {frames done, checksum, max elapsed}.

Reference: the `-numthreads` fan-out of VkResample.cpp:1959-1969; thread t of T processes files
f*T + t + 1 for f < numLocalFiles (VkResample.cpp:1622-1629).  One rank (= one GPU) plays the role of one
reference thread."""
import math


def local_frame_count(num_files, num_threads, thread_id):
    """numLocalFiles of VkResample.cpp:1622-1625."""
    n = math.ceil(num_files / float(num_threads))
    if (n - 1) * num_threads + thread_id > num_files - 1:
        n -= 1
    return max(n, 0)


def frames_for_rank(num_files, world, rank):
    """0-based frame indices of this rank, in processing order (the reference's file numbers are these + 1)."""
    return [f * world + rank for f in range(local_frame_count(num_files, world, rank))]


def reduce_summary(dist, frames_done, checksum, elapsed_s, device=None):
    """All-reduce {sum frames, sum checksum, max elapsed} over the job.  `dist` is torch.distributed (or None
    for a single process).  Returns python scalars; exact for checksums < 2**53."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(frames_done), int(checksum), float(elapsed_s)
    import torch
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    s = torch.tensor([float(frames_done), float(checksum)], dtype=torch.float64, device=dev)
    m = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(round(s[0].item())), int(round(s[1].item())), float(m[0].item())

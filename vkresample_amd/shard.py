"""Frame sharding for batched mode: frames are independent, so N ranks = N independent shards and the
only communication is one end-of-run reduction of {frames done, checksum, max elapsed}.

Reference: the `-numthreads` fan-out of VkResample.cpp:1959-1969; thread t of T processes files
f*T + t + 1 for f < numLocalFiles (VkResample.cpp:1622-1629).  One rank (= one GPU) plays the role of one
reference thread."""
import math
import threading


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
    for a single process).  Returns python scalars; the sums are 64-bit integer sums (exact while the job's total stays below
    2**63: 2**11 ranks with checksums below 2**52)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(frames_done), int(checksum), float(elapsed_s)
    import torch
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    s = torch.tensor([int(frames_done), int(checksum)], dtype=torch.int64, device=dev)
    m = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(s[0].item()), int(s[1].item()), float(m[0].item())


class FrameQueue:
    """Shared frame counter of a job: the dynamic alternative to the stripe (north_star's "work-queue"; the CLI's -workqueue
    does the same between the host threads of one process).  Ranks claim `chunk` consecutive frame numbers at a time from ONE
    counter in torch.distributed's key-value store (the rendezvous TCPStore: `add` is atomic) -- no data-path collective, no
    GPU traffic; the next claim is fetched by a helper thread while the current chunk runs, so the store's round trip
    (~0.1 ms) hides behind the chunk's GPU time (8 frames of 2048x1024 = 0.5 ms).  A rank whose GPU runs faster (the pool's
    boards differ by +- 4 % at the power limit) simply claims more chunks; the stripe would leave it idle at the end.
    One queue object per job step: `key` must be new for every step (all ranks use the same sequence of keys)."""

    def __init__(self, store, total, chunk=8, key="fftup/frames"):
        self.store, self.total, self.chunk, self.key = store, int(total), max(1, int(chunk)), key
        self._next = None
        self._thread = None
        self._prefetch()

    @staticmethod
    def default_store(dist):
        """the process group's own store (torchrun's TCPStore), prefixed so the keys cannot collide with c10d's"""
        import torch.distributed.distributed_c10d as c10d
        return dist.PrefixStore("fftup_queue", c10d._get_default_store())

    def _claim(self):
        end = self.store.add(self.key, self.chunk)          # atomic fetch-and-add on the store's server (rank 0's process)
        start = end - self.chunk
        self._next = (start, min(end, self.total)) if start < self.total else None

    def _prefetch(self):
        self._thread = threading.Thread(target=self._claim, daemon=True)
        self._thread.start()

    def claim(self):
        """(first, end) of the next chunk of frame numbers for this rank, or None when the job is handed out"""
        self._thread.join()
        got = self._next
        if got is not None:
            self._prefetch()
        return got

    def __iter__(self):
        while True:
            c = self.claim()
            if c is None:
                return
            yield c

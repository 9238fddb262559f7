"""One process per GPU: rank discovery, stream sharding and timing reduction.

The hot path shards by stream (frames of one stream are strictly sequential,
streams are independent), so there is NO data-path collective: ranks only meet at
the timing barrier and at one max-reduction of the elapsed time.  `backend` is
"nccl" (= RCCL over xGMI on ROCm) on GPUs and "gloo" in the CPU tests."""
import os


class Rank:
    def __init__(self, rank, local_rank, world, dist=None, device=None):
        self.rank, self.local_rank, self.world, self.dist, self.device = rank, local_rank, world, dist, device

    @property
    def is_root(self):
        return self.rank == 0

    def stream_seeds(self, streams_per_rank, base=0x5EED0000):
        """global stream ids owned by this rank (seed-addressed synthetic streams)"""
        first = self.rank * streams_per_rank
        return [base + first + s for s in range(streams_per_rank)]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def init(backend="nccl"):
    """reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)"""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return Rank(0, local_rank, 1)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = None
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = "cuda"
    dist.init_process_group(backend, rank=rank, world_size=world)
    return Rank(rank, local_rank, world, dist, device)

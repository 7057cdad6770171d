"""Multi-GPU plumbing of the decode path: one process per GPU, streams partitioned across ranks.

The Opus decode path has no cross-stream data flow: every stream's state lives on
exactly one GPU and a decode step never exchanges payload or PCM between ranks.  The only collectives are
the ones the benchmark contract needs -- a barrier around the timed region and a MAX over the ranks'
elapsed times -- so this module is deliberately small.  `torch.distributed` is plumbing here: backend
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os


class Ranks:
    """Rank/world bookkeeping + the two collectives the path uses."""

    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch
            import torch.distributed as dist
            backend = backend or "nccl"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group(backend="nccl", device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend=backend)
            self.dist = dist

    # ---- partition -------------------------------------------------------------------------------
    def stream_range(self, streams_per_rank):
        """Global stream ids owned by this rank: [rank * n, (rank + 1) * n) -- weak scaling."""
        return self.rank * streams_per_rank, (self.rank + 1) * streams_per_rank

    def seed_base(self, base=0x9E3779B9):
        """Per-rank seed of the synthetic payload generator (differs per global stream id)."""
        return (base ^ (self.rank * 0x01000193)) & 0xFFFFFFFF

    def owner_of(self, global_stream, streams_per_rank):
        return global_stream // streams_per_rank

    # ---- collectives -----------------------------------------------------------------------------
    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            if self.device is not None and self.device.type == "cuda":
                import torch
                torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def aggregate_throughput(ranks, frames_local, elapsed_local):
    """Whole-job throughput the bench contract asks for: frames of ALL ranks / MAX elapsed time."""
    total = ranks.sum_over_ranks(frames_local)
    dt = ranks.max_over_ranks(elapsed_local)
    return total / dt, dt, total

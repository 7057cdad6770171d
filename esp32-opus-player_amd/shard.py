"""Multi-GPU plumbing of the decode path: one process per GPU, streams partitioned across ranks.

The Opus decode path has no cross-stream data flow: every stream's state lives on
exactly one GPU and a decode step never exchanges payload or PCM between ranks.  The collectives are the
ones the benchmark contract needs -- a barrier around the timed region and a MAX over the ranks' elapsed
times -- plus the one exchange the path does have when work arrives at a single ingest point: the
work-queue scatter (`pack_work` / `Ranks.scatter_bytes`), which hands every rank its decode steps (descriptor
tables + packet arena, made from Ogg pages by opusgpu_pages_demux) in one buffer, straight into its HBM.
`torch.distributed` is plumbing here: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os

import numpy as np


def usable_cpus():
    """CPUs this process can actually use: the smaller of the affinity mask and the cgroup CPU quota (a GPU box shows all
    of the host's logical CPUs but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2: "<quota> <period>" or "max <period>"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


# ---- work queue: one rank's decode steps in one buffer ------------------------------------------------------
# [ header: int64 x WORK_HEADER_WORDS | descriptor tables of all steps, back to back | packet arena ], every part
# starting on a 256-byte boundary.  header = magic, n_steps, total descriptors, arena bytes, then one count per step, then --
# tables grouped by mode (OPUSGPU_PAGES_GROUP_BY_MODE: SILK-only, hybrid, CELT-only frames in that order) -- per step the number
# of SILK-only and of hybrid frames (-1, -1: the step's table is not grouped), so that the receiving rank can issue a step as
# declared sub-steps (Context.decode_work_step) without reading the table back.
WORK_MAGIC = 0x4F475752
WORK_HEADER_WORDS = 1024
WORK_HEADER_BYTES = 8 * WORK_HEADER_WORDS
WORK_MAX_STEPS = (WORK_HEADER_WORDS - 4) // 3


def mode_counts(table):
    """(SILK-only, hybrid) frame counts of a step table that is grouped by mode, (-1, -1) if it is not grouped."""
    m = table["flags"] & 3
    if len(m) > 1 and (np.diff(m.astype(np.int8)) < 0).any():
        return -1, -1
    return int((m == 0).sum()), int((m == 1).sum())


def _pad256(n):
    return (n + 255) & ~255


def pack_work(batch):
    """PageBatch (host) -> uint8 buffer in the layout above."""
    n_steps = batch.n_steps
    if n_steps > WORK_MAX_STEPS:
        raise ValueError(f"{n_steps} steps do not fit the work header ({WORK_MAX_STEPS})")
    tables = [batch.step(k)[0] for k in range(n_steps)]
    counts = [len(t) for t in tables]
    total = sum(counts)
    arena = batch.arena if batch.arena is not None else np.zeros(0, np.uint8)
    desc_at = WORK_HEADER_BYTES
    arena_at = desc_at + _pad256(16 * total)
    buf = np.zeros(arena_at + _pad256(arena.size), dtype=np.uint8)
    hdr = buf[:WORK_HEADER_BYTES].view(np.int64)
    hdr[0:4] = (WORK_MAGIC, n_steps, total, arena.size)
    hdr[4:4 + n_steps] = counts
    hdr[4 + n_steps:4 + 3 * n_steps] = np.array([mode_counts(t) for t in tables], dtype=np.int64).reshape(-1)
    at = desc_at
    for t in tables:
        buf[at:at + 16 * len(t)] = t.view(np.uint8).reshape(-1)
        at += 16 * len(t)
    buf[arena_at:arena_at + arena.size] = arena
    return buf


class WorkLayout:
    """Where the parts of a packed work buffer are (byte offsets from its start)."""

    def __init__(self, header_bytes):
        hdr = np.frombuffer(bytes(header_bytes[:WORK_HEADER_BYTES]), dtype=np.int64)
        if hdr[0] != WORK_MAGIC:
            raise ValueError("not a packed work buffer")
        self.n_steps, self.total, self.arena_bytes = int(hdr[1]), int(hdr[2]), int(hdr[3])
        self.counts = [int(c) for c in hdr[4:4 + self.n_steps]]
        mc = hdr[4 + self.n_steps:4 + 3 * self.n_steps].reshape(-1, 2)
        self.mode_counts = [(int(a), int(b)) for a, b in mc]  # per step: (SILK-only, hybrid) frames, (-1, -1) = not grouped
        self.desc_at = []
        at = WORK_HEADER_BYTES
        for c in self.counts:
            self.desc_at.append(at)
            at += 16 * c
        self.arena_at = WORK_HEADER_BYTES + _pad256(16 * self.total)
        self.nbytes = self.arena_at + _pad256(self.arena_bytes)


# ---- raw pages for one rank in one buffer (the scalable ingest: the receiving rank demuxes its own share) ---------------
# [ int64: magic, n_pages, blob bytes | int32 lens[n_pages] | int32 stream_ids[n_pages] | page bytes ], parts 256-aligned
PAGES_MAGIC = 0x4F475047


def pack_pages(blob, lens, stream_ids):
    """Pages back to back in `blob` (page i is lens[i] bytes long), their decoder streams -> one uint8 buffer."""
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    n = len(lens)
    if len(ids) != n or int(lens.sum()) != blob.size:
        raise ValueError("lens / stream_ids / blob do not describe the same pages")
    lens_at = 256
    ids_at = lens_at + _pad256(4 * n)
    blob_at = ids_at + _pad256(4 * n)
    buf = np.zeros(blob_at + _pad256(blob.size), dtype=np.uint8)
    buf[:24].view(np.int64)[:] = (PAGES_MAGIC, n, blob.size)
    buf[lens_at:lens_at + 4 * n] = lens.view(np.uint8)
    buf[ids_at:ids_at + 4 * n] = ids.view(np.uint8)
    buf[blob_at:blob_at + blob.size] = blob
    return buf


def unpack_pages(buf):
    """-> (blob, offsets, lens, stream_ids) views into a pack_pages buffer (host memory)."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    magic, n, nbytes = (int(x) for x in buf[:24].view(np.int64))
    if magic != PAGES_MAGIC:
        raise ValueError("not a packed page buffer")
    lens_at = 256
    ids_at = lens_at + _pad256(4 * n)
    blob_at = ids_at + _pad256(4 * n)
    lens = buf[lens_at:lens_at + 4 * n].view(np.int32)
    ids = buf[ids_at:ids_at + 4 * n].view(np.int32)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]]) if n else np.zeros(0, np.int64)
    return buf[blob_at:blob_at + nbytes], offs, lens, ids


class Ranks:
    """Rank/world bookkeeping + the collectives the path uses."""

    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = device
        # OPUSGPU_FORCE_DIST=1: take the torch.distributed path with a single rank too (rehearsal of the N > 1 code on a
        # one-GPU box: process-group init, barrier, all-reduce and scatter over RCCL)
        if self.world > 1 or os.environ.get("OPUSGPU_FORCE_DIST") == "1":
            import torch
            import torch.distributed as dist
            backend = backend or "nccl"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                # RCCL prints its version banner (NCCL_DEBUG=VERSION) straight to stdout when the first communicator is
                # created; this process's stdout carries the result line only, so file descriptor 1 points at stderr
                # until the group and its communicator exist.
                import sys
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
                try:
                    dist.init_process_group(backend="nccl", device_id=self.device)
                    dist.barrier()
                    torch.cuda.synchronize()
                finally:
                    sys.stdout.flush()
                    os.dup2(saved, 1)
                    os.close(saved)
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

    def scatter_bytes(self, buffers=None, src=0):
        """The work-queue scatter: rank `src` passes one uint8 array per rank, every rank gets its own back -- as a numpy
        array (one rank, or a CPU backend) or as a uint8 tensor in this rank's HBM (nccl: the bytes travel GPU to GPU).
        One small collective for the sizes, then one point-to-point send per destination, each of the destination's exact
        size: on the source's device only the buffer in flight and the next one (being copied up from pinned host memory)
        exist at any time -- not one padded copy per rank, as a scatter collective of device tensors needs."""
        if self.dist is None:
            return buffers[0]
        import torch
        sizes = torch.zeros(1, dtype=torch.int64, device=self.device)
        if self.rank == src:
            if len(buffers) != self.world:
                raise ValueError("one buffer per rank")
            all_sizes = [torch.tensor([b.size], dtype=torch.int64, device=self.device) for b in buffers]
        else:
            all_sizes = None
        self.dist.scatter(sizes, all_sizes, src=src)
        n = int(sizes.item())
        on_gpu = self.device is not None and self.device.type == "cuda"

        def staged(b):  # the source's copy of one destination's buffer, where the backend can send it from
            t = torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint8))
            return t.pin_memory().to(self.device, non_blocking=True) if on_gpu else t

        if self.rank == src:
            in_flight = []
            for dst, b in enumerate(buffers):
                if dst == src or b.size == 0:
                    continue
                t = staged(b)
                in_flight.append((self.dist.isend(t, dst), t))
                if len(in_flight) > 1:  # at most two destinations' buffers on the device
                    in_flight.pop(0)[0].wait()
            for w, _ in in_flight:
                w.wait()
            recv = staged(buffers[src])
        else:
            recv = torch.empty(n, dtype=torch.uint8, device=self.device)
            if n:
                self.dist.recv(recv, src=src)
        return recv.numpy() if recv.device.type == "cpu" else recv

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

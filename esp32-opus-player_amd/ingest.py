"""Ogg pages in, PCM out, with the ingest of the next batch of pages UNDER the decode of this one (BASELINE config 5).

A batch is a set of whole pages (typically the next page of every stream).  One host thread turns batch b + 1 into decode steps
(`PageBatch`: opusgpu_pages_demux_into on the host's cores, checksums included) and queues the step tables and packet bytes for
upload on the context's copy stream (`opusgpu_upload_async`), while the calling thread has batch b's steps in flight on the decode
stream.  Memory is a ring of `depth` slots, each a page-locked host buffer the demux writes into and a device buffer of the same
layout ([step tables | arena]): one copy per batch, from memory that was made DMA-able ONCE.  (Page-locking per batch -- whether by
opusgpu_host_register on the demux's own output or by the runtime behind a copy from pageable memory -- was measured to slow the
decode kernels that run meanwhile by 20 %: mapping host memory into the GPU's address space and out of it again is not free for
the kernels in flight.)  A slot is reused once the steps that read it have finished.  The two streams meet only at events: the
decode stream waits for a batch's upload fence, the ingest thread waits (on the host) for the event behind the last step of the
batch whose slot it takes over.

Reference: what the reference does per page in `OggS` parsing + `opus_decoder` calls (src/ogg.cpp:439-480, src/opus_decoder.cpp:931)
one stream at a time on one core; here both halves are batched and the host half hides behind the device half.
"""
import ctypes as C
import queue
import threading
import time

import numpy as np

from . import PageBatch, BufferTooSmall


def _pad256(x):
    return (x + 255) // 256 * 256


class OverlappedPageDecode:
    def __init__(self, ctx, threads=1, depth=3, page_flags=None, keeps_mode=False, by_kind=False):
        """keeps_mode: no stream ever changes its mode (OPUSGPU_STEP_KEEPS_MODE): every step then carries that promise, which lets
        opusgpu_set_pipeline run its entropy kernels ahead; by_kind (with it): a step whose table is grouped by mode goes as three
        declared sub-steps instead of one step of all modes."""
        self.ctx, self.threads, self.depth = ctx, max(1, threads), max(2, depth)
        self.page_flags = page_flags
        self.keeps_mode, self.by_kind = keeps_mode, by_kind
        self.dev = [None] * self.depth   # (device address, capacity)
        self.host = [None] * self.depth  # page-locked uint8 arrays

    def reserve(self, nbytes):
        """Make every slot at least nbytes large now (page-locking and hipMalloc then stay out of the run) -- and let the copy stream
        carry its first upload and fence here: creating the stream and the first transfer out of a slot cost milliseconds once."""
        for i in range(self.depth):
            host, dev = self._slot(i, nbytes)
            base = dev.value if isinstance(dev, C.c_void_p) else int(dev)
            self.ctx.upload_async(C.c_void_p(base), host[:4096])
        e = self.ctx.event()
        self.ctx.upload_fence(e)
        self.ctx.event_synchronize(e)
        self.ctx.event_destroy(e)

    def _slot(self, i, need):
        if self.host[i] is None or self.host[i].nbytes < need:
            if self.host[i] is not None:
                self.ctx.host_unregister(self.host[i])
                self.ctx.dev_free(self.dev[i])
            cap = _pad256(need + need // 8)
            raw = np.empty(cap + 4096, dtype=np.uint8)
            off = (-raw.ctypes.data) % 4096
            self.host[i] = raw[off:off + cap]
            self.ctx.host_register(self.host[i])
            self.dev[i] = self.ctx.dev_alloc(cap)
        return self.host[i], self.dev[i]

    def close(self):
        for i in range(self.depth):
            if self.host[i] is not None:
                self.ctx.host_unregister(self.host[i])
                self.ctx.dev_free(self.dev[i])
        self.dev, self.host = [None] * self.depth, [None] * self.depth

    def run(self, batches, d_pcm, d_result, on_batch_done=None):
        """batches: iterable of (blob uint8, offsets int64, lens int32, stream_ids int32), in stream order of time.  Decodes every
        step of every batch into d_pcm / d_result (each step overwrites them, as opusgpu_decode_step_device does).
        on_batch_done(b): called on the calling thread after batch b's steps have been queued.
        -> statistics: wall time, per-batch ingest times, the time until the first batch was ready, GPU time from batch end to
        batch end."""
        ctx = self.ctx
        ready = queue.Queue()
        done_ev, done_flag = {}, {}
        err = []
        stop = threading.Event()  # set when the decode loop raises: the ingest thread leaves without touching another slot
        stats = {"ingest_s": [], "demux_s": [], "slot_wait_s": [], "pages": 0, "steps": 0}
        batches = list(batches)
        for b in range(len(batches)):
            done_ev[b], done_flag[b] = ctx.event(), threading.Event()

        def ingest():
            try:
                for b, (blob, offs, lens, sids) in enumerate(batches):
                    t0 = time.perf_counter()
                    i = b % self.depth
                    if b >= self.depth:  # the slot's previous tenant must have been decoded
                        done_flag[b - self.depth].wait()
                        if stop.is_set():  # (the decode side gave up: its flags were set to let this thread go, no event behind them)
                            return
                        ctx.event_synchronize(done_ev[b - self.depth])
                    if stop.is_set():
                        return
                    t1 = time.perf_counter()
                    kw = {} if self.page_flags is None else {"flags": self.page_flags}
                    host, dev = self._slot(i, int(np.sum(lens, dtype=np.int64)) + 32 * len(lens) + 4096)
                    try:
                        pb = PageBatch(blob, offs, lens, sids, threads=self.threads, out_mem=host, **kw)
                    except BufferTooSmall as e:  # pages of very many short packets: 16 bytes of table per packet
                        host, dev = self._slot(i, e.need)
                        pb = PageBatch(blob, offs, lens, sids, threads=self.threads, out_mem=host, **kw)
                    t2 = time.perf_counter()
                    counts, mc = [], []
                    for k in range(pb.n_steps):
                        fl = pb.step(k)[0]["flags"]
                        counts.append(len(fl))
                        if self.keeps_mode and self.by_kind:  # (per-mode counts: only the sub-step flow reads them -- and this thread
                            m = fl & 3                        #  shares the interpreter with the one that launches the decode steps)
                            grouped = len(m) < 2 or not (np.diff(m.astype(np.int8)) < 0).any()
                            mc.append((int((m == 0).sum()), int((m == 1).sum())) if grouped else (-1, -1))
                        else:
                            mc.append((-1, -1))
                    base = dev.value if isinstance(dev, C.c_void_p) else int(dev)
                    ctx.upload_async(C.c_void_p(base), pb.image)
                    fence = ctx.event()
                    ctx.upload_fence(fence)
                    good = int((pb.info["status"] > 0).sum())
                    arena_at = pb.arena_offset
                    pb.close()  # (the tables and the arena live in the slot, not in the batch object)
                    stats["slot_wait_s"].append(t1 - t0)
                    stats["demux_s"].append(t2 - t1)
                    stats["ingest_s"].append(time.perf_counter() - t1)
                    ready.put((b, base, counts, arena_at, fence, good, mc))
            except BaseException as e:  # noqa: BLE001 -- handed to the caller's thread
                err.append(e)
                ready.put(None)

        t_start = time.perf_counter()
        th = threading.Thread(target=ingest, name="opusgpu-ingest")
        th.start()
        first_wait = None
        fences = []
        try:
            for b in range(len(batches)):
                item = ready.get()
                if item is None:
                    raise err[0]
                if first_wait is None:
                    first_wait = time.perf_counter() - t_start
                _, base, counts, arena_at, fence, good, mc = item
                ctx.stream_wait_event(fence)
                fences.append(fence)
                at = 0
                for k, n in enumerate(counts):
                    if n:
                        if self.keeps_mode and self.by_kind and mc[k][0] >= 0:  # declared sub-steps by kind (Context.decode_step_by_kind)
                            ctx.decode_step_by_kind(mc[k][0], mc[k][1], n - mc[k][0] - mc[k][1], C.c_void_p(base + at),
                                                    C.c_void_p(base + arena_at), d_pcm, d_result, keeps_kind=True)
                        else:
                            ctx.decode_step_device(n, C.c_void_p(base + at), C.c_void_p(base + arena_at), d_pcm, d_result,
                                                   modes=15 if self.keeps_mode else 0)  # (15: frames of any mode + the promise)
                        stats["steps"] += 1
                    at += 16 * n
                ctx.event_record(done_ev[b])
                done_flag[b].set()
                stats["pages"] += good
                if on_batch_done:
                    on_batch_done(b)
            ctx.synchronize()
        except BaseException:
            stop.set()
            raise
        finally:
            for f in done_flag.values():
                f.set()
            th.join()
            if stop.is_set():  # (the error path: whatever was queued ends before its events go; fences the thread made but
                try:           #  never handed over are in the queue)
                    ctx.synchronize()
                except Exception:  # noqa: BLE001
                    pass
                while not ready.empty():
                    item = ready.get_nowait()
                    if item is not None and item[4] not in fences:
                        fences.append(item[4])
                for e in list(done_ev.values()) + fences:
                    ctx.event_destroy(e)
        stats["wall_s"] = time.perf_counter() - t_start
        stats["first_batch_ready_s"] = first_wait
        stats["gpu_ms_between_batch_ends"] = [ctx.event_elapsed_ms(done_ev[b - 1], done_ev[b]) for b in range(1, len(batches))]
        for e in list(done_ev.values()) + fences:
            ctx.event_destroy(e)
        return stats

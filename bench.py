#!/usr/bin/env python3
"""bench.py -- throughput of the Opus decode hot path on MI355X.

One "step" = one pass of the hot path over one batch: every stream of the batch decodes one 20 ms
frame (stream state persists in HBM from step to step).  Workload at N=1 = BASELINE.json configs[1]:
65,536 synthetic CELT-only fullband stereo streams, 160-byte LCG payloads.
All packets of all timed steps are resident in HBM before the timed region starts.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU under torch.distributed.run (RCCL).  When the driver has already started the ranks
(WORLD_SIZE in the environment) this process is one of them and WORLD_SIZE must equal --gpus; when it has not,
this process -- before it touches HIP or torch.cuda -- starts `python -m torch.distributed.run --nproc-per-node N
bench.py ...` as a CHILD process, relays rank 0's JSON line and exits with the child's status.

Prints ONE JSON line on rank 0.  `value` = frames decoded by all ranks / max-over-ranks wall time.
`roofline` prices the decode kernel against HBM peak using the ALGORITHMIC bytes per frame
(B_celt = 21,633 B); `cpu_baseline` times the CPU oracle (a bit-identical port of
the reference path) on a bounded sample of the same workload on this host's cores.
"""
import argparse
import importlib.util
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (toc, payload bytes, algorithmic bytes per frame [SURVEY 8d], default streams per GPU)
    "celt_fb_stereo_64k": (0xFC, 160, 21633, 65536),
    "silk_nb_stereo_64k": (0x0C, 40, 5953, 65536),
    "hybrid_fb_stereo_256k": (0x7C, 120, 24945, 262144),
    # BASELINE config 5: Ogg pages of 10 packets, modes 1:1:1 across streams, 2 M pages over 8 GPUs = 262,144 per GPU.
    # Rank 0 ingests (page demux on the host) and scatters every rank's decode steps (the one collective of the path).
    "mixed_pages_2m": (None, None, (5953 + 24945 + 21633) / 3.0, 262144),
}
# RFC mode (include/opusgpu.h OPUSGPU_MODE_RFC): stream s keeps TOC configuration s % 32 -- all modes, bandwidths and
# frame durations 2.5 ... 60 ms -- one stereo code-0 packet per step, 5 % of the packets lost: concealed, or (SILK-only / hybrid
# streams whose next packet arrived) recovered from that packet's forward error correction data.  Not a BASELINE config.
RFC_WORKLOAD = "rfc_mixed_64k"
RFC_LOSS = 0.05
MIX = ((0x0C, 40), (0x7C, 120), (0xFC, 160))  # mode of local stream s = s % 3: SILK-NB, hybrid FB, CELT FB
PACKETS_PER_PAGE = 10
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def load_pkg():
    name = "esp32_opus_player_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "esp32-opus-player_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_shard():
    pkg = load_pkg()
    name = pkg.__name__ + ".shard"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "esp32-opus-player_amd", "shard.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(toc, L, seconds_target=12.0):
    """Time the CPU oracle (bit-identical port of the reference decode path) on a bounded sample."""
    import oracle_py
    o = oracle_py.load()
    pkg = load_pkg()
    cores = load_shard().usable_cpus()  # the CPUs granted to this process, not the host's logical CPU count
    frames = 16
    # calibrate on one core, then size the sample to ~seconds_target of total CPU work
    pay = pkg.lcg_payloads(64, frames, L)
    t0 = time.perf_counter()
    o.batch_decode(2, toc, pay, want_pcm=False)
    per_frame = (time.perf_counter() - t0) / (64 * frames)
    streams = int(max(cores * 8, min(65536, seconds_target / per_frame / frames)))
    streams -= streams % cores
    pay = pkg.lcg_payloads(streams, frames, L)
    oks = [0] * cores
    chunk = streams // cores

    def work(t):
        _, oks[t] = o.batch_decode(2, toc, pay, s0=t * chunk, s1=(t + 1) * chunk, want_pcm=False)

    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    total = sum(oks)
    return {
        "value": total / dt, "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"{streams} streams x {frames} frames of the same workload, {cores} threads = the CPUs granted to this "
                  f"process (host shows {os.cpu_count()} logical CPUs), one oracle decoder per stream, {dt:.2f} s wall",
        "single_core_frames_per_s": 1.0 / per_frame, "host_logical_cpus": os.cpu_count(),
    }


def mixed_cpu_baseline():
    """CPU oracle on the 1:1:1 mode mix: one bounded sample per mode, combined as frames / total time."""
    parts = [cpu_baseline(toc, L, seconds_target=5.0) for toc, L in MIX]
    value = 3.0 / sum(1.0 / p["value"] for p in parts)
    return {"value": value, "unit": "frames/s", "cores": parts[0]["cores"], "kind": "port",
            "sample": "equal numbers of SILK-NB, hybrid and CELT frames: 3 / sum(1 / rate) of " +
                      " | ".join(p["sample"] for p in parts),
            "per_mode_frames_per_s": [p["value"] for p in parts], "host_logical_cpus": parts[0]["host_logical_cpus"]}


def prepare_pages_work(pkg, shard, ranks, ctx, n, frames, ingest="rank0", page_crc="host"):
    """mixed_pages workload: rank 0 builds the Ogg pages of every rank's streams (`frames` packets per stream in pages of
    PACKETS_PER_PAGE, chained) and routes them by owner.  ingest = "rank0": it also turns them into decode steps
    (opusgpu_pages_demux, host threads) and scatters the packed work -- one demux for the whole job.  ingest = "per-rank":
    it scatters the raw pages and every rank demuxes its own share on its own host CPUs -- demux capacity grows with the
    rank count; with page_crc = "gpu" the checksums of the raw pages are verified where they arrived, in HBM
    (opusgpu_pages_crc_device), and the host demux skips its CRC pass.
    Returns (device address of this rank's work, its layout, ingest statistics, keep-alive object)."""
    threads = shard.usable_cpus()
    per_rank = ingest == "per-rank"
    buffers, stats = None, None
    RAW_PAGES.clear()
    if ranks.rank == 0:
        buffers, n_pages, page_bytes, t_demux, t_gen = [], 0, 0, 0.0, 0.0
        for r in range(ranks.world):
            t_g0 = time.perf_counter()
            mats, ids = [], []
            for m, (toc, L) in enumerate(MIX):
                sid = np.arange(m, n, 3, dtype=np.int32)
                seed = (0x9E3779B9 ^ (r * 0x01000193) ^ (m * 0x5bd1e995)) & 0xFFFFFFFF
                pay = pkg.lcg_payloads(len(sid), frames, L, seed_base=seed)
                for q, lo in enumerate(range(0, frames, PACKETS_PER_PAGE)):
                    pg = pkg.build_pages(toc, pay[lo:lo + PACKETS_PER_PAGE], sid.astype(np.uint32) + r * n, seqno=2 + q)
                    mats.append((q, pg))
                    ids.append((q, sid))
            order = sorted(range(len(mats)), key=lambda i: mats[i][0])  # a stream's pages in order
            blob = np.concatenate([mats[i][1].reshape(-1) for i in order])
            lens = np.concatenate([np.full(mats[i][1].shape[0], mats[i][1].shape[1], dtype=np.int32) for i in order])
            offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
            sids = np.concatenate([ids[i][1] for i in order])
            del mats
            n_pages += len(lens)
            page_bytes += int(blob.size)
            t_gen += time.perf_counter() - t_g0  # making the synthetic pages is not ingest
            if per_rank:
                buffers.append(shard.pack_pages(blob, lens, sids))
                continue
            t0 = time.perf_counter()
            batch = pkg.PageBatch(blob, offs, lens, sids, threads=threads, flags=pkg.PAGES_VERIFY_CRC | PAGE_ORDER)
            t_demux += time.perf_counter() - t0
            if not (batch.info["status"] > 0).all():
                raise SystemExit("page demux rejected synthetic pages")
            buffers.append(shard.pack_work(batch))
            batch.close()
        stats = {"mode": ingest, "pages": n_pages, "page_bytes": page_bytes, "demux_threads": threads, "generate_s": t_gen,
                 "page_crc": page_crc if per_rank else "host"}
        if not per_rank:
            stats.update({"demux_s": t_demux, "pages_per_s": n_pages / t_demux, "demux_GB_per_s": page_bytes / t_demux / 1e9})
    t0 = time.perf_counter()
    mine = ranks.scatter_bytes(buffers, src=0)
    if ranks.dist is not None:
        import torch
        torch.cuda.synchronize()
    t_scatter = time.perf_counter() - t0
    if per_rank:
        # this rank's raw pages -> host (they may have arrived in HBM), demux here, steps -> HBM
        raw = mine if isinstance(mine, np.ndarray) else mine.cpu().numpy()
        blob, offs, lens, sids = shard.unpack_pages(raw)
        RAW_PAGES.update({"blob": blob, "offs": offs, "lens": lens, "sids": sids, "raw": raw})  # for overlapped_end_to_end()
        if page_crc == "gpu":
            import ctypes
            if isinstance(mine, np.ndarray):  # one rank, no scatter: put the raw pages where a scatter would have put them
                d_raw = ctx.dev_alloc(raw.size)
                ctx.h2d(d_raw, raw)
                d_raw_at = d_raw.value
            else:
                d_raw, d_raw_at = None, mine.data_ptr()
            d_blob = ctypes.c_void_p(d_raw_at + (blob.ctypes.data - raw.ctypes.data))
            # one page first: the checksum tables and the kernel's code are set up once per context, not per batch
            pkg.PageBatch.with_gpu_crc(ctx, d_blob, blob, offs[:1], lens[:1], sids[:1]).close()
            t0 = time.perf_counter()
            batch = pkg.PageBatch.with_gpu_crc(ctx, d_blob, blob, offs, lens, sids, threads=threads, flags=PAGE_ORDER)
            t_local = time.perf_counter() - t0
            if d_raw is not None:
                ctx.dev_free(d_raw)
        else:
            t0 = time.perf_counter()
            batch = pkg.PageBatch(blob, offs, lens, sids, threads=threads, flags=pkg.PAGES_VERIFY_CRC | PAGE_ORDER)
            t_local = time.perf_counter() - t0
        if not (batch.info["status"] > 0).all():
            raise SystemExit("page demux rejected synthetic pages")
        mine = shard.pack_work(batch)
        batch.close()
        # the job's demux rate: all pages over the slowest rank's demux time
        t_max = ranks.max_over_ranks(t_local)
        if stats is not None:
            stats.update({"demux_s": t_max, "pages_per_s": stats["pages"] / t_max, "demux_GB_per_s": stats["page_bytes"] / t_max / 1e9})
    if isinstance(mine, np.ndarray):  # one rank: host -> HBM directly
        lay = shard.WorkLayout(mine)
        base = ctx.dev_alloc(mine.size)
        ctx.h2d(base, mine)
        keep = None
    else:  # the scatter delivered into this rank's HBM: only the header comes back to the host
        lay = shard.WorkLayout(mine[:shard.WORK_HEADER_BYTES].cpu().numpy())
        base, keep = mine.data_ptr(), mine
    if stats is not None:
        stats["scatter_s"] = t_scatter if ranks.dist is not None else None
        stats["work_bytes_per_rank"] = int(lay.nbytes)
    return base, lay, stats, keep


# mixed_pages_2m: how the demux orders a step's table -- grouped by mode, and (BENCH_ORDER_BY_HEADER=0: not) inside the SILK-only and
# hybrid groups by the frames' LBRR flags (OPUSGPU_PAGES_ORDER_BY_HEADER)
PAGE_ORDER = 2 | (4 if os.environ.get("BENCH_ORDER_BY_HEADER", "1") != "0" else 0)
MIXED_FLOW = os.environ.get("BENCH_MIXED_FLOW", "keeps")  # mixed_pages_2m: keeps | substeps | inorder (see run_workload)
RAW_PAGES = {}  # this rank's share of the raw pages, as prepare_pages_work (per-rank ingest) received it


def overlapped_end_to_end(pkg, ranks, ctx, n, d_pcm, d_res, threads):
    """Config 5 end to end with the ingest UNDER the decode (esp32-opus-player_amd/ingest.py): this rank's raw pages, in batches
    of one page per stream, are demuxed (checksums verified on the host's cores) and uploaded on the copy stream by a second host
    thread while the batch before decodes.  Streams are reset first, so the last step leaves the same PCM as the timed run did.
    -> (statistics of this rank, PCM of the last step)"""
    spec = importlib.util.spec_from_file_location(pkg.__name__ + ".ingest", os.path.join(ROOT, "esp32-opus-player_amd", "ingest.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    blob, offs, lens, sids = (RAW_PAGES[k] for k in ("blob", "offs", "lens", "sids"))
    if len(lens) % n:
        raise SystemExit("raw pages are not whole batches of one page per stream")
    batches = [(blob, offs[q * n:(q + 1) * n], lens[q * n:(q + 1) * n], sids[q * n:(q + 1) * n]) for q in range(len(lens) // n)]
    split = os.environ.get("BENCH_E2E_SPLIT_FIRST", "8")
    if split != "0":
        # the first batch's ingest is the one nothing hides: it goes in growing pieces of 1/8, 1/8, 1/4, 1/2 (by page order), so that
        # decoding starts after an eighth of it and every later piece is demuxed under the decode of the one before ("1": two halves).
        # Works since the demux of a batch takes less than half its decode (round 4: 0.92 of the decode-only rate with halves, 0.94 so)
        b0 = batches[0]
        cuts = [0, n // 8, n // 4, n // 2, n] if split != "1" else [0, n // 2, n]
        batches = [(blob, b0[1][a:b], b0[2][a:b], b0[3][a:b]) for a, b in zip(cuts[:-1], cuts[1:])] + batches[1:]
    ctx.streams_reset(0, n)
    ctx.synchronize()
    pipe = mod.OverlappedPageDecode(ctx, threads=int(os.environ.get("BENCH_E2E_THREADS", threads)), depth=int(os.environ.get("BENCH_E2E_DEPTH", "3")),
                                    keeps_mode=MIXED_FLOW != "inorder", by_kind=MIXED_FLOW == "substeps", page_flags=1 | PAGE_ORDER)  # (a stream's mode is fixed in this workload)
    pipe.reserve(int(max(int(b[2].sum()) + 32 * len(b[2]) for b in batches)) + 4096)  # a service sets its slots up once, not per job
    ranks.barrier()
    st = pipe.run(batches, d_pcm, d_res)
    ranks.barrier()
    pipe.close()
    out = np.zeros((n, 960, 2), dtype=np.int16)
    ctx.d2h(out, d_pcm)
    return st, out


def run_host_path(args, ranks, pkg, ctx):
    """The host-buffer entry (opusgpu_decode_packets, what opus_multistream_decode callers get): packets in host memory -> PCM in
    host memory, 65,536 CELT-FB stereo packets per call, page-locked caller buffer.  PCIe-inclusive (10.6 MB in, 252 MB out per
    call): reported beside the device-resident figures, never as the headline `value`.  -> dict (rank 0)."""
    n, steps, L, toc = 65536, 9, 160, pkg.TOC_CELT_FB_STEREO  # (the first two calls grow the library's staging buffers: not timed)
    ctx.set_pipeline(False)
    ctx.streams_alloc(n, 2)
    pay = pkg.lcg_payloads(n, steps + 1, L, seed_base=ranks.seed_base())
    arenas = [np.concatenate([np.full((n, 1), toc, dtype=np.uint8), pay[s]], axis=1).reshape(-1) for s in range(steps + 1)]
    offs, lens, ids = np.arange(n, dtype=np.int64) * (L + 1), np.full(n, L + 1, dtype=np.int32), np.arange(n, dtype=np.int32)
    raw = np.zeros(n * 960 * 2 + 4096, dtype=np.int16)
    off = ((-raw.ctypes.data) % 4096) // 2
    out = raw[off:off + n * 960 * 2].reshape(n, 960, 2)
    ctx.host_register(out)
    try:
        ptrs = [(np.uint64(a.ctypes.data) + offs.astype(np.uint64)).astype(np.uint64) for a in arenas]  # (a C caller has its pointers)
        res = np.zeros(n, dtype=np.int32)
        for s in range(2):
            ctx.decode_packets_raw(ids, ptrs[s], lens, out, res)
        ranks.barrier()
        t0 = time.perf_counter()
        for s in range(2, steps + 1):
            ctx.decode_packets_raw(ids, ptrs[s], lens, out, res)
        dt = ranks.max_over_ranks(time.perf_counter() - t0)
        timed = steps - 1
        if not (res == 960).all():
            raise SystemExit("host path: decode failed")
        import oracle_py
        pick = np.arange(0, n, n // 128)
        ref, _ = oracle_py.load().batch_decode(2, toc, np.ascontiguousarray(pay[:, pick]))
        if not np.array_equal(out[pick], ref[:, steps]):
            raise SystemExit("host path: PCM differs from the CPU oracle")
    finally:
        ctx.host_unregister(out)
        ctx.set_pipeline(args.pipeline == "on")
    if ranks.rank != 0:
        return None
    return {"name": "host_path_64k", "value": n * timed * ranks.world / dt, "unit": "frames/s", "ms_per_step": dt / timed * 1e3,
            "config": {"workload": f"host_path_64k: opusgpu_decode_packets, {n} CELT-FB stereo packets of {L + 1} bytes per call from host memory, "
                                   "PCM into a page-locked host buffer (PCIe-inclusive: 10.6 MB in, 251.7 MB out per call)", "streams_per_gpu": n},
            "pcie_floor_ms": n * 3840 / 55.8e9 * 1e3,
            "parity_check": {"streams_checked": len(pick), "frames_of_history": steps + 1, "result": "last call bit-exact vs the CPU oracle"}}


def launch_ranks(args, argv):
    """--gpus N > 1 and no rank environment: start the N ranks as a child job (torch.distributed.run, one process per
    GPU) and relay its output.  Nothing in THIS process has touched HIP / torch.cuda: it only waits for the child."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def check_against_oracle(pkg, ctx, name, n, frames, d_pcm, pay=None, pages_seed=None, slot_stream=None):
    """After the timed region: the last step's PCM of >= 256 streams spread over the batch against the CPU oracle (which
    decodes those streams' whole history: warm-up + timed steps).  Raises on a mismatch.
    slot_stream: stream id of every PCM slot of the last step (step tables grouped by mode are not in stream order)."""
    import zlib
    import oracle_py
    o = oracle_py.load()
    pick = np.unique(np.concatenate([np.arange(0, n, max(1, n // 320)), [n - 1]])).astype(np.int64)
    out = np.zeros((n, 960, 2), dtype=np.int16)
    ctx.d2h(out, d_pcm)
    slot_of = np.arange(n) if slot_stream is None else np.argsort(slot_stream)
    refs = {}
    if pay is not None:
        toc = WORKLOADS[name][0]
        ref, _ = o.batch_decode(2, toc, np.ascontiguousarray(pay[:frames][:, pick]))
        refs = {int(s): ref[i, frames - 1] for i, s in enumerate(pick)}
    else:  # mixed pages: stream s has mode s % 3 and payload row s // 3 of that mode's generator
        for m, (toc, L) in enumerate(MIX):
            sel = pick[pick % 3 == m]
            if len(sel):
                full = pkg.lcg_payloads(int(sel.max()) // 3 + 1, frames, L, seed_base=pages_seed(m))
                ref, _ = o.batch_decode(2, toc, np.ascontiguousarray(full[:, sel // 3]))
                refs.update({int(s): ref[i, frames - 1] for i, s in enumerate(sel)})
    bad = [s for s in sorted(refs) if not (out[slot_of[s]] == refs[s]).all()]
    if bad and os.environ.get("BENCH_ABLATION") == "1":  # timing of builds that leave work out on purpose (tools/ab.sh with -DOG_RABL=..): no result
        return {"streams_checked": len(refs), "frames_of_history": frames, "result": "NOT bit-exact: ablation build (BENCH_ABLATION=1), the figure is not a measurement of the decoder"}
    if bad:
        raise SystemExit(f"{name}: GPU PCM of the last timed step differs from the CPU oracle for streams {bad[:8]} "
                         f"({len(bad)} of {len(refs)} checked)")
    crc = 0
    for s in sorted(refs):
        crc = zlib.crc32(out[slot_of[s]].tobytes(), crc)
    return {"streams_checked": len(refs), "frames_of_history": frames, "result": "last timed step bit-exact vs the CPU oracle",
            "pcm_crc32": f"{crc:08x}"}


def traffic_for(name, n):
    """HBM bytes per step from the committed PMC passes of THIS round's build (tools/prof_pmc.sh -> profiles/r05/), taken
    at this batch size; null when no such file exists.  The file names the build it was measured on."""
    for rnd in ("r05",):
        tpath = os.path.join(ROOT, "profiles", rnd, f"traffic_{name}.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("frames_per_launch") == n:
                return tj["hbm_bytes_per_step"], tj.get("measured_on", f"profiles/{rnd}")
    return None, None


def valu_issue_for(name, n, step_ms):
    """SURVEY 8(d), caveat H1: the decode is bound by instruction issue, not HBM -- report that next to the HBM fraction.  From the
    same committed PMC passes: vector-ALU wave-instructions of the step's kernels, the time the chip needs just to issue them
    (a wave64 vector instruction occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz), and that time over the measured
    step.  None when no counters of this round exist for the workload."""
    tpath = os.path.join(ROOT, "profiles", "r05", f"traffic_{name}.json")
    if not os.path.exists(tpath):
        return None
    with open(tpath) as fh:
        tj = json.load(fh)
    ks = {k: v for k, v in tj.get("kernels", {}).items() if v.get("valu_insts_per_launch")}
    if tj.get("frames_per_launch") != n or not ks:
        return None
    to_ms = 4.0 / (256 * 4 * 2.4e9) * 1e3
    total = sum(v["valu_insts_per_launch"] for v in ks.values())
    top = max(ks, key=lambda k: ks[k]["valu_insts_per_launch"])
    return {"valu_wave_instructions_per_step": total, "issue_ms": total * to_ms, "frac_of_step": total * to_ms / step_ms,
            "dominant_kernel": {"name": top, "valu_per_frame": ks[top]["valu_insts_per_launch"] / n,
                                "issue_ms": ks[top]["valu_insts_per_launch"] * to_ms, "measured_ms": ks[top].get("avg_ms"),
                                "frac": ks[top]["valu_insts_per_launch"] * to_ms / ks[top]["avg_ms"] if ks[top].get("avg_ms") else None},
            "model": "wave64 VALU instruction = 4 cycles of one SIMD; 1024 SIMDs at 2.4 GHz (MI355X_MICROARCH.md)",
            "source": tj.get("measured_on")}


def kernels_of(name):
    split = os.environ.get("OPUSGPU_SPLIT", "1") != "0"
    split_silk = split and os.environ.get("OPUSGPU_SPLIT_HYBRID", "1") != "0"
    if name == "mixed_pages_2m":
        return "k_silk_parse + k_silk_params + k_celt_parse + k_silk_synth (narrowband SILK-only frames: k_silk_synth_nb) + k_celt_recon + k_celt_post + k_decode_step (Q4 pass)"
    if name.startswith("celt"):
        return "k_celt_parse + k_celt_recon + k_celt_post" if split else "k_decode_step"
    if name.startswith("silk"):
        return "k_silk_parse + k_silk_params + k_silk_synth_nb (narrowband; k_silk_synth otherwise)" if split_silk else "k_decode_step"
    return "k_silk_parse + k_silk_params + k_silk_synth + k_celt_parse + k_celt_recon + k_celt_post" if split_silk else "k_decode_step"


def run_workload(name, args, ranks, pkg, ctx, n_override=0, cpu=True):
    """One workload: W warm-up steps, K timed steps between barriers, PCM check, roofline, CPU baseline.  -> dict (rank 0)."""
    shard = load_shard()
    rank, world = ranks.rank, ranks.world
    toc, L, bytes_per_frame, default_streams = WORKLOADS[name]
    n = n_override or default_streams
    K, W = args.steps, args.warmup
    mixed = name == "mixed_pages_2m"
    if mixed:  # stream s has mode s % 3: 262,144 streams per GPU are 87,382 SILK-NB + 87,381 hybrid + 87,381 CELT (8 ranks: 2,097,152 pages)
        bytes_per_frame = sum(len(range(m, n, 3)) * b for m, b in enumerate((5953, 24945, 21633))) / n
    ctx.streams_alloc(n, 2)
    ingest, pay, e2e = None, None, None
    frees = []
    d_pcm = ctx.dev_alloc(n * 960 * 2 * 2)
    d_res = ctx.dev_alloc(4 * n)
    frees += [d_pcm, d_res]
    if mixed:
        # work arrives at rank 0 as Ogg pages and is scattered: the path's one exchange step, before the timed region
        t_in0 = time.perf_counter()
        base, lay, ingest, _keep = prepare_pages_work(pkg, shard, ranks, ctx, n, K + W, args.ingest, args.page_crc)
        t_ingest = ranks.max_over_ranks(time.perf_counter() - t_in0)
        if lay.counts != [n] * (K + W):
            raise SystemExit(f"unexpected step tables: {lay.counts[:4]}...")
        if _keep is None:
            frees.append(base)

        # a stream's mode is fixed in this workload, which is what OPUSGPU_STEP_KEEPS_MODE promises: every step -- frames of all three
        # modes -- then runs its entropy kernels ahead, next to the arithmetic kernels of the step before (BENCH_MIXED_FLOW=substeps:
        # every step as three declared sub-steps over the parts of its table; =inorder: no promise, in order in two halves)
        keeps = args.pipeline == "on" and MIXED_FLOW != "inorder"

        def step(f):
            ctx.decode_work_step(base, lay, f, d_pcm, d_res, by_kind=keeps and MIXED_FLOW == "substeps", keeps_kind=keeps)
    else:
        # streams are sharded across ranks with no data-path exchange: each rank owns streams
        # [rank*n, (rank+1)*n) of the global id space (seeds differ per global stream id)
        pay = pkg.lcg_payloads(n, K + W, L, seed_base=ranks.seed_base())
        d_arena, d_desc = [], []
        # SILK-only / hybrid workloads: every step's table in the order of the frames' LBRR flags (the top bits of their first byte),
        # so that the 32 frames of a parse wave agree on how many extra frames of side information they read past -- framing work,
        # like grouping a page step's table by mode; BENCH_ORDER_BY_HEADER=0: stream order
        by_header = os.environ.get("BENCH_ORDER_BY_HEADER", "1") != "0" and not (toc & 0x80)
        last_slot_stream = None
        for f in range(K + W):
            arena, descs = pkg.build_step(toc, pay[f], order_by_header=by_header)
            last_slot_stream = descs["stream"].astype(np.int64) if by_header else None
            a = ctx.dev_alloc(arena.nbytes + 16)
            d = ctx.dev_alloc(descs.nbytes)
            ctx.h2d(a, arena)
            ctx.h2d(d, descs)
            d_arena.append(a)
            d_desc.append(d)
        frees += d_arena + d_desc

        def step(f):
            ctx.decode_step_device(n, d_desc[f], d_arena[f], d_pcm, d_res, modes=pkg.toc_modes(toc))

        def window(f0, f1):  # steps f0 .. f1-1 queued by ONE call: the library sees which step follows which
            ctx.decode_steps_device([n] * (f1 - f0), d_desc[f0:f1], d_arena[f0:f1], [d_pcm] * (f1 - f0), [d_res] * (f1 - f0),
                                    modes=pkg.toc_modes(toc))

    def barrier():
        ranks.barrier()
        ctx.synchronize()

    windowed = (not mixed) and args.window == "on"
    if windowed and W:
        window(0, W)
    else:
        for f in range(W):
            step(f)
    ctx.synchronize()
    ev = [ctx.event() for _ in range(K + 1)]
    barrier()
    t0 = time.perf_counter()
    ctx.event_record(ev[0])
    if windowed:
        window(W, W + K)
        ctx.event_record(ev[K])
    else:
        for f in range(K):
            step(W + f)
            ctx.event_record(ev[f + 1])
    ctx.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = ([ctx.event_elapsed_ms(ev[0], ev[K]) / K] * K if windowed else
                 [ctx.event_elapsed_ms(ev[f], ev[f + 1]) for f in range(K)])
    for e in ev:
        ctx.event_destroy(e)
    res = np.zeros(n, dtype=np.int32)
    ctx.d2h(res, d_res)
    if not (res == 960).all():
        raise SystemExit(f"{name}: decode failed for {(res != 960).sum()} frames in the last step")
    # parity of what was just timed: last step's PCM of >= 256 streams vs the CPU oracle (every rank checks its own)
    if mixed:
        r = rank
        last = np.zeros(n, dtype=pkg.DESC_DTYPE)
        import ctypes
        at = base.value if isinstance(base, ctypes.c_void_p) else int(base)
        ctx.d2h(last, ctypes.c_void_p(at + lay.desc_at[K + W - 1]))
        parity = check_against_oracle(pkg, ctx, name, n, K + W, d_pcm, slot_stream=last["stream"].astype(np.int64),
                                      pages_seed=lambda m: (0x9E3779B9 ^ (r * 0x01000193) ^ (m * 0x5bd1e995)) & 0xFFFFFFFF)
        if RAW_PAGES and os.environ.get("BENCH_E2E", "1") != "0":  # (BENCH_E2E=0: profiling runs count the timed steps' kernels only)
            timed_pcm = np.zeros((n, 960, 2), dtype=np.int16)
            ctx.d2h(timed_pcm, d_pcm)
            e2e_stats, e2e_pcm = overlapped_end_to_end(pkg, ranks, ctx, n, d_pcm, d_res, max(1, shard.usable_cpus() - 2))
            if not np.array_equal(timed_pcm, e2e_pcm):
                raise SystemExit(f"{name}: the overlapped end-to-end run left other PCM than the timed run")
            e2e = {"wall_s": ranks.max_over_ranks(e2e_stats["wall_s"]), "pages": ranks.sum_over_ranks(e2e_stats["pages"]),
                   "rank0": e2e_stats}
            del timed_pcm, e2e_pcm
    else:
        parity = check_against_oracle(pkg, ctx, name, n, K + W, d_pcm, pay=pay, slot_stream=last_slot_stream)

    value, dt, total_frames = shard.aggregate_throughput(ranks, n * K, dt)
    for p in frees:
        ctx.dev_free(p)
    if rank != 0:
        return None
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    achieved = bytes_per_frame * n / avg_kernel_s / 1e9
    traffic, traffic_src = traffic_for(name, n)
    out = {
        "value": value, "unit": "frames/s", "ms_per_step": dt / K * 1e3,
        "config": ({"workload": f"{name}: {n} streams/GPU, one Ogg page of {PACKETS_PER_PAGE} packets per stream and "
                                f"10 steps, modes SILK-NB : hybrid FB : CELT FB = 1:1:1 across streams (TOC 0x0C / 0x7C / "
                                f"0xFC, 40 / 120 / 160-byte LCG payloads), 48 kHz stereo, step tables grouped by mode" +
                                (" and, within the SILK-only and hybrid groups, ordered by the frames' LBRR flags" if PAGE_ORDER & 4 else "") +
                                ("; steps carry OPUSGPU_STEP_KEEPS_MODE (a stream's mode is fixed) and are pipelined: the entropy "
                                 "kernels of step k + 1 next to the arithmetic kernels of step k" +
                                 (", three declared sub-steps per step" if MIXED_FLOW == "substeps" else "")
                                 if keeps else "; undeclared steps, in order, in two halves"),
                    "streams_per_gpu": n,
                    "sharding": ("rank 0 ingests the pages (host demux) and scatters every rank's decode steps "
                                 if args.ingest == "rank0" else
                                 "rank 0 routes the raw pages and scatters them, every rank demuxes its own share "
                                 ) + "(torch.distributed scatter = RCCL), before the timed region; no collective inside it"}
                   if mixed else
                   {"workload": f"{name}: {n} streams/GPU x 20 ms frames, 48 kHz stereo, "
                                f"TOC 0x{toc:02X}, {L}-byte LCG payloads, state persistent across steps" +
                                ("; step tables ordered by the frames' LBRR flags (top bits of the first payload byte)" if by_header else "") +
                                (f"; a window of {K} steps queued by one call, declared {'CELT-only' if toc & 0x80 else 'mode'} mask"
                                 if windowed else ""),
                    "streams_per_gpu": n, "sharding": "streams partitioned across ranks, no data-path collective"}),
        "x_realtime_per_gpu": value / world / 50.0,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": f"decode step = {kernels_of(name)} (launched back to back" +
                               ("; pipelined steps, opusgpu_set_pipeline: the next step's k_celt_parse of CELT-only frames runs on "
                                "the library's second stream next to this step's k_celt_recon_fb / k_celt_post)"
                                if args.pipeline == "on" else ")"),
                     "pipeline": args.pipeline, "window": "on" if windowed else "off", "avg_launch_ms": avg_kernel_s * 1e3,
                     "algorithmic_bytes_per_frame": bytes_per_frame, "frames_per_launch": n,
                     "valu_issue": valu_issue_for(name, n, avg_kernel_s * 1e3)},
        "parity_check": parity,
    }
    if ingest is not None:
        # host page demux + scatter happen before the timed region and are never part of `value`; the end-to-end figure
        # prices them in: all pages of the job / (ingest wall time + the decode steps those pages make)
        t_in = max(t_ingest - ingest["generate_s"], 1e-9)
        ingest["ingest_wall_s"] = t_in
        pages = ingest["pages"]
        decode_s = (dt / K) * (K + W)
        ingest["serial_end_to_end_pages_per_s"] = pages / (t_in + decode_s)
        ingest["serial_end_to_end_note"] = ("pages of all ranks / (routing + packing + scatter + demux + upload of the steps, wall time, "
                                            "max over ranks; making the synthetic pages excluded) + (their decode steps at the measured "
                                            "step time): nothing overlapped")
        ingest["decode_only_pages_per_s"] = pages / decode_s
        if e2e is not None:
            r0 = e2e["rank0"]
            nb = len(r0["ingest_s"])
            ingest["end_to_end_pages_per_s"] = e2e["pages"] / e2e["wall_s"]
            ingest["end_to_end_over_decode_only"] = ingest["end_to_end_pages_per_s"] / ingest["decode_only_pages_per_s"]
            ingest["end_to_end"] = {
                "what": "every rank, from its share of the raw pages (where the scatter left them, host side) to the last PCM: batches of "
                        "one page per stream; a host thread demuxes batch b + 1 (checksums verified on the host's cores) and uploads its "
                        "step tables and packets on the copy stream while batch b decodes (esp32-opus-player_amd/ingest.py); wall time "
                        "of the whole job incl. the first batch's ingest, which nothing hides; final PCM identical to the timed run's",
                "wall_s": e2e["wall_s"], "batches": nb, "steps": r0["steps"], "first_batch_ready_s": r0["first_batch_ready_s"],
                "ingest_s_per_batch": r0["ingest_s"], "demux_s_per_batch": r0["demux_s"], "slot_wait_s_per_batch": r0["slot_wait_s"],
                "gpu_ms_between_batch_ends": r0["gpu_ms_between_batch_ends"],
                }
        out["ingest"] = ingest
    if cpu and not args.no_cpu_baseline:
        out["cpu_baseline"] = mixed_cpu_baseline() if mixed else cpu_baseline(toc, L, seconds_target=args.cpu_seconds)
        out["cpu_baseline"]["calibration"] = ("oracle only (a bit-identical C restatement of the reference path); the reference "
                                              "itself is not buildable in this image (needs <Arduino.h>), so no oracle/reference "
                                              "speed ratio exists")
    return out


def rfc_frame_bytes(toc, L, lost):
    """Algorithmic bytes of one RFC-mode frame, by SURVEY 8d's rule (packet in, PCM out, the minimal state a bit-exact decoder reads
    and writes back) at the frame's own duration D: CELT history read 8,672 + 400 B, write (D + 60) samples x 4 B x 2 ch + 400;
    SILK state 2,072 (NB) / 2,712 (MB) / 3,352 B (WB, hybrid) read + write; a concealed frame has no packet bytes."""
    import rfc_common
    D = rfc_common.dur(toc)
    mode, bw = rfc_common.mode_bw(toc)
    b = (0 if lost else L + 1) + 4 * D
    if mode != rfc_common.MODE_SILK:
        b += 9072 + 8 * (D + 60) + 400
    if mode != rfc_common.MODE_CELT:
        b += 3352 if mode == rfc_common.MODE_HYBRID else {1101: 2072, 1102: 2712, 1103: 3352}[bw]
    return b


def run_rfc_workload(args, ranks, pkg, ctx, n_override=0, cpu=True):
    """RFC mode, one measured line: every TOC configuration, lost packets and forward error correction through the device-resident
    entry (opusgpu_decode_step_device in OPUSGPU_MODE_RFC: one k_decode_rfc launch per step).  -> dict (rank 0)."""
    import zlib
    import rfc_common
    import oracle_py
    shard = load_shard()
    rank, world = ranks.rank, ranks.world
    n = n_override or 65536
    K, W = min(args.steps, 8), min(args.warmup, 2)
    F = K + W
    rng = np.random.default_rng(0xC0DEC + rank)
    cfg = np.arange(n) % 32
    toc_of = ((np.arange(32) << 3) | 4).astype(np.uint8)
    dur_of = np.array([rfc_common.dur(int(t)) for t in toc_of])
    mode_of = np.array([rfc_common.mode_bw(int(t))[0] for t in toc_of])
    # bytes per frame: about 64 kbit/s at every duration, SILK-only at about 24 kbit/s
    L_of = np.array([max(8, min(600, int((24000 if mode_of[c] == rfc_common.MODE_SILK else 64000) * dur_of[c] / 48000 / 8))) for c in range(32)])
    flags_of = np.zeros(32, dtype=np.int32)
    for c in range(32):  # the library's own framing names the descriptor flags of a configuration
        d = (pkg.FrameDesc * 48)()
        pkt = bytes([int(toc_of[c])]) + bytes(int(L_of[c]))
        if pkg.load_lib().opusgpu_packet_to_frames_mode(pkt, len(pkt), 0, 1, d) != 1 or d[0].offset != 1 or d[0].len != L_of[c]:
            raise SystemExit("rfc workload: unexpected framing of a code-0 packet")
        flags_of[c] = d[0].flags
    Ls, tocs, durs = L_of[cfg], toc_of[cfg], dur_of[cfg]
    size = Ls + 1
    at = np.concatenate([[0], np.cumsum(size)]).astype(np.int64)  # packet s of a step lies at at[s] of that step's arena
    # ops[f, s]: 0 decode, 1 lost and concealed, 2 lost and recovered from packet f + 1 (never at step 0: nothing to continue from)
    lost = rng.random((F + 1, n)) < RFC_LOSS
    lost[0] = False
    lost[F] = False
    can_fec = (mode_of[cfg] != rfc_common.MODE_CELT)[None, :] & ~np.roll(lost, -1, axis=0)
    ops = np.where(lost, np.where(can_fec & (rng.random((F + 1, n)) < 0.5), 2, 1), 0).astype(np.uint8)[:F]
    arenas = []
    for f in range(F + 1):  # packet bytes of every step (step F: only read by the recoveries of step F - 1)
        a = rng.integers(0, 256, int(at[n]), dtype=np.uint8)
        a[at[:-1]] = tocs
        arenas.append(a)
    ctx.set_mode(True)
    ctx.streams_alloc(n, 2)
    stride = 2880 * 2
    d_pcm = ctx.dev_alloc(n * stride * 2)
    d_res = ctx.dev_alloc(4 * n)
    frees = [d_pcm, d_res]
    d_arena, d_desc = [], []
    host_arena, offs = [], np.zeros((F, n), dtype=np.int64)
    arena_at = 0
    for f in range(F):
        if (ops[f] == 2).any():  # a recovery reads the NEXT step's packet: this step's arena carries a copy of it
            a = arenas[f].copy()
            for s_ in np.nonzero(ops[f] == 2)[0]:
                a[at[s_]:at[s_ + 1]] = arenas[f + 1][at[s_]:at[s_ + 1]]
        else:
            a = arenas[f]
        descs = np.zeros(n, dtype=pkg.DESC_DTYPE)
        descs["stream"] = np.arange(n, dtype=np.int32)
        descs["offset"] = np.where(ops[f] == 1, 0, at[:-1] + 1)
        descs["len"] = np.where(ops[f] == 1, 0, Ls)
        descs["flags"] = flags_of[cfg] | np.where(ops[f] == 2, 1 << 10, 0)
        da, dd = ctx.dev_alloc(a.nbytes + 16), ctx.dev_alloc(descs.nbytes)
        ctx.h2d(da, a)
        ctx.h2d(dd, descs)
        d_arena.append(da)
        d_desc.append(dd)
        host_arena.append(a)
        offs[f] = arena_at + at[:-1]
        arena_at += a.nbytes
    frees += d_arena + d_desc
    lens = np.where(ops == 1, 0, size[None, :]).astype(np.int32)

    def step(f):
        ctx.decode_step_device(n, d_desc[f], d_arena[f], d_pcm, d_res)

    for f in range(W):
        step(f)
    ctx.synchronize()
    ev = [ctx.event() for _ in range(K + 1)]
    ranks.barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.event_record(ev[0])
    for f in range(K):
        step(W + f)
        ctx.event_record(ev[f + 1])
    ctx.synchronize()
    ranks.barrier()
    dt = time.perf_counter() - t0
    kernel_ms = [ctx.event_elapsed_ms(ev[f], ev[f + 1]) for f in range(K)]
    for e in ev:
        ctx.event_destroy(e)
    res = np.zeros(n, dtype=np.int32)
    ctx.d2h(res, d_res)
    if not (res == durs).all():
        raise SystemExit(f"{RFC_WORKLOAD}: {(res != durs).sum()} frames of the last step did not return their TOC's duration")
    # parity of what was timed: the last step's PCM of 10 streams per configuration against the oracle's RFC mode, which decodes,
    # conceals and recovers those streams' whole history
    pick = np.unique(np.concatenate([np.arange(0, n, max(1, n // 320)), np.arange(n - 32, n)])).astype(np.int64)
    out = np.zeros((n, 2880, 2), dtype=np.int16)
    ctx.d2h(out, d_pcm)
    o = oracle_py.load()
    big = np.concatenate(host_arena)
    ref, rets = o.batch_decode_rfc(2, big, offs[:, pick], lens[:, pick], ops[:, pick])
    crc, bad = 0, []
    for i, s_ in enumerate(pick):
        D = int(durs[s_])
        if rets[i, F - 1] != D or not np.array_equal(out[s_, :D], ref[i, :D]):
            bad.append(int(s_))
        crc = zlib.crc32(out[s_, :D].tobytes(), crc)
    if bad:
        raise SystemExit(f"{RFC_WORKLOAD}: GPU PCM of the last timed step differs from the CPU oracle's RFC mode for streams {bad[:8]} "
                         f"({len(bad)} of {len(pick)} checked)")
    parity = {"streams_checked": len(pick), "frames_of_history": F,
              "lost_or_recovered_in_last_step": int((ops[F - 1, pick] != 0).sum()),
              "result": "last timed step bit-exact vs the CPU oracle's RFC mode (itself parity-unpinned: DESIGN.md section 11)",
              "pcm_crc32": f"{crc:08x}"}
    timed = ops[W:]
    alg = sum(rfc_frame_bytes(int(toc_of[c]), int(L_of[c]), False) * int(((cfg == c)[None, :] & (timed != 1)).sum()) +
              rfc_frame_bytes(int(toc_of[c]), int(L_of[c]), True) * int(((cfg == c)[None, :] & (timed == 1)).sum()) for c in range(32))
    audio_s = float(durs.sum()) * K / 48000.0
    value, dt, total_frames = shard.aggregate_throughput(ranks, n * K, dt)
    for p in frees:
        ctx.dev_free(p)
    if os.environ.get("BENCH_PROF") and hasattr(pkg.load_lib(), "opusgpu_debug_prof"):  # (a -DOG_PROF build: section cycles of k_decode_rfc)
        import ctypes
        buf = (ctypes.c_ulonglong * 64)()
        pkg.load_lib().opusgpu_debug_prof(buf, 1)
        tot = sum(buf) or 1
        sys.stderr.write("k_decode_rfc sections (OG_MARK id: cycles per frame, share):\n" + "".join(
            f"  {i:2d} {buf[i] / (n * (K + W)):10.0f} {100.0 * buf[i] / tot:5.1f}%\n" for i in range(64) if buf[i]))
    ctx.set_mode(False)
    if rank != 0:
        return None
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    achieved = alg / K / avg_kernel_s / 1e9
    out_line = {
        "name": RFC_WORKLOAD, "value": value, "unit": "frames/s", "ms_per_step": dt / K * 1e3, "steps": K, "warmup": W,
        "config": {"workload": f"{RFC_WORKLOAD}: RFC mode (frames at the durations their TOC names), {n} streams/GPU, stream s keeps TOC "
                               f"configuration s % 32 (SILK NB/MB/WB 10-60 ms, hybrid SWB/FB 10/20 ms, CELT NB-FB 2.5-20 ms), one "
                               f"stereo code-0 packet per stream and step (about 64 kbit/s, SILK-only 24 kbit/s, random bytes), "
                               f"{100 * RFC_LOSS:.0f} % of the packets lost: concealed, or for half of the SILK-only / hybrid losses "
                               f"whose next packet arrived recovered from its forward error correction data",
                   "streams_per_gpu": n, "sharding": "streams partitioned across ranks, no data-path collective",
                   "lost_frames": int((timed == 1).sum()), "recovered_frames": int((timed == 2).sum())},
        "x_realtime_per_gpu": audio_s / dt,
        "audio_seconds_per_step": audio_s / K,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "decode step = k_decode_rfc (one frame per wave at its true duration; og_rfc.hip)",
                     "avg_launch_ms": avg_kernel_s * 1e3, "algorithmic_bytes_per_frame": alg / K / n, "frames_per_launch": n},
        "parity_check": parity,
    }
    if cpu and not args.no_cpu_baseline:
        cores = shard.usable_cpus()
        ns = min(n, 4096)
        t1 = time.perf_counter()
        _, r2 = o.batch_decode_rfc(2, big, offs[:, :ns], lens[:, :ns], ops[:, :ns], threads=cores, want_pcm=False)
        dt_cpu = time.perf_counter() - t1
        out_line["cpu_baseline"] = {
            "value": float((r2 > 0).sum()) / dt_cpu, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"the first {ns} streams x {F} steps of the same workload (losses and recoveries included), {cores} threads, one "
                      f"oracle decoder in RFC mode per stream, {dt_cpu:.2f} s wall",
            "calibration": "oracle only; its RFC mode has no reference-origin vector (parity-unpinned)"}
    return out_line


# The driver keeps the TAIL of what a run prints: the one JSON line has to fit it.  By default the line carries every number and
# short texts only (COMPACT below); --verbose prints the explanatory strings as well (what a sample was, how a figure was taken).
_DROP = {"traffic_source", "model", "source", "calibration", "serial_end_to_end_note", "what", "ingest_s_per_batch", "demux_s_per_batch",
         "slot_wait_s_per_batch", "gpu_ms_between_batch_ends", "host_logical_cpus", "generate_s", "valu_wave_instructions_per_step",
         "issue_ms", "demux_threads", "page_bytes", "work_bytes_per_rank", "pipeline", "window", "frames_of_history"}
_CUT = {"workload": ":", "kernel": " (", "sample": ",", "result": " (", "sharding": ","}


def compact(o, key=None, depth=0):
    """The line without its prose: long strings cut at their first clause, explanatory keys dropped, floats to 6 digits."""
    if isinstance(o, dict):
        # (below the line's own objects -- the other configurations' -- what repeats the headline's: the peak, its unit, the bound)
        return {k: compact(v, k, depth + 1) for k, v in o.items()
                if k not in _DROP and not (depth > 1 and k in ("sharding", "peak", "bound", "frames_per_launch"))}
    if isinstance(o, list):
        return [compact(v, key, depth + 1) for v in o]
    if isinstance(o, float):
        return float(f"{o:.6g}")
    if isinstance(o, str) and key in _CUT and len(o) > 64:
        return o.split(_CUT[key])[0][:96]
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--verbose", action="store_true", help="the JSON line with its explanatory texts (default: numbers and short texts, "
                                                           "so that the whole line fits the driver's tail)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)   # (pipelined steps: the first parse and the last de-emphasis of a run are not hidden)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="celt_fb_stereo_64k", choices=sorted(WORKLOADS) + [RFC_WORKLOAD])
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: the workload's)")
    ap.add_argument("--page-crc", default="gpu", choices=["host", "gpu"],
                    help="mixed_pages_2m with --ingest per-rank: who verifies the page checksums")
    ap.add_argument("--ingest", default="per-rank", choices=["rank0", "per-rank"],
                    help="mixed_pages_2m: who demuxes the Ogg pages (rank 0 for all, or every rank its own share)")
    ap.add_argument("--pipeline", default="on", choices=["on", "off"],
                    help="opusgpu_set_pipeline: step k+1's CELT parse next to step k's reconstruction (tables resident, as here)")
    ap.add_argument("--window", default="on", choices=["on", "off"],
                    help="queue the K timed steps with ONE opusgpu_decode_steps_device call (the library then orders the kernels of "
                         "neighbouring pipelined steps by dependencies); off: one opusgpu_decode_step_device_modes call per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work of the headline workload's cpu_baseline sample")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start / join the ranks, check the rank count, print a line and stop: no GPU work (the CPU test of "
                         "the launcher path runs this with OPUSGPU_DIST_BACKEND=gloo)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="only the selected workload (default: the other BASELINE configs follow as other_configs)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by a launcher: start the ranks ourselves, as a child job, before anything here touches the GPU
        sys.exit(launch_ranks(args, sys.argv[1:]))

    # one process per GPU; streams are partitioned across the ranks (esp32-opus-player_amd/shard.py): RCCL carries
    # nothing but the barrier around the timed region and the MAX / SUM of the per-rank figures
    ranks = load_shard().Ranks(backend=os.environ.get("OPUSGPU_DIST_BACKEND", "nccl"))
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report a "
                         f"{world}-GPU figure as a {args.gpus}-GPU one")
    if world > 1 and rank == 0:
        print(f"[bench] {world} ranks over RCCL (torch.distributed backend nccl), one process per GPU", file=sys.stderr, flush=True)
    K, W = args.steps, args.warmup
    if args.rendezvous_only:
        total = ranks.sum_over_ranks(1)
        ranks.barrier()
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "ranks_counted": int(total)}), flush=True)
        ranks.close()
        return

    pkg = load_pkg()
    ctx = pkg.Context(local_rank)
    ctx.set_pipeline(args.pipeline == "on")
    if args.workload == RFC_WORKLOAD:  # not a BASELINE config: its own line, same fields
        o = run_rfc_workload(args, ranks, pkg, ctx, n_override=args.streams)
        if rank == 0:
            o.update({"metric": "decoded 48 kHz stereo frames/sec/GPU, RFC mode (x real-time); HBM GB/s vs roofline", "n_gpus": world,
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                      "dtype": "int32 fixed-point (int16/int32 with 64-bit products)"})
            print(json.dumps(o if args.verbose else compact(o), separators=(",", ":")), flush=True)
        ranks.close()
        ctx.close()
        return
    main_out = run_workload(args.workload, args, ranks, pkg, ctx, n_override=args.streams)
    if args.pipeline == "on" and not args.no_other_configs:
        # the same workload with nothing queued ahead: one in-order step per call (what a caller gets that hands over one step at a
        # time and needs it back before the next -- the headline `value` is the pipelined window, K x 20 ms of audio buffered per stream)
        saved = (args.pipeline, args.steps, args.warmup)
        args.pipeline, args.steps, args.warmup = "off", min(args.steps, 8), 2
        ctx.set_pipeline(False)
        o = run_workload(args.workload, args, ranks, pkg, ctx, n_override=args.streams, cpu=False)
        args.pipeline, args.steps, args.warmup = saved
        ctx.set_pipeline(True)
        if rank == 0 and o is not None:
            main_out["config"]["in_order_ms_per_step"] = o["ms_per_step"]
            main_out["config"]["workload"] += f"; in order, one step per call: {o['ms_per_step']:.3f} ms per step"
    others = []
    if not args.no_other_configs and args.workload == "celt_fb_stereo_64k" and not args.streams:
        # the other BASELINE configs (3, 4, and one GPU's share of 5), timed the same way in the same run; their CPU
        # baselines get a smaller sample so that the default run stays within a few minutes
        saved = args.cpu_seconds
        args.cpu_seconds = 5.0
        for name in ("silk_nb_stereo_64k", "hybrid_fb_stereo_256k", "mixed_pages_2m"):
            o = run_workload(name, args, ranks, pkg, ctx)
            if o is not None:
                o["name"] = name
                others.append(o)
        o = run_rfc_workload(args, ranks, pkg, ctx)
        if o is not None:
            others.append(o)
        o = run_host_path(args, ranks, pkg, ctx)
        if o is not None:
            others.append(o)
        args.cpu_seconds = saved
    if rank == 0:
        line = {
            "metric": "decoded 48 kHz stereo frames/sec/GPU (x real-time); HBM GB/s vs roofline",
            "value": main_out["value"], "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": main_out["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32 fixed-point (int16/int32 with 64-bit products)", "data": "synthetic",
        }
        for k in ("config", "x_realtime_per_gpu", "roofline", "parity_check", "ingest", "cpu_baseline"):
            if k in main_out:
                line[k] = main_out[k]
        if others:
            line["other_configs"] = others
        print(json.dumps(line if args.verbose else compact(line), separators=(",", ":")), flush=True)
    ranks.close()
    ctx.close()


if __name__ == "__main__":
    main()

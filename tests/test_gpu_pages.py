"""BASELINE config 5 on one GPU: mixed-mode Ogg pages -> host demux (opusgpu_pages_demux) -> packed work
(shard.pack_work, what the work-queue scatter delivers) -> HBM -> decode steps; every PCM sample of every stream and step
compared with the CPU oracle.  Modes 1:1:1 across streams (fixed within a stream), two chained pages per stream."""
import ctypes
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shard():
    spec = importlib.util.spec_from_file_location("og_shard", os.path.join(ROOT, "esp32-opus-player_amd", "shard.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _run(pkg, oracle, ctx, n, pages_per_stream, packets_per_page, threads, pipeline=False, by_kind=False, keeps=False):
    """pipeline: all steps queued back to back with opusgpu_set_pipeline on (each step's PCM and results in buffers of their
    own), compared after one synchronisation -- the CELT-only third of every step parses ahead of the step before.
    keeps: every step carries OPUSGPU_STEP_KEEPS_MODE (a stream's mode is fixed here) -- a step of all three modes runs ahead like a
    declared one; by_kind: with the promise, every step as three declared sub-steps: the sub-steps of both pipelined kinds run
    ahead of each other, nothing drains between them."""
    shard = _shard()
    modes = ((pkg.TOC_SILK_NB_STEREO, 40), (pkg.TOC_HYBRID_FB_STEREO, 120), (pkg.TOC_CELT_FB_STEREO, 160))
    frames = pages_per_stream * packets_per_page
    ids_of = [np.arange(m, n, 3, dtype=np.int32) for m in range(3)]
    ref = [None] * 3
    parts = []  # (page matrix, stream ids) in the order the pages enter the queue
    for m, (toc, L) in enumerate(modes):
        pay = pkg.lcg_payloads(len(ids_of[m]), frames, L, seed_base=0x9E3779B9 + m)
        ref[m], ok = oracle.batch_decode_threads(2, toc, pay)
        assert ok == len(ids_of[m]) * frames
        for q in range(pages_per_stream):
            pg = pkg.build_pages(toc, pay[q * packets_per_page:(q + 1) * packets_per_page], ids_of[m].astype(np.uint32) + 77, seqno=2 + q)
            parts.append((q, pg, ids_of[m]))
    parts.sort(key=lambda x: x[0])  # all first pages, then all second pages ...
    blob = np.concatenate([pg.reshape(-1) for _, pg, _ in parts])
    lens = np.concatenate([np.full(pg.shape[0], pg.shape[1], dtype=np.int32) for _, pg, _ in parts])
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
    ids = np.concatenate([i for _, _, i in parts])
    batch = pkg.PageBatch(blob, offs, lens, ids, threads=threads)
    assert (batch.info["status"] == packets_per_page).all()
    assert batch.n_steps == frames
    work = shard.pack_work(batch)
    batch.close()
    lay = shard.WorkLayout(work)
    assert lay.counts == [n] * frames
    ctx.streams_alloc(n, 2)
    d_work = ctx.dev_alloc(work.size)
    nbuf = frames if pipeline else 1
    d_pcms = [ctx.dev_alloc(n * 960 * 2 * 2) for _ in range(nbuf)]
    d_ress = [ctx.dev_alloc(4 * n) for _ in range(nbuf)]
    out = np.zeros((n, 960, 2), dtype=np.int16)
    res = np.zeros(n, dtype=np.int32)
    try:
        ctx.h2d(d_work, work)
        if pipeline:
            ctx.set_pipeline(True)
            if by_kind:
                assert all(mc == (len(ids_of[0]), len(ids_of[1])) for mc in lay.mode_counts)
            for k in range(frames):
                ctx.decode_work_step(d_work, lay, k, d_pcms[k], d_ress[k], by_kind=by_kind, keeps_kind=by_kind or keeps)
            ctx.synchronize()
        for k in range(frames):
            d_pcm, d_res = d_pcms[k % nbuf], d_ress[k % nbuf]
            if not pipeline:
                ctx.decode_work_step(d_work, lay, k, d_pcm, d_res)
                ctx.synchronize()
            ctx.d2h(out, d_pcm)
            ctx.d2h(res, d_res)
            assert (res == 960).all(), f"step {k}: {(res != 960).sum()} frames failed"
            descs = np.frombuffer(work[lay.desc_at[k]:lay.desc_at[k] + 16 * n].tobytes(), dtype=pkg.DESC_DTYPE)
            stream = descs["stream"]
            assert (np.sort(stream) == np.arange(n)).all()
            mode = descs["flags"] & 3
            assert (np.diff(mode) >= 0).all()  # grouped by mode
            for m in range(3):
                slots = np.nonzero(mode == m)[0]
                assert (stream[slots] % 3 == m).all()
                want = ref[m][stream[slots] // 3, k]
                bad = np.nonzero((out[slots] != want).reshape(len(slots), -1).any(axis=1))[0]
                assert bad.size == 0, f"step {k} mode {m}: {bad.size} streams differ from the oracle, first {stream[slots][bad[:5]]}"
    finally:
        if pipeline:
            ctx.set_pipeline(False)
        for p in [d_work] + d_pcms + d_ress:
            ctx.dev_free(p)


def test_mixed_mode_pages_small(pkg, oracle, gpu_ctx):
    _run(pkg, oracle, gpu_ctx, 3 * 1024, 2, 5, threads=2)
    _run(pkg, oracle, gpu_ctx, 3 * 1024, 2, 5, threads=2, pipeline=True)
    _run(pkg, oracle, gpu_ctx, 3 * 1024 + 1, 2, 5, threads=2, pipeline=True, by_kind=True)
    _run(pkg, oracle, gpu_ctx, 3 * 1024 + 2, 2, 5, threads=2, pipeline=True, keeps=True)


def test_mixed_mode_pages_c5_share_pipelined(pkg, oracle, gpu_ctx):
    """The same share with pipelined steps, all ten queued back to back, every step with the promise that its streams keep their
    mode (what bench.py's mixed_pages_2m workload times)."""
    _run(pkg, oracle, gpu_ctx, 262144, 1, 10, threads=16, pipeline=True, keeps=True)


def test_mixed_mode_pages_c5_share_pipelined_substeps(pkg, oracle, gpu_ctx):
    """... every step as three declared sub-steps by kind."""
    _run(pkg, oracle, gpu_ctx, 262144, 1, 10, threads=16, pipeline=True, by_kind=True)


def test_mixed_mode_pages_c5_share_pipelined_undeclared(pkg, oracle, gpu_ctx):
    """... and as undeclared steps (in order, in two halves; only their CELT-only third would run ahead if it were declared)."""
    _run(pkg, oracle, gpu_ctx, 262144, 1, 10, threads=16, pipeline=True)


def test_mixed_mode_pages_c5_share(pkg, oracle, gpu_ctx):
    """One GPU's share of config C5 (2,097,152 pages over 8 GPUs): 262,144 pages of 10 packets, one page per stream (modes by
    stream id mod 3: the SILK-NB third is one stream larger)."""
    _run(pkg, oracle, gpu_ctx, 262144, 1, 10, threads=16)


def test_ragged_pages_random_modes(pkg, oracle, gpu_ctx):
    """Ragged input on the page path: every stream has its own number of pages (1 - 3, chained) and packets per page (1 - 8),
    its own mode / bandwidth / mono-stereo per packet and payload lengths from 2 to 400 bytes, so later steps are only
    partly filled, groups differ from step to step and chained pages start at different steps.  Every frame against the
    oracle (frames of a stream in order: page by page, packet by packet)."""
    import random
    import ogg_util
    shard = _shard()
    rng = random.Random(77)
    n = 1500
    cfgs = [1, 5, 9, 13, 15, 19, 23, 27, 31]
    streams = []  # per stream: list of packets in decode order
    pages, ids = [], []
    for s in range(n):
        home = rng.choice(cfgs)
        pk = []
        for q in range(rng.randrange(1, 4)):
            page_pk = []
            for _ in range(rng.randrange(1, 9)):
                cfg = home if rng.random() < 0.8 else rng.choice(cfgs)
                toc = cfg << 3 | (4 if rng.random() < 0.8 else 0)
                page_pk.append(bytes([toc]) + bytes(rng.getrandbits(8) for _ in range(rng.choice([2, 3, 10, 40, 120, 160, 400]))))
            pages.append((q, s, ogg_util.page(1000 + s, 2 + q, 0, page_pk)))
            pk += page_pk
        streams.append(pk)
    pages.sort(key=lambda x: (x[0], rng.random()))  # a stream's pages in order, streams shuffled within each wave of pages
    blob = np.frombuffer(b"".join(p for _, _, p in pages), dtype=np.uint8)
    lens = np.array([len(p) for _, _, p in pages], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
    sids = np.array([s for _, s, _ in pages], dtype=np.int32)
    batch = pkg.PageBatch(blob, offs, lens, sids, threads=4)
    assert (batch.info["status"] > 0).all()  # all pages accepted
    frames = max(len(pk) for pk in streams)
    assert batch.n_steps == frames
    work = shard.pack_work(batch)
    batch.close()
    lay = shard.WorkLayout(work)
    assert lay.counts == [sum(1 for pk in streams if len(pk) > k) for k in range(frames)]
    # oracle: every stream's packets in order
    plen = np.zeros((frames, n), dtype=np.int64)
    for s, pk in enumerate(streams):
        plen[:len(pk), s] = [len(p) for p in pk]
    aoffs = np.concatenate([[0], np.cumsum(plen.reshape(-1))[:-1]]).reshape(frames, n)
    arena = np.zeros(int(plen.sum()) + 16, dtype=np.uint8)
    for s, pk in enumerate(streams):
        for f, p in enumerate(pk):
            arena[aoffs[f, s]:aoffs[f, s] + len(p)] = np.frombuffer(p, dtype=np.uint8)
    # (streams that have run out contribute zero-length "packets": the oracle returns an error for those and they are skipped)
    ref, rets = oracle.batch_decode_var(2, arena, aoffs, plen.astype(np.int32))
    gpu_ctx.streams_alloc(n, 2)
    d_work = gpu_ctx.dev_alloc(work.size)
    d_pcm, d_res = gpu_ctx.dev_alloc(n * 960 * 2 * 2), gpu_ctx.dev_alloc(4 * n)
    try:
        gpu_ctx.h2d(d_work, work)
        for k in range(frames):
            m = lay.counts[k]
            out = np.zeros((m, 960, 2), dtype=np.int16)
            res = np.zeros(m, dtype=np.int32)
            gpu_ctx.decode_work_step(d_work, lay, k, d_pcm, d_res)
            gpu_ctx.synchronize()
            gpu_ctx.d2h(out, d_pcm)
            gpu_ctx.d2h(res, d_res)
            descs = np.frombuffer(work[lay.desc_at[k]:lay.desc_at[k] + 16 * m].tobytes(), dtype=pkg.DESC_DTYPE)
            stream = descs["stream"]
            assert len(set(stream.tolist())) == m and (np.diff(descs["flags"] & 3) >= 0).all()
            assert sorted(stream.tolist()) == [s for s, pk in enumerate(streams) if len(pk) > k]
            assert (res == rets[stream, k]).all(), (k, res[res != rets[stream, k]][:4])
            for j in np.nonzero(res == 960)[0]:
                s = stream[j]
                toc = streams[s][k][0]
                ncmp = 960 if (not toc & 0x80 and (toc & 0x60) != 0x60 and not toc & 4) else 1920  # Q3
                assert (out[j].reshape(-1)[:ncmp] == ref[s, k].reshape(-1)[:ncmp]).all(), (k, s, hex(toc))
    finally:
        for p in (d_work, d_pcm, d_res):
            gpu_ctx.dev_free(p)


def test_page_crc_on_the_gpu(pkg, gpu_ctx):
    """opusgpu_pages_crc_device against the independent pure-Python CRC of tests/ogg_util.py: pages at arbitrary byte
    alignment (junk between them), empty and tiny pages, a maximum-size page (255 segments of 255 bytes), bit flips in the
    header, the body and the checksum field, truncated pages, wrong capture pattern / version."""
    import random
    import ogg_util
    rng = random.Random(123)

    def expect(pg):
        if len(pg) < 27 or pg[:4] != b"OggS" or pg[4] != 0:
            return pkg.PAGE_BAD_CAPTURE
        total = 27 + pg[26] + sum(pg[27:27 + pg[26]]) if len(pg) >= 27 + pg[26] else None
        if total is None or len(pg) < total:
            return pkg.PAGE_BAD_CAPTURE
        want = int.from_bytes(pg[22:26], "little")
        return int(ogg_util.ogg_crc(pg[:22] + bytes(4) + pg[26:total]) == want)

    pages = []
    for i in range(2500):
        packets = [bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 1, 7, 30, 161, 254, 255, 256, 700]))) for _ in range(rng.randrange(0, 9))]
        pg = bytearray(ogg_util.page(rng.getrandbits(32), i, rng.getrandbits(40), packets))
        how = rng.random()
        if how < 0.15:
            pg[rng.randrange(len(pg))] ^= 1 << rng.randrange(8)          # anywhere, incl. capture pattern and checksum
        elif how < 0.2:
            pg = pg[:rng.randrange(0, len(pg))]                           # truncated
        elif how < 0.23:
            pg[4] = 1                                                     # stream structure version
        elif how < 0.3:
            pg += bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 50)))  # trailing bytes beyond the page: ignored
        pages.append(bytes(pg))
    pages.append(ogg_util.page(7, 7, 7, [bytes(rng.getrandbits(8) for _ in range(255 * 254 + 200))]))  # 255 lacing values: the largest page
    assert len(pages[-1]) > 65000
    blob = bytearray()
    offs, lens = [], []
    for pg in pages:
        blob += bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 9)))  # arbitrary alignment
        offs.append(len(blob))
        lens.append(len(pg))
        blob += pg
    want = np.array([expect(pg) for pg in pages], dtype=np.int32)
    assert (want == 1).sum() > 1000 and (want == 0).sum() > 100 and (want == pkg.PAGE_BAD_CAPTURE).sum() > 50
    n = len(pages)
    ctx = gpu_ctx
    d_blob, d_offs, d_lens, d_st = ctx.dev_alloc(len(blob) + 64), ctx.dev_alloc(8 * n), ctx.dev_alloc(4 * n), ctx.dev_alloc(4 * n)
    try:
        ctx.h2d(d_offs, np.array(offs, dtype=np.int64))
        ctx.h2d(d_lens, np.array(lens, dtype=np.int32))
        # the blob may start at any address (a view into a larger buffer), and nothing follows its last page
        for shift in (0, 1, 13, 16, 63):
            base = ctypes.c_void_p(d_blob.value + shift)
            ctx.h2d(base, np.frombuffer(bytes(blob), dtype=np.uint8))
            got = np.full(n, 99, dtype=np.int32)
            ctx.h2d(d_st, got)
            ctx.pages_crc_device(n, base, d_offs, d_lens, d_st)
            ctx.synchronize()
            ctx.d2h(got, d_st)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (shift, bad[:5], got[bad[:5]], want[bad[:5]], [lens[i] for i in bad[:5]])
    finally:
        for p in (d_blob, d_offs, d_lens, d_st):
            ctx.dev_free(p)


def test_demux_with_gpu_checksums_equals_demux_with_host_checksums(pkg, gpu_ctx):
    """PageBatch.with_gpu_crc (checksums by opusgpu_pages_crc_device, host demux without its CRC pass) makes the same
    decode steps, arena and page reports as the host demux with OPUSGPU_PAGES_VERIFY_CRC -- on pages some of which are
    damaged (body, header, checksum field, capture pattern, truncation) or carry a bad stream id."""
    rng = np.random.default_rng(5)
    n = 3000
    modes = ((pkg.TOC_SILK_NB_STEREO, 40), (pkg.TOC_HYBRID_FB_STEREO, 120), (pkg.TOC_CELT_FB_STEREO, 160))
    mats, ids = [], []
    for m, (toc, L) in enumerate(modes):
        sid = np.arange(m, n, 3, dtype=np.int32)
        pay = pkg.lcg_payloads(len(sid), 4, L, seed_base=31 + m)
        mats.append(pkg.build_pages(toc, pay, sid.astype(np.uint32), seqno=2))
        ids.append(sid)
    blob = np.concatenate([x.reshape(-1) for x in mats]).copy()
    lens = np.concatenate([np.full(x.shape[0], x.shape[1], dtype=np.int32) for x in mats])
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
    sids = np.concatenate(ids).copy()
    hit = rng.choice(n, size=600, replace=False)
    for j, i in enumerate(hit):
        kind = j % 6
        if kind == 0:
            blob[offs[i] + rng.integers(27 + 4, lens[i])] ^= 1 << rng.integers(8)   # body (or lacing table)
        elif kind == 1:
            blob[offs[i] + rng.integers(5, 22)] ^= 1 << rng.integers(8)           # header fields
        elif kind == 2:
            blob[offs[i] + rng.integers(22, 26)] ^= 1 << rng.integers(8)          # the checksum itself
        elif kind == 3:
            blob[offs[i] + rng.integers(0, 5)] ^= 1 << rng.integers(8)            # capture pattern / version
        elif kind == 4:
            lens[i] = rng.integers(0, lens[i])                                     # truncated
        else:
            sids[i] = -1                                                           # no decoder stream
    ctx = gpu_ctx
    d_blob = ctx.dev_alloc(blob.size)
    try:
        ctx.h2d(d_blob, blob)
        a = pkg.PageBatch(blob, offs, lens, sids, flags=pkg.PAGES_VERIFY_CRC | pkg.PAGES_GROUP_BY_MODE, threads=4)
        b = pkg.PageBatch.with_gpu_crc(ctx, d_blob, blob, offs, lens, sids, flags=pkg.PAGES_GROUP_BY_MODE, threads=4)
        st = a.info["status"]
        assert (st == pkg.PAGE_BAD_CRC).sum() >= 250 and (st == pkg.PAGE_BAD_CAPTURE).sum() >= 100 and (st == 4).sum() >= n - 600
        for name in a.info.dtype.names:
            assert np.array_equal(a.info[name], b.info[name]), name
        assert a.n_steps == b.n_steps == 4
        assert np.array_equal(a.arena, b.arena)
        for k in range(a.n_steps):
            (da, pa), (db, pb) = a.step(k), b.step(k)
            assert np.array_equal(da, db) and np.array_equal(pa, pb)
        a.close()
        b.close()
    finally:
        ctx.dev_free(d_blob)


def _ingest_mod(pkg):
    import sys
    name = pkg.__name__ + ".ingest"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "esp32-opus-player_amd", "ingest.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("n,depth,keeps_kind,by_kind", [(3 * 2048, 2, False, False), (3 * 700, 3, False, False), (3 * 2048 + 2, 3, True, False),
                                                        (3 * 2048 + 1, 2, True, True)])
def test_overlapped_ingest_matches_the_oracle(pkg, oracle, gpu_ctx, n, depth, keeps_kind, by_kind):
    """esp32-opus-player_amd/ingest.py: five batches of one page per stream, demuxed and uploaded by a second host thread on the copy
    stream while the batch before decodes (ring of `depth` device slots, so slots are reused); after every batch the PCM of its
    last step -- which depends on every step before it -- is compared with the oracle, every stream, every sample."""
    ctx = gpu_ctx
    modes = ((pkg.TOC_SILK_NB_STEREO, 40), (pkg.TOC_HYBRID_FB_STEREO, 120), (pkg.TOC_CELT_FB_STEREO, 160))
    nb, ppp = 5, 4
    frames = nb * ppp
    ids_of = [np.arange(m, n, 3, dtype=np.int32) for m in range(3)]
    ref, per_batch = [None] * 3, [[] for _ in range(nb)]
    for m, (toc, L) in enumerate(modes):
        pay = pkg.lcg_payloads(len(ids_of[m]), frames, L, seed_base=0x1234567 + m)
        ref[m], ok = oracle.batch_decode_threads(2, toc, pay)
        assert ok == len(ids_of[m]) * frames
        for q in range(nb):
            per_batch[q].append((pkg.build_pages(toc, pay[q * ppp:(q + 1) * ppp], ids_of[m].astype(np.uint32) + 5, seqno=2 + q), ids_of[m]))
    batches = []
    for q in range(nb):
        blob = np.concatenate([pg.reshape(-1) for pg, _ in per_batch[q]])
        lens = np.concatenate([np.full(pg.shape[0], pg.shape[1], dtype=np.int32) for pg, _ in per_batch[q]])
        offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
        batches.append((blob, offs, lens, np.concatenate([i for _, i in per_batch[q]])))
    ctx.streams_alloc(n, 2)
    d_pcm, d_res = ctx.dev_alloc(n * 960 * 2 * 2), ctx.dev_alloc(4 * n)
    # keeps_kind: pipelined steps that carry OPUSGPU_STEP_KEEPS_MODE (a stream's mode is fixed here); by_kind: as declared sub-steps
    ctx.set_pipeline(keeps_kind)
    pipe = _ingest_mod(pkg).OverlappedPageDecode(ctx, threads=3, depth=depth, keeps_mode=keeps_kind, by_kind=by_kind,
                                                 page_flags=pkg.PAGES_VERIFY_CRC | pkg.PAGES_GROUP_BY_MODE)  # (check() below knows the slots' streams from the page order)
    seen = []

    def check(b):  # (called on the decoding thread between batches: reading back waits for the batch's steps)
        out, res = np.zeros((n, 960, 2), dtype=np.int16), np.zeros(n, dtype=np.int32)
        ctx.d2h(out, d_pcm)
        ctx.d2h(res, d_res)
        assert (res == 960).all()
        k = (b + 1) * ppp - 1
        # step tables are grouped by mode, streams in page order within a mode: SILK-NB, hybrid, CELT
        order = np.concatenate(ids_of)
        for m in range(3):
            slots = np.nonzero(order % 3 == m)[0]
            assert np.array_equal(out[slots], ref[m][order[slots] // 3, k]), (b, m)
        seen.append(b)

    try:
        st = pipe.run(batches, d_pcm, d_res, on_batch_done=check)
    finally:
        pipe.close()
        ctx.set_pipeline(False)
        ctx.dev_free(d_pcm)
        ctx.dev_free(d_res)
    assert seen == list(range(nb)) and st["steps"] == frames and st["pages"] == nb * n

"""Host-side Ogg page ingest (opusgpu_pages_demux, include/opusgpu.h): pages -> decode steps.  No GPU involved.
The pages come from tests/ogg_util.py, an independent pure-Python model of the page layout and CRC."""
import random

import numpy as np
import pytest

import ogg_util


def _batch(pkg, pages, ids, **kw):
    blob = np.frombuffer(b"".join(pages), dtype=np.uint8)
    lens = np.array([len(p) for p in pages], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]) if len(pages) else np.zeros(0, np.int64)
    return pkg.PageBatch(blob, offs, lens, ids, **kw)


def _packet(rng, toc, n):
    return bytes([toc]) + bytes(rng.getrandbits(8) for _ in range(n))


def test_pages_to_steps_match_the_packet_split(pkg):
    """Every frame the per-packet splitter finds shows up in the right step with the right bytes, for code-0/1/2/3
    packets, packets longer than one lacing value, and several packets per page."""
    rng = random.Random(1)
    pk_a = [_packet(rng, 0xFC, 160), _packet(rng, 0xFD, 200), _packet(rng, 0x0C, 40)]          # 1 + 2 + 1 frames
    pk_b = [_packet(rng, 0x7C, 300), bytes([0x7F, 0x03]) + bytes(rng.getrandbits(8) for _ in range(90))]  # 1 + 3 (CBR code 3)
    pk_c = [bytes([0xFE, 30]) + bytes(rng.getrandbits(8) for _ in range(75))]                   # code 2: 30 + 45
    pages = [ogg_util.page(0x11, 5, 1234, pk_a), ogg_util.page(0x22, 9, -1, pk_b, eos=True), ogg_util.page(0x33, 2, 77, pk_c, bos=True)]
    ids = [7, 3, 5]
    b = _batch(pkg, pages, ids, flags=pkg.PAGES_VERIFY_CRC)
    assert list(b.info["status"]) == [4, 4, 2]
    assert list(b.info["packets"]) == [3, 2, 1]
    assert list(b.info["serial"]) == [0x11, 0x22, 0x33] and list(b.info["seqno"]) == [5, 9, 2]
    assert list(b.info["granulepos"]) == [1234, -1, 77] and list(b.info["header_type"]) == [0, 4, 2]
    assert b.n_steps == 4
    want = {}  # (page, frame index) -> (bytes, flags)
    for i, pks in enumerate((pk_a, pk_b, pk_c)):
        k = 0
        for p in pks:
            for off, ln, fl in pkg.packet_to_frames(p, ids[i]):
                want[(i, k)] = (p[off:off + ln], fl)
                k += 1
    seen = {}
    for s in range(b.n_steps):
        descs, slot_pages = b.step(s)
        assert len(set(descs["stream"])) == len(descs)  # a stream at most once per step
        assert list(slot_pages) == sorted(slot_pages)   # input order without grouping
        for d, pg in zip(descs, slot_pages):
            assert d["stream"] == ids[pg]
            seen[(int(pg), s)] = (bytes(b.arena[d["offset"]:d["offset"] + d["len"]]), int(d["flags"]))
    assert seen == want
    b.close()


def test_order_by_header_sorts_each_mode_group_by_the_lbrr_flags(pkg):
    """OPUSGPU_PAGES_ORDER_BY_HEADER: per step the table is grouped by mode, and inside the SILK-only and the hybrid group the
    frames come in the order of their LBRR flags (bits 6 and 4 of the first payload byte; bit 6 alone for a mono frame), pages in
    input order among equals; the CELT-only group keeps input order; no frame is lost or changed."""
    rng = random.Random(5)
    pages, ids = [], []
    for s in range(300):
        toc = rng.choice([0x0C, 0x08, 0x7C, 0x78, 0xFC, 0x4C])
        pages.append(ogg_util.page(100 + s, 2, 0, [_packet(rng, toc, rng.choice([0, 1, 30, 60])) for _ in range(rng.randrange(1, 4))]))
        ids.append(s)
    plain = _batch(pkg, pages, ids, flags=pkg.PAGES_VERIFY_CRC)
    b = _batch(pkg, pages, ids, flags=pkg.PAGES_VERIFY_CRC | pkg.PAGES_ORDER_BY_HEADER, threads=3)
    assert b.n_steps == plain.n_steps
    for s in range(b.n_steps):
        descs, slot_pages = b.step(s)
        d0, p0 = plain.step(s)
        frame = lambda bb, d: (int(d["stream"]), bytes(bb.arena[d["offset"]:d["offset"] + d["len"]]), int(d["flags"]))
        assert sorted(frame(b, d) for d in descs) == sorted(frame(plain, d) for d in d0)
        mode = descs["flags"] & 3
        assert (np.diff(mode.astype(int)) >= 0).all()
        for m in range(3):
            sel = np.nonzero(mode == m)[0]
            keys = []
            for j in sel:
                d = descs[j]
                b0 = int(b.arena[d["offset"]]) if d["len"] > 0 else 0
                keys.append((((b0 >> 6) & 1) | ((((b0 >> 4) & 1) << 1) if d["flags"] & 32 else 0)) if (m < 2 and d["len"] > 0) else 0)
            order = list(zip(keys, slot_pages[sel]))
            assert order == sorted(order), (s, m)  # by key, input order among equals
    plain.close()
    b.close()


def test_bad_pages_are_reported_per_page(pkg):
    rng = random.Random(2)
    good = ogg_util.page(1, 0, 0, [_packet(rng, 0xFC, 50)])
    bad_crc = bytearray(good); bad_crc[40] ^= 1
    bad_cap = b"OggX" + good[4:]
    bad_ver = good[:4] + b"\x01" + good[5:]
    truncated = good[:-3]
    continued = ogg_util.page(1, 0, 0, [_packet(rng, 0xFC, 50)], continued=True)
    spans = ogg_util.page(1, 0, 0, [_packet(rng, 0xFC, 254)])  # 255 bytes: lacing 255, 0 -- terminated, fine
    open_end = bytearray(ogg_util.page(1, 0, 0, [_packet(rng, 0xFC, 254)]))
    # drop the terminating 0 lacing value: the packet now continues on the next page
    open_end = bytes(open_end[:26]) + bytes([1]) + bytes([255]) + bytes(open_end[29:])
    open_end = bytearray(open_end); open_end[22:26] = b"\0\0\0\0"
    open_end[22:26] = ogg_util.ogg_crc(bytes(open_end)).to_bytes(4, "little")
    bad_split = ogg_util.page(1, 0, 0, [bytes([0xFE, 200]) + bytes(10)])      # code 2, first frame longer than the packet
    skipped = ogg_util.page(1, 0, 0, [bytes([0xFF, 0x00]), _packet(rng, 0xFC, 20), b""])  # code 3 with 0 frames; empty packet
    pages = [good, bytes(bad_crc), bad_cap, bad_ver, truncated, continued, spans, bytes(open_end), bad_split, skipped, good]
    ids = list(range(len(pages) - 1)) + [-4]
    b = _batch(pkg, pages, ids, flags=pkg.PAGES_VERIFY_CRC)
    assert list(b.info["status"]) == [1, pkg.PAGE_BAD_CRC, pkg.PAGE_BAD_CAPTURE, pkg.PAGE_BAD_CAPTURE, pkg.PAGE_BAD_CAPTURE,
                                      pkg.PAGE_SPANS, 1, pkg.PAGE_SPANS, pkg.PAGE_BAD_PACKET, 1, pkg.PAGE_BAD_STREAM]
    assert b.info["packets"][9] == 3  # three packets seen, one kept
    descs, slot_pages = b.step(0)
    assert list(slot_pages) == [0, 6, 9] and b.n_steps == 1
    b.close()
    b = _batch(pkg, pages[:2], ids[:2], flags=0)  # without verification the flipped body bit goes through
    assert list(b.info["status"]) == [1, 1]
    b.close()


def test_pages_of_one_stream_chain_and_modes_group(pkg):
    rng = random.Random(3)
    tocs = {0: 0x0C, 1: 0x7C, 2: 0xFC}  # SILK-NB, hybrid FB, CELT FB (stereo, 20 ms)
    pages, ids, modes = [], [], []
    for i in range(40):
        m = rng.randrange(3)
        pages.append(ogg_util.page(100 + i, 0, 0, [_packet(rng, tocs[m], 30) for _ in range(3)]))
        ids.append(i)
        modes.append(m)
    pages.append(ogg_util.page(100 + 5, 1, 0, [_packet(rng, tocs[modes[5]], 30) for _ in range(2)]))  # second page of stream 5
    ids.append(5)
    modes.append(modes[5])
    b = _batch(pkg, pages, ids)  # default flags: verify + group by mode
    assert b.n_steps == 5 and b.info["first_step"][40] == 3 and (b.info["first_step"][:40] == 0).all()
    for s in range(5):
        descs, slot_pages = b.step(s)
        assert len(descs) == (40 if s < 3 else 1)
        got = [int(f) & 3 for f in descs["flags"]]
        assert got == sorted(got) and got == [modes[p] for p in slot_pages]
        for m in range(3):  # stable inside a group
            grp = [int(p) for p, g in zip(slot_pages, got) if g == m]
            assert grp == sorted(grp)
        assert len(set(descs["stream"])) == len(descs)
    b.close()


def test_vectorised_page_builder_and_threads(pkg):
    """build_pages (numpy, used by the bench) writes the same bytes as the pure-Python model, and the demux gives the
    same steps with 1 and with 4 threads."""
    n, npk, L = 1500, 4, 21
    pay = pkg.lcg_payloads(n, npk, L)
    serials = np.arange(n, dtype=np.uint32) + 1000
    pages = pkg.build_pages(0xFC, pay, serials, seqno=2)
    for i in (0, 1, 777, n - 1):
        ref = ogg_util.page(int(serials[i]), 2, npk * 960, [bytes([0xFC]) + pay[k, i].tobytes() for k in range(npk)])
        assert pages[i].tobytes() == ref
    lens = np.full(n, pages.shape[1], dtype=np.int32)
    offs = np.arange(n, dtype=np.int64) * pages.shape[1]
    ids = np.arange(n, dtype=np.int32)[::-1].copy()
    b1 = pkg.PageBatch(pages.reshape(-1), offs, lens, ids, threads=1)
    b4 = pkg.PageBatch(pages.reshape(-1), offs, lens, ids, threads=4)
    assert (b1.info == b4.info).all() and (b1.info["status"] == npk).all() and b1.n_steps == b4.n_steps == npk
    assert (b1.arena == b4.arena).all()
    for s in range(npk):
        d1, p1 = b1.step(s)
        d4, p4 = b4.step(s)
        assert (d1 == d4).all() and (p1 == p4).all() and len(d1) == n
        k = 123
        assert bytes(b1.arena[d1["offset"][k]:d1["offset"][k] + L]) == pay[s, p1[k]].tobytes() and d1["stream"][k] == ids[p1[k]]
    b1.close(); b4.close()


def test_empty_batch(pkg):
    b = pkg.PageBatch(np.zeros(0, np.uint8), [], [], [])
    assert b.n_steps == 0
    with pytest.raises(IndexError):
        b.step(0)
    b.close()


def test_page_with_more_than_32_frames_groups_correctly(pkg):
    """The demux keeps the modes of a page's first 32 frames from its first scan and re-scans longer pages: a page of 45
    frames with alternating modes must still come out grouped, frame k in step k."""
    rng = random.Random(5)
    tocs = [0x0C, 0x7C, 0xFC]
    long_modes = [rng.randrange(3) for _ in range(45)]
    long_page = ogg_util.page(9, 0, 0, [_packet(rng, tocs[m], 5) for m in long_modes])
    short_modes = [[rng.randrange(3) for _ in range(3)] for _ in range(6)]
    pages = [ogg_util.page(20 + i, 0, 0, [_packet(rng, tocs[m], 5) for m in ms]) for i, ms in enumerate(short_modes)]
    pages.insert(3, long_page)
    ids = list(range(len(pages)))
    b = _batch(pkg, pages, ids)
    assert b.info["status"][3] == 45 and b.n_steps == 45
    for s in range(45):
        descs, slot_pages = b.step(s)
        got = [int(f) & 3 for f in descs["flags"]]
        assert got == sorted(got)
        want = {3: long_modes[s]}
        for i, ms in enumerate(short_modes):
            if s < 3:
                want[i if i < 3 else i + 1] = ms[s]
        assert {int(p): g for p, g in zip(slot_pages, got)} == want
    b.close()


@pytest.mark.parametrize("id_scale", [1, 100003])  # dense stream ids (table) and sparse ones (hash map)
def test_slot_order_is_the_stable_order_for_any_thread_count(pkg, id_scale):
    """Mixed modes, 0 - 6 packets per page, up to three pages per stream, some pages damaged: slot s of step k is what a
    stable sort of the frames by (step, mode) gives, whatever the number of threads (the slot numbering is done by
    page ranges in parallel)."""
    rng = random.Random(11)
    tocs = {0: 0x0C, 1: 0x7C, 2: 0xFC}
    n_streams = 1200
    mode_of_stream = [rng.randrange(3) for _ in range(n_streams)]
    pages, ids = [], []
    for rnd in range(3):
        for s in rng.sample(range(n_streams), n_streams - 200 * rnd):
            pg = bytearray(ogg_util.page(s, rnd, 0, [_packet(rng, tocs[mode_of_stream[s]], rng.randrange(2, 60)) for _ in range(rng.randrange(0, 7))]))
            if rng.random() < 0.05:
                pg[rng.randrange(len(pg))] ^= 0x10
            pages.append(bytes(pg))
            ids.append(s * id_scale)
    b1 = _batch(pkg, pages, ids, threads=1)
    nxt, want = {}, {}
    for i, st in enumerate(b1.info["status"]):
        if st <= 0:
            continue
        at = nxt.get(ids[i], 0)
        assert b1.info["first_step"][i] == at
        for k in range(st):
            want.setdefault((at + k, mode_of_stream[ids[i] // id_scale]), []).append(i)
        nxt[ids[i]] = at + st
    assert (b1.info["status"] < 0).sum() > 50 and b1.n_steps == max(k for k, _ in want) + 1 and b1.n_steps > 10
    for threads in (1, 3, 7):
        b = b1 if threads == 1 else _batch(pkg, pages, ids, threads=threads)
        assert (b.info == b1.info).all() and b.n_steps == b1.n_steps and (b.arena == b1.arena).all()
        for k in range(b.n_steps):
            descs, slot_pages = b.step(k)
            assert list(slot_pages) == [i for m in range(3) for i in want.get((k, m), [])]
            d1, _ = b1.step(k)
            assert (descs == d1).all()
        if b is not b1:
            b.close()
    b1.close()


def test_demux_into_caller_memory_makes_the_same_tables_and_arena(pkg):
    """opusgpu_pages_demux_into (the overlapped ingest's demux: output in the caller's page-locked slot, uploadable in one copy)
    against opusgpu_pages_demux; a slot that is too small is reported with the size needed and left untouched."""
    n, ppp = 300, 7
    parts = []
    for m, (toc, L) in enumerate(((pkg.TOC_SILK_NB_STEREO, 40), (pkg.TOC_HYBRID_FB_STEREO, 120), (pkg.TOC_CELT_FB_STEREO, 160))):
        ids = np.arange(m, n, 3, dtype=np.int32)
        pay = pkg.lcg_payloads(len(ids), ppp, L, seed_base=99 + m)
        parts.append((pkg.build_pages(toc, pay, ids.astype(np.uint32) + 3), ids))
    blob = np.concatenate([pg.reshape(-1) for pg, _ in parts])
    lens = np.concatenate([np.full(pg.shape[0], pg.shape[1], dtype=np.int32) for pg, _ in parts])
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
    ids = np.concatenate([i for _, i in parts])
    a = pkg.PageBatch(blob, offs, lens, ids, threads=2)
    small = np.full(1024, 0xAB, dtype=np.uint8)
    with pytest.raises(pkg.BufferTooSmall) as e:
        pkg.PageBatch(blob, offs, lens, ids, threads=2, out_mem=small)
    assert (small == 0xAB).all()
    raw = np.zeros(e.value.need + 64, dtype=np.uint8)
    mem = raw[(-raw.ctypes.data) % 16:][:e.value.need]
    b = pkg.PageBatch(blob, offs, lens, ids, threads=2, out_mem=mem)
    assert b.n_steps == a.n_steps == ppp and np.array_equal(a.info, b.info)
    assert b.arena_offset % 256 == 0 and b.image.nbytes == e.value.need
    assert np.array_equal(a.arena, b.arena) and np.array_equal(b.arena, mem[b.arena_offset:b.arena_offset + b.arena.size])
    at = 0
    for k in range(ppp):
        da, pa = a.step(k)
        db, pb_ = b.step(k)
        assert np.array_equal(da, db) and np.array_equal(pa, pb_)
        assert np.array_equal(mem[at:at + 16 * len(db)].view(pkg.DESC_DTYPE), db)  # tables lie step after step from the start
        at += 16 * len(db)
    a.close()
    b.close()

"""Worker of test_multirank_gloo.py::test_work_queue_scatter: one rank of a world_size-2 gloo job on CPU.

Rank 0 is the ingest point: it holds Ogg pages for the streams of BOTH ranks (modes mixed across streams), routes
them by owner, turns each rank's pages into decode steps (opusgpu_pages_demux) and scatters the packed work.  Every
rank then rebuilds what its share must be from the same seeds and compares byte for byte."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

pkg = bench.load_pkg()
shard = bench.load_shard()
ranks = shard.Ranks(backend="gloo")
m, npk = int(os.environ.get("OG_TEST_STREAMS", "30")), 4  # streams per rank, packets per page
TOCS = (pkg.TOC_SILK_NB_STEREO, pkg.TOC_HYBRID_FB_STEREO, pkg.TOC_CELT_FB_STEREO)
LENS = (40, 120, 160)


def pages_of(global_ids):
    """The pages of these global streams, modes 1:1:1 by global id: [(page bytes, global id)] in id order."""
    out = []
    for mode in range(3):
        ids = [g for g in global_ids if g % 3 == mode]
        if not ids:
            continue
        pay = pkg.lcg_payloads(ranks.world * m, npk, LENS[mode])[:, ids]
        pg = pkg.build_pages(TOCS[mode], pay, np.array(ids, dtype=np.uint32) + 5000)
        out += [(pg[k].tobytes(), g) for k, g in enumerate(ids)]
    return sorted(out, key=lambda x: x[1])


def work_for(rank):
    items = pages_of(range(rank * m, (rank + 1) * m))
    blob = np.frombuffer(b"".join(p for p, _ in items), dtype=np.uint8)
    lens = np.array([len(p) for p, _ in items], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    local = np.array([g - rank * m for _, g in items], dtype=np.int32)
    assert all(ranks.owner_of(g, m) == rank for _, g in items)
    b = pkg.PageBatch(blob, offs, lens, local)
    assert (b.info["status"] == npk).all()
    buf = shard.pack_work(b)
    b.close()
    return buf


def raw_for(rank):
    items = pages_of(range(rank * m, (rank + 1) * m))
    blob = np.frombuffer(b"".join(p for p, _ in items), dtype=np.uint8)
    lens = np.array([len(p) for p, _ in items], dtype=np.int32)
    return shard.pack_pages(blob, lens, np.array([g - rank * m for _, g in items], dtype=np.int32))


# second ingest mode: raw pages are scattered and every rank demuxes its own share; the result must be the same work
raw = ranks.scatter_bytes([raw_for(r) for r in range(ranks.world)] if ranks.rank == 0 else None, src=0)
blob, offs, lens, local = shard.unpack_pages(raw)
b2 = pkg.PageBatch(blob, offs, lens, local)
per_rank_work = shard.pack_work(b2)
b2.close()

buffers = [work_for(r) for r in range(ranks.world)] if ranks.rank == 0 else None
mine = ranks.scatter_bytes(buffers, src=0)
want = work_for(ranks.rank)
lay = shard.WorkLayout(mine)
modes = []
for k in range(lay.n_steps):
    d = np.frombuffer(mine[lay.desc_at[k]:lay.desc_at[k] + 16 * lay.counts[k]].tobytes(), dtype=pkg.DESC_DTYPE)
    modes.append([int(f) & 3 for f in d["flags"]])
out = {"rank": ranks.rank, "same": bool(mine.size == want.size and (mine == want).all()), "size": int(mine.size),
       "n_steps": lay.n_steps, "counts": lay.counts, "grouped": all(x == sorted(x) for x in modes),
       "crc": zlib.crc32(mine.tobytes()), "nbytes": lay.nbytes,
       "per_rank_same": bool(per_rank_work.size == want.size and (per_rank_work == want).all())}
with open(os.path.join(os.environ["OG_TEST_OUT"], f"pages_rank{ranks.rank}.json"), "w") as fh:
    json.dump(out, fh)
ranks.close()

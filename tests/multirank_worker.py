"""Worker of test_multirank_gloo.py: one rank of a world_size-2 gloo job on CPU.

Exercises the N>1 plumbing of bench.py (esp32-opus-player_amd/shard.py): stream partition, per-rank payload
seeds, barrier + MAX-over-ranks timing, whole-job aggregation.  The decode itself is played by the CPU oracle
(test infrastructure) on a handful of streams -- there is no GPU here."""
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import oracle_py  # noqa: E402

pkg = bench.load_pkg()
shard = bench.load_shard()
ranks = shard.Ranks(backend="gloo")
n, frames, L = 6, 3, 160
lo, hi = ranks.stream_range(n)
pay = pkg.lcg_payloads(n, frames, L, seed_base=ranks.seed_base())
o = oracle_py.load()
ranks.barrier()
t0 = time.perf_counter()
pcm, ok = o.batch_decode(2, pkg.TOC_CELT_FB_STEREO, pay, want_pcm=True)
time.sleep(0.05 * (ranks.rank + 1))  # make the ranks' elapsed times differ
dt = time.perf_counter() - t0
ranks.barrier()
value, dt_max, total = shard.aggregate_throughput(ranks, ok, dt)
out = {"rank": ranks.rank, "world": ranks.world, "lo": lo, "hi": hi, "ok": int(ok), "dt": dt, "dt_max": dt_max,
       "total": total, "value": value, "crc": zlib.crc32(np.ascontiguousarray(pcm).tobytes()),
       "pay_crc": zlib.crc32(pay.tobytes())}
with open(os.path.join(os.environ["OG_TEST_OUT"], f"rank{ranks.rank}.json"), "w") as fh:
    json.dump(out, fh)
ranks.close()

import sys, os, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import load_pkg
import oracle_py
pkg = load_pkg(); o = oracle_py.load()
ctx = pkg.Context(0)
for name, toc, L in (("celt", 0xFC, 160), ("silk", 0x0C, 40), ("hyb", 0x7C, 120)):
    n = 8
    pay = pkg.lcg_payloads(n, 2, L)
    ref, ok = o.batch_decode(2, toc, pay)
    ctx.streams_alloc(n, 2)
    for f in range(2):
        pk = [bytes([toc]) + pay[f, s].tobytes() for s in range(n)]
        pcm, res = ctx.decode_packets(np.arange(n), pk)
        d = (pcm != ref[:, f])
        print(name, "frame", f, "res", res[:4], "ndiff per stream", d.reshape(n, -1).sum(axis=1))
        if d.any():
            s = int(np.nonzero(d.reshape(n, -1).any(axis=1))[0][0])
            idx = np.nonzero(d[s].reshape(-1))[0]
            print("  stream", s, "first diffs", idx[:10], "gpu", pcm[s].reshape(-1)[idx[:6]], "ref", ref[s, f].reshape(-1)[idx[:6]])

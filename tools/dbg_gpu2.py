import sys, os, ctypes as C, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import load_pkg
import oracle_py
pkg = load_pkg(); o = oracle_py.load()
o.lib.oc_taps_enable.argtypes=[C.c_void_p]; o.lib.oc_taps_copy.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_void_p]
ctx = pkg.Context(0)
n=4; toc=0xFC; L=160
pay = pkg.lcg_payloads(n, 1, L)
ctx.streams_alloc(n, 2)
pk = [bytes([toc]) + pay[0, s].tobytes() for s in range(n)]
pcm, res = ctx.decode_packets(np.arange(n), pk)
for s in range(n):
    d = o.decoder(2); o.lib.oc_taps_enable(d.h); d.decode(pk[s])
    X = np.zeros(1920, dtype=np.int16); o.lib.oc_taps_copy(d.h, 0, 0, X.ctypes.data)
    g = pcm[s].reshape(-1)
    diff = np.nonzero(g != X)[0]
    print("stream", s, "X diffs", diff.size, "first", diff[:12], "gpu", g[diff[:6]], "ref", X[diff[:6]])

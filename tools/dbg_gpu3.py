import sys, os, ctypes as C, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import load_pkg
pkg = load_pkg()
ctx = pkg.Context(0)
n=2; toc=0xFC; L=160
pay = pkg.lcg_payloads(n, 1, L)
ctx.streams_alloc(n, 2)
pk = [bytes([toc]) + pay[0, s].tobytes() for s in range(n)]
pcm, res = ctx.decode_packets(np.arange(n), pk)
for s in range(n):
    g = pcm[s].reshape(-1)
    print("stream", s, "pulses", g[0:21].tolist()); print(" fine_quant", g[32:53].tolist()); print(" tf_res", g[64:85].tolist()); print(" cap", g[96:117].tolist()); print(" offsets", g[128:149].tolist())
    print(" coded,intensity,dual,spread,transient,balance,trim,tell,intra,pitch,silence", g[160:171].tolist()); print(" bandE", g[192:234].tolist())
    print(" thresh", g[300:321].tolist()); print(" trim_off", g[332:353].tolist()); print(" bits1", g[364:385].tolist()); print(" bits2", g[396:417].tolist())
    print(" total,lo,hi,skip_start,int_rsv,ds_rsv,skip_rsv,trim, band_alloc[5][3], log2frac[21], eband[21]", g[428:439].tolist())

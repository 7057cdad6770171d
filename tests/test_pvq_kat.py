"""Unit known-answer test ("cwrsi over all (N,K) reachable"): the kernels' PVQ leaf decode -- codeword index ->
pulses -> scaled to the leaf gain -> spreading rotation undone -> collapse mask -- in host emulation against the oracle's
alg_unquant, for every (N, K) the pulse cache of the 48 kHz mode can produce, over several indices (first, last, random),
block counts, spread settings and gains.  CPU only."""
import ctypes as C
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tables():
    spec = importlib.util.spec_from_file_location("gen_rom", os.path.join(ROOT, "tools", "gen_rom_tables.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _get_pulses(q):
    return q if q < 8 else (8 + (q & 7)) << ((q >> 3) - 1)


def reachable_nk():
    """(N, K) pairs of the pulse cache (compute_pulse_cache layout: index[(LM + 1) * nbEBands + band], LM = -1 .. 3)."""
    t = _tables()
    nb = 21
    out = set()
    for lm in range(-1, 4):
        for i in range(nb):
            width = t.EBAND[i + 1] - t.EBAND[i]
            n = width << lm if lm >= 0 else width >> 1
            if n < 1 or (lm < 0 and width & 1):
                continue
            idx = t.PULSE_IDX[(lm + 1) * nb + i]
            if idx < 0:
                continue
            for q in range(1, t.PULSE_BITS[idx] + 1):
                out.add((n, _get_pulses(q)))
    return sorted(out)


def _v(n, k, memo={}):  # V(n, k) = number of PVQ codewords = U(n, k) + U(n, k + 1)
    def u(a, b):
        if (a, b) in memo:
            return memo[(a, b)]
        if a == 0 or b == 0:
            r = 1 if (a == 0 and b == 0) else 0
        else:
            r = u(a - 1, b) + u(a, b - 1) + u(a - 1, b - 1)
        memo[(a, b)] = r
        return r
    import sys
    sys.setrecursionlimit(10000)
    return u(n, k) + u(n, k + 1)


def test_pvq_leaf_over_all_reachable_n_k(oracle):
    emu = C.CDLL(os.path.join(ROOT, "tests", "emul", "libog_emul.so"))
    emu.emu_pvq_leaf.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p]
    emu.emu_pvq_leaf.restype = C.c_uint
    oracle.lib.oc_test_pvq_leaf.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p]
    oracle.lib.oc_test_pvq_leaf.restype = C.c_uint
    rng = np.random.default_rng(8)
    pairs = [(n, k) for n, k in reachable_nk() if n >= 2]  # N = 1 bands carry a sign bit, not a PVQ codeword
    assert len(pairs) > 300 and max(n for n, _ in pairs) == 176
    a, b = np.zeros(176, dtype=np.int16), np.zeros(176, dtype=np.int16)
    checked = 0
    for n, k in pairs:
        v = _v(n, k)
        assert 0 < v <= 0xFFFFFFFF, (n, k, v)  # a legal (N, K) has a 32-bit codebook
        idxs = {0, v - 1, v // 2} | {int(x) for x in rng.integers(0, v, 4)}
        blocks = [bb for bb in (1, 2, 4, 8, 16) if n % bb == 0 and n // bb >= 1]
        for index in idxs:
            for B in blocks:
                for spread in (0, 1, 2, 3):
                    gain = int(rng.choice([32767, 16384, 23170, 1, 12345]))
                    ma = emu.emu_pvq_leaf(n, k, index, B, gain, spread, a.ctypes.data)
                    mb = oracle.lib.oc_test_pvq_leaf(n, k, index, spread, B, gain, b.ctypes.data)
                    assert ma == mb, ("mask", n, k, index, B, spread, gain, ma, mb)
                    assert (a[:n] == b[:n]).all(), ("X", n, k, index, B, spread, gain)
                    checked += 1
    assert checked > 20000


def test_pvq_sparse_leaves_many_indices(oracle):
    """The leaf walk skips runs of zeros in sparse leaves by bisection (og_celt_split.hpp, pvq_leaf_lane): every reachable
    (N, K) with more than one dimension per pulse, many indices each -- random ones, the first and last, and the ones around
    every U(N, j) boundary, where a run ends or a sign flips."""
    emu = C.CDLL(os.path.join(ROOT, "tests", "emul", "libog_emul.so"))
    emu.emu_pvq_leaf.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p]
    emu.emu_pvq_leaf.restype = C.c_uint
    oracle.lib.oc_test_pvq_leaf.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p]
    oracle.lib.oc_test_pvq_leaf.restype = C.c_uint
    rng = np.random.default_rng(18)
    a, b = np.zeros(176, dtype=np.int16), np.zeros(176, dtype=np.int16)
    memo = {}
    _v(2, 2, memo)

    def u(x, y):
        _v(x, y, memo)
        return memo[(x, y)]
    checked = 0
    for n, k in [(n, k) for n, k in reachable_nk() if n >= 4 and n > k]:
        v = _v(n, k, memo)
        idxs = {0, 1, v - 1, v - 2} | {int(x) for x in rng.integers(0, v, 60)}
        for j in range(1, k + 2):
            for d in (-1, 0, 1):
                idxs.add(u(n, j) + d)
                idxs.add(u(n - 1, j) + d)
        for index in sorted(x for x in idxs if 0 <= x < v):
            ma = emu.emu_pvq_leaf(n, k, index, 1, 32767, 0, a.ctypes.data)
            mb = oracle.lib.oc_test_pvq_leaf(n, k, index, 0, 1, 32767, b.ctypes.data)
            assert ma == mb and np.array_equal(a[:n], b[:n]), (n, k, index)
            checked += 1
    assert checked > 10000


def test_zero_run_identity_in_exact_integers():
    """What the kernel's zero-run skip rests on (tools/pvq_zero_run.py): with V(a) = U(a, k) + U(a, k + 1) the dimensions n .. a+1
    of a codeword all decode to zero exactly when V(n) - V(a) <= 2 i < V(n) + V(a), and skipping them subtracts (V(n) - V(a)) / 2
    -- checked against the step-by-step walk (cwrsi, src/celt.cpp:2545) in Python's exact integers."""
    import random
    spec = importlib.util.spec_from_file_location("pvq_zero_run", os.path.join(ROOT, "tools", "pvq_zero_run.py"))
    z = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(z)
    random.seed(5)
    n_cases = 0
    for n in (3, 4, 6, 8, 11, 16, 22, 24, 36, 48, 72, 96, 144, 176):
        for k in (1, 2, 3, 4, 5, 7, 10, 13, 14, 20, 40):
            v = z.V(n, k)
            if v >= 2 ** 32:
                continue
            picks = [0, v - 1, z.U(n, k), z.U(n, k) - 1, z.U(n, k + 1), z.U(n, k + 1) - 1] + [random.randrange(v) for _ in range(12)]
            for i in picks:
                if not 0 <= i < v:
                    continue
                ref = z.cwrsi_ref(n, k, i)
                assert sum(abs(x) for x in ref) == k and len(ref) == n
                for ratio in (1, 2):
                    assert z.cwrsi_skip2(n, k, i, ratio) == ref, (n, k, i, ratio)
                n_cases += 1
    assert n_cases > 1500

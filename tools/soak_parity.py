#!/usr/bin/env python3
"""Randomised parity soak on the GPU (device-resident path), far larger than the test suite: every stream gets its own
random sequence of 20 ms packets -- mode / bandwidth / mono-stereo switches between frames, payloads of 0 .. 1275 bytes incl.
all-zero and all-ones -- and every PCM sample and return code is compared with the CPU oracle.
usage (GPU box): python3 tools/soak_parity.py [streams] [frames] [rounds] [seed]
       --host: through opusgpu_decode_packets instead, packets of every frame-count code (1 - 48 frames, padding, VBR / CBR,
       some malformed), every configuration incl. the non-20 ms ones, room for three frames per call
       --rfc: RFC mode (true frame durations, all 32 configurations x codes 0..3) with 25 % lost packets, 6 % DTX packets and 10 % of the packets preceded by a recovery from their FEC data,
       through opusgpu_decode_packets against one oracle decoder per stream (tests/test_gpu_rfc.py's comparison, larger)
       --pipeline: the device-resident path with pipelined steps (opusgpu_set_pipeline): the tables of every step are uploaded
       first, all steps of a round are queued back to back (one PCM buffer per step), compared afterwards
       --pipeline --masks: ... and every step declares a mode mask and holds only frames of those modes: runs of SILK-only, hybrid,
       CELT-only and mixed steps follow each other, so every kind of pipelined step and every switch between kinds is exercised"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py  # noqa: E402

spec = importlib.util.spec_from_file_location("opusgpu_pkg", os.path.join(ROOT, "esp32-opus-player_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pkg)

HOST = "--host" in sys.argv
if HOST:
    sys.argv.remove("--host")
RFC = "--rfc" in sys.argv
if RFC:
    sys.argv.remove("--rfc")
PIPE = "--pipeline" in sys.argv
if PIPE:
    sys.argv.remove("--pipeline")
MASKS = "--masks" in sys.argv  # with --pipeline: every step declares a mode mask (runs of SILK-only, hybrid, CELT-only, mixed steps)
if MASKS:
    sys.argv.remove("--masks")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
CONFIGS = np.array([1, 5, 9, 13, 15, 19, 23, 27, 31])  # the 20 ms configuration of every mode / bandwidth
LENS = np.array([0, 1, 2, 3, 5, 8, 13, 20, 30, 40, 60, 80, 100, 120, 160, 200, 320, 500, 800, 1275])
oracle = oracle_py.load()
ctx = pkg.Context(0)
total = bad_total = errs = 0
t_start = time.time()


def random_packet(rng, cfg, stereo):
    """A packet of any frame-count code (RFC 6716 section 3.2), frames of random length; sometimes malformed on purpose."""
    code = int(rng.choice([0, 0, 0, 1, 2, 3, 3]))
    toc = bytes([cfg << 3 | (4 if stereo else 0) | code])
    fl = lambda: int(rng.choice([0, 1, 2, 10, 40, 80, 160, 300]))
    body = lambda k: rng.integers(0, 256, k, dtype=np.uint8).tobytes()
    if code == 0:
        return toc + body(fl())
    if code == 1:
        return toc + body(2 * fl() + int(rng.random() < 0.05))       # odd length: invalid
    if code == 2:
        a = fl()
        size = bytes([a]) if a < 252 else bytes([252 + (a & 3), (a - 252 - (a & 3)) >> 2])
        return toc + size + body(a + fl() if rng.random() < 0.95 else max(a - 1, 0))
    # code 3; m = 0 is invalid.  Not more frames than the call has room for: the reference checks the room against the
    # TOC's frame duration but writes 960 samples per frame (Q6), i.e. overruns the caller's buffer on such packets;
    # the library returns OPUSGPU_BUFFER_TOO_SMALL instead, and there is nothing defined to compare.
    m = int(rng.choice([0, 1, 2, 3, 3, 2]))
    vbr, pad = rng.random() < 0.5, rng.random() < 0.3
    out = toc + bytes([m | (0x80 if vbr else 0) | (0x40 if pad else 0)])
    padlen = int(rng.choice([0, 1, 5, 254, 255, 300])) if pad else 0
    if pad:
        k = padlen
        while k >= 255:
            out += b"\xff"
            k -= 254
        out += bytes([k])
    if vbr:
        sizes = [fl() for _ in range(max(m, 1))]
        for a in sizes[:-1]:
            out += bytes([a]) if a < 252 else bytes([252 + (a & 3), (a - 252 - (a & 3)) >> 2])
        out += body(sum(sizes))
    else:
        out += body(max(m, 1) * fl())
    return out + bytes(padlen)


if RFC:
    import test_gpu_rfc  # noqa: E402  (tests/ is on the path)
    for rnd in range(rounds):
        for channels in (2, 1):
            state = {}

            def pick(s, f, rng):
                if s not in state or rng.random() < 0.3:
                    state[s] = int(rng.integers(32))
                return state[s], int(rng.choice([0, 0, 0, 1, 2, 3]))

            got = test_gpu_rfc._run(pkg, oracle, ctx, channels, {"streams": n, "steps": frames, "pick": pick, "p_redundant": 0.04},
                                    seed * 1000 + 900 + rnd * 2 + channels, p_loss=0.25, p_dtx=0.06, p_fec=0.1)  # asserts on any mismatch
            total += got
            print(f"rfc round {rnd} channels {channels}: {got} packets with PCM compared (of {n * frames}), 0 mismatches, {time.time() - t_start:.0f} s", flush=True)
    print(f"SOAK (RFC mode with lost packets and DTX frames): {total} packets compared sample by sample, 0 mismatches")
    sys.exit(0)

if HOST:
    ALLCFG = list(range(32))
    CAP = 3
    for rnd in range(rounds):
        for channels in (2, 1):
            rng = np.random.default_rng(seed * 1000 + 500 + rnd * 2 + channels)
            home = rng.choice(CONFIGS, n)
            pk = [[None] * n for _ in range(frames)]
            for f in range(frames):
                for s in range(n):
                    r = rng.random()
                    cfg = int(home[s]) if r < 0.75 else (int(rng.choice(CONFIGS)) if r < 0.9 else int(rng.choice(ALLCFG)))
                    pk[f][s] = random_packet(rng, cfg, bool(rng.random() < (0.85 if channels == 2 else 0.15)))
            lens = np.array([[len(pk[f][s]) for s in range(n)] for f in range(frames)], dtype=np.int64)
            offs = np.concatenate([[0], np.cumsum(lens.reshape(-1))[:-1]]).reshape(frames, n)
            arena = np.frombuffer(b"".join(pk[f][s] for f in range(frames) for s in range(n)) + bytes(16), dtype=np.uint8)
            ref, rets = oracle.batch_decode_var(channels, arena, offs, lens.astype(np.int32), cap_frames=CAP)
            ctx.streams_alloc(n, channels)
            for f in range(frames):
                pcm, res = ctx.decode_packets(np.arange(n), pk[f], frame_capacity=CAP)
                code_bad = np.nonzero(res != rets[:, f])[0]
                nb = code_bad.size
                for s in np.nonzero(rets[:, f] > 0)[0]:
                    t = pk[f][s][0]
                    if channels == 2 and not (t & 0x80) and (t & 0x60) != 0x60 and not (t & 4):
                        continue  # Q3: mono SILK-only packet in a stereo decoder: half of the output is undefined
                    r = rets[s, f]
                    if not (pcm[s, :r] == ref[s, f, :r]).all():
                        nb += 1
                if nb:
                    print(f"host round {rnd} channels {channels} frame {f}: {nb} differ (codes: {code_bad[:3]} gpu {res[code_bad[:3]]} "
                          f"oracle {rets[code_bad[:3], f]})", flush=True)
                bad_total += nb
                total += n
                errs += int((rets[:, f] <= 0).sum())
            print(f"host round {rnd} channels {channels}: {n * frames} packets done, {bad_total} mismatches so far, {time.time() - t_start:.0f} s", flush=True)
    print(f"SOAK (host path, multi-frame packets): {total} packets compared ({errs} of them error returns, compared as codes), {bad_total} mismatches")
    sys.exit(1 if bad_total else 0)

for rnd in range(rounds):
    for channels in (2, 1):
        rng = np.random.default_rng(seed * 1000 + rnd * 2 + channels)
        # per stream a "home" configuration; each frame keeps it with probability 0.8, else any configuration
        home = rng.choice(CONFIGS, n)
        cfg = np.where(rng.random((frames, n)) < 0.8, home[None, :], rng.choice(CONFIGS, (frames, n)))
        masks = [0] * frames
        if MASKS:  # step f holds only frames of the modes in masks[f]: bit 0 SILK-only, 1 hybrid, 2 CELT-only (the library's mask)
            by_mode = [np.array([1, 5, 9]), np.array([13, 15]), np.array([19, 23, 27, 31])]
            homes = [rng.choice(c, n) for c in by_mode]
            m = int(rng.choice([1, 2, 3, 4, 6, 7]))
            for f in range(frames):
                if rng.random() > 0.7:
                    m = int(rng.choice([1, 1, 2, 2, 3, 4, 6, 7]))
                masks[f] = m
                allowed = [k for k in range(3) if m >> k & 1]
                pick = rng.choice(allowed, n)
                away = rng.random(n) > 0.8
                for k in allowed:
                    sel = pick == k
                    cfg[f, sel] = np.where(away[sel], rng.choice(by_mode[k], int(sel.sum())), homes[k][sel])
        stereo = (rng.random((frames, n)) < (0.85 if channels == 2 else 0.15))
        toc = (cfg << 3 | np.where(stereo, 4, 0)).astype(np.uint8)
        lens = rng.choice(LENS, (frames, n), p=np.r_[np.full(4, 0.02), np.full(15, 0.06), 0.02])
        plen = (lens + 1).astype(np.int64)                      # packet = TOC + payload
        offs = np.concatenate([[0], np.cumsum(plen.reshape(-1))[:-1]]).reshape(frames, n)
        arena = rng.integers(0, 256, int(plen.sum()) + 16, dtype=np.uint8)
        kind = rng.integers(0, 10, (frames, n))
        for f, s in zip(*np.nonzero(kind == 0)):                # some all-zero / all-ones payloads
            arena[offs[f, s] + 1: offs[f, s] + plen[f, s]] = 0
        for f, s in zip(*np.nonzero(kind == 1)):
            arena[offs[f, s] + 1: offs[f, s] + plen[f, s]] = 255
        arena[offs.reshape(-1)] = toc.reshape(-1)
        ref, rets = oracle.batch_decode_var(channels, arena, offs, plen.astype(np.int32))
        # device path: one step per frame index
        mode = np.where(toc & 0x80, 2, np.where((toc & 0x60) == 0x60, 1, 0)).astype(np.int32)
        bw_c = ((toc >> 5) & 3).astype(np.int32)
        bw = np.where(mode == 2, np.where(bw_c == 0, 0, bw_c + 1), np.where(mode == 1, np.where(toc & 0x10, 4, 3), bw_c))
        flags = (mode | bw << 2 | np.where(toc & 4, 32, 0)).astype(np.int32)
        ctx.streams_alloc(n, channels)
        d_arena = ctx.dev_alloc(arena.size)
        ctx.h2d(d_arena, arena)
        out = np.zeros((n, 960, channels), dtype=np.int16)
        res = np.zeros(n, dtype=np.int32)
        descs = np.zeros(n, dtype=pkg.DESC_DTYPE)
        descs["stream"] = np.arange(n, dtype=np.int32)
        nbuf = frames if PIPE else 1
        d_descs = [ctx.dev_alloc(16 * n) for _ in range(nbuf)]
        d_pcms = [ctx.dev_alloc(n * 960 * channels * 2) for _ in range(nbuf)]
        d_ress = [ctx.dev_alloc(4 * n) for _ in range(nbuf)]

        def upload(f, d):
            descs["offset"] = (offs[f] + 1).astype(np.int32)
            descs["len"] = lens[f].astype(np.int32)
            descs["flags"] = flags[f]
            ctx.h2d(d, descs)

        ctx.set_pipeline(PIPE)
        if PIPE:
            for f in range(frames):
                upload(f, d_descs[f])
            for f in range(frames):
                ctx.decode_step_device(n, d_descs[f], d_arena, d_pcms[f], d_ress[f], modes=masks[f])
            ctx.synchronize()
        for f in range(frames):
            d_desc, d_pcm, d_res = d_descs[f % nbuf], d_pcms[f % nbuf], d_ress[f % nbuf]
            if not PIPE:
                upload(f, d_desc)
                ctx.decode_step_device(n, d_desc, d_arena, d_pcm, d_res)
                ctx.synchronize()
            ctx.d2h(out, d_pcm)
            ctx.d2h(res, d_res)
            code_bad = np.nonzero(res != rets[:, f])[0]
            ok = rets[:, f] == 960
            # Q3: a mono SILK-only packet in a stereo decoder defines only the first 960 interleaved entries
            half = ok & (mode[f] == 0) & ((toc[f] & 4) == 0) & (channels == 2)
            full = ok & ~half
            a, b = out.reshape(n, -1), ref[:, f].reshape(n, -1)
            pcm_bad = np.nonzero((a[full] != b[full]).any(axis=1))[0]
            pcm_bad_h = np.nonzero((a[half][:, :960] != b[half][:, :960]).any(axis=1))[0] if half.any() else np.zeros(0, int)
            nb = code_bad.size + pcm_bad.size + pcm_bad_h.size
            if nb:
                print(f"round {rnd} channels {channels} frame {f}: {code_bad.size} return codes, {pcm_bad.size + pcm_bad_h.size} PCM blocks differ"
                      f" (first code diff stream {code_bad[:3]}, gpu {res[code_bad[:3]]}, oracle {rets[code_bad[:3], f]})", flush=True)
            bad_total += nb
            total += n
            errs += int((rets[:, f] != 960).sum())
        for p in [d_arena] + d_descs + d_pcms + d_ress:
            ctx.dev_free(p)
        print(f"round {rnd} channels {channels}: {n * frames} frames done, {bad_total} mismatches so far, {time.time() - t_start:.0f} s", flush=True)
print(f"SOAK{' (pipelined steps' + (', declared mode masks' if MASKS else '') + ')' if PIPE else ''}: {total} frames compared ({errs} of them error returns, compared as codes), {bad_total} mismatches")
sys.exit(1 if bad_total else 0)

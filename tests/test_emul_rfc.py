"""CPU: RFC mode of the kernel source (host emulation of decode_frame_rfc) against the oracle's RFC mode, all 32 TOC
configurations x codes 0..3 with configuration switches, lost packets and DTX frames (tools/fuzz_emul_rfc.py at a small size);
and the oracle's RFC mode against what is defined independently of it: the sample count of every packet, and how a
concealment behaves (its length, silence before the first packet, the decay over a burst of losses)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_rfc_mode_matches_oracle():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emul"), "-s"])
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_emul_rfc.py"), "250", "8", "3"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert " 0 mismatches" in p.stdout


def test_oracle_rfc_mode_returns_true_durations(oracle):
    rng = np.random.default_rng(4)
    for channels in (1, 2):
        d = oracle.decoder(channels)
        d.init()
        d.set_rfc(True)
        for cfg in range(32):
            toc = (cfg << 3) | (4 if channels == 2 else 0)
            if toc & 0x80:
                want = (48000 << ((toc >> 3) & 3)) // 400
            elif (toc & 0x60) == 0x60:
                want = 960 if toc & 8 else 480
            else:
                want = [480, 960, 1920, 2880][(toc >> 3) & 3]
            for count in (1, 2):
                if want * count > 5760:
                    continue
                pkt = bytes([toc | 3, count]) + rng.integers(0, 256, 60 * count, dtype=np.uint8).tobytes()
                _, r = d.decode(pkt)
                assert r == want * count, (hex(toc), count, r)
        d.set_rfc(False)
        _, r = d.decode(bytes([0xE0]) + bytes(60))  # reference mode again: a 2.5 ms TOC decodes as 20 ms (Q6)
        assert r == 960


def test_oracle_conceals_the_duration_asked_for(oracle):
    rng = np.random.default_rng(8)
    for channels in (1, 2):
        d = oracle.decoder(channels)
        d.init()
        d.set_rfc(True)
        buf, r = d.conceal(960)  # nothing decoded yet: zeros
        assert r == 960 and not buf[:960].any()
        for cfg in (1, 9, 11, 13, 15, 17, 23, 29, 31):  # SILK NB 20, WB 20, WB 60, hybrid SWB 20, FB 20, CELT NB 5, WB 20, FB 5, FB 20
            toc = (cfg << 3) | (4 if channels == 2 else 0)
            fs = __import__("rfc_common").dur(toc)
            for _ in range(3):
                _, r = d.decode(bytes([toc]) + rng.integers(0, 256, 70, dtype=np.uint8).tobytes())
                assert r == fs
            for want in (fs, 2 * fs, fs):
                if want > 5760:
                    continue
                _, r = d.conceal(want)
                assert r == want, (cfg, want, r)
            _, r = d.decode(bytes([toc]) + rng.integers(0, 256, 70, dtype=np.uint8).tobytes())  # and decoding goes on
            assert r == fs
        assert d.conceal(100)[1] < 0  # not a multiple of 2.5 ms
        d.set_rfc(False)
        assert d.conceal(960)[1] == -18  # reference mode: no concealment (Q8) -- the reference's empty-packet branch instead, which
        #                                  after a CELT-only packet ends in celt_decode_with_ec's refusal (tests/test_empty_packets.py)


def test_oracle_celt_concealment_decays(oracle):
    """a burst of lost CELT frames: the first five are extrapolated from the pitch period of the last output (each period a little
    quieter, never louder than what it continues); from the sixth on the noise-based branch takes over from the energies the last
    DECODED frame left and fades towards the noise floor (0.5 dB per frame, never below the floor).  Within either phase the level
    does not grow, and over the noise phase it falls where the last decoded frame was well above the floor."""
    fell = quieter = 0
    for seed in range(12):
        rng = np.random.default_rng(seed)
        d = oracle.decoder(2)
        d.init()
        d.set_rfc(True)
        for _ in range(6):
            ref, _r = d.decode(bytes([0xFC]) + rng.integers(0, 256, 120, dtype=np.uint8).tobytes())
        last = float(np.sqrt(np.mean(ref[:960].astype(np.float64) ** 2)))
        rms = []
        for _ in range(24):
            buf, r = d.conceal(960)
            assert r == 960
            rms.append(float(np.sqrt(np.mean(buf[:960].astype(np.float64) ** 2))))
        assert max(rms[:5]) <= 1.3 * last + 1, (seed, last, rms[:5])  # the pitch phase continues the last frame, no louder
        assert all(rms[k + 1] <= 1.05 * rms[k] + 1 for k in range(4)), (seed, rms[:5])
        quieter += rms[4] < 0.8 * rms[0]
        early, late = np.mean(rms[6:10]), np.mean(rms[-4:])  # the noise phase
        assert late <= 1.15 * early + 1, (seed, rms)
        fell += late < 0.7 * early
    assert fell >= 3 and quieter >= 6, (fell, quieter)


def test_oracle_mode_transitions_start_from_the_old_modes_concealment(oracle):
    """RFC 6716 section 4.5: a CELT-only frame after SILK-only / hybrid frames (and the other way round) with no redundant frame in
    between starts with 2.5 ms of the OLD mode's concealment, cross-fades over the next 2.5 ms and is the new frame's own audio
    from 5 ms on.  Checked against a twin decoder with the same history that conceals 5 ms instead of decoding the packet:
    the first 2.5 ms are identical; and against the frame decoded after a same-mode history: from 5 ms on ... the latter cannot be
    asked of the oracle (it has no switch for the smoothing), so only the first property is pinned here."""
    rng = np.random.default_rng(12)
    # (old configuration, new configuration): SILK NB 20 -> CELT FB 20, hybrid FB 20 -> CELT FB 10, CELT FB 20 -> SILK WB 20,
    # CELT WB 10 -> hybrid SWB 20, SILK WB 20 -> CELT FB 2.5 (a frame shorter than 5 ms: cross-faded from its start)
    for channels in (1, 2):
        for old, new in ((1, 31), (15, 30), (31, 9), (22, 13), (9, 28)):
            st = 4 if channels == 2 else 0
            twins = [oracle.decoder(channels) for _ in range(2)]
            for d in twins:
                d.init()
                d.set_rfc(True)
            for _ in range(3):
                pkt = bytes([(old << 3) | st]) + rng.integers(0, 256, 8 if old < 12 else 90, dtype=np.uint8).tobytes()
                a, ra = twins[0].decode(pkt)
                b, rb = twins[1].decode(pkt)
                assert ra == rb > 0 and np.array_equal(a[:ra], b[:rb])
            # a hybrid / SILK frame that carries a redundant frame has no transition: keep the payload so short that it cannot
            # (SILK-only: whatever follows the SILK data would be one)
            pkt = bytes([(new << 3) | st]) + rng.integers(0, 256, 8 if new < 12 else 60, dtype=np.uint8).tobytes()
            got, r = twins[0].decode(pkt)
            plc, rp = twins[1].conceal(min(240, r))
            assert r > 0 and rp == min(240, r)
            head = 120 if r >= 240 else 0
            if head:
                assert np.array_equal(got[:head], plc[:head]), (channels, old, new)
            # inside the cross-fade the frame moves away from the concealment: at its end the concealment's weight is ~ 0
            # (the last sample's window value squared is 32767 * 32767 >> 15)
            assert not np.array_equal(got[:r], plc[:r]) or not plc[:rp].any()

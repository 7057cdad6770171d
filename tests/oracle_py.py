"""ctypes binding of the CPU oracle (oracle/liboc_oracle.so).  TEST INFRASTRUCTURE ONLY: used by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboc_oracle.so")

MODE_SILK, MODE_HYBRID, MODE_CELT = 1000, 1001, 1002


def usable_cpus():
    """CPUs this process can actually use (esp32-opus-player_amd/shard.py: affinity mask and cgroup quota)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("og_shard_for_oracle", os.path.join(ROOT, "esp32-opus-player_amd", "shard.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.usable_cpus()


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp = C.c_void_p
        lib.oc_decoder_create.restype = vp
        lib.oc_decoder_create.argtypes = [C.c_int]
        lib.oc_decoder_destroy.argtypes = [vp]
        lib.oc_decoder_destroy.restype = None
        lib.oc_decoder_init.argtypes = [vp, C.c_int]
        lib.oc_decoder_init.restype = None
        lib.oc_decoder_reset.argtypes = [vp]
        lib.oc_decoder_reset.restype = None
        lib.oc_decode.argtypes = [vp, C.c_char_p, C.c_int32, vp, C.c_int]
        lib.oc_decode.restype = C.c_int
        lib.oc_decoder_final_range.argtypes = [vp]
        lib.oc_decoder_final_range.restype = C.c_uint32
        lib.oc_decoder_ctl_final_range.argtypes = [vp]
        lib.oc_decoder_ctl_final_range.restype = C.c_uint32
        lib.oc_decoder_ctl_pitch.argtypes = [vp, C.POINTER(C.c_int32)]
        lib.oc_decoder_ctl_pitch.restype = C.c_int
        lib.oc_decoder_set_rfc.argtypes = [vp, C.c_int]
        lib.oc_decoder_set_rfc.restype = None
        lib.oc_batch_decode.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        lib.oc_batch_decode.restype = C.c_long

    def batch_decode(self, channels, toc, payloads, s0=0, s1=None, want_pcm=True):
        """payloads uint8 [frames, streams, L] -> (pcm int16 [streams, frames, 960, ch] or None, frames_ok)."""
        nf, ns, L = payloads.shape
        s1 = ns if s1 is None else s1
        pay = np.ascontiguousarray(payloads)
        pcm = np.zeros((ns, nf, 960, channels), dtype=np.int16) if want_pcm else None
        ok = self.lib.oc_batch_decode(channels, toc, pay.ctypes.data, ns, nf, L, s0, s1,
                                      pcm.ctypes.data if want_pcm else None)
        return pcm, ok

    def batch_decode_threads(self, channels, toc, payloads, threads=None):
        """batch_decode over all streams, the stream range split over `threads` host threads (the C call releases the GIL).
        -> (pcm int16 [streams, frames, 960, ch], frames decoded without error)."""
        import threading
        nf, ns, L = payloads.shape
        threads = max(1, min(threads or usable_cpus(), ns))
        pay = np.ascontiguousarray(payloads)
        pcm = np.zeros((ns, nf, 960, channels), dtype=np.int16)
        oks = [0] * threads
        cuts = [ns * t // threads for t in range(threads + 1)]

        def work(t):
            oks[t] = self.lib.oc_batch_decode(channels, toc, pay.ctypes.data, ns, nf, L, cuts[t], cuts[t + 1], pcm.ctypes.data)

        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        return pcm, sum(oks)

    def batch_decode_var(self, channels, arena, offs, lens, threads=None, cap_frames=1):
        """Arbitrary packets: frame f of stream s = arena[offs[f, s] : offs[f, s] + lens[f, s]] (TOC first).  offs / lens:
        [frames, streams].  -> (pcm int16 [streams, frames, 960 * cap_frames, ch], return codes int32 [streams, frames]);
        cap_frames: room (and frame_size) for that many 20 ms frames per call.  Streams are split over host threads."""
        import threading
        nf, ns = offs.shape
        threads = max(1, min(threads or usable_cpus(), ns))
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        pcm = np.zeros((ns, nf, 960 * cap_frames, channels), dtype=np.int16)
        rets = np.zeros((ns, nf), dtype=np.int32)
        self.lib.oc_batch_decode_var_cap.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_int]
        self.lib.oc_batch_decode_var_cap.restype = C.c_long
        cuts = [ns * t // threads for t in range(threads + 1)]

        def work(t):
            self.lib.oc_batch_decode_var_cap(channels, arena.ctypes.data, offs.ctypes.data, lens.ctypes.data, ns, nf, cuts[t], cuts[t + 1],
                                             pcm.ctypes.data, rets.ctypes.data, cap_frames)

        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        return pcm, rets

    def batch_decode_rfc(self, channels, arena, offs, lens, ops, threads=None, want_pcm=True):
        """RFC mode with losses (oracle/oc_batch.c oc_batch_decode_rfc): offs / lens / ops [frames, streams]; ops 0 decode, 1 lost
        (concealed), 2 lost and recovered from the next packet's FEC data (offs / lens name THAT packet).
        -> (pcm of every stream's last step int16 [streams, 5760, ch] or None, return values int32 [streams, frames])."""
        import threading
        nf, ns = offs.shape
        threads = max(1, min(threads or usable_cpus(), ns))
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        ops = np.ascontiguousarray(ops, dtype=np.uint8)
        pcm = np.zeros((ns, 5760, channels), dtype=np.int16) if want_pcm else None
        rets = np.zeros((ns, nf), dtype=np.int32)
        fn = self.lib.oc_batch_decode_rfc
        fn.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 2
        fn.restype = C.c_long
        cuts = [ns * t // threads for t in range(threads + 1)]

        def work(t):
            fn(channels, arena.ctypes.data, offs.ctypes.data, lens.ctypes.data, ops.ctypes.data, ns, nf, cuts[t], cuts[t + 1],
               pcm.ctypes.data if want_pcm else None, rets.ctypes.data)

        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        return pcm, rets

    def decoder(self, channels):
        return OracleDecoder(self, channels)

    def decode_streams(self, channels, packets_per_stream):
        """packets_per_stream: list (streams) of list (frames) of bytes -> int16 [streams, frames, 960, ch]
        plus return codes [streams, frames].  Fresh state per stream."""
        ns = len(packets_per_stream)
        nf = max(len(p) for p in packets_per_stream)
        pcm = np.zeros((ns, nf, 960, channels), dtype=np.int16)
        rets = np.zeros((ns, nf), dtype=np.int32)
        d = self.decoder(channels)
        for s, pk in enumerate(packets_per_stream):
            d.init()
            for f, p in enumerate(pk):
                out, r = d.decode(p)
                rets[s, f] = r
                if r > 0:
                    pcm[s, f, :r] = out[:r]
        return pcm, rets


class OracleDecoder:
    def __init__(self, o, channels):
        self.o, self.channels = o, channels
        self.h = o.lib.oc_decoder_create(channels)
        # (+ 960: a stereo packet in a mono decoder mixes 960 * 2 entries per frame, Q3 -- the last pass of an empty packet too)
        self.buf = np.zeros((5760 + 960, channels), dtype=np.int16)

    def init(self):
        self.o.lib.oc_decoder_init(self.h, self.channels)

    def reset(self):
        self.o.lib.oc_decoder_reset(self.h)

    def set_rfc(self, on=True):
        """RFC mode (oracle/oc_opus.h: frames at the durations their TOC names; parity-unpinned)."""
        self.o.lib.oc_decoder_set_rfc(self.h, 1 if on else 0)

    def decode(self, packet: bytes):
        r = self.o.lib.oc_decode(self.h, bytes(packet), len(packet), self.buf.ctypes.data, 5760)
        return self.buf, r

    def prev_mode(self):
        """Mode of the last frame decoded or concealed (0 before the first): the mode a concealment runs in."""
        self.o.lib.oc_decoder_prev_mode.argtypes = [C.c_void_p]
        return int(self.o.lib.oc_decoder_prev_mode(self.h))

    def conceal(self, samples):
        """A lost packet (RFC mode only): conceal `samples` per channel, as opus_decode(data = NULL) would."""
        r = self.o.lib.oc_decode(self.h, None, 0, self.buf.ctypes.data, samples)
        return self.buf, r

    def decode_cap(self, packet: bytes, cap_frames):
        """Like a caller with room for cap_frames 20 ms frames (frame_size = 960 * cap_frames)."""
        r = self.o.lib.oc_decode(self.h, bytes(packet), len(packet), self.buf.ctypes.data, 960 * cap_frames)
        return self.buf, r

    def __del__(self):
        try:
            self.o.lib.oc_decoder_destroy(self.h)
        except Exception:
            pass


def fnv1a_u16(samples, h=2166136261):
    """FNV-1a over uint16 units (the hash the survey's KATs were recorded with)."""
    u = np.ascontiguousarray(samples).view(np.uint16).reshape(-1)
    for v in u.tolist():
        h = ((h ^ v) * 16777619) & 0xFFFFFFFF
    return h


def load():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return Oracle(C.CDLL(LIB))

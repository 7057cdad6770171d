"""ctypes binding of libopusgpu.so (the MI355X batched Opus decoder) for tests and bench.py.

This module is plumbing only: every decode goes through the C ABI declared in include/opusgpu.h and
runs on the GPU.  There is no CPU fallback -- if the HIP library is missing or no GPU is usable the
calls raise.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)

The directory name has a hyphen, so import it with `importlib` (see tests/conftest.py: load_pkg()).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OPUSGPU_LIB", os.path.join(HERE, "libopusgpu.so"))  # override: experiments only

OPUSGPU_ERR_NO_DEVICE = -100
FRAME = 960

EXPORTS = [
    "opusgpu_version", "opusgpu_ctx_create", "opusgpu_ctx_destroy", "opusgpu_last_error",
    "opusgpu_streams_alloc", "opusgpu_streams_reset", "opusgpu_stream_count", "opusgpu_stream_channels",
    "opusgpu_stream_state_bytes", "opusgpu_decode_packets", "opusgpu_packet_to_frames",
    "opusgpu_dev_alloc", "opusgpu_dev_free", "opusgpu_memcpy_h2d", "opusgpu_memcpy_d2h",
    "opusgpu_decode_step_device", "opusgpu_synchronize", "opusgpu_event_create", "opusgpu_event_record",
    "opusgpu_event_elapsed_ms", "opusgpu_event_destroy", "opusgpu_stream_state_get",
]


class FrameDesc(C.Structure):
    _fields_ = [("stream", C.c_int32), ("offset", C.c_int32), ("len", C.c_int32), ("flags", C.c_int32)]


DESC_DTYPE = np.dtype([("stream", "<i4"), ("offset", "<i4"), ("len", "<i4"), ("flags", "<i4")])

_lib = None


def load_lib():
    """Load libopusgpu.so; raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the decode path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32p, u8pp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_char_p)
    lib.opusgpu_version.restype = C.c_int
    lib.opusgpu_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.opusgpu_ctx_destroy.argtypes = [vp]
    lib.opusgpu_ctx_destroy.restype = None
    lib.opusgpu_last_error.argtypes = [vp]
    lib.opusgpu_last_error.restype = C.c_char_p
    lib.opusgpu_streams_alloc.argtypes = [vp, C.c_int, C.c_int]
    lib.opusgpu_streams_reset.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.opusgpu_stream_count.argtypes = [vp]
    lib.opusgpu_stream_channels.argtypes = [vp]
    lib.opusgpu_stream_state_bytes.restype = C.c_size_t
    lib.opusgpu_decode_packets.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    lib.opusgpu_packet_to_frames.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(FrameDesc)]
    lib.opusgpu_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.opusgpu_dev_free.argtypes = [vp, vp]
    lib.opusgpu_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.opusgpu_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.opusgpu_decode_step_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.opusgpu_synchronize.argtypes = [vp]
    lib.opusgpu_event_create.argtypes = [vp, C.POINTER(vp)]
    lib.opusgpu_event_record.argtypes = [vp, vp]
    lib.opusgpu_event_elapsed_ms.argtypes = [vp, vp, vp, C.POINTER(C.c_float)]
    lib.opusgpu_event_destroy.argtypes = [vp, vp]
    lib.opusgpu_stream_state_get.argtypes = [vp, C.c_int, vp, C.c_size_t]
    _lib = lib
    return lib


class OpusGpuError(RuntimeError):
    pass


def packet_to_frames(packet: bytes, stream: int = 0):
    """Host-only: frame descriptors of one packet (list of (offset, len, flags)) or a negative code."""
    lib = load_lib()
    d = (FrameDesc * 48)()
    n = lib.opusgpu_packet_to_frames(packet, len(packet), stream, d)
    if n < 0:
        return n
    return [(d[i].offset, d[i].len, d[i].flags) for i in range(n)]


class Context:
    """One GPU context: per-stream state in HBM plus the batched decode entry points."""

    def __init__(self, device: int = 0):
        self.lib = load_lib()
        h = C.c_void_p()
        rc = self.lib.opusgpu_ctx_create(device, C.byref(h))
        if rc != 0:
            raise OpusGpuError(f"opusgpu_ctx_create(device={device}) failed with {rc}"
                               + (" (no usable HIP device; no CPU fallback exists)" if rc == OPUSGPU_ERR_NO_DEVICE else ""))
        self.h = h
        self.channels = 0
        self.n_streams = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.opusgpu_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise OpusGpuError(f"{what} failed: {rc} ({self.lib.opusgpu_last_error(self.h).decode()})")

    def streams_alloc(self, n, channels):
        self._chk(self.lib.opusgpu_streams_alloc(self.h, n, channels), "opusgpu_streams_alloc")
        self.n_streams, self.channels = n, channels

    def streams_reset(self, first, count, full=True):
        self._chk(self.lib.opusgpu_streams_reset(self.h, first, count, 1 if full else 0), "opusgpu_streams_reset")

    def decode_packets(self, stream_ids, packets, frame_capacity=1):
        """Batched opus_multistream_decode: returns (pcm[n, cap*960, ch] int16, result[n] int32)."""
        n = len(packets)
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        lens = np.array([len(p) for p in packets], dtype=np.int32)
        bufs = [C.create_string_buffer(bytes(p), max(len(p), 1)) for p in packets]
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        pcm = np.zeros((n, frame_capacity * FRAME, self.channels), dtype=np.int16)
        res = np.zeros(n, dtype=np.int32)
        self._chk(self.lib.opusgpu_decode_packets(self.h, n, ids.ctypes.data, C.addressof(ptrs), lens.ctypes.data,
                                                  pcm.ctypes.data, frame_capacity, res.ctypes.data),
                  "opusgpu_decode_packets")
        return pcm, res

    # ---- device-resident path -------------------------------------------------------------------
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.opusgpu_dev_alloc(self.h, nbytes, C.byref(p)), "opusgpu_dev_alloc")
        return p

    def dev_free(self, p):
        self._chk(self.lib.opusgpu_dev_free(self.h, p), "opusgpu_dev_free")

    def h2d(self, dptr, arr):
        a = np.ascontiguousarray(arr)
        self._chk(self.lib.opusgpu_memcpy_h2d(self.h, dptr, a.ctypes.data, a.nbytes), "opusgpu_memcpy_h2d")

    def d2h(self, arr, dptr):
        self._chk(self.lib.opusgpu_memcpy_d2h(self.h, arr.ctypes.data, dptr, arr.nbytes), "opusgpu_memcpy_d2h")

    def decode_step_device(self, n, d_descs, d_arena, d_pcm, d_result, stream=None):
        self._chk(self.lib.opusgpu_decode_step_device(self.h, n, d_descs, d_arena, d_pcm, d_result, stream),
                  "opusgpu_decode_step_device")

    def synchronize(self):
        self._chk(self.lib.opusgpu_synchronize(self.h), "opusgpu_synchronize")

    def event(self):
        e = C.c_void_p()
        self._chk(self.lib.opusgpu_event_create(self.h, C.byref(e)), "opusgpu_event_create")
        return e

    def event_record(self, e):
        self._chk(self.lib.opusgpu_event_record(self.h, e), "opusgpu_event_record")

    def event_elapsed_ms(self, a, b):
        ms = C.c_float()
        self._chk(self.lib.opusgpu_event_elapsed_ms(self.h, a, b, C.byref(ms)), "opusgpu_event_elapsed_ms")
        return ms.value

    def event_destroy(self, e):
        self.lib.opusgpu_event_destroy(self.h, e)


# ---- synthetic workloads ---------------------------------------------------
TOC_CELT_FB_STEREO = 0xFC
TOC_SILK_NB_STEREO = 0x0C
TOC_HYBRID_FB_STEREO = 0x7C


def lcg_payloads(n_streams, n_frames, payload_len, seed_base=0x9E3779B9):
    """Per-stream LCG payload bytes: x <- 1664525 x + 1013904223 (mod 2^32), byte = x >> 24,
    seed = seed_base ^ stream_id, running continuously over the stream's frames.
    Returns uint8 [n_frames, n_streams, payload_len]."""
    x = (np.uint32(seed_base) ^ np.arange(n_streams, dtype=np.uint32)).astype(np.uint32)
    out = np.empty((n_frames, n_streams, payload_len), dtype=np.uint8)
    a, c = np.uint32(1664525), np.uint32(1013904223)
    with np.errstate(over="ignore"):
        for f in range(n_frames):
            for i in range(payload_len):
                x = x * a + c
                out[f, :, i] = (x >> np.uint32(24)).astype(np.uint8)
    return out


def build_step(toc, payloads):
    """Arena + descriptors for one decode step: payloads uint8 [n_streams, L], one code-0 packet each.
    The arena holds TOC + payload per stream; descriptors point past the TOC byte."""
    n, L = payloads.shape
    arena = np.empty((n, L + 1), dtype=np.uint8)
    arena[:, 0] = toc
    arena[:, 1:] = payloads
    if toc & 0x80:
        mode, bw = 2, ((toc >> 5) & 3)
        bw = 0 if bw == 0 else bw + 1
    elif (toc & 0x60) == 0x60:
        mode, bw = 1, (4 if toc & 0x10 else 3)
    else:
        mode, bw = 0, (toc >> 5) & 3
    flags = mode | (bw << 2) | (32 if toc & 4 else 0)
    descs = np.zeros(n, dtype=DESC_DTYPE)
    descs["stream"] = np.arange(n, dtype=np.int32)
    descs["offset"] = np.arange(n, dtype=np.int32) * (L + 1) + 1
    descs["len"] = L
    descs["flags"] = flags
    return arena.reshape(-1), descs

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
    "opusgpu_stream_state_bytes", "opusgpu_debug_stage_taps", "opusgpu_decode_packets", "opusgpu_decode_packets_fec", "opusgpu_packet_to_frames",
    "opusgpu_dev_alloc", "opusgpu_dev_free", "opusgpu_memcpy_h2d", "opusgpu_memcpy_d2h",
    "opusgpu_decode_step_device", "opusgpu_decode_step_device_modes", "opusgpu_decode_steps_device", "opusgpu_synchronize", "opusgpu_event_create", "opusgpu_event_record",
    "opusgpu_event_elapsed_ms", "opusgpu_event_destroy", "opusgpu_event_synchronize", "opusgpu_stream_state_get", "opusgpu_stream_pitch_get",
    "opusgpu_upload_async", "opusgpu_upload_fence", "opusgpu_stream_wait_event", "opusgpu_host_register", "opusgpu_host_unregister",
    "opusgpu_pages_demux", "opusgpu_pages_demux_into", "opusgpu_page_batch_arena_offset", "opusgpu_page_batch_steps", "opusgpu_page_batch_step", "opusgpu_page_batch_arena",
    "opusgpu_page_batch_free", "opusgpu_pages_crc_device", "opusgpu_output_stage_device",
    "opusgpu_set_mode", "opusgpu_get_mode", "opusgpu_set_pipeline", "opusgpu_get_pipeline", "opusgpu_packet_to_frames_mode",
    "opusgpu_empty_packet_to_frames",
]


class OutputCfg(C.Structure):
    """opusgpu_output_cfg (include/opusgpu.h): the player's output settings, src/main.cpp m_vol / m_f_forceMono /
    m_bitsPerSample / m_channels."""
    _fields_ = [("volume", C.c_uint8), ("force_mono", C.c_uint8), ("bits", C.c_uint8), ("channels", C.c_uint8)]


OUTPUT_CFG_DTYPE = np.dtype([("volume", "u1"), ("force_mono", "u1"), ("bits", "u1"), ("channels", "u1")])


class _SilkChTaps(C.Structure):
    _fields_ = [("pitchL", C.c_int32 * 4), ("Gains_Q16", C.c_int32 * 4), ("PredCoef_Q12", (C.c_int16 * 16) * 2),
                ("LTPCoef_Q14", C.c_int16 * 20), ("LTP_scale_Q14", C.c_int32), ("signalType", C.c_int32), ("quantOffsetType", C.c_int32)]


class StageTaps(C.Structure):
    """opusgpu_stage_taps (include/opusgpu.h): what the kernels of the last decode step left between the stages."""
    _fields_ = [("celt_valid", C.c_int32), ("celt_ret", C.c_int32), ("silence", C.c_int32), ("transient", C.c_int32), ("lm", C.c_int32),
                ("spread", C.c_int32), ("dual_stereo", C.c_int32), ("anti_collapse_on", C.c_int32), ("intensity", C.c_int32),
                ("pf_pitch", C.c_int32), ("pf_gain", C.c_int32), ("pf_tapset", C.c_int32), ("n_leaves", C.c_int32),
                ("celt_rng_final", C.c_uint32), ("bandE", C.c_int16 * 42), ("pulses", C.c_int16 * 21), ("tf_res", C.c_int8 * 21),
                ("pad0", C.c_int8 * 3), ("syn_post", (C.c_int32 * 960) * 2), ("overlap_tail", (C.c_int32 * 60) * 2),
                ("state_bandE", C.c_int16 * 42), ("state_logE1", C.c_int16 * 42), ("state_logE2", C.c_int16 * 42), ("pad1", C.c_int16),
                ("state_rng", C.c_uint32), ("pf_period", C.c_int32), ("pf_gain_state", C.c_int32), ("pf_tapset_state", C.c_int32),
                ("silk_valid", C.c_int32), ("silk_ret", C.c_int32), ("decode_only_middle", C.c_int32), ("ms_pred_q13", C.c_int32 * 2),
                ("silk_ch", _SilkChTaps * 2), ("silk_out", (C.c_int16 * 320) * 2), ("silk_sLPC_Q14", (C.c_int32 * 16) * 2),
                ("silk_fs_kHz", C.c_int32 * 2)]


class FrameDesc(C.Structure):
    _fields_ = [("stream", C.c_int32), ("offset", C.c_int32), ("len", C.c_int32), ("flags", C.c_int32)]


HAS_SILK, HAS_HYBRID, HAS_CELT = 1, 2, 4  # opusgpu_decode_step_device_modes
STEP_KEEPS_MODE = 8  # OPUSGPU_STEP_KEEPS_MODE: no stream of the step has decoded a frame of another mode since its last reset


def toc_modes(toc):
    """The mode mask of a step whose frames all carry this TOC byte."""
    return HAS_CELT if toc & 0x80 else (HAS_HYBRID if (toc & 0x60) == 0x60 else HAS_SILK)


DESC_DTYPE = np.dtype([("stream", "<i4"), ("offset", "<i4"), ("len", "<i4"), ("flags", "<i4")])
# opusgpu_page_info (include/opusgpu.h)
PAGE_INFO_DTYPE = np.dtype([("status", "<i4"), ("packets", "<i4"), ("first_step", "<i4"), ("header_type", "<i4"),
                            ("serial", "<u4"), ("seqno", "<u4"), ("granulepos", "<i8")])
PAGE_BAD_CAPTURE, PAGE_BAD_CRC, PAGE_SPANS, PAGE_BAD_PACKET, PAGE_BAD_STREAM = -200, -201, -202, -203, -204
PAGES_VERIFY_CRC, PAGES_GROUP_BY_MODE, PAGES_ORDER_BY_HEADER = 1, 2, 4

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
    lib.opusgpu_decode_packets_fec.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    lib.opusgpu_packet_to_frames.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(FrameDesc)]
    lib.opusgpu_packet_to_frames_mode.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int, C.POINTER(FrameDesc)]
    lib.opusgpu_empty_packet_to_frames.argtypes = [C.c_int32, C.c_int32, C.c_int, C.c_int, C.POINTER(FrameDesc)]
    lib.opusgpu_set_mode.argtypes = [vp, C.c_int]
    lib.opusgpu_get_mode.argtypes = [vp]
    lib.opusgpu_set_pipeline.argtypes = [vp, C.c_int]
    lib.opusgpu_get_pipeline.argtypes = [vp]
    lib.opusgpu_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.opusgpu_dev_free.argtypes = [vp, vp]
    lib.opusgpu_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.opusgpu_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.opusgpu_decode_step_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.opusgpu_decode_step_device_modes.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    lib.opusgpu_decode_steps_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.opusgpu_synchronize.argtypes = [vp]
    lib.opusgpu_event_create.argtypes = [vp, C.POINTER(vp)]
    lib.opusgpu_event_record.argtypes = [vp, vp]
    lib.opusgpu_event_elapsed_ms.argtypes = [vp, vp, vp, C.POINTER(C.c_float)]
    lib.opusgpu_event_destroy.argtypes = [vp, vp]
    lib.opusgpu_event_synchronize.argtypes = [vp, vp]
    lib.opusgpu_upload_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.opusgpu_upload_fence.argtypes = [vp, vp]
    lib.opusgpu_stream_wait_event.argtypes = [vp, vp, vp]
    lib.opusgpu_host_register.argtypes = [vp, vp, C.c_size_t]
    lib.opusgpu_host_unregister.argtypes = [vp, vp]
    lib.opusgpu_stream_state_get.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.opusgpu_stream_pitch_get.argtypes = [vp, C.c_int, vp]
    lib.opusgpu_debug_stage_taps.argtypes = [vp, C.c_int, vp]
    lib.opusgpu_pages_demux.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    lib.opusgpu_pages_demux_into.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(vp)]
    lib.opusgpu_page_batch_arena_offset.argtypes = [vp]
    lib.opusgpu_page_batch_arena_offset.restype = C.c_size_t
    lib.opusgpu_page_batch_steps.argtypes = [vp]
    lib.opusgpu_page_batch_step.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]
    lib.opusgpu_page_batch_arena.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.opusgpu_page_batch_arena.restype = vp
    lib.opusgpu_page_batch_free.argtypes = [vp]
    lib.opusgpu_page_batch_free.restype = None
    lib.opusgpu_pages_crc_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.opusgpu_output_stage_device.argtypes = [vp, C.c_int, C.c_int, vp, C.c_longlong, vp, C.c_int, vp, OutputCfg, vp,
                                                C.c_longlong, vp]
    _lib = lib
    return lib


class OpusGpuError(RuntimeError):
    code = 0  # the negative OPUSGPU_* value the call returned


class BufferTooSmall(OpusGpuError):
    def __init__(self, need):
        super().__init__(f"output memory too small: {need} bytes needed")
        self.need = need


def packet_to_frames(packet: bytes, stream: int = 0):
    """Host-only: frame descriptors of one packet (list of (offset, len, flags)) or a negative code."""
    lib = load_lib()
    d = (FrameDesc * 48)()
    n = lib.opusgpu_packet_to_frames(packet, len(packet), stream, d)
    if n < 0:
        return n
    return [(d[i].offset, d[i].len, d[i].flags) for i in range(n)]


def empty_packet_to_frames(last_flags, decoder_channels, frame_size, stream: int = 0):
    """Host-only: the frames of an EMPTY packet in reference mode (opusgpu_empty_packet_to_frames): `last_flags` the flags of the
    stream's last accepted packet, negative when it has had none since its reset.  List of (offset, len, flags) or a negative code."""
    lib = load_lib()
    d = (FrameDesc * 48)()
    n = lib.opusgpu_empty_packet_to_frames(stream, last_flags, decoder_channels, frame_size, d)
    if n < 0:
        return n
    return [(d[i].offset, d[i].len, d[i].flags) for i in range(n)]


class PageBatch:
    """Decode steps made from a batch of Ogg pages by opusgpu_pages_demux (host only, include/opusgpu.h).
    `blob` holds the pages back to back, page i = blob[offsets[i] : offsets[i] + lens[i]]; stream_ids[i] is the decoder
    stream page i belongs to.  info: one PAGE_INFO_DTYPE record per page (status = frames contributed or PAGE_*)."""

    def __init__(self, blob, offsets, lens, stream_ids, flags=PAGES_VERIFY_CRC | PAGES_GROUP_BY_MODE, threads=1, out_mem=None):
        """out_mem: a uint8 array (16-byte aligned, e.g. page-locked by Context.host_register) that receives the step tables and the
        arena (opusgpu_pages_demux_into); raises BufferTooSmall(need) when it is too small.  Then `self.image` is the part of out_mem
        that holds [tables | arena] and `self.arena_offset` where the arena begins in it: one upload carries the whole batch."""
        lib = load_lib()
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.asarray(offsets, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        n = len(offsets)
        if not (len(lens) == n and len(ids) == n):
            raise ValueError("offsets, lens and stream_ids must have one entry per page")
        if n and (offsets.min() < 0 or (offsets + lens).max() > blob.size):
            raise ValueError("a page lies outside the blob")
        ptrs = (np.uint64(blob.ctypes.data) + offsets.astype(np.uint64)).astype(np.uint64)
        self.info = np.zeros(n, dtype=PAGE_INFO_DTYPE)
        h = C.c_void_p()
        self.image, self.arena_offset = None, 0
        if out_mem is None:
            r = lib.opusgpu_pages_demux(n, ptrs.ctypes.data, lens.ctypes.data, ids.ctypes.data, flags, threads,
                                        self.info.ctypes.data, C.byref(h))
        else:
            need = C.c_size_t()
            r = lib.opusgpu_pages_demux_into(n, ptrs.ctypes.data, lens.ctypes.data, ids.ctypes.data, flags, threads,
                                             self.info.ctypes.data, out_mem.ctypes.data, out_mem.nbytes, C.byref(need), C.byref(h))
            if r == -2:
                raise BufferTooSmall(need.value)
        if r != 0:
            raise OpusGpuError(f"opusgpu_pages_demux failed: {r}")
        self.lib, self.h = lib, h
        self.n_steps = lib.opusgpu_page_batch_steps(h)
        nbytes = C.c_size_t()
        a = lib.opusgpu_page_batch_arena(h, C.byref(nbytes))
        self.arena = np.ctypeslib.as_array((C.c_uint8 * nbytes.value).from_address(a)) if nbytes.value else np.zeros(0, np.uint8)
        if out_mem is not None:
            self.arena_offset = lib.opusgpu_page_batch_arena_offset(h)
            self.image = out_mem[:self.arena_offset + nbytes.value]
            self._out_mem = out_mem

    @classmethod
    def with_gpu_crc(cls, ctx, d_blob, blob, offsets, lens, stream_ids, flags=PAGES_GROUP_BY_MODE, threads=1):
        """The steps PageBatch(..., flags | PAGES_VERIFY_CRC) makes, with the checksums computed on the GPU
        (opusgpu_pages_crc_device) from the copy of the pages that lies in HBM at `d_blob` (same offsets as in `blob`,
        e.g. raw pages delivered by the work-queue scatter); the host demux then skips its own CRC pass.  Pages whose
        checksum does not match are kept out of the demux and reported as PAGE_BAD_CRC."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = len(offsets)
        st = np.zeros(n, dtype=np.int32)
        if n:
            bufs = [ctx.dev_alloc(8 * n), ctx.dev_alloc(4 * n), ctx.dev_alloc(4 * n)]
            try:
                ctx.h2d(bufs[0], offsets)
                ctx.h2d(bufs[1], lens)
                ctx.pages_crc_device(n, d_blob, bufs[0], bufs[1], bufs[2])
                ctx.synchronize()
                ctx.d2h(st, bufs[2])
            finally:
                for b in bufs:
                    ctx.dev_free(b)
        batch = cls(blob, offsets, np.where(st == 1, lens, 0), stream_ids, flags & ~PAGES_VERIFY_CRC, threads)
        bad_crc = (st == 0) & (batch.info["status"] != PAGE_BAD_STREAM)  # (the demux looks at the stream id first)
        batch.info["status"][bad_crc] = PAGE_BAD_CRC
        hdr = np.ascontiguousarray(blob, dtype=np.uint8)
        for i in np.nonzero(bad_crc)[0]:  # the header fields the demux reports for such a page
            h = hdr[offsets[i]:offsets[i] + 27]
            batch.info[i]["header_type"] = h[5]
            batch.info[i]["granulepos"] = h[6:14].view("<i8")[0]
            batch.info[i]["serial"], batch.info[i]["seqno"] = h[14:18].view("<u4")[0], h[18:22].view("<u4")[0]
        return batch

    def step(self, k):
        """-> (descriptors [DESC_DTYPE], page index of every slot [int32]); views into the batch, valid until close()."""
        d, sp = C.c_void_p(), C.c_void_p()
        n = self.lib.opusgpu_page_batch_step(self.h, k, C.byref(d), C.byref(sp))
        if n < 0:
            raise IndexError(k)
        if n == 0:
            return np.zeros(0, dtype=DESC_DTYPE), np.zeros(0, dtype=np.int32)
        descs = np.frombuffer((C.c_uint8 * (16 * n)).from_address(d.value), dtype=DESC_DTYPE)
        pages = np.frombuffer((C.c_uint8 * (4 * n)).from_address(sp.value), dtype=np.int32)
        return descs, pages

    def close(self):
        if self.h:
            self.lib.opusgpu_page_batch_free(self.h)
            self.h = None
            self.arena = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
            e = OpusGpuError(f"{what} failed: {rc} ({self.lib.opusgpu_last_error(self.h).decode()})")
            e.code = rc
            raise e

    def streams_alloc(self, n, channels):
        self._chk(self.lib.opusgpu_streams_alloc(self.h, n, channels), "opusgpu_streams_alloc")
        self.n_streams, self.channels = n, channels

    def set_mode(self, rfc):
        """RFC mode on / off (include/opusgpu.h, OPUSGPU_MODE_RFC): frames at the durations their TOC names."""
        self._chk(self.lib.opusgpu_set_mode(self.h, 1 if rfc else 0), "opusgpu_set_mode")

    def set_pipeline(self, on):
        """Pipelined decode steps (include/opusgpu.h, opusgpu_set_pipeline): step k+1's CELT parse next to step k's
        reconstruction.  The tables of a decode_step_device call must then be complete in device memory at the call."""
        self._chk(self.lib.opusgpu_set_pipeline(self.h, 1 if on else 0), "opusgpu_set_pipeline")

    def streams_reset(self, first, count, full=True):
        self._chk(self.lib.opusgpu_streams_reset(self.h, first, count, 1 if full else 0), "opusgpu_streams_reset")

    def decode_packets_fec(self, stream_ids, packets, frame_capacity=1):
        """RFC mode: packets[i] FOLLOWS a lost packet of its stream; produces the lost packet's audio from packets[i]'s forward
        error correction data where it has any, by concealment otherwise (opusgpu_decode_packets_fec).  Decode the packets
        themselves with decode_packets afterwards."""
        return self.decode_packets(stream_ids, packets, frame_capacity, _fn="opusgpu_decode_packets_fec")

    def decode_packets(self, stream_ids, packets, frame_capacity=1, _fn="opusgpu_decode_packets"):
        """Batched opus_multistream_decode: returns (pcm[n, cap*960, ch] int16, result[n] int32).
        RFC mode: an empty (or None) packet is a LOST packet, concealed for as long as the stream's last packet was.
        Reference mode: an empty packet is what the reference makes of one (include/opusgpu.h "EMPTY PACKETS"): passes of an empty
        frame in the stream's last mode, 960 samples each, frame_capacity of them or until one fails."""
        packets = [b"" if p is None else p for p in packets]
        n = len(packets)
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        lens = np.array([len(p) for p in packets], dtype=np.int32)
        bufs = [C.create_string_buffer(bytes(p), max(len(p), 1)) for p in packets]
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        pcm = np.zeros((n, frame_capacity * FRAME, self.channels), dtype=np.int16)
        res = np.zeros(n, dtype=np.int32)
        self._chk(getattr(self.lib, _fn)(self.h, n, ids.ctypes.data, C.addressof(ptrs), lens.ctypes.data,
                                         pcm.ctypes.data, frame_capacity, res.ctypes.data), _fn)
        return pcm, res

    def decode_packets_arena(self, stream_ids, arena, offsets, lens, frame_capacity=1, pcm=None):
        """opusgpu_decode_packets for packets that lie in one uint8 array (packet i = arena[offsets[i] : offsets[i] +
        lens[i]]): the pointer table is made by numpy, not packet by packet -- at 65,536 packets per call the Python-side
        marshalling of decode_packets costs several times the call itself.  `pcm`: an int16 array to decode into
        (n, frame_capacity * 960, channels), reused between calls; a fresh one otherwise."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offsets = np.asarray(offsets, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        n = len(lens)
        if len(offsets) != n or len(ids) != n:
            raise ValueError("stream_ids, offsets and lens must have one entry per packet")
        if n and (offsets.min() < 0 or (offsets + lens).max() > arena.size):
            raise ValueError("a packet lies outside the arena")
        ptrs = (np.uint64(arena.ctypes.data) + offsets.astype(np.uint64)).astype(np.uint64)
        shape = (n, frame_capacity * FRAME, self.channels)
        if pcm is None:
            pcm = np.empty(shape, dtype=np.int16)
            pcm[...] = 0
        elif pcm.shape != shape or pcm.dtype != np.int16 or not pcm.flags.c_contiguous:
            raise ValueError(f"pcm must be a C-contiguous int16 array of shape {shape}")
        res = np.zeros(n, dtype=np.int32)
        self._chk(self.lib.opusgpu_decode_packets(self.h, n, ids.ctypes.data, ptrs.ctypes.data, lens.ctypes.data,
                                                  pcm.ctypes.data, frame_capacity, res.ctypes.data),
                  "opusgpu_decode_packets")
        return pcm, res

    def decode_packets_raw(self, ids, ptrs, lens, pcm, res, frame_capacity=1):
        """opusgpu_decode_packets with every table made by the caller beforehand (int32 ids / lens, uint64 packet addresses, int16
        pcm, int32 res; all C-contiguous, one entry per packet): nothing but the call itself -- what a C caller pays."""
        self._chk(self.lib.opusgpu_decode_packets(self.h, len(lens), ids.ctypes.data, ptrs.ctypes.data, lens.ctypes.data,
                                                  pcm.ctypes.data, frame_capacity, res.ctypes.data), "opusgpu_decode_packets")

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

    def debug_stage_taps(self, slot):
        """Stage taps of slot `slot` of the last decode_step_device call (test / debug entry, include/opusgpu.h)."""
        t = StageTaps()
        self._chk(self.lib.opusgpu_debug_stage_taps(self.h, slot, C.byref(t)), "opusgpu_debug_stage_taps")
        return t

    def decode_step_device(self, n, d_descs, d_arena, d_pcm, d_result, stream=None, modes=0):
        """modes: 0 = not known, else a mask of HAS_SILK / HAS_HYBRID / HAS_CELT (opusgpu_decode_step_device_modes)."""
        if modes:
            self._chk(self.lib.opusgpu_decode_step_device_modes(self.h, n, d_descs, d_arena, d_pcm, d_result, stream, modes),
                      "opusgpu_decode_step_device_modes")
        else:
            self._chk(self.lib.opusgpu_decode_step_device(self.h, n, d_descs, d_arena, d_pcm, d_result, stream),
                      "opusgpu_decode_step_device")

    def decode_steps_device(self, n, d_descs, d_arena, d_pcm, d_result, stream=None, modes=0):
        """A window of consecutive steps in one call (opusgpu_decode_steps_device): n is a list of frame counts, the others lists
        of device pointers, one entry per step."""
        k = len(n)

        def ptrs(v):
            return (C.c_void_p * k)(*[p.value if isinstance(p, C.c_void_p) else int(p) for p in v])
        counts = (C.c_int32 * k)(*[int(x) for x in n])
        self._chk(self.lib.opusgpu_decode_steps_device(self.h, k, counts, ptrs(d_descs), ptrs(d_arena), ptrs(d_pcm), ptrs(d_result), stream, modes),
                  "opusgpu_decode_steps_device")

    def pages_crc_device(self, n_pages, d_blob, d_offsets, d_lens, d_status, stream=None):
        """Page checksums on the GPU (include/opusgpu.h): d_status[i] = 1 match, 0 mismatch, PAGE_BAD_CAPTURE malformed."""
        self._chk(self.lib.opusgpu_pages_crc_device(self.h, n_pages, d_blob, d_offsets, d_lens, d_status, stream),
                  "opusgpu_pages_crc_device")

    def output_stage_device(self, n_blocks, block_samples, d_pcm, pcm_stride, d_i2s, i2s_stride, d_valid=None, valid_all=0,
                            d_cfgs=None, volume=64, force_mono=False, bits=16, channels=2, stream=None):
        """The player's output stage on the GPU (include/opusgpu.h): PCM blocks -> 32-bit I2S words."""
        cfg = OutputCfg(volume, 1 if force_mono else 0, bits, channels)
        self._chk(self.lib.opusgpu_output_stage_device(self.h, n_blocks, block_samples, d_pcm, pcm_stride, d_valid, valid_all,
                                                       d_cfgs, cfg, d_i2s, i2s_stride, stream), "opusgpu_output_stage_device")

    def decode_step_by_kind(self, n_silk, n_hybrid, n_celt, d_descs, d_arena, d_pcm, d_result, keeps_kind=False, stream=None):
        """One step whose table is grouped by mode (OPUSGPU_PAGES_GROUP_BY_MODE: SILK-only, hybrid, CELT-only frames in that
        order), issued as up to three DECLARED sub-steps over the parts of the table, of d_pcm and of d_result: declared steps are
        what opusgpu_set_pipeline lets run ahead.  keeps_kind: OPUSGPU_STEP_KEEPS_MODE (include/opusgpu.h) -- the caller's word that
        a stream's mode never crosses between CELT-only and SILK-only / hybrid (config 5: it is fixed)."""
        def at(p, off):
            return C.c_void_p((p.value if isinstance(p, C.c_void_p) else int(p)) + off)
        if self.lib.opusgpu_get_mode(self.h) != 0:
            raise OpusGpuError("decode_step_by_kind: reference mode only (a frame's PCM block is 960 samples here; RFC mode's is 2880)")
        f0 = 0
        extra = STEP_KEEPS_MODE if keeps_kind else 0
        for cnt, mode in ((n_silk, HAS_SILK), (n_hybrid, HAS_HYBRID), (n_celt, HAS_CELT)):
            if cnt:
                self.decode_step_device(cnt, at(d_descs, 16 * f0), d_arena, at(d_pcm, f0 * 960 * self.channels * 2), at(d_result, 4 * f0),
                                        stream=stream, modes=mode | extra)
            f0 += cnt

    def decode_work_step(self, base, layout, k, d_pcm, d_result, by_kind=False, keeps_kind=False):
        """Step k of a packed work buffer (shard.pack_work / shard.WorkLayout) resident in HBM at address `base`: the
        step's descriptor table and the arena are used where they lie.  keeps_kind: OPUSGPU_STEP_KEEPS_MODE (a stream's mode is
        fixed): the step -- frames of every mode -- runs ahead like a declared one; by_kind: as three declared sub-steps
        (decode_step_by_kind), if the step's table is grouped by mode."""
        base = base.value if isinstance(base, C.c_void_p) else int(base)
        n = layout.counts[k]
        ns, nh = layout.mode_counts[k] if by_kind else (-1, -1)
        if ns >= 0:
            self.decode_step_by_kind(ns, nh, n - ns - nh, C.c_void_p(base + layout.desc_at[k]), C.c_void_p(base + layout.arena_at),
                                     d_pcm, d_result, keeps_kind=keeps_kind)
        else:
            self.decode_step_device(n, C.c_void_p(base + layout.desc_at[k]), C.c_void_p(base + layout.arena_at), d_pcm, d_result,
                                    modes=(HAS_SILK | HAS_HYBRID | HAS_CELT | STEP_KEEPS_MODE) if keeps_kind else 0)

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

    def event_synchronize(self, e):
        self._chk(self.lib.opusgpu_event_synchronize(self.h, e), "opusgpu_event_synchronize")

    # uploads next to the decode (include/opusgpu.h: opusgpu_upload_async and friends; ingest.py drives them)
    def upload_async(self, dptr, arr):
        """Queue a host -> HBM copy on the context's copy stream; `arr` must stay alive until a later fence has passed."""
        arr = np.ascontiguousarray(arr)
        self._chk(self.lib.opusgpu_upload_async(self.h, dptr, arr.ctypes.data, arr.nbytes), "opusgpu_upload_async")
        return arr

    def upload_fence(self, e):
        self._chk(self.lib.opusgpu_upload_fence(self.h, e), "opusgpu_upload_fence")

    def stream_wait_event(self, e, stream=None):
        self._chk(self.lib.opusgpu_stream_wait_event(self.h, e, stream), "opusgpu_stream_wait_event")

    def host_register(self, arr):
        self._chk(self.lib.opusgpu_host_register(self.h, arr.ctypes.data, arr.nbytes), "opusgpu_host_register")

    def host_unregister(self, arr):
        self._chk(self.lib.opusgpu_host_unregister(self.h, arr.ctypes.data), "opusgpu_host_unregister")


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


_CRC_T = None


def ogg_crc_rows(rows):
    """Ogg page CRC-32 (polynomial 0x04c11db7, MSB first, zero start; src/ogg.cpp:439) of every row of a uint8 matrix,
    all rows in step: one table lookup per byte column."""
    global _CRC_T
    if _CRC_T is None:
        t = np.arange(256, dtype=np.uint64) << np.uint64(24)
        for _ in range(8):
            t = np.where(t & np.uint64(0x80000000), (t << np.uint64(1)) ^ np.uint64(0x04C11DB7), t << np.uint64(1)) & np.uint64(0xFFFFFFFF)
        _CRC_T = t.astype(np.uint32)
    crc = np.zeros(rows.shape[0], dtype=np.uint32)
    for j in range(rows.shape[1]):
        crc = (crc << np.uint32(8)) ^ _CRC_T[(crc >> np.uint32(24)) ^ rows[:, j]]
    return crc


def build_pages(toc, payloads, serials, seqno=2, granule_step=960):
    """Synthetic Ogg pages, one per stream: page s carries payloads[:, s] as code-0 packets (TOC + payload each).
    payloads uint8 [packets, n, L] with L + 1 < 255 (one lacing value per packet).  Returns uint8 [n, page_len]:
    "OggS", version 0, header type 0, granule position packets * granule_step, serial number, page sequence number,
    CRC, segment table, packets (src/ogg.cpp:439-480 for the layout the reference checks)."""
    npk, n, L = payloads.shape
    if L + 1 >= 255 or npk > 255:
        raise ValueError("one lacing value per packet, at most 255 packets")
    hdr = 27 + npk
    pages = np.zeros((n, hdr + npk * (L + 1)), dtype=np.uint8)
    pages[:, 0:4] = np.frombuffer(b"OggS", dtype=np.uint8)
    pages[:, 6:14] = np.frombuffer(np.int64(npk * granule_step).tobytes(), dtype=np.uint8)
    pages[:, 14:18] = np.asarray(serials, dtype="<u4").reshape(n, 1).view(np.uint8)
    pages[:, 18:22] = np.frombuffer(np.uint32(seqno).tobytes(), dtype=np.uint8)
    pages[:, 26] = npk
    pages[:, 27:hdr] = L + 1
    body = pages[:, hdr:].reshape(n, npk, L + 1)
    body[:, :, 0] = toc
    body[:, :, 1:] = payloads.transpose(1, 0, 2)
    pages[:, 22:26] = ogg_crc_rows(pages).astype("<u4").reshape(n, 1).view(np.uint8)
    return pages


def silk_header_key(first_payload_bytes, stereo):
    """What a SILK-only / hybrid frame's header will make its decoder do, read off the frame's first byte: the VAD and LBRR flags
    are the range coder's first symbols, each of probability 1/2 -- exactly the byte's top bits (reference src/silk.cpp:1568-1573:
    VAD flag and LBRR flag of the mid channel, then of the side channel).  -> 2-bit key: bit 0 the mid channel carries an LBRR
    frame, bit 1 the side channel does (each is a whole extra frame of side information and pulses that the decoder must read
    past, src/silk.cpp:1590-1616).  Frames handed to the lane-per-frame parse kernel in key order make its waves uniform: a wave
    whose 32 frames have no LBRR data skips those passes instead of idling through them.  The bit positions are those of
    a frame decoded as 20 ms -- every frame in reference mode (Q6), the only mode with such steps."""
    b = np.asarray(first_payload_bytes, dtype=np.uint8)
    key = (b >> 6) & 1
    if stereo:
        key = key | (((b >> 4) & 1) << 1)
    return key.astype(np.uint8)


def build_step(toc, payloads, order_by_header=False):
    """Arena + descriptors for one decode step: payloads uint8 [n_streams, L], one code-0 packet each.
    The arena holds TOC + payload per stream; descriptors point past the TOC byte.
    order_by_header: the table in the order of silk_header_key (stable; SILK-only / hybrid TOCs only) -- slot j of the step's
    PCM and results then belongs to stream descs["stream"][j]."""
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
    if order_by_header and mode != 2 and L > 0:
        descs = descs[np.argsort(silk_header_key(payloads[:, 0], bool(toc & 4)), kind="stable")]
    return arena.reshape(-1), descs

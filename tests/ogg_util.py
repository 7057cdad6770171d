"""Test helper: build Ogg Opus files in memory (pages, lacing, CRC) -- used for survey KAT 3 and container tests."""
import random
import struct


def _crc_table():
    t = []
    for i in range(256):
        r = i << 24
        for _ in range(8):
            r = ((r << 1) ^ 0x04C11DB7) if r & 0x80000000 else (r << 1)
            r &= 0xFFFFFFFF
        t.append(r)
    return t


_T = _crc_table()


def ogg_crc(data: bytes) -> int:
    crc = 0
    for b in data:
        crc = ((crc << 8) & 0xFFFFFFFF) ^ _T[((crc >> 24) & 0xFF) ^ b]
    return crc


def page(serial, seqno, granule, packets, bos=False, eos=False, continued=False):
    lacing = bytearray()
    body = bytearray()
    for p in packets:
        n = len(p)
        while n >= 255:
            lacing.append(255)
            n -= 255
        lacing.append(n)
        body += p
    assert len(lacing) <= 255
    flags = (1 if continued else 0) | (2 if bos else 0) | (4 if eos else 0)
    hdr = bytearray(b"OggS\x00" + bytes([flags]) + struct.pack("<qIII", granule, serial, seqno, 0) + bytes([len(lacing)]) + lacing)
    crc = ogg_crc(bytes(hdr) + bytes(body))
    hdr[22:26] = struct.pack("<I", crc)
    return bytes(hdr) + bytes(body)


def opus_head(channels=2, pre_skip=312, rate=48000, gain=0, family=0):
    return b"OpusHead" + bytes([1, channels]) + struct.pack("<HIhB", pre_skip, rate, gain, family)


def opus_tags(vendor=b"test"):
    return b"OpusTags" + struct.pack("<I", len(vendor)) + vendor + struct.pack("<I", 0)


def kat3_file():
    """SURVEY.md appendix B, smoke KAT 3."""
    random.seed(7)
    serial = 0x1234
    out = page(serial, 0, 0, [opus_head()], bos=True) + page(serial, 1, 0, [opus_tags()])
    gp = 0
    for pg in range(10):
        pk = [bytes([0xFC]) + bytes(random.getrandbits(8) for _ in range(160)) for _ in range(10)]
        gp += 9600
        out += page(serial, 2 + pg, gp, pk, eos=(pg == 9))
    return out

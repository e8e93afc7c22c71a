"""AFSK-1200 / HDLC decoder: crc_good against the reference's ax25.c build, and the oracle decoder on
synthesized AX.25 frames with known content."""
import numpy as np
import pytest

import kq_oracle as ko
from common import afsk_audio, afsk_bits, ax25_fcs

FRAME_A = bytes([0x82, 0xA0, 0xA4, 0xA6, 0x40, 0x40, 0x60, 0x96, 0x82, 0x72, 0xA2, 0x40, 0x40, 0x61, 0x03, 0xF0]) + \
    b"!4903.50N/07201.75W-ka9q-radio on MI355X"
FRAME_B = bytes(range(1, 60))
FRAME_C = bytes([0xFF] * 5 + [0x7E, 0x7D, 0x00, 0x3E, 0xF8]) + b"stuffing \xff\xff\xff\x1f"


@pytest.mark.skipif(ko.ref_ax25_lib() is None, reason="oracle/_ref not built (no reference tree)")
def test_crc_good_matches_reference_build():
    R = ko.ref_ax25_lib()
    rng = np.random.default_rng(3)
    for i in range(300):
        n = int(rng.integers(1, 400))
        f = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        if i % 3 == 0:
            f = f + ax25_fcs(f)
        assert ko.crc_good(f) == R.crc_good(f, len(f))


def test_crc_known_answers():
    assert ko.crc_good(FRAME_A + ax25_fcs(FRAME_A)) == 1
    bad = bytearray(FRAME_A + ax25_fcs(FRAME_A))
    bad[5] ^= 0x10
    assert ko.crc_good(bad) == 0
    # CRC-16/X.25 check value of "123456789" is 0x906e
    assert ax25_fcs(b"123456789") == bytes([0x6E, 0x90])


def test_decodes_clean_frames():
    x = afsk_audio(afsk_bits([FRAME_A, FRAME_B, FRAME_C]))
    d = ko.Afsk()
    d.push(np.concatenate([np.zeros(777, np.float32), x, np.zeros(3000, np.float32)]))
    got = d.frames()
    assert got == [f + ax25_fcs(f) for f in (FRAME_A, FRAME_B, FRAME_C)]


def test_decodes_with_noise_and_clock_offset_in_odd_chunks():
    x = afsk_audio(afsk_bits([FRAME_B, FRAME_A], lead_flags=20), noise=0.05, seed=4, clock_ppm=800.0)
    d = ko.Afsk()
    rng = np.random.default_rng(0)
    i = 0
    while i < len(x):                       # packet.c:204-211: input arrives in arbitrary packet sizes
        n = int(rng.integers(1, 700))
        d.push(x[i:i + n])
        i += n
    d.push(np.zeros(2000, np.float32))
    assert d.frames() == [FRAME_B + ax25_fcs(FRAME_B), FRAME_A + ax25_fcs(FRAME_A)]


def test_bad_fcs_and_abort_are_dropped():
    bits = afsk_bits([FRAME_A])
    # flip one payload bit well inside the frame: CRC fails, nothing is emitted
    bits2 = list(bits)
    bits2[8 * 8 + 50] ^= 1
    d = ko.Afsk()
    d.push(afsk_audio(bits2))
    d.push(np.zeros(2000, np.float32))
    assert d.frames() == []
    # seven ones in a row abort the frame (packet.c:391-398); a following good frame still decodes
    flag = [0, 1, 1, 1, 1, 1, 1, 0]
    body = afsk_bits([FRAME_B], lead_flags=0, gap_flags=0)
    stream = flag * 6 + body[:100] + [1] * 9 + flag * 4 + afsk_bits([FRAME_C], lead_flags=0, gap_flags=2)
    d = ko.Afsk()
    d.push(afsk_audio(stream))
    d.push(np.zeros(2000, np.float32))
    assert d.frames() == [FRAME_C + ax25_fcs(FRAME_C)]


def test_pcm_ingest_keeps_the_unsigned_quirk():
    # packet.c:207: ntohs() is unsigned, so a negative PCM word w arrives as (w + 65536) / 32768
    x = afsk_audio(afsk_bits([FRAME_A]), amp=0.4)
    pcm = np.round(x * 32767).astype(">i2")
    a, b = ko.Afsk(), ko.Afsk()
    a.push_pcm_be(pcm.tobytes())
    w = pcm.astype(np.int32)
    b.push(((w + 65536 * (w < 0)) * np.float32(1.0 / 32768)).astype(np.float32))
    np.testing.assert_array_equal(a.filter_output(), b.filter_output())
    assert a.state() == b.state()
    a.push(np.zeros(2000, np.float32))
    b.push(np.zeros(2000, np.float32))
    assert a.frames() == b.frames()

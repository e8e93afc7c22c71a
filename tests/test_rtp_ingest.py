"""I/Q packet ingest rules: RTP header, payload types, sequence / timestamp bookkeeping
(multicast.c:242-277, 305-340; main.c:315-341; radio.c:62-104) -- oracle restatement on hand-built packets with known
outcomes (multicast.c needs <bsd/string.h>, so the reference itself cannot be built here: parity unpinned)."""
import struct

import numpy as np

import kq_oracle as ko


def rtp_packet(seq, ts, ssrc, payload, ptype=97, csrc=(), ext=None, pad=0, status=b"\0" * 24):
    b0 = (2 << 6) | ((1 if pad else 0) << 5) | ((1 if ext is not None else 0) << 4) | len(csrc)
    pkt = struct.pack(">BBHII", b0, ptype, seq & 0xFFFF, ts & 0xFFFFFFFF, ssrc)
    for c in csrc:
        pkt += struct.pack(">I", c)
    if ext is not None:                      # the reference skips 4 + (4 + length) bytes, length taken as bytes
        pkt += struct.pack(">HH", 0xBEEF, len(ext) - 4) + ext
    pkt += status + payload
    if pad:
        pkt += b"\0" * (pad - 1) + bytes([pad])
    return pkt


def iq16(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, 2)) * 3000).astype("<i2").tobytes()


def test_in_sequence_gap_and_duplicate():
    ing = ko.IqIngest()
    n = 240
    # 13 packets in sequence, then one whose timestamp jumps by 100 lost samples, then a stale duplicate
    ts, seq, total = 1000, 65530, 0                       # sequence number wraps on the way
    for k in range(13):
        r = ing.packet(rtp_packet(seq, ts, 0xABCD, iq16(n, k)))
        assert r == (0, 12 + 24, n, 1)
        ts += n
        seq += 1
        total += n
    r = ing.packet(rtp_packet(seq, ts + 100, 0xABCD, iq16(n, 99)))
    assert r == (100, 36, n, 1)
    total += 100 + n
    assert ing.packet(rtp_packet(seq - 3, ts - 3 * n, 0xABCD, iq16(n, 5))) is None     # stale duplicate
    assert ing.samples == total == 13 * 240 + 100 + 240
    assert ing.rtp.dupes == 1 and ing.rtp.drops == 0 and ing.rtp.packets == 15
    # a packet two sequence numbers ahead (one datagram lost) with the matching timestamp gap
    seq, ts = seq + 1 + 1, ts + 100 + n + n
    r = ing.packet(rtp_packet(seq, ts, 0xABCD, iq16(n, 7)))
    assert r == (n, 36, n, 1) and ing.rtp.drops == 1


def test_header_variants_and_payload_types():
    ing = ko.IqIngest()
    p8 = bytes(range(100))                                # 50 int8 I/Q samples
    assert ing.packet(rtp_packet(1, 0, 7, p8, ptype=98)) == (0, 36, 50, 2)
    assert ing.packet(rtp_packet(2, 50, 7, p8, ptype=98, csrc=(1, 2, 3))) == (0, 36 + 12, 50, 2)
    assert ing.packet(rtp_packet(3, 100, 7, p8, ptype=98, ext=b"\1" * 8)) == (0, 36 + 4 + 8, 50, 2)
    assert ing.packet(rtp_packet(4, 150, 7, p8, ptype=98, pad=6)) == (0, 36, 50, 2)
    assert ing.packet(rtp_packet(5, 200, 7, p8, ptype=96)) is None            # not an I/Q payload type
    assert ing.packet(b"\x80\x61\0\1") is None                               # shorter than an RTP header
    assert ing.rtp.packets == 4                                               # ignored datagrams never reach rtp_process


def test_ssrc_change_and_wild_jumps():
    ing = ko.IqIngest()
    assert ing.packet(rtp_packet(10, 5000, 1, iq16(100, 0))) == (0, 36, 100, 1)
    assert ing.samples == 100
    # new SSRC: counters restart (radio.c:73-77, multicast.c:306-321)
    assert ing.packet(rtp_packet(777, 42, 2, iq16(100, 1))) == (0, 36, 100, 1)
    assert ing.samples == 100 and ing.rtp.packets == 1 and ing.rtp.ssrc == 2
    # a jump of more than one second of 192 kHz samples is dropped (radio.c:79-82) but the state has advanced
    assert ing.packet(rtp_packet(778, 142 + 192001, 2, iq16(100, 2))) is None
    assert ing.packet(rtp_packet(779, 142 + 192001 + 100, 2, iq16(100, 3))) == (0, 36, 100, 1)
    # timestamp going backwards with a fresh sequence number: dropped, sequence state still advances
    assert ing.packet(rtp_packet(780, 10, 2, iq16(100, 4))) is None
    assert ing.rtp.seq == 781

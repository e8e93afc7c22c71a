"""Pins the oracle's PCM packetiser (kq_chan.c kqo_pcm_rtp; the checker of kq_bank_pull_pcm / kq_bank_pull_rtp_audio)
on the reference: /root/reference/audio.c compiled in place (oracle/_ref/libref_audio.so) and driven through a
socketpair by oracle/ref_audio_capture.c.  Pinned: scaleclip (audio.c:22-28), the 480-word packets, the silent
packets that are not sent while the timestamp advances, the marker on resume, seq / timestamp / packets / bytes
(audio.c:32-132), payload bytes.  Not pinned: the byte layout of the RTP header (hton_rtp is multicast.c's, which
this image cannot build): the harness records the header FIELDS audio.c filled in, and they are compared with the
fields parsed from the oracle's RFC 3550 header.

tests/golden/audio_ref.npz holds inputs and the reference's datagrams (generated here: run this file as a script)
for boxes without oracle/_ref."""
import ctypes as C
import os

import numpy as np
import pytest

import kq_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_audio.so")
GOLD = os.path.join(ROOT, "tests", "golden", "audio_ref.npz")


class RefState(C.Structure):
    _fields_ = [("ssrc", C.c_uint32), ("seq", C.c_uint16), ("timestamp", C.c_uint32), ("silent", C.c_int),
                ("packets", C.c_longlong), ("bytes", C.c_longlong)]


def _cases():
    rng = np.random.default_rng(20181202)
    out = []
    a = (0.4 * rng.standard_normal(2048)).astype(np.float32)
    a[100:110] = [1.0, -1.0, 1.5, -1.5, 0.99999, -0.99999, 3.0e-5, -3.0e-5, 3.06e-5, -0.0]   # clip edges, sub-LSB values
    out.append(("mono_clip", a, 0))
    b = (0.2 * rng.standard_normal(3000)).astype(np.float32)
    b[480:1440] = 0                       # two whole packets of silence, then the marker
    b[1440:1920] = 1e-6                   # below one LSB: truncates to zero words, silent as well
    out.append(("mono_silence", b, 0))
    out.append(("mono_short", (0.1 * rng.standard_normal(37)).astype(np.float32), 0))
    out.append(("mono_all_zero", np.zeros(1000, np.float32), 0))
    c = (0.3 * rng.standard_normal(2 * 1111)).astype(np.float32)
    c[480:960] = 0
    out.append(("stereo_ragged", c, 1))
    out.append(("mono_long", (0.25 * rng.standard_normal(48000)).astype(np.float32), 0))
    return out


def _reference(state, audio, stereo):
    lib = C.CDLL(REF)
    buf = C.create_string_buffer(4 * len(audio) + 65536)
    used = C.c_int()
    n = lib.ref_audio_send(C.byref(state), audio.ctypes.data_as(C.POINTER(C.c_float)), len(audio), stereo,
                           C.cast(buf, C.POINTER(C.c_ubyte)), len(buf), C.byref(used))
    assert n >= 0
    return ko.split_packets(buf.raw[:used.value])


def _fields_capture(p):      # ref_audio_capture.c's record
    assert p[0] == 0xA5
    return dict(marker=p[1] >> 7, type=p[1] & 0x7f, seq=p[2] | p[3] << 8, timestamp=int.from_bytes(p[4:8], "little"),
                ssrc=int.from_bytes(p[8:12], "little"))


def _fields_rtp(p):          # RFC 3550 header as the oracle (and multicast.c:282-294) writes it
    assert p[0] == 0x80
    return dict(marker=p[1] >> 7, type=p[1] & 0x7f, seq=int.from_bytes(p[2:4], "big"),
                timestamp=int.from_bytes(p[4:8], "big"), ssrc=int.from_bytes(p[8:12], "big"))


def _check(ref_packets, ref_state, audio, stereo, start):
    o = ko.OutRtp(start["ssrc"], start["seq"], start["timestamp"], start["silent"], 0, 0)
    got = o.packetize(audio, stereo)
    assert len(got) == len(ref_packets)
    for g, r in zip(got, ref_packets):
        assert _fields_rtp(g) == _fields_capture(r)
        assert g[12:] == r[12:]                                   # payload: big-endian clipped words
    for k in ("seq", "timestamp", "silent", "packets", "bytes"):
        assert getattr(o, k) == ref_state[k], k


START = dict(ssrc=0x6B613971, seq=65530, timestamp=0xFFFFFF00, silent=1)   # both counters wrap inside the cases


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libref_audio.so not built (needs the reference tree)")
def test_oracle_packetiser_matches_reference_build():
    st = RefState(START["ssrc"], START["seq"], START["timestamp"], START["silent"], 0, 0)
    o_start = dict(START)
    for name, audio, stereo in _cases():        # state carried from case to case, as from block to block
        before = dict(ssrc=st.ssrc, seq=st.seq, timestamp=st.timestamp, silent=st.silent)
        p0, b0 = st.packets, st.bytes
        pk = _reference(st, audio, stereo)
        after = dict(seq=st.seq, timestamp=st.timestamp, silent=st.silent, packets=st.packets - p0, bytes=st.bytes - b0)
        _check(pk, after, audio, stereo, before)
    assert o_start == START


def test_oracle_packetiser_matches_committed_reference_vectors():
    z = np.load(GOLD, allow_pickle=False)
    names = [str(n) for n in z["names"]]
    assert names == [c[0] for c in _cases()]
    for i, (name, audio, stereo) in enumerate(_cases()):
        np.testing.assert_array_equal(audio, z["in_%d" % i])                   # the generator has not drifted
        start = dict(zip(("ssrc", "seq", "timestamp", "silent"), (int(v) for v in z["start_%d" % i])))
        after = dict(zip(("seq", "timestamp", "silent", "packets", "bytes"), (int(v) for v in z["after_%d" % i])))
        _check(ko.split_packets(z["out_%d" % i].tobytes()), after, audio, stereo, start)


if __name__ == "__main__":
    st = RefState(START["ssrc"], START["seq"], START["timestamp"], START["silent"], 0, 0)
    d = {"names": np.array([c[0] for c in _cases()])}
    for i, (name, audio, stereo) in enumerate(_cases()):
        d["in_%d" % i] = audio
        d["start_%d" % i] = np.array([st.ssrc, st.seq, st.timestamp, st.silent], np.int64)
        p0, b0 = st.packets, st.bytes
        pk = _reference(st, audio, stereo)
        d["after_%d" % i] = np.array([st.seq, st.timestamp, st.silent, st.packets - p0, st.bytes - b0], np.int64)
        blob = b"".join(len(p).to_bytes(2, "little") + p for p in pk)
        d["out_%d" % i] = np.frombuffer(blob, np.uint8)
        print(name, len(pk), "datagrams")
    np.savez_compressed(GOLD, **d)
    print("wrote", GOLD, os.path.getsize(GOLD), "bytes")

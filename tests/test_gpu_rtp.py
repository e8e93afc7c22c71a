"""kq_bank_push_rtp: datagrams straight off the wire (int16 and int8 I/Q, a timestamp gap, a duplicate, header
extras) against the oracle fed by the same decisions; sample counts exact, audio to the parity tolerance."""
import numpy as np
import pytest

import ka9q_sdr_amd as kq
import kq_oracle as ko
from common import bank_cfg, oracle_cfg, rel_rms
from ka9q_sdr_amd import workload as wl
from test_rtp_ingest import rtp_packet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fs,L,M,D", [(192000, 512, 513, 4),
                                       (192000, 3840, 4353, 4),      # the reference's default -L / -M: M - 1 > L
                                       (240000, 4800, 4801, 5),      # N = 9600
                                       (48000, 960, 961, 1)])        # decimate 1
def test_packet_stream_matches_oracle(gpu, fs, L, M, D):
    g = dict(samprate=fs, L=L, M=M, D=D)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-(wl.emitter_freq(24, fs) + 3.0))]
    iq = wl.make_iq(g["samprate"], 16 * g["L"] + 2000, seed=31, emitters=range(20, 30))
    i16 = np.stack([np.round(iq.real * 20000), np.round(iq.imag * 20000)], axis=1).astype("<i2")
    i8 = np.stack([np.round(iq.real * 100), np.round(iq.imag * 100)], axis=1).astype("i1")
    p = dict(plan[0])
    bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], 1, 32)
    bank.add_channel(bank_cfg(p))
    ch = ko.Channel(oracle_cfg(p, g["samprate"], g["L"], g["M"], g["D"]))
    ing = ko.IqIngest()
    want, got = [], []
    n, pos, seq, ts = 240, 0, 65500, 4000000000                  # sequence and timestamp both wrap on the way
    packets = []
    npk = 16 * g["L"] // n
    for k in range(npk):
        kind = 98 if 10 <= k < 14 else 97                        # a few int8 packets in between
        body = (i8 if kind == 98 else i16)[pos:pos + n].tobytes()
        extra = dict(csrc=(5,)) if k == 3 else dict(pad=4) if k == 4 else {}
        if k == 17:
            ts += 1000                                           # 1000 samples lost: zero fill crosses two block ends
        packets.append(rtp_packet(seq, ts, 0x1234, body, ptype=kind, **extra))
        if k == 20:
            packets.append(packets[-3])                          # a stale duplicate arrives late
        pos += n
        ts += n
        seq += 1
    for pkt in packets:
        r = ing.packet(pkt)
        nb_before = bank.blocks_ready()
        added = bank.push_rtp(pkt)
        if r is None:
            assert added == 0 and bank.blocks_ready() == nb_before
            continue
        zeros, off, count, fmt = r
        assert added == zeros + count
        if zeros:
            want += ch.zero_fill(zeros)
        want += ch.push_raw(pkt[off:off + count * (4 if fmt == 1 else 2)], count, fmt)
        nb = bank.process()
        got += [(bank.audio(0, b), bank.status(0, b)) for b in range(nb)]
    c = bank.rtp_counters()
    assert (c["samples"], c["packets"], c["dupes"], c["drops"]) == (ing.samples, ing.rtp.packets, ing.rtp.dupes, ing.rtp.drops)
    assert c["samples"] == npk * n + 1000 and c["dupes"] == 1
    assert (c["next_seq"], c["next_timestamp"], c["ssrc"]) == (ing.rtp.seq, ing.rtp.timestamp, 0x1234)
    assert len(got) == len(want) == (npk * n + 1000) // g["L"]
    for (ga, gs), (wa, ws) in zip(got, want):
        assert gs["nout"] == ws["nout"] and gs["squelch_count"] == ws["squelch_count"]
        np.testing.assert_allclose(gs["if_power"], ws["if_power"], rtol=2e-4, atol=1e-12)
    assert rel_rms(np.concatenate([a for a, _ in got]), np.concatenate([a for a, _ in want])) < 1e-5
    bank.close()


def test_recording_file_playback(gpu, tmp_path):
    """iqrecord-style file (s16le + xattrs) played into the bank == the same int16 samples pushed directly"""
    from ka9q_sdr_amd import iqfile
    g = dict(samprate=192000, L=512, M=513, D=4)
    p = wl.channel_plan("cfg1", 1)[0]
    iq = wl.make_iq(g["samprate"], 7 * g["L"] + 100, seed=33)
    i16 = np.stack([np.round(iq.real * 20000), np.round(iq.imag * 20000)], axis=1).astype(np.int16)
    path = str(tmp_path / "rec")
    iqfile.write_recording(path, i16, g["samprate"], frequency=10.0e6)
    a = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], 1, 4)
    b = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], 1, 8)
    a.add_channel(bank_cfg(p))
    b.add_channel(bank_cfg(p))
    got = []
    for nb in iqfile.play_into(a, path, chunk_blocks=3):
        got += [a.audio(0, k) for k in range(nb)]
    b.push_iq(i16)
    assert b.process() == 7
    want = [b.audio(0, k) for k in range(7)]
    assert len(got) == 7
    assert rel_rms(np.concatenate(got), np.concatenate(want)) < 1e-6
    a.close()
    b.close()


def test_random_packet_sequences_keep_the_books_like_the_oracle(gpu):
    """Differential test of the datagram bookkeeping (two independent implementations of multicast.c:305-340 and
    radio.c:62-104): 600 datagrams with random duplicates, reordering, losses, timestamp jumps in both directions,
    SSRC changes, both payload types, padding / CSRC / extension headers, and junk; after every datagram the library's
    counters and the number of samples it queued must equal the oracle's."""
    g = dict(samprate=192000, L=512, M=513, D=4)
    bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], 1, 64)
    bank.add_channel(bank_cfg(wl.channel_plan("cfg1", 1)[0]))
    ing = ko.IqIngest()
    rng = np.random.default_rng(77)
    seq, ts, ssrc = 65000, 0xFFFFF000, 5
    queued = 0
    for k in range(600):
        n = int(rng.integers(1, 200))
        kind = 98 if rng.random() < 0.3 else 97
        body = rng.integers(-100, 100, n * 2).astype("i1" if kind == 98 else "<i2").tobytes()
        r = rng.random()
        s_seq, s_ts, s_ssrc, ptype = seq, ts, ssrc, kind
        if r < 0.08:
            s_seq = seq - int(rng.integers(1, 5))                 # duplicate / late
        elif r < 0.16:
            s_seq = seq + int(rng.integers(1, 4))                 # datagrams lost
            s_ts = ts + int(rng.integers(0, 3000))
        elif r < 0.22:
            s_ts = ts + int(rng.integers(1, 5000))                # timestamp gap
        elif r < 0.26:
            s_ts = ts - int(rng.integers(1, 5000))                # timestamp going backwards
        elif r < 0.28:
            s_ts = ts + 192001 + int(rng.integers(0, 1000))       # wild jump
        elif r < 0.30:
            s_ssrc = ssrc = int(rng.integers(1, 1 << 31))         # sender restarted
            s_seq, s_ts = int(rng.integers(0, 65536)), int(rng.integers(0, 1 << 32))
        elif r < 0.33:
            ptype = 96                                            # not I/Q
        extra = {}
        if rng.random() < 0.1:
            extra["csrc"] = tuple(range(int(rng.integers(1, 4))))
        if rng.random() < 0.1:
            extra["pad"] = int(rng.integers(1, 9))
        if rng.random() < 0.05:
            extra["ext"] = bytes(8)
        pkt = rtp_packet(s_seq, s_ts, s_ssrc, body, ptype=ptype, **extra)
        if rng.random() < 0.02:
            pkt = pkt[:int(rng.integers(0, 12))]                  # truncated junk
        want = ing.packet(pkt)
        added = bank.push_rtp(pkt)
        assert added == (0 if want is None else want[0] + want[2]), k
        queued += added
        if want is not None:                                      # the sender moves on from what was accepted
            seq, ts = (s_seq + 1) & 0xFFFF, (s_ts + n) & 0xFFFFFFFF
        c = bank.rtp_counters()
        assert (c["samples"], c["packets"], c["dupes"], c["drops"], c["next_seq"], c["next_timestamp"], c["ssrc"]) == \
               (ing.samples, ing.rtp.packets, ing.rtp.dupes, ing.rtp.drops, ing.rtp.seq, ing.rtp.timestamp, ing.rtp.ssrc), k
        if bank.blocks_ready() >= 32:
            queued -= bank.process() * g["L"]
        assert bank.blocks_ready() == queued // g["L"]
    bank.close()


def test_gap_larger_than_the_free_ring_is_retried_not_lost(gpu):
    """A timestamp gap whose zero fill does not fit the ring right now, with whole blocks waiting: the datagram is
    refused with nothing consumed (it must not count as a duplicate when handed in again), and after a process call it
    goes through with the reference's sample count (radio.c:83-100 always injects the zeros)."""
    g = dict(samprate=192000, L=512, M=513, D=4)
    bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], 1, 2)          # 2 blocks + 511 samples of slack = 1535
    bank.add_channel(bank_cfg(wl.channel_plan("cfg1", 1)[0]))
    ing = ko.IqIngest()
    body = lambda n: np.full(2 * n, 1000, "<i2").tobytes()
    first = rtp_packet(10, 1000, 5, body(1100))
    assert bank.push_rtp(first) == 1100 and ing.packet(first) is not None
    gap = rtp_packet(11, 2100 + 300, 5, body(200))                        # 300 lost samples + 200: 1600 > 1535
    before = bank.rtp_counters()
    assert bank.push_rtp(gap) is None
    assert bank.rtp_counters() == before                                   # sequence / timestamp / counters untouched
    assert bank.process() == 2                                             # two blocks out, 76 samples stay
    want = ing.packet(gap)
    assert bank.push_rtp(gap) == want[0] + want[2] == 500
    c = bank.rtp_counters()
    assert (c["samples"], c["packets"], c["dupes"], c["next_seq"], c["next_timestamp"]) == (ing.samples, 2, 0, 12, 2600)
    bank.close()


@pytest.mark.parametrize("max_blocks", [1, 2])
def test_partly_filled_block_never_blocks_a_packet_that_fits_the_ring(gpu, max_blocks):
    """The two stuck states of the earlier room check: (pending % L) + need > max_blocks * L with need <= max_blocks * L
    and no whole block waiting (a retry could never succeed), e.g. pending = 500, need = 600 with two blocks of 512; and a
    bank of one block, where any packet that straddles a block end did this.  The ring's L - 1 samples of slack take it."""
    g = dict(samprate=192000, L=512, M=513, D=4)
    bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], 1, max_blocks)
    bank.add_channel(bank_cfg(wl.channel_plan("cfg1", 1)[0]))
    ing = ko.IqIngest()
    body = lambda n: np.full(2 * n, 500, "<i2").tobytes()
    seq, ts, total = 1, 100, 0
    need = 600 if max_blocks == 2 else 400
    for i, n in enumerate((500, need - 100, 300, 512, 37, 480)):
        gapn = 100 if i == 1 else 0                                           # the second packet also brings a 100-sample gap
        ts += gapn
        pkt = rtp_packet(seq, ts, 9, body(n))
        want = ing.packet(pkt)
        tries = 0
        while True:
            got = bank.push_rtp(pkt)
            if got is not None:
                break
            tries += 1
            assert tries < 4 and bank.process() >= 1                          # -2 only ever with a whole block waiting
        assert got == want[0] + want[2] == gapn + n
        total += got
        seq, ts = seq + 1, ts + n
        c = bank.rtp_counters()
        assert (c["samples"], c["packets"], c["drops"], c["next_seq"], c["next_timestamp"]) == \
               (ing.samples, ing.rtp.packets, ing.rtp.drops, ing.rtp.seq, ing.rtp.timestamp)
    assert total == ing.samples
    bank.close()


def test_gap_larger_than_the_whole_ring_is_filled_ring_by_ring(gpu):
    """A 5000-sample gap into a ring of 2 x 512 (+ 511): rtp state must not freeze (the earlier check refused the packet,
    and every later one, until the gap passed 192000).  Each round fills the ring with zeros and moves the timestamp on;
    the audio is the oracle's, whose zero fill runs the oscillators and the filter through the gap (radio.c:83-100)."""
    g = dict(samprate=192000, L=512, M=513, D=4)
    fs, L = g["samprate"], g["L"]
    plan = wl.channel_plan("cfg1", 1)
    bank = kq.Bank(fs, L, g["M"], g["D"], 1, 2)
    bank.add_channel(bank_cfg(plan[0]))
    ch = ko.Channel(oracle_cfg(plan[0], fs, L, g["M"], g["D"]))
    ing = ko.IqIngest()
    rng = np.random.default_rng(5)
    seq, ts = 100, 50000
    got, want = [], []
    for k in range(8):
        n = 700
        raw = rng.integers(-3000, 3000, 2 * n).astype("<i2")
        if k == 3:
            ts += 5000                                                        # lost: far more than the ring holds
        if k == 6:
            seq += 2                                                          # and two packets lost later on (800 samples)
            ts += 800
        pkt = rtp_packet(seq, ts, 77, raw.tobytes())
        r = ing.packet(pkt)
        zeros, off, count, fmt = r
        if zeros:
            want += ch.zero_fill(zeros)
        want += ch.push_raw(pkt[off:off + 4 * count], count, fmt)
        appended, rounds = 0, 0
        while True:
            before = bank.rtp_counters()
            res = bank.push_rtp(pkt)
            if res is not None:
                appended += res
                break
            after = bank.rtp_counters()
            appended += after["samples"] - before["samples"]
            rounds += 1
            assert rounds < 12
            nb = bank.process()
            assert nb >= 1
            got += [bank.audio(0, b) for b in range(nb)]
        assert rounds >= 3 if k == 3 else rounds <= (1 if k == 6 else 0), (k, rounds)
        c = bank.rtp_counters()
        assert (c["samples"], c["packets"], c["dupes"], c["drops"], c["next_seq"], c["next_timestamp"]) == \
               (ing.samples, ing.rtp.packets, ing.rtp.dupes, ing.rtp.drops, ing.rtp.seq, ing.rtp.timestamp), k
        seq, ts = seq + 1, ts + n
        nb = bank.process()
        got += [bank.audio(0, b) for b in range(nb)]
    assert len(got) == len(want) == ing.samples // L
    for b, (ga, (wa, _ws)) in enumerate(zip(got, want)):
        assert rel_rms(ga, wa) < 1e-5 or np.abs(wa).max() < 1e-9, b
    # a payload that can never fit (a ring of one block against a 2000-sample packet) is an error, and the stream goes on
    small = kq.Bank(fs, L, g["M"], g["D"], 1, 1)
    small.add_channel(bank_cfg(plan[0]))
    with pytest.raises(kq.KqError, match="does not fit"):
        small.push_rtp(rtp_packet(1, 10, 3, np.zeros(4000, "<i2").tobytes()))
    nxt, total = rtp_packet(2, 2010, 3, np.zeros(400, "<i2").tobytes()), 0      # its 2000 samples are a gap for the next one
    for _ in range(8):
        before = small.rtp_counters()["samples"]
        res = small.push_rtp(nxt)
        total += (res if res is not None else small.rtp_counters()["samples"] - before)
        if res is not None:
            break
        assert small.process() >= 1
    assert total == 2000 + 200 and small.rtp_counters()["next_timestamp"] == 2210
    small.close()
    bank.close()
    ch.close()

"""I/Q recording files as `iqrecord` writes and `iqplay` reads them: raw interleaved int16 little-endian
I/Q with the stream parameters in extended attributes `user.samplerate`, `user.channels`, `user.ssrc`,
`user.sampleformat` ("s16le"), `user.frequency`, `user.source_timestamp` (iqrecord.c:263-271, attr.c:22-76;
iqplay.c:88-96 reads them back).  Host-side I/O only: the samples go to the bank as int16 I/Q (KQ_IQ_S16) and are converted on the device.
"""
import os

import numpy as np

_ATTRS = ("samplerate", "channels", "ssrc", "sampleformat", "frequency", "source_timestamp")


def read_attributes(path):
    """-> dict of the attributes present (strings as stored; absent ones are left out, as attrscanf's -1 return)"""
    out = {}
    for name in _ATTRS:
        try:
            out[name] = os.getxattr(path, "user." + name).decode()
        except OSError:
            pass
    return out


def write_recording(path, iq_int16, samprate, frequency=0.0, ssrc=0, source_timestamp=0):
    """Write a recording the way iqrecord does (iqrecord.c:263-271, 302); iq_int16: int16[n, 2].  Attributes are
    skipped silently where the filesystem has no xattr support, as the reference ignores attrprintf's result."""
    np.ascontiguousarray(iq_int16, "<i2").tofile(path)
    for name, value in (("samplerate", "%lu" % samprate), ("channels", "2"), ("ssrc", "%lx" % ssrc),
                        ("sampleformat", "s16le"), ("frequency", "%.3f" % frequency),
                        ("source_timestamp", "%d" % source_timestamp)):
        try:
            os.setxattr(path, "user." + name, value.encode())
        except OSError:
            pass


def open_recording(path):
    """-> (attributes, np.memmap int16[n, 2])"""
    attrs = read_attributes(path)
    fmt = attrs.get("sampleformat", "s16le")
    if fmt != "s16le":
        raise ValueError("I/Q recordings are s16le; %s holds %s" % (path, fmt))
    n = os.path.getsize(path) // 4
    return attrs, np.memmap(path, dtype="<i2", mode="r", shape=(n, 2))


def play_into(bank, path, chunk_blocks=None):
    """Feed a recording through `bank` (already configured for the file's sample rate); yields the number of
    blocks each process call completed.  Trailing samples short of a block stay pending, as in proc_samples."""
    attrs, data = open_recording(path)
    if "samplerate" in attrs and int(attrs["samplerate"]) != bank.samprate:
        raise ValueError("recording is %s Hz, bank runs at %d Hz" % (attrs["samplerate"], bank.samprate))
    step = (chunk_blocks or bank.max_blocks) * bank.L
    for pos in range(0, len(data), step):
        part = np.ascontiguousarray(data[pos:pos + step], np.int16)   # the bank converts int16 I/Q on the device
        bank.push_iq(part)
        yield bank.process()

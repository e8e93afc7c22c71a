"""I/Q recording files (iqrecord.c:263-271, attr.c:22-76): raw s16le I/Q + user.* extended attributes."""
import os

import numpy as np
import pytest

from ka9q_sdr_amd import iqfile


def _xattr_ok(tmp_path):
    probe = tmp_path / "probe"
    probe.write_bytes(b"x")
    try:
        os.setxattr(str(probe), "user.test", b"1")
        return True
    except OSError:
        return False


def test_round_trip_samples_and_attributes(tmp_path):
    rng = np.random.default_rng(0)
    iq = rng.integers(-20000, 20000, (1000, 2)).astype(np.int16)
    path = str(tmp_path / "iqrecord-2m")
    iqfile.write_recording(path, iq, 192000, frequency=147.435e6, ssrc=0xBEEF, source_timestamp=1234567890123)
    attrs, data = iqfile.open_recording(path)
    assert data.shape == (1000, 2) and np.array_equal(np.asarray(data), iq)
    assert open(path, "rb").read(4) == iq[0].astype("<i2").tobytes()      # little endian, I then Q
    if _xattr_ok(tmp_path):
        assert attrs == {"samplerate": "192000", "channels": "2", "ssrc": "beef", "sampleformat": "s16le",
                         "frequency": "147435000.000", "source_timestamp": "1234567890123"}
    else:
        assert attrs == {}                                                # like attrscanf returning -1 everywhere


def test_wrong_sample_format_is_refused(tmp_path):
    if not _xattr_ok(tmp_path):
        pytest.skip("no xattr support on this filesystem")
    path = str(tmp_path / "pcm")
    iqfile.write_recording(path, np.zeros((4, 2), np.int16), 48000)
    os.setxattr(path, "user.sampleformat", b"s16be")                      # what iqrecord writes for PCM streams
    with pytest.raises(ValueError):
        iqfile.open_recording(path)

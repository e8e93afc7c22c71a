"""ka9q_sdr_amd -- MI355X (gfx950) implementation of ka9q-radio's per-channel DSP hot path.

The compute path is libka9q_hip.so (hand-written HIP kernels behind the C ABI of
include/ka9q_hip.h).  This package is the thin Python mirror of that ABI used by the tests and
by bench.py; there is no CPU or PyTorch fallback: without the built library, or without a GPU,
every compute entry point raises.
"""
from .bank import (  # noqa: F401
    Bank,
    BankConfig,
    ChannelConfig,
    ChanStatus,
    FanoutInfo,
    KQ_AM_DEMOD,
    KQ_FM_DEMOD,
    KQ_FWD_AUTO,
    KQ_FWD_FULL,
    KQ_FWD_PRUNED,
    KQ_IQ_CF32,
    KQ_IQ_S8,
    KQ_IQ_S16,
    KQ_LINEAR_DEMOD,
    KqError,
    build_library,
    channel_config,
    device_count,
    HostBuffer,
    library_path,
    load_library,
)
from .decimate import Decimator  # noqa: F401,E402
from .packet import AfskBank, KQ_PCM_F32, KQ_PCM_S16BE  # noqa: F401,E402
from . import iqfile  # noqa: F401,E402

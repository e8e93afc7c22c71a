"""kq_fanout_* (include/ka9q_hip.h): the front-end fan-out a C host uses to shard channels over GPUs -- the
reference's UDP multicast (multicast.c:143-237) replaced by ncclBroadcast on a side stream, two slots.  One GPU is
all a test box has, so the communicator here has one rank; it still goes through librccl (ncclGetUniqueId,
ncclCommInitRank, ncclBroadcast on the side stream), the slot / event protocol is the one N ranks run, and the
batches come out of the bank equal to the oracle's.  The N-rank layout itself is covered on CPU
(tests/test_distributed_gloo.py) and measured by bench.py --gpus N."""
import ctypes as C

import numpy as np
import pytest
import torch

import ka9q_sdr_amd as kq
from common import bank_cfg, rel_rms, run_oracle
from ka9q_sdr_amd import workload as wl

pytestmark = pytest.mark.gpu


def test_fanout_of_one_through_rccl_feeds_a_bank(gpu):
    lib = kq.load_library()
    lib.kq_fanout_create.restype = C.c_void_p
    lib.kq_fanout_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.kq_fanout_post.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    lib.kq_fanout_acquire.restype = C.c_void_p
    lib.kq_fanout_acquire.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    lib.kq_fanout_release.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.kq_fanout_destroy.argtypes = [C.c_void_p]
    ident = C.create_string_buffer(128)
    assert lib.kq_fanout_unique_id(ident) == 0, lib.kq_last_error()
    assert any(ident.raw)

    g = dict(samprate=192000, L=512, M=513, D=4)
    fs, L, M = g["samprate"], g["L"], g["M"]
    plan = wl.channel_plan("cfg1", 2)
    nblocks, nbatches = 3, 4
    iq = wl.make_iq(fs, nbatches * nblocks * L, seed=77)
    want = run_oracle(plan, g, iq, nbatches * nblocks)

    stream = torch.cuda.Stream()
    bank = kq.Bank(fs, L, M, g["D"], len(plan), nblocks, stream=stream.cuda_stream)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    nwin = (M - 1) + nblocks * L
    f = lib.kq_fanout_create(0, 0, 1, 0, ident, nwin)
    assert f, lib.kq_last_error()
    # a slot that has not been posted carries nothing; an oversized post is refused
    assert lib.kq_fanout_post(f, 0, None, nwin, 0) == -1
    hist = np.zeros(M - 1, np.complex64)
    got = [[] for _ in plan]
    windows = []
    for k in range(nbatches):          # window layout of kq_bank_process_resident: M-1 history + the batch's blocks
        batch = iq[k * nblocks * L:(k + 1) * nblocks * L]
        windows.append(np.ascontiguousarray(np.concatenate([hist, batch])))
        hist = batch[-(M - 1):]
    assert lib.kq_fanout_post(f, 0, windows[0].ctypes.data, nwin, 0) == 0     # prime: batch 0 travels
    for k in range(nbatches):
        slot = k & 1
        if k + 1 < nbatches:           # batch k+1 travels while batch k is processed
            assert lib.kq_fanout_post(f, slot ^ 1, windows[k + 1].ctypes.data, nwin, 0) == 0, lib.kq_last_error()
        n = C.c_size_t()
        p = lib.kq_fanout_acquire(f, slot, C.c_void_p(stream.cuda_stream), C.byref(n))
        assert p and n.value == nwin
        assert bank.process_resident(p, nblocks) == nblocks
        assert lib.kq_fanout_release(f, slot, C.c_void_p(stream.cuda_stream)) == 0
        bank.sync()
        for c in range(len(plan)):
            got[c] += [bank.audio(c, b) for b in range(nblocks)]
    for c in range(len(plan)):
        assert rel_rms(np.concatenate(got[c]), np.concatenate(want[c][0])) < 1e-5
    assert lib.kq_fanout_destroy(f) == 0
    bank.close()

"""N > 1 path on CPU: gloo, world sizes 2 and 4 (the latter with uneven shards).  Rank 0 owns the front-end stream and fans it out with
FrontEndFanout (the same class bench.py uses over RCCL); each rank processes its own channel shard.
The per-shard compute here is the CPU oracle (the GPU library needs a device); what is under test is the
sharding + fan-out logic: every rank sees identical I/Q, shards cover the plan exactly once, and a sharded
channel's output equals the unsharded run."""
import os
import sys

import numpy as np
import pytest


def _worker(rank, world, port, out_dir, total=6):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      KQ_TEST_TOTAL=str(total))
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [root, os.path.join(root, "oracle"), here]
    import torch
    import torch.distributed as dist
    import kq_oracle as ko
    from ka9q_sdr_amd import workload as wl
    from ka9q_sdr_amd.shard import FrontEndFanout, shard_range
    from common import oracle_cfg

    dist.init_process_group("gloo", rank=rank, world_size=world)
    geom = dict(samprate=192000, L=512, M=513, D=4)
    total, nblocks, nbatches = int(os.environ.get("KQ_TEST_TOTAL", "6")), 3, 2
    first, count = shard_range(total, world, rank)
    fs, L = geom["samprate"], geom["L"]
    plan = []
    for c in range(first, first + count):
        e = 28 + c
        p = wl._mode_params(wl.emitter_kind(e), e)
        p.update(second_lo=-(wl.emitter_freq(e, fs) + 1.5))
        plan.append(p)
    chans = [ko.Channel(oracle_cfg(p, fs, L, geom["M"], geom["D"])) for p in plan]
    bufs = [torch.zeros(nblocks * L, dtype=torch.complex64) for _ in range(2)]
    fan = FrontEndFanout(bufs, src=0)
    audio = [[] for _ in plan]
    sums = []
    for k in range(nbatches):
        i = k & 1
        if rank == 0:
            bufs[i].copy_(torch.from_numpy(wl.make_iq(fs, nblocks * L, seed=40, start=k * nblocks * L, emitters=range(24, 40))))
        fan.post(i)
        x = fan.acquire(i).numpy()
        sums.append(float(np.abs(x).sum()))
        for ci, ch in enumerate(chans):
            for b in range(nblocks):
                a, _, _, _ = ch.block(x[b * L:(b + 1) * L])
                audio[ci].append(a)
        fan.release(i)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), first=first, count=count, sums=np.array(sums),
             **{"audio%d" % (first + ci): np.concatenate(a) for ci, a in enumerate(audio)})
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_fanout_world2(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["sums"], r1["sums"])                   # both ranks saw the same front-end samples
    assert int(r0["first"]) == 0 and int(r1["first"]) == int(r0["count"]) and int(r0["count"]) + int(r1["count"]) == 6
    # unsharded run of channel 4 (owned by rank 1) on the same stream
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here]
    import kq_oracle as ko
    from ka9q_sdr_amd import workload as wl
    from common import oracle_cfg
    fs, L = 192000, 512
    e = 28 + 4
    p = wl._mode_params(wl.emitter_kind(e), e)
    p.update(second_lo=-(wl.emitter_freq(e, fs) + 1.5))
    x = np.concatenate([wl.make_iq(fs, 3 * L, seed=40, start=k * 3 * L, emitters=range(24, 40)) for k in range(2)])
    auds, _, _ = ko.run_chain(oracle_cfg(p, fs, L, 513, 4), x.reshape(6, L))
    assert np.array_equal(np.concatenate(auds), r1["audio4"])


def test_sharded_fanout_world4_uneven_shards(tmp_path):
    """10 channels over 4 ranks: shards of 3, 3, 2, 2 -- contiguous, disjoint, covering the plan; every rank sees the
    same samples; the C host's kq_shard_range (kq_fanout.cpp) agrees with the Python one."""
    import ctypes as C
    import torch.multiprocessing as mp
    import ka9q_sdr_amd as kq
    from ka9q_sdr_amd.shard import shard_range
    total, world = 10, 4
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path), total), nprocs=world, join=True)
    rs = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    assert [int(r["count"]) for r in rs] == [3, 3, 2, 2]
    nxt = 0
    for r in rs:
        assert int(r["first"]) == nxt
        nxt += int(r["count"])
        assert np.array_equal(r["sums"], rs[0]["sums"])
        assert sorted(k for k in r.files if k.startswith("audio")) == sorted("audio%d" % c for c in
                                                                              range(int(r["first"]), int(r["first"]) + int(r["count"])))
    assert nxt == total
    lib = kq.load_library()
    for tot, w in ((10, 4), (1024, 8), (8192, 8), (5, 8), (0, 3)):
        for rank in range(w):
            f, c = C.c_uint(), C.c_uint()
            assert lib.kq_shard_range(tot, w, rank, C.byref(f), C.byref(c)) == 0
            assert (f.value, c.value) == shard_range(tot, w, rank)
    assert lib.kq_shard_range(4, 2, 2, C.byref(f), C.byref(c)) == -1


def _id_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here)]
    import torch.distributed as dist
    from ka9q_sdr_amd.shard import ID_BYTES, share_unique_id

    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def make_id():                      # stands for kq_fanout_unique_id: must run on the root rank only
        calls.append(rank)
        return bytes((7 * i + 3) & 0xFF for i in range(ID_BYTES))

    ident = share_unique_id(make_id, rank, 0, dist, "cpu")
    with open(os.path.join(out_dir, "id%d.bin" % rank), "wb") as f:
        f.write(ident)
    with open(os.path.join(out_dir, "calls%d.txt" % rank), "w") as f:
        f.write(",".join(map(str, calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_unique_id_hand_off_world3(tmp_path):
    """bench.py --gpus N: rank 0 makes the fan-out's 128-byte identifier (kq_fanout_unique_id) and every rank needs the
    same bytes before the collective kq_fanout_create -- share_unique_id carries them over the process group."""
    import torch.multiprocessing as mp
    from ka9q_sdr_amd.shard import ID_BYTES, share_unique_id
    port = 29500 + ((os.getpid() + 977) % 2000)
    mp.spawn(_id_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    want = bytes((7 * i + 3) & 0xFF for i in range(ID_BYTES))
    for r in range(3):
        assert (tmp_path / ("id%d.bin" % r)).read_bytes() == want
        assert (tmp_path / ("calls%d.txt" % r)).read_text() == ("0" if r == 0 else "")
    # one rank, no process group: the maker is called directly
    assert share_unique_id(lambda: want, 0) == want


def test_c_fanout_wrapper_refuses_without_a_device():
    """CFanout (the ctypes front of kq_fanout_*) fails loudly where there is no GPU: no silent fallback to the twin."""
    import ka9q_sdr_amd as kq
    from ka9q_sdr_amd.shard import CFanout
    lib = kq.load_library()
    if kq.device_count() > 0:
        pytest.skip("a GPU is present: covered by tests/test_gpu_fanout.py")
    with pytest.raises(RuntimeError):
        CFanout(lib, 0, 0, 1, 4096)

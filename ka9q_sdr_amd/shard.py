"""Multi-GPU layout of the channel bank: one process per GPU, channels sharded, front-end I/Q fanned out.

The reference runs one `radio` process per channel and fans the front-end stream out by UDP multicast
(multicast.c:143-237, README.md:470-477).  Here rank r owns a contiguous range of channels and every
batch of front-end samples is broadcast from the ingest rank over torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Channels are independent, so there is no
other collective anywhere on the path.

Two hosts for the same two-slot protocol: `CFanout` drives the product's C ABI (kq_fanout_*, ncclBroadcast on the
library's own side stream -- what a C host links against, and what bench.py measures by default); `FrontEndFanout` is
the torch.distributed twin (any backend: "gloo" in the CPU tests).
"""
import ctypes as C

ID_BYTES = 128     # KQ_FANOUT_ID_BYTES


def share_unique_id(make_id, rank, src=0, dist=None, device=None):
    """The hand-off kq_fanout_unique_id's header comment asks of the host: rank `src` makes the 128-byte identifier
    (make_id() -> bytes), every rank returns it.  It travels as a uint8 tensor over the process group the launcher has
    set up anyway (any backend; `device` = where that backend wants its tensors); without a group there is one rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return bytes(make_id())
    import torch
    t = torch.zeros(ID_BYTES, dtype=torch.uint8, device=device or "cpu")
    if rank == src:
        ident = bytes(make_id())
        if len(ident) != ID_BYTES:
            raise ValueError("identifier must be %d bytes" % ID_BYTES)
        t.copy_(torch.frombuffer(bytearray(ident), dtype=torch.uint8))
    dist.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


class CFanout:
    """kq_fanout_* through ctypes: two device slots owned by the library, ncclBroadcast (RCCL) on its side stream.

    The root fills both slots once with fill(); after that post(i) re-broadcasts slot i in place (the synthetic bench
    sends the same batch every time; a live receiver would hand post() the next batch's pointer)."""

    def __init__(self, lib, device, rank, world, nsamples, ident=None, src=0):
        self.lib, self.rank, self.world, self.src, self.n = lib, rank, world, src, nsamples
        buf = C.create_string_buffer(bytes(ident), ID_BYTES) if ident is not None else None
        self.h = lib.kq_fanout_create(device, rank, world, src, buf, nsamples)
        if not self.h:
            raise RuntimeError("kq_fanout_create: " + (lib.kq_last_error() or b"").decode())
        self.ptr = [None, None]

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(what + ": " + (self.lib.kq_last_error() or b"").decode())

    def fill(self, i, src_ptr, is_device=1):
        """root: copy a batch into slot i and broadcast it; other ranks: join that broadcast"""
        self._chk(self.lib.kq_fanout_post(self.h, i, src_ptr, self.n, is_device), "kq_fanout_post")

    def post(self, i, compute_stream=None):
        if self.ptr[i] is None:
            raise RuntimeError("slot %d was never acquired" % i)
        self._chk(self.lib.kq_fanout_post(self.h, i, self.ptr[i], self.n, 1), "kq_fanout_post")

    def acquire(self, i, compute_stream):
        """compute_stream (a hipStream_t as an integer) waits for slot i's batch; returns the slot's device pointer"""
        n = C.c_size_t()
        p = self.lib.kq_fanout_acquire(self.h, i, C.c_void_p(compute_stream), C.byref(n))
        if not p:
            raise RuntimeError("kq_fanout_acquire: " + (self.lib.kq_last_error() or b"").decode())
        self.ptr[i] = p
        return p

    def release(self, i, compute_stream):
        self._chk(self.lib.kq_fanout_release(self.h, i, C.c_void_p(compute_stream)), "kq_fanout_release")

    def stats(self):
        from .bank import FanoutInfo
        info = FanoutInfo()
        self._chk(self.lib.kq_fanout_stats(self.h, C.byref(info)), "kq_fanout_stats")
        return {k: getattr(info, k) for k, _ in info._fields_}

    def enable_timing(self, on=True):
        """time the consumer stream's waits for a batch (kq_fanout_stats: waits, wait_ms)"""
        self._chk(self.lib.kq_fanout_enable_timing(self.h, 1 if on else 0), "kq_fanout_enable_timing")

    def close(self):
        if self.h:
            self.lib.kq_fanout_destroy(self.h)
            self.h = None


def shard_range(total_channels, world, rank):
    """Contiguous, balanced channel range [first, first+count) of `rank`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(total_channels, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


class FrontEndFanout:
    """Double-buffered broadcast of front-end I/Q batches from `src` to every rank.

    buffers: two tensors of identical shape on this rank's device.  With CUDA tensors the broadcast runs
    on a side stream so that batch k+1 travels while batch k is being processed; events order producer and
    consumer.  With CPU tensors (gloo) everything is synchronous.
    """

    def __init__(self, buffers, src=0, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.bufs = buffers
        # collectives move the real view of complex buffers (same storage): every backend handles float32
        self.wire = [torch.view_as_real(b) if b.is_complex() else b for b in buffers]
        self.src = src
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = buffers[0].is_cuda
        if self.cuda and self.world > 1:
            self.side = torch.cuda.Stream(device=buffers[0].device)
            self.ready = [torch.cuda.Event() for _ in buffers]
            self.freed = [torch.cuda.Event() for _ in buffers]
        else:
            self.side = None

    def post(self, i, compute_stream=None):
        """Start broadcasting buffer i (after its previous consumer, recorded by release(i), is done)."""
        if self.world == 1:
            return
        if self.side is None:
            self.dist.broadcast(self.wire[i], src=self.src, group=self.group)
            return
        with self.torch.cuda.stream(self.side):
            self.side.wait_event(self.freed[i])
            self.dist.broadcast(self.wire[i], src=self.src, group=self.group)
            self.ready[i].record(self.side)

    def acquire(self, i, compute_stream=None):
        """Make the compute stream wait until buffer i has arrived; returns the buffer."""
        if self.side is not None:
            (compute_stream or self.torch.cuda.current_stream()).wait_event(self.ready[i])
        return self.bufs[i]

    def release(self, i, compute_stream=None):
        """Mark buffer i as consumed (recorded on the compute stream)."""
        if self.side is not None:
            self.freed[i].record(compute_stream or self.torch.cuda.current_stream())

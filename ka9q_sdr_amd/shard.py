"""Multi-GPU layout of the channel bank: one process per GPU, channels sharded, front-end I/Q fanned out.

The reference runs one `radio` process per channel and fans the front-end stream out by UDP multicast
(multicast.c:143-237, README.md:470-477).  Here rank r owns a contiguous range of channels and every
batch of front-end samples is broadcast from the ingest rank over torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Channels are independent, so there is no
other collective anywhere on the path.
"""


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

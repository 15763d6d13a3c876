"""Batch-sharded inference across the GPUs of one node.

The decoder path shards only over frames: one process per GPU, the model replicated at start-up, rank r takes
global frames r, r + world, ... (bs-1 streams) or a contiguous B/world slice of a batch (HyperSeg-L: 32 -> 4 per GPU).
The single collective is an RCCL ``all_gather_into_tensor`` of every rank's logits (BASELINE.json's north star; the
reference's counterpart is nn.DataParallel's gather onto device 0, test_fps.py:155-156, available as ``mode='gather'``),
issued asynchronously on a ring of three buffers so that RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests) moves
step i over xGMI while step i+1 is computed.
"""
import torch
import torch.distributed as dist

RING = 3


def shard_frames(n_frames, rank, world):
    """Global frame indices processed by ``rank`` (round robin)."""
    return list(range(rank, n_frames, world))


def shard_batch(batch, rank, world):
    """Contiguous slice [lo, hi) of a batch of ``batch`` frames owned by ``rank`` (batch % world == 0)."""
    if batch % world != 0:
        raise ValueError(f'batch {batch} does not shard over {world} ranks')
    per = batch // world
    return rank * per, (rank + 1) * per


class LogitsGatherer:
    """Asynchronous collection of one tensor per rank and step over a ring of three send/receive buffers.

    ``mode='allgather'`` (default, the north star's collective): ``all_gather_into_tensor`` -- every rank ends up with
    every rank's result.  ``mode='gather'``: every rank sends to ``dst`` only (nn.DataParallel's semantics,
    test_fps.py:155-156); on point-to-point xGMI that is one transfer per peer link and only ``dst`` pays the inbound
    bandwidth.

    ``submit(i, y)`` starts the collective of step i and returns ``(i-2, collected)`` -- the result of the collective
    issued two steps earlier, now complete -- or None during the first two steps.  The returned (world, *shape) tensor
    lives in ring slot (i-2) % 3, which the NEXT submit (step i+1) re-targets: it is valid until then, and on the
    nccl path the wait() inside submit orders the caller's stream after the transfer, so kernels enqueued before the
    next submit may read it safely.  Ranks that own no result (gather mode, rank != dst) get ``collected = None``."""

    def __init__(self, world, shape, dtype, device, mode='allgather', dst=0):
        if mode not in ('gather', 'allgather'):
            raise ValueError(mode)
        self.world, self.mode, self.dst = world, mode, dst
        self.rank = dist.get_rank()
        self.shape = tuple(shape)
        self.bytes_per_step = int(torch.empty((), dtype=dtype).element_size()) * int(torch.Size(shape).numel())
        self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(RING)]
        self.recv = [None] * RING
        if mode == 'allgather' or self.rank == dst:
            self.recv = [torch.empty((world * shape[0],) + tuple(shape[1:]), dtype=dtype, device=device)
                         for _ in range(RING)]
        self.work = [None] * RING
        self.step_of = [None] * RING
        self.completed = 0

    def _finish(self, k):
        if self.work[k] is None:
            return None
        self.work[k].wait()
        self.work[k] = None
        self.completed += 1
        out = self.recv[k].view((self.world,) + self.shape) if self.recv[k] is not None else None
        return self.step_of[k], out

    def submit(self, step, y):
        k = step % RING
        if self.work[k] is not None:                 # a caller that skipped steps: never overwrite a live slot
            self._finish(k)
        done = self._finish((step - 2) % RING) if step >= 2 else None
        self.send[k].copy_(y)
        self.step_of[k] = step
        if self.mode == 'allgather':
            self.work[k] = dist.all_gather_into_tensor(self.recv[k], self.send[k], async_op=True)
        else:
            parts = list(self.recv[k].view((self.world,) + self.shape).unbind(0)) if self.rank == self.dst else None
            self.work[k] = dist.gather(self.send[k], gather_list=parts, dst=self.dst, async_op=True)
        return done

    def drain(self):
        """Completes every outstanding collective; returns their (step, collected) pairs in step order."""
        live = sorted((k for k in range(RING) if self.work[k] is not None), key=lambda q: self.step_of[q])
        return [self._finish(k) for k in live]

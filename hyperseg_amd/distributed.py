"""Batch-sharded inference across the GPUs of one node.

The decoder path shards only over frames: one process per GPU, the model replicated at start-up, rank r takes
global frames r, r + world, ... .  The single collective is the gather of every rank's logits onto rank 0 (the reference's
counterpart: nn.DataParallel's gather onto device 0, test_fps.py:155-156), double-buffered and issued asynchronously so that
RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests) moves frame i over xGMI while frame i+1 is computed.
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """Global frame indices processed by ``rank`` (round robin)."""
    return list(range(rank, n_frames, world))


class LogitsGatherer:
    """Double-buffered asynchronous collection of one tensor per rank and step.

    ``mode='gather'`` (default): every rank sends its result to ``dst`` -- what nn.DataParallel does with the replicas'
    outputs (test_fps.py:155-156).  On xGMI that is one direct point-to-point transfer per peer link (39.8 MB of logits
    per frame and link at HyperSeg-M: ~40 GB/s per link at 1000 frames/s, against ~153 GB/s), and only ``dst`` pays the
    inbound bandwidth.  ``mode='allgather'``: ``all_gather_into_tensor`` -- every rank ends up with every result; at
    8 GPUs each rank would have to absorb 7 x 39.8 MB per millisecond, which is what bounds the step then.
    ``submit`` / ``drain`` return the collected (world, *shape) tensor on ranks that own one, else None."""

    def __init__(self, world, shape, dtype, device, mode='gather', dst=0):
        assert mode in ('gather', 'allgather')
        self.world, self.mode, self.dst = world, mode, dst
        self.rank = dist.get_rank()
        self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(2)]
        self.shape = tuple(shape)
        # one contiguous (world*B, ...) buffer per parity, handed out as a (world, B, ...) view; in gather mode the
        # gather_list entries are its per-rank slices, and only dst allocates it
        self.recv = [None, None]
        if mode == 'allgather' or self.rank == dst:
            self.recv = [torch.empty((world * shape[0],) + tuple(shape[1:]), dtype=dtype, device=device) for _ in range(2)]
        self.work = [None, None]
        self.step_of = [None, None]
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
        """Start collecting ``y`` for ``step``; returns (step, collected) of the transfer that used this buffer pair two
        steps ago (now complete), or None.  The returned tensor is valid until the next submit on the same parity."""
        k = step & 1
        done = self._finish(k)
        if done is not None and done[1] is not None and y.device.type == 'cpu':
            done = (done[0], done[1].clone())
        self.send[k].copy_(y)
        self.step_of[k] = step
        if self.mode == 'allgather':
            self.work[k] = dist.all_gather_into_tensor(self.recv[k], self.send[k], async_op=True)
        else:
            parts = list(self.recv[k].view((self.world,) + self.shape).unbind(0)) if self.rank == self.dst else None
            try:
                self.work[k] = dist.gather(self.send[k], gather_list=parts, dst=self.dst, async_op=True)
            except (RuntimeError, NotImplementedError) as e:      # a backend without gather: same error on every rank
                if self.completed or any(w is not None for w in self.work):
                    raise
                import warnings
                warnings.warn(f'dist.gather unavailable ({e}); falling back to all_gather_into_tensor')
                self.mode = 'allgather'
                self.recv = [torch.empty((self.world * self.shape[0],) + self.shape[1:], dtype=self.send[0].dtype,
                                         device=self.send[0].device) for _ in range(2)]
                self.work[k] = dist.all_gather_into_tensor(self.recv[k], self.send[k], async_op=True)
        return done

    def drain(self):
        out = []
        for k in sorted((0, 1), key=lambda q: (self.step_of[q] is None, self.step_of[q] or 0)):
            d = self._finish(k)
            if d is not None:
                out.append(d)
        return out

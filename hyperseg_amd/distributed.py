"""Batch-sharded inference across the GPUs of one node.

The decoder path shards only over frames: one process per GPU, the model replicated at start-up, rank r takes
global frames r, r + world, ... .  The single collective is an all-gather of every rank's logits, double-buffered and
issued asynchronously so that RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests) moves frame i over xGMI while
frame i+1 is computed.  The reference's counterpart is nn.DataParallel's gather onto device 0 (test_fps.py:155-156).
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """Global frame indices processed by ``rank`` (round robin)."""
    return list(range(rank, n_frames, world))


class LogitsGatherer:
    """Double-buffered asynchronous ``all_gather_into_tensor`` of one tensor per step."""

    def __init__(self, world, shape, dtype, device):
        self.world = world
        self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(2)]
        # concatenation layout (world*B, ...): accepted by both RCCL and gloo; handed out as a (world, B, ...) view
        self.recv = [torch.empty((world * shape[0],) + tuple(shape[1:]), dtype=dtype, device=device) for _ in range(2)]
        self.shape = tuple(shape)
        self.work = [None, None]
        self.step_of = [None, None]
        self.completed = 0

    def _finish(self, k):
        if self.work[k] is None:
            return None
        self.work[k].wait()
        self.work[k] = None
        self.completed += 1
        return self.step_of[k], self.recv[k].view((self.world,) + self.shape)

    def submit(self, step, y):
        """Start gathering ``y`` for ``step``; returns (step, gathered) of the transfer that used this buffer pair two
        steps ago (now complete), or None.  The returned tensor is valid until the next submit on the same parity."""
        k = step & 1
        done = self._finish(k)
        if done is not None:
            done = (done[0], done[1].clone() if y.device.type == 'cpu' else done[1])
        self.send[k].copy_(y)
        self.step_of[k] = step
        self.work[k] = dist.all_gather_into_tensor(self.recv[k], self.send[k], async_op=True)
        return done

    def drain(self):
        out = []
        for k in sorted((0, 1), key=lambda q: (self.step_of[q] is None, self.step_of[q] or 0)):
            d = self._finish(k)
            if d is not None:
                out.append(d)
        return out

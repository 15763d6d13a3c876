"""Batch-sharded inference across the GPUs of one node.

The decoder path shards only over frames: one process per GPU, the model replicated at start-up, rank r takes
global frames r, r + world, ... (bs-1 streams) or a contiguous B/world slice of a batch (HyperSeg-L: 32 -> 4 per GPU).
The single collective is an RCCL ``all_gather_into_tensor`` of every rank's logits (BASELINE.json's north star; the
reference's counterpart is nn.DataParallel's gather onto device 0, test_fps.py:155-156, available as ``mode='gather'``),
issued asynchronously on a ring of three buffers so that RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests) moves
step i over xGMI while step i+1 is computed.  Two ways to get that overlap:
  * :class:`LogitsGatherer` -- the collective on RCCL's own stream, ordered against the compute stream with events
    (``submit`` per step).  Works everywhere; on ROCm 7.2 an event hand-off between two HIP-graph replays costs a bubble
    (measured at world 1: +14 % per step for an all-gather that moves nothing);
  * :class:`InGraphAllGather` -- the collective CAPTURED into the step's HIP graph as a branch parallel to the forward
    (replay i = forward(i) || all_gather(i - 1)): no host call and no cross-stream event between replays.
bench.py times both at N > 1 in a short calibration and keeps the faster one (``--collective auto``).
"""
import torch
import torch.distributed as dist

RING = 3
# xGMI on an MI355X node: every GPU pair has its own link (7 per GPU), ~153 GB/s per link counting both directions --
# ~76.5 GB/s each way.  A collective policy "fits" when the bytes it moves across its busiest link in one direction per step,
# times the step rate, stay under LINK_HEADROOM of that.
XGMI_LINK_GBS_PER_DIRECTION = 76.5
LINK_HEADROOM = 0.8


def link_schedule(policy, world, payload_bytes):
    """Bytes per step that ``policy`` moves for one rank's payload of ``payload_bytes`` on a fully connected xGMI node:
      per_link_bytes     across the BUSIEST link in one direction;
      bytes_in_per_gpu   received by the busiest GPU (all of its links together);
      links_per_gpu      links of a GPU that carry data.
    ``allgather`` / ``ingraph``: priced as ONE ring (RCCL's all-gather forwards every shard around the ring: (world - 1) shards
    cross each ring link; RCCL may stripe the payload over several rings on different links -- the single ring is the conservative
    figure, per_link_bytes_if_striped the optimistic one: every shard over its own direct link).  ``direct``: the all-pairs schedule
    of LogitsGatherer(mode='direct'), one shard per link and direction.  ``gather``: everything onto one rank, one shard per link."""
    n, s = int(world), int(payload_bytes)
    if n <= 1 or policy in (None, 'none'):
        return dict(per_link_bytes=0, per_link_bytes_if_striped=0, bytes_in_per_gpu=0, links_per_gpu=0)
    if policy in ('allgather', 'ingraph'):
        return dict(per_link_bytes=(n - 1) * s, per_link_bytes_if_striped=s, bytes_in_per_gpu=(n - 1) * s, links_per_gpu=2 if n > 2 else 1)
    if policy == 'direct':
        return dict(per_link_bytes=s, per_link_bytes_if_striped=s, bytes_in_per_gpu=(n - 1) * s, links_per_gpu=n - 1)
    if policy == 'gather':
        return dict(per_link_bytes=s, per_link_bytes_if_striped=s, bytes_in_per_gpu=(n - 1) * s, links_per_gpu=n - 1)
    raise ValueError(policy)


def link_gbs_needed(policy, world, payload_bytes, steps_per_s):
    """GB/s the busiest link has to sustain in one direction for ``policy`` to keep up with ``steps_per_s``."""
    return link_schedule(policy, world, payload_bytes)['per_link_bytes'] * float(steps_per_s) / 1e9


def fitting_policy(world, payload_bytes, steps_per_s, link_gbs=XGMI_LINK_GBS_PER_DIRECTION, headroom=LINK_HEADROOM):
    """The first policy of (allgather, direct) whose busiest link stays under ``headroom * link_gbs`` at ``steps_per_s``, and
    whether it actually fits: ('allgather' | 'direct', fits).  HyperSeg-M at 8 ranks (39.8 MB of logits, ~1290 steps/s per GPU):
    the ring needs 7 x 39.8 MB x 1290 = 359 GB/s per link -- no; all-pairs 51 GB/s -- yes.  When not even the all-pairs schedule
    fits the caller can move uint8 masks instead (76x fewer bytes; bench.py --gather masks / auto)."""
    budget = headroom * link_gbs
    for policy in ('allgather', 'direct'):
        if link_gbs_needed(policy, world, payload_bytes, steps_per_s) <= budget:
            return policy, True
    return 'direct', False


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

    ``mode='allgather'`` (default, the north star's collective): RCCL ``all_gather_into_tensor`` -- every rank ends up with
    every rank's result.  On point-to-point xGMI RCCL's ring moves (world - 1) shards through EVERY link per step: at 8 ranks x
    ~1100 frames/s x 39.8 MB that is ~300 GB/s per link against ~153 GB/s.  ``mode='direct'``: the all-pairs
    schedule -- one grouped batch of RCCL point-to-point sends / receives (``batch_isend_irecv``: a single ncclGroup), every
    shard crossing exactly ONE link, the one between its producer and its consumer: 7 x 39.8 MB in, 7 x out per GPU and step,
    each of the 7 links carrying one shard each way.  ``mode='gather'``: every rank sends to ``dst`` only (nn.DataParallel's
    semantics, test_fps.py:155-156).

    Zero-copy: the slot a rank's own result belongs in is part of the receive buffer (``slot(step)``: rows [rank] of ring entry
    step % 3).  A producer that writes there -- bench.py captures one HIP graph per ring entry with the decoder's
    ``output_buffer`` pointing at it -- hands ``submit`` that very tensor and no copy is made (allgather then runs in place,
    NCCL's sendbuff == recvbuff + rank * count form); any other tensor is copied in first, as in rounds 1-2.

    ``submit(i, y)`` starts the collective of step i and returns ``(i-2, collected)`` -- the result of the collective
    issued two steps earlier, now complete -- or None during the first two steps.  The returned (world, *shape) tensor
    lives in ring slot (i-2) % 3 = (i+1) % 3, i.e. THE SLOT STEP i+1 IS PRODUCED INTO.  Lifetime: with payloads copied in, it
    is valid until ``submit(i+1)``; with zero-copy producers (``slot()``) the rank's OWN row is overwritten as soon as the
    producer of step i+1 starts writing -- before ``submit(i+1)`` is called -- so a consumer must have read (or enqueued
    its reads of) the collected tensor BEFORE launching the next forward.  On the nccl path the wait() inside submit
    orders the caller's stream after the transfer, so kernels enqueued right after ``submit`` returns read it safely.
    Ranks that own no result (gather mode, rank != dst) get ``collected = None``.

    ``probe_load``: N = 1 measurement aid only -- that many extra out-of-place all-gathers of the slot per step (at
    world 1 each is one RCCL copy kernel of the payload), so that RCCL kernels actually run next to the forward."""

    def __init__(self, world, shape, dtype, device, mode='allgather', dst=0, probe_load=0):
        if mode not in ('gather', 'allgather', 'direct'):
            raise ValueError(mode)
        self.world, self.mode, self.dst = world, mode, dst
        self.rank = dist.get_rank()
        self.shape = tuple(shape)
        self.bytes_per_step = int(torch.empty((), dtype=dtype).element_size()) * int(torch.Size(shape).numel())
        self.recv = [None] * RING
        if mode in ('allgather', 'direct') or self.rank == dst:
            self.recv = [torch.empty((world * shape[0],) + tuple(shape[1:]), dtype=dtype, device=device)
                         for _ in range(RING)]
        if mode in ('allgather', 'direct'):
            # the rank's own rows of the receive buffer ARE its send buffer
            self.send = [r.view((world,) + self.shape)[self.rank] for r in self.recv]
        else:
            self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(RING)]
        self.work = [None] * RING
        self.step_of = [None] * RING
        self.completed = 0
        self.copies = 0                          # submits that had to copy their payload into the slot
        self.scratch = [torch.empty((world * shape[0],) + tuple(shape[1:]), dtype=dtype, device=device)
                        for _ in range(probe_load)] if mode == 'allgather' else []

    def slot(self, step):
        """Where the result of ``step`` should be produced to be sent without a copy."""
        return self.send[step % RING]

    def _finish(self, k):
        if self.work[k] is None:
            return None
        for w in self.work[k]:
            w.wait()
        self.work[k] = None
        self.completed += 1
        out = self.recv[k].view((self.world,) + self.shape) if self.recv[k] is not None else None
        return self.step_of[k], out

    def submit(self, step, y):
        k = step % RING
        if self.work[k] is not None:                 # a caller that skipped steps: never overwrite a live slot
            self._finish(k)
        done = self._finish((step - 2) % RING) if step >= 2 else None
        if not (y.data_ptr() == self.send[k].data_ptr() and tuple(y.shape) == self.shape):
            self.send[k].copy_(y)
            self.copies += 1
        self.step_of[k] = step
        if self.mode == 'allgather':
            self.work[k] = [dist.all_gather_into_tensor(self.recv[k], self.send[k], async_op=True)]
            self.work[k] += [dist.all_gather_into_tensor(t, self.send[k], async_op=True) for t in self.scratch]
        elif self.mode == 'direct':
            rows = self.recv[k].view((self.world,) + self.shape)
            ops = []
            for d in range(1, self.world):           # peer at distance d: every rank sends "forward" and receives "backward"
                to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
                ops.append(dist.P2POp(dist.isend, self.send[k], to))
                ops.append(dist.P2POp(dist.irecv, rows[frm], frm))
            self.work[k] = dist.batch_isend_irecv(ops) if ops else []
        else:
            parts = list(self.recv[k].view((self.world,) + self.shape).unbind(0)) if self.rank == self.dst else None
            self.work[k] = [dist.gather(self.send[k], gather_list=parts, dst=self.dst, async_op=True)]
        return done

    def drain(self):
        """Completes every outstanding collective; returns their (step, collected) pairs in step order."""
        live = sorted((k for k in range(RING) if self.work[k] is not None), key=lambda q: self.step_of[q])
        return [self._finish(k) for k in live]


class InGraphAllGather:
    """The per-step all-gather as a branch of the step's HIP graph, parallel to the forward.

    One graph per ring slot k: ``fork -> [forward -> logits into slot k] || [all_gather_into_tensor(slot k-1), in place] -> join``.
    Replay i therefore computes step i while RCCL moves step i-1's logits; both are nodes of ONE graph, so there is no host
    call per collective and no event between two replays (what the stream form pays for).  After replay i the collected
    result of step i-1 is in ``collected(i-1)``; ``drain(last)`` gathers the final step's logits eagerly.  The first replay
    gathers a slot nobody produced yet (harmless).  ``capture_forward(out)`` must run the forward with its result written
    to ``out`` (zero copy) and return that tensor.

    Capturing an RCCL collective needs the communicator to be up: one eager collective is issued first.  If the capture
    raises (a stack that cannot capture RCCL), the caller falls back to the stream form."""

    def __init__(self, gatherer, capture_forward, probe_load=0):
        import torch.cuda
        g = self.g = gatherer
        if g.mode != 'allgather':
            raise ValueError("InGraphAllGather needs a LogitsGatherer(mode='allgather')")
        dist.all_gather_into_tensor(g.recv[0], g.send[0])          # communicator + its streams exist before capture
        torch.cuda.synchronize()
        self.scratch = [torch.empty_like(g.recv[0]) for _ in range(probe_load)]
        self.graphs = []
        self.side = torch.cuda.Stream()
        pool = None
        for k in range(RING):
            prev = (k - 1) % RING
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, pool=pool):
                cur = torch.cuda.current_stream()
                self.side.wait_stream(cur)
                with torch.cuda.stream(self.side):
                    dist.all_gather_into_tensor(g.recv[prev], g.send[prev])
                    for t in self.scratch:
                        dist.all_gather_into_tensor(t, g.send[prev])
                yk = capture_forward(g.send[k])
                cur.wait_stream(self.side)
            if yk.data_ptr() != g.send[k].data_ptr():
                raise RuntimeError('capture_forward did not produce its result in the slot it was given')
            pool = pool or gk.pool()
            self.graphs.append((gk, yk))
        self.bytes_per_step = g.bytes_per_step
        self.completed = 0

    def step(self, i):
        gk, yk = self.graphs[i % RING]
        gk.replay()
        self.completed += 1
        return yk

    def collected(self, step):
        """(world, *shape) result of ``step``; valid once the replay of step+1 (or ``drain``) has completed."""
        return self.g.recv[step % RING].view((self.g.world,) + self.g.shape)

    def drain(self, last_step):
        if last_step is not None:
            dist.all_gather_into_tensor(self.g.recv[last_step % RING], self.g.send[last_step % RING])

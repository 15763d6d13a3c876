"""The timed step loop (barrier + synchronize brackets, MAX over ranks), one-region replays, two frames in flight, per-leg wall seconds.

Part of bench.py's measurement harness (round 6: bench.py was one 1 100-line file running ten legs; the legs live here, bench.py is the
driver entry).  Nothing in this package imports oracle/: the CPU-baseline leg, the only one that may, stays in bench.py."""
import time

import torch
import torch.distributed as dist


class StepLoop:
    """One rank's step / drain / fence triple.  ``forward()`` produces the rank's output tensor (a graph replay returns
    the captured static output); ``comm`` is a hyperseg_amd.distributed.LogitsGatherer or None."""

    def __init__(self, forward, comm=None, to_payload=None, world=1, device=None, forward_takes_step=False, on_drain=None):
        self.forward, self.comm, self.world = forward, comm, world
        self.on_drain, self.last_step = on_drain, None     # on_drain(last step): a collective carried by the step's own graph
        self.forward_takes_step = forward_takes_step      # forward(i): one HIP graph per ring slot (zero-copy collective)
        self.to_payload = to_payload or (lambda y: y)
        self.device = device
        self.last = None              # the most recent collected (step, tensor) pair, for checks outside the timing
        self.cuda = device is not None and device.type == 'cuda'

    def step(self, i):
        y = self.forward(i) if self.forward_takes_step else self.forward()
        self.last_step = i
        if self.comm is not None:
            done = self.comm.submit(i, self.to_payload(y))
            if done is not None:
                self.last = done
        return y

    def drain(self):
        if self.comm is not None:
            for done in self.comm.drain():
                self.last = done
        if self.on_drain is not None and self.last_step is not None:
            self.on_drain(self.last_step)
            self.last_step = None

    def fence(self):
        if self.cuda:
            torch.cuda.synchronize(self.device)
        if self.world > 1:
            dist.barrier()
            if self.cuda:
                torch.cuda.synchronize(self.device)


def run_timed(loop, steps, warmup, repeats=1):
    """W untimed steps, then ``repeats`` regions of exactly ``steps`` steps; returns the per-region wall time, MAX over
    ranks.  Step indices keep increasing across regions (the gatherer's ring is indexed by them)."""
    i = 0
    for _ in range(warmup):
        loop.step(i)
        i += 1
    loop.drain()
    times = []
    for _ in range(repeats):
        loop.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loop.step(i)
            i += 1
        loop.drain()
        loop.fence()
        elapsed = time.perf_counter() - t0
        if loop.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=loop.device if loop.cuda else None)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        times.append(elapsed)
    return times


def time_replayed(forward, x, steps, warmup, batch):
    """One timed region of a fresh HIP graph of ``forward(x)`` (side numbers only): (frames/s, ms per step, output)."""
    for _ in range(3):
        y = forward(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = forward(x)
    for _ in range(max(1, warmup)):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return round(steps * batch / el, 2), round(1e3 * el / steps, 4), y, g


def two_in_flight(forward, x, y_ref, steps, warmup, batch):
    """A serving-style side number, never ``value``: two independent requests of the benched batch in flight.  Each is a
    HIP graph of the same forward, captured and replayed on ITS OWN stream (own capture stream => own library workspaces,
    own graph memory pool; no fork/join inside a graph), launched alternately.  The bs-1 frame is a chain of ~200
    latency-bound launches that each fill a fraction of the 256 CUs, so a second chain COULD overlap almost for free.
    Measured (round 2, ROCm 7.2): it does not -- 1026 vs 1018 frames/s at HyperSeg-M, graph replays issued from two streams
    execute back to back -- so the number documents that there is nothing to gain this way.  Outputs are compared with the
    single-stream run."""
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    xs = [x, x.clone()]
    graphs, outs = [], []
    for s, xi in zip(streams, xs):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            forward(xi)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            yi = forward(xi)
        graphs.append(g)
        outs.append(yi)

    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i & 1]):
                graphs[i & 1].replay()

    run(2 * max(1, warmup))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    diff = max(float((o.float() - y_ref.float()).abs().max()) for o in outs)
    return {'value': round(steps * batch / el, 2), 'unit': 'frames/s', 'ms_per_step': round(1e3 * el / steps, 4),
            'max_abs_diff_vs_benched': diff,
            'note': 'NOT the headline: 2 requests in flight on 2 streams (one HIP graph each); value above = 1 in flight'}


class Legs:
    """Wall seconds per leg of a run (`legs_s` on the line: where a default run's minutes go -- the timed regions are milliseconds)."""

    def __init__(self):
        self.t, self.out = time.perf_counter(), {}

    def mark(self, name):
        now = time.perf_counter()
        self.out[name] = round(self.out.get(name, 0.0) + now - self.t, 1)
        self.t = now

"""N > 1 path of bench.py's batch-sharded inference harness, world_size 2 on CPU with the gloo backend: the asynchronous
all_gather_into_tensor (default, the north star's collective; or gather onto rank 0) of per-rank results (the logits of
the rank's frames) on a ring of three buffers, overlapped with the "next frame", must deliver every rank's frames in rank
order, and a returned buffer must stay valid until the next submit.  The GPU run uses the same code with backend "nccl"
(= RCCL over xGMI)."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # spawned workers re-import this file

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hyperseg_amd.distributed import LogitsGatherer, shard_frames


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, steps, q, mode='allgather', zero_copy=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        shape = (1, 3, 4, 5)
        g = LogitsGatherer(world, shape, torch.float32, torch.device('cpu'), mode=mode)
        owner = mode in ('allgather', 'direct') or rank == 0
        ok = True
        frames = shard_frames(steps * world, rank, world)
        assert frames == list(range(rank, steps * world, world))
        seen = []

        def check(item):
            nonlocal ok
            step, out = item
            seen.append(step)
            want = torch.tensor([float(step * world + r) for r in range(world)]).view(world, 1, 1, 1, 1)
            ok &= (out is not None) == owner
            if owner:
                ok &= bool(torch.equal(out, want.expand(world, *shape)))
        for i, f in enumerate(frames):
            if zero_copy:                                    # the producer writes straight into the ring slot: no copy in submit
                y = g.slot(i)
                y.fill_(float(f))
            else:
                y = torch.full(shape, float(f))              # stands for the logits of global frame f
            prev = g.submit(i, y)                            # starts the collective of step i, returns step i-2's result
            ok &= (prev is None) == (i < 2)
            if prev is not None:
                # the returned buffer must stay intact until the NEXT submit (it used to be re-targeted by
                # the very submit that returned it)
                held = prev[1].clone() if prev[1] is not None else None
                check(prev)
                if held is not None:
                    ok &= bool(torch.equal(prev[1], held))
        for item in g.drain():
            check(item)
        ok &= seen == list(range(steps))
        ok &= g.copies == (0 if zero_copy else steps)
        q.put((rank, ok, g.completed))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode,world,zero_copy', [('allgather', 2, False), ('gather', 2, False), ('direct', 2, False), ('direct', 3, True),
                                                  ('allgather', 2, True)])
@pytest.mark.parametrize('steps', [1, 5])
def test_batch_sharded_gather_gloo(steps, mode, world, zero_copy):
    """allgather / gather-to-0 / direct all-pairs (grouped point-to-point sends and receives: every shard crosses one link), with
    the payload copied into the ring slot or produced there (zero copy): ordering, ownership and buffer lifetime."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, q, mode, zero_copy)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True, steps) for r in range(world)]


def _bench_worker(rank, world, port, q, mode):
    """bench.py's own step / drain / fence loop (StepLoop + run_timed), driven with a CPU 'forward' over gloo."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import importlib.util
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        shape = (2, 3, 4, 4)                                   # (frames per rank, classes, h, w)
        calls = [0]

        def forward():                                         # "logits" that encode (rank, call index)
            calls[0] += 1
            return torch.full(shape, float(100 * rank + calls[0]))
        comm = LogitsGatherer(world, shape, torch.float32, torch.device('cpu'), mode=mode)
        loop = bench.StepLoop(forward, comm, None, world, torch.device('cpu'))
        steps, warmup, repeats = 4, 2, 3
        times = bench.run_timed(loop, steps, warmup, repeats)
        total = warmup + steps * repeats
        ok = len(times) == repeats and all(t > 0 for t in times) and calls[0] == total and comm.completed == total
        owner = mode in ('allgather', 'direct') or rank == 0
        step, out = loop.last                                  # the last collected step must be the last one issued
        ok &= step == total - 1 and (out is not None) == owner
        if owner:
            want = torch.tensor([float(100 * r + total) for r in range(world)]).view(world, 1, 1, 1, 1)
            ok &= bool(torch.equal(out, want.expand(world, *shape)))
        # every rank reports the same (max-over-ranks) region times
        t = torch.tensor(times, dtype=torch.float64)
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        ok &= all(bool(torch.equal(ts[0], u)) for u in ts)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['allgather', 'gather', 'direct'])
def test_bench_step_loop_gloo(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_shard_batch():
    from hyperseg_amd.distributed import shard_batch
    assert [shard_batch(32, r, 8) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]
    assert shard_batch(32, 0, 1) == (0, 32)
    with pytest.raises(ValueError):
        shard_batch(32, 0, 5)


@pytest.mark.gpu
def test_gatherer_on_the_rccl_path_single_gpu():
    """The nccl (= RCCL) code path of LogitsGatherer on the one GPU a test box has (world size 1): asynchronous
    all_gather_into_tensor / gather on RCCL's stream, the returned tensor READ by kernels enqueued after submit (ADVICE r1:
    nothing exercised the CUDA path), ring slots re-targeted only after their consumer was served."""
    if not torch.cuda.is_available():
        pytest.fail('needs the MI355X')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        for mode in ('allgather', 'gather', 'direct'):
            shape = (2, 19, 32, 64)
            g = LogitsGatherer(1, shape, torch.float32, dev, mode=mode)
            sums = []
            for i in range(7):
                if mode != 'gather' and i % 2:               # odd steps: produced in the ring slot itself (zero copy)
                    y = g.slot(i)
                    y.fill_(float(i))
                    prev = g.submit(i, y)
                else:
                    y = torch.full(shape, float(i), device=dev)
                    y.mul_(1.0)                               # "compute" of step i on the caller's stream
                    prev = g.submit(i, y)
                    y.fill_(-1.0)                             # the graph's static output is overwritten by the next replay
                if prev is not None:
                    step, out = prev
                    assert tuple(out.shape) == (1,) + shape
                    sums.append((step, out.sum()))           # a kernel on the caller's stream reads the collected tensor
            for step, out in g.drain():
                sums.append((step, out.sum()))
            torch.cuda.synchronize()
            n = float(torch.Size(shape).numel())
            assert [s for s, _ in sums] == list(range(7))
            assert [float(v) for _, v in sums] == [n * i for i in range(7)]
            assert g.completed == 7 and g.copies == (7 if mode == 'gather' else 4)
    finally:
        dist.destroy_process_group()

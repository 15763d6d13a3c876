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


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_workload_plan_and_device_selection():
    """What the driver's N = 1, 2, 4, 8 launches make every rank do: --model l shards BASELINE config 4's bs-32 batch into
    32 / N contiguous frames per rank (strong scaling, the shards tile [0, 32)), m / s / sc keep bs 1 per rank (weak);
    LOCAL_RANK r -> cuda:r, refusing two ranks per GPU and a GPU-less box."""
    bench = _load_bench()
    for world in (1, 2, 4, 8):
        plans = [bench.plan_workload('l', r, world) for r in range(world)]
        assert all(p['batch'] == 32 // world and p['global_batch'] == 32 and p['scaling'] == 'strong' for p in plans)
        assert [p['frames'] for p in plans] == [(r * 32 // world, (r + 1) * 32 // world) for r in range(world)]
        assert (plans[0]['h'], plans[0]['w'], plans[0]['spec']['num_classes']) == (512, 512, 21)
        for key, size in (('m', (512, 1024)), ('s', (768, 1536)), ('sc', (576, 768))):
            p = bench.plan_workload(key, world - 1, world)
            assert (p['batch'], p['global_batch'], p['scaling'], (p['h'], p['w'])) == (1, world, 'weak', size)
    with pytest.raises(ValueError):
        bench.plan_workload('l', 0, 3)
    assert [bench.select_device(r, visible=8) for r in range(8)] == [torch.device('cuda', r) for r in range(8)]
    lc = bench.plan_workload('lc', 3, 8)                        # CamVid HyperSeg-L (round 6): bs 1 per GPU like m / s / sc
    assert (lc['cfg'], lc['h'], lc['w'], lc['batch'], lc['global_batch'], lc['scaling']) == ('hyperseg-l-camvid', 768, 1024, 1, 8, 'weak')
    for bad in (dict(local_rank=8, visible=8), dict(local_rank=1, visible=1), dict(local_rank=0, visible=0)):
        with pytest.raises(SystemExit):
            bench.select_device(bad['local_rank'], visible=bad['visible'])
    assert bench.select_device(5, stub=True) == torch.device('cpu')


@pytest.mark.parametrize('model,extra', [('m', []), ('l', []), ('m', ['--collective', 'auto']), ('m', ['--collective', 'direct']), ('m', ['--gather', 'masks']),
                                         ('m', ['--link-gbs', '1e-9']), ('m', ['--link-gbs', '1e-9', '--gather', 'auto']),
                                         ('m', ['--collective', 'auto', '--gather', 'auto'])])
def test_bench_main_under_torch_distributed_run(model, extra):
    """bench.py's main() ITSELF, launched exactly as the driver launches it (python -m torch.distributed.run --nproc-per-node 2
    ... bench.py --gpus 2 --steps K --warmup W), with the model stubbed out (HS_BENCH_STUB=1: gloo, CPU): rank / world from
    the environment, the shard plan, the policy calibration of --collective auto, the timed regions with their barriers, the
    all-gather of the per-rank rates, and ONE JSON line on stdout (RCCL / gloo chatter goes to stderr) carrying the
    contract's keys."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HS_BENCH_STUB='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '3',
           '--repeats', '2', '--calib-steps', '4', '--model', model] + extra
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config'):
        assert k in d, k
    assert d['n_gpus'] == 2 and d['steps'] == 6 and d['warmup'] == 3 and d['unit'] == 'frames/s' and d['higher_is_better'] is True
    assert d['scaling'] == ('strong' if model == 'l' else 'weak') and len(d['per_rank_frames_per_s']) == 2
    per_step = 32 if model == 'l' else 2                       # global frames per step
    assert abs(d['value'] - per_step / (d['ms_per_step'] * 1e-3)) < 0.01 * d['value']
    assert abs(sum(d['per_rank_frames_per_s']) - d['value']) < 0.01 * d['value']
    c = d['collective']
    starved = '--link-gbs' in extra                            # a link budget nothing fits: the all-pairs schedule, and masks under --gather auto
    auto = ['--collective', 'auto'] == extra[:2]
    want_policy = 'direct' if 'direct' in extra or starved else 'allgather'      # auto on CPU: the in-graph form needs HIP graphs
    want_payload = 'masks' if 'masks' in extra or (starved and 'auto' in extra[2:]) else 'logits'
    if auto:                                                   # whichever calibrated fastest among the eligible candidates
        assert c['policy'] in ('allgather', 'direct') and c['requested'] == 'auto'
        cal = c['calibration_ms_per_step']
        assert cal['allgather'] > 0 and cal['direct'] > 0 and cal['allgather:masks'] > 0 and 'ingraph' not in cal
        if 'auto' not in extra[2:]:
            assert c['payload'] == 'logits'                    # the masks candidate is timed, not eligible, unless --gather auto
        want_policy, want_payload = c['policy'], c['payload']
    assert c['policy'] == want_policy and c['payload'] == want_payload
    assert c['completed'] >= 6 * 2 + 3
    if not extra or starved:
        assert c['requested'] == 'fit' and c['calibration_ms_per_step'] is None
        assert c['fit']['steps_per_s_without_collective'] > 0 and c['fit']['fits'] is (not starved)
    classes, (h, w) = (21, (512, 512)) if model == 'l' else (19, (512, 1024))
    frames = 16 if model == 'l' else 1
    elems = frames * (h // 16) * (w // 16) * (1 if want_payload == 'masks' else classes * 4)
    assert c['bytes_sent_per_rank_per_step'] == elems
    # the per-link accounting on the line IS the schedule of hyperseg_amd.distributed.link_schedule for the policy that ran
    from hyperseg_amd.distributed import link_gbs_needed, link_schedule
    sched = link_schedule(want_policy, 2, elems)
    assert {k: c[k] for k in sched} == sched
    assert c['per_link_bytes'] == elems                        # world 2: ring and all-pairs both move one shard over the one link
    rate = 1e3 / d['ms_per_step']
    assert abs(c['link_gbs_needed_at_this_rate'] - link_gbs_needed(want_policy, 2, elems, rate)) <= 0.02 + 0.02 * c['link_gbs_needed_at_this_rate']


def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT torch.distributed.run around it (the shape of the driver's N = 1 command; VERDICT r5 #2;
    the reference's counterpart is one command for N GPUs as well: nn.DataParallel, test_fps.py:155-156): bench.py re-launches itself
    as 2 ranks and the one JSON line on stdout says n_gpus 2.  Stubbed model (gloo, CPU)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HS_BENCH_STUB='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', '--repeats', '2']
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['warmup'] == 2 and len(d['per_rank_frames_per_s']) == 2
    assert d['scaling'] == 'weak' and abs(d['value'] - 2 / (d['ms_per_step'] * 1e-3)) < 0.01 * d['value']
    # a failing rank is a failing command: an unknown flag reaches the ranks' own argument parser
    bad = subprocess.run(cmd + ['--model', 'nope'], env=env, cwd=root, capture_output=True, timeout=240)
    assert bad.returncode != 0


def test_link_schedule_and_fitting_policy():
    """The per-link arithmetic behind bench.py's N > 1 default: at 8 x HyperSeg-M the RCCL ring all-gather of fp32 logits
    asks ~360 GB/s of one xGMI link and direction (76.5 available), the all-pairs schedule 51 -- the default must be the latter -- while
    HyperSeg-L's strong-scaled batch and any masks payload fit the ring."""
    from hyperseg_amd.distributed import LINK_HEADROOM, XGMI_LINK_GBS_PER_DIRECTION, fitting_policy, link_gbs_needed, link_schedule
    m_logits = 19 * 512 * 1024 * 4
    assert link_schedule('allgather', 8, m_logits) == dict(per_link_bytes=7 * m_logits, per_link_bytes_if_striped=m_logits,
                                                           bytes_in_per_gpu=7 * m_logits, links_per_gpu=2)
    assert link_schedule('direct', 8, m_logits) == dict(per_link_bytes=m_logits, per_link_bytes_if_striped=m_logits,
                                                        bytes_in_per_gpu=7 * m_logits, links_per_gpu=7)
    assert link_schedule('gather', 4, 10)['per_link_bytes'] == 10 and link_schedule('none', 8, 10)['per_link_bytes'] == 0
    assert link_schedule('allgather', 1, 10)['per_link_bytes'] == 0
    assert abs(link_gbs_needed('allgather', 8, m_logits, 1290.0) - 359.8) < 0.5 and abs(link_gbs_needed('direct', 8, m_logits, 1290.0) - 51.4) < 0.1
    assert fitting_policy(8, m_logits, 1290.0) == ('direct', True)
    assert fitting_policy(2, m_logits, 1290.0) == ('allgather', True)          # one shard over the one link either way: 51 GB/s
    assert fitting_policy(8, m_logits // 76, 1290.0) == ('allgather', True)    # uint8 masks
    assert fitting_policy(8, m_logits, 1290.0, link_gbs=10.0) == ('direct', False)
    s_logits = 19 * 768 * 1536 * 4                                            # HyperSeg-S at ~810 steps/s: all-pairs needs 72.6 > 0.8 x 76.5
    assert fitting_policy(8, s_logits, 810.0) == ('direct', False) and LINK_HEADROOM * XGMI_LINK_GBS_PER_DIRECTION < 72.6
    with pytest.raises(ValueError):
        link_schedule('ring', 8, 1)


def _zero_copy_lifetime_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        shape = (1, 2, 3)
        g = LogitsGatherer(world, shape, torch.float32, torch.device('cpu'), mode='allgather')
        ok = True
        for i in range(6):
            g.slot(i).fill_(float(10 * i + rank))
            done = g.submit(i, g.slot(i))
            if done is None:
                continue
            step, out = done
            want = torch.tensor([float(10 * step + r) for r in range(world)]).view(world, 1, 1, 1).expand(world, *shape)
            ok &= bool(torch.equal(out, want))                                     # read BEFORE the next forward: intact
            # the producer of step i+1 now writes its slot -- which is this rank's own row of the tensor just returned
            g.slot(i + 1).fill_(-1.0)
            ok &= bool((out[rank] == -1.0).all())                                  # own row: gone (documented lifetime)
            others = [r for r in range(world) if r != rank]
            ok &= bool(torch.equal(out[others], want[others]))                     # the peers' rows stay until submit(i+1)
        g.drain()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_zero_copy_slot_lifetime_gloo():
    """with zero-copy slots the tensor ``submit(i)`` returns (step i-2) shares its own-rank row with the slot
    step i+1 is produced into.  The documented contract -- consume it before launching the next forward -- is what this
    pins: intact right after submit, own row replaced once the next producer has written, peers' rows untouched."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_zero_copy_lifetime_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


@pytest.mark.gpu
def test_in_graph_all_gather_single_gpu():
    """InGraphAllGather on the RCCL path (world 1): the all-gather captured into the step's HIP graph as a branch parallel
    to the forward.  Replay i must leave step i's result in its slot and step i-1's collected; drain() collects the last."""
    if not torch.cuda.is_available():
        pytest.fail('needs the MI355X')
    from hyperseg_amd.distributed import InGraphAllGather
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        shape = (1, 19, 64, 128)
        g = LogitsGatherer(1, shape, torch.float32, dev, mode='allgather')
        counter = torch.zeros((), device=dev)

        def capture_forward(out):                       # "forward": a counter kernel + the result written into the slot
            counter.add_(1.0)
            out.copy_(counter.expand(shape))
            return out
        ing = InGraphAllGather(g, capture_forward, probe_load=2)
        counter.zero_()
        for i in range(7):
            y = ing.step(i)
            torch.cuda.synchronize()
            assert float(y.mean()) == float(i + 1)
            if i >= 1:
                assert float(ing.collected(i - 1).mean()) == float(i)
                assert all(float(t.mean()) == float(i) for t in ing.scratch)       # the probe's out-of-place copies ran too
        ing.drain(6)
        torch.cuda.synchronize()
        assert float(ing.collected(6).mean()) == 7.0 and ing.completed == 7
    finally:
        dist.destroy_process_group()


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

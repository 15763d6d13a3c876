"""Launch plumbing: what a rank computes, which device it owns, self-launch of N ranks under torch.distributed.run, the CPU / gloo stub model.

Part of bench.py's measurement harness (round 6: bench.py was one 1 100-line file running ten legs; the legs live here, bench.py is the
driver entry).  Nothing in this package imports oracle/: the CPU-baseline leg, the only one that may, stays in bench.py."""
import os
import sys

import torch

from .constants import MODELS

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # the repository root: bench.py lives there


def plan_workload(model_key, rank, world):
    """What ``rank`` of ``world`` computes per step: m / s / sc keep bs 1 per GPU (weak scaling), l shards BASELINE config 4's
    bs-32 batch into 32 / world contiguous frames per GPU (strong scaling; world must divide 32)."""
    from hyperseg_amd import configs
    from hyperseg_amd.distributed import shard_batch
    cfg = MODELS[model_key]
    spec = configs.MODELS[cfg]
    h, w = spec['size']
    if model_key == 'l':
        lo, hi = shard_batch(spec['batch'], rank, world)
        return dict(cfg=cfg, spec=spec, h=h, w=w, batch=hi - lo, global_batch=spec['batch'], scaling='strong', frames=(lo, hi))
    return dict(cfg=cfg, spec=spec, h=h, w=w, batch=spec['batch'], global_batch=spec['batch'] * world, scaling='weak', frames=None)


def select_device(local_rank, stub=False, visible=None):
    """LOCAL_RANK -> this rank's device: one process per GPU, rank r of the node on cuda:r (torch.distributed.run exports
    LOCAL_RANK).  Refuses to run two ranks on one GPU or without a GPU; ``stub``: the CPU / gloo plumbing test."""
    if stub:
        return torch.device('cpu')
    n = torch.cuda.device_count() if visible is None else visible
    if n < 1:
        raise SystemExit('bench.py needs an MI355X (no GPU visible)')
    if not 0 <= local_rank < n:
        raise SystemExit(f'LOCAL_RANK={local_rank} but {n} GPU(s) visible: launch one rank per GPU (--nproc-per-node <= {n})')
    return torch.device('cuda', local_rank)


def launch_ranks(n, argv):
    """Re-runs this file as ``n`` ranks of one node under torch.distributed.run (127.0.0.1 rendezvous on a free port) and returns
    the launcher's exit code; the children inherit stdout, so rank 0's one JSON line is the only thing printed there."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(REPO, 'bench.py'), *argv]
    sys.stdout.flush()
    rc = subprocess.run(cmd, env=env).returncode
    if rc != 0:
        sys.exit(rc)
    return rc


class StubModel:
    """HS_BENCH_STUB=1 (CPU / gloo plumbing test of this file's main(), tests/test_distributed.py): stands in for the model;
    ``forward`` writes a rank- and step-dependent pattern of the logits' shape at 1/16 of the resolution."""

    def __init__(self, plan, rank):
        self.shape = (plan['batch'], plan['spec']['num_classes'], plan['h'] // 16, plan['w'] // 16)
        self.rank, self.calls = rank, 0

    def __call__(self, x):
        self.calls += 1
        return torch.full(self.shape, float(self.rank * 1000 + self.calls % 7), dtype=torch.float32)

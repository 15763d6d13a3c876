#!/usr/bin/env python
"""bench.py -- frames/sec of a BASELINE HyperSeg configuration on MI355X (default: HyperSeg-M, EfficientNet-B1, 1024x512, bs=1).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--model m|s|sc|l]

A "step" is one forward of the whole model on one resident synthetic batch per rank: encoder + context head
(hyperseg_amd.utils.inference.prepare_for_inference; --stock-encoder: plain PyTorch-ROCm), then the HIP decoder (hot
path), captured once in a HIP graph and replayed.  N > 1 is batch-sharded inference, one process per GPU:
  --model m / s / sc   bs 1 per GPU (weak scaling: BASELINE configs 2, 3);
  --model l            the bs-32 batch of BASELINE config 4 sharded 32/N frames per GPU (strong scaling);
the single collective is the north star's all-gather of every rank's logits over RCCL / xGMI, asynchronous over a ring of three
buffers whose slots the decoder's last kernel writes directly (zero copy: one HIP graph per slot), so it overlaps the next step:
  --collective allgather (default at N > 1)  RCCL all_gather_into_tensor, in place, on RCCL's stream;
  --collective ingraph | auto             the same collective captured into the step's HIP graph | both, calibrated, the faster one kept;
  --collective direct                     grouped RCCL point-to-point sends / receives, all pairs: one shard per link and direction;
  --collective gather                     onto rank 0 only (nn.DataParallel's semantics);  --gather masks: uint8 argmax masks instead.
Given explicitly at N = 1 the collective runs on a one-rank group and `collective.overhead_pct` reports its own per-step cost.
Rank 0 writes ONE JSON line to stdout (everything else that writes to fd 1 -- RCCL's banner -- is sent to stderr).

W warm-up steps, then `--repeats` (default 5) timed regions of EXACTLY K steps each, every region bracketed by a barrier
+ torch.cuda.synchronize() on both sides, MAX over ranks per region; `value` uses the MEDIAN region (all are listed).

Extra objects on the line:
  roofline      the dominant decoder kernel against the roof that binds it (the f16-split level 4: HBM; `f32_mfma_frac` beside it for
                comparison with rounds 1-2): algorithmic FLOPs or bytes per launch / its average duration measured with
                HIP events on the launch stream over instrumented eager passes right after the timed regions (a graph
                replay cannot host events).  `traffic` is null unless --traffic-dir names rocprofv3 --pmc FETCH_SIZE /
                WRITE_SIZE passes of THIS command made in the same session (tools/gpu_full_visit.sh does that).
  parity        the replayed output of the benched configuration vs the eager STOCK-encoder model on the same batch
                (outside the timed regions): max tensor-relative error, argmax flips where the stock margin > 1e-4.
  exact_f32     the same step (one timed region) with the fused inverted-residual levels on the exact-f32 matrix cores
                (hs_ir_math = f32); the headline `value` uses config.ir_math (auto: f16 split products at f32-class
                accuracy on the level-4 block).  N = 1 only.
  two_frames_in_flight  serving-style side number (never `value`): two requests of the benched batch in flight, one HIP
                graph each on its own stream.  N = 1 only.
  fps_reference_protocol  hyperseg/test_fps.py's protocol (per iteration sync -> H2D of a pinned batch -> eager forward ->
                sync; hyperseg_amd/fps.py) on the same model, N = 1 only.
  cpu_baseline  the CPU oracle ("port": stock encoder on CPU + oracle/cpu_port.py decoder) timed on the host cores of
                this box on a bounded sample of the same workload (N = 1, rank 0, --model m only).
  other_configs the other BASELINE configurations as side objects of the default run (N = 1, --model m): `s` = HyperSeg-S 1536x768
                (config 3: whole-model frames/s, decoder launch table, its dominant launch's roofline), `train_sc` = the config-5
                training step of the CamVid-S decoder replayed as one HIP graph, fp32 and bf16 (ms per step, step-level roofline).
"""
import argparse
import json
import os
import statistics
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL peer-to-peer needs it on this driver

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from hyperseg_amd.benchlib.constants import F16_PEAK_TFLOPS, FP32_PEAK_TFLOPS, HBM_PEAK_GBS, LABELS, MODELS  # noqa: E402,F401
from hyperseg_amd.benchlib.decoder_probe import (EVENT_REPS, decoder_launches_text, decoder_levels, instrumented_decoder,  # noqa: E402,F401
                                                 pmc_traffic, roofline_of, self_traffic_passes)
from hyperseg_amd.benchlib.launch import StubModel, launch_ranks, plan_workload, select_device  # noqa: E402,F401
from hyperseg_amd.benchlib.side_configs import side_model, side_train_step  # noqa: E402,F401
from hyperseg_amd.benchlib.timing import Legs, StepLoop, run_timed, time_replayed, two_in_flight  # noqa: E402,F401


# --------------------------------------------------------------------------------------------- the CPU baseline (the one leg that may import oracle/)
def cpu_baseline(model_cpu, size, budget_s=12.0):
    """CPU 'port' baseline on this box's host cores: stock encoder + context head on CPU, then the reference's
    ATen op sequence for the decoder (oracle/cpu_port.py, pinned to the oracle).  The thread count is chosen by a
    one-frame calibration over {8, 16, 32, 64, all} (more threads than that only slows these small ops down) and
    reported as `cores`; bounded sample of ~budget_s seconds."""
    from oracle import hyperseg_oracle as O
    from oracle import cpu_port as P
    plan = O.config_plan('M')
    params = {k: v for k, v in model_cpu.decoder.state_dict().items()}
    x = torch.rand(1, 3, *size)
    torch.set_flush_denormal(True)

    def frame():
        t0 = time.perf_counter()
        feats = model_cpu.backbone(x)
        s = model_cpu.weight_mapper(feats[-1])
        t1 = time.perf_counter()
        P.decoder_v1_0(plan, params, [x] + feats[:-1], s)
        return t1 - t0, time.perf_counter() - t1
    ncpu = os.cpu_count() or 1
    with torch.no_grad():
        best = None
        for nt in sorted({min(n, ncpu) for n in (8, 16, 32, 64, ncpu)}):
            torch.set_num_threads(nt)
            t_first = sum(frame())
            # the candidates are ascending and the time is unimodal in the thread count: once a count is clearly slower than the best the
            # larger ones are not tried (round 6: at 256 threads ONE frame took over a minute -- 165 of the default run's 180 s)
            if best is not None and t_first > 1.5 * best[0]:
                break
            t = sum(frame())
            if best is None or t < best[0]:
                best = (t, nt)
        torch.set_num_threads(best[1])
        frame()
        t_start, enc, dec, n = time.perf_counter(), 0.0, 0.0, 0
        while n < 3 or (time.perf_counter() - t_start < budget_s and n < 200):
            e, d = frame()
            enc, dec, n = enc + e, dec + d, n + 1
    total = enc + dec
    return {'value': round(n / total, 3), 'unit': 'frames/s', 'cores': best[1], 'kind': 'port',
            'sample': f'{n} frames of HyperSeg-M 1024x512 bs1 ({total:.1f} s) on {best[1]} of {ncpu} host threads '
                      f'(thread count = the fastest of {{8, 16, 32, 64, all {ncpu}}} in an ascending one-frame calibration that stops at the first count '
                      f'1.5x slower than the best: more threads slow these small ops down, so the choice favours the CPU): stock encoder + context head on CPU + oracle/cpu_port.py decoder',
            'decoder_ms': round(1e3 * dec / n, 2), 'encoder_ms': round(1e3 * enc / n, 2)}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--repeats', type=int, default=5, help='timed regions of --steps steps each; value = their median')
    ap.add_argument('--model', default='m', choices=sorted(MODELS),
                    help='m: HyperSeg-M 1024x512 bs1/GPU (default, the headline metric); s: HyperSeg-S 1536x768 bs1/GPU; '
                         'sc: CamVid-S 768x576; l: HyperSeg-L 512x512, global batch 32 sharded over the GPUs; lc: CamVid HyperSeg-L 1024x768 '
                         '(configs/train/camvid_efficientnet_b1_hyperseg-l.py: six-level v1_0 decoder)')
    ap.add_argument('--output', default='logits', choices=['logits', 'masks'],
                    help="what a step produces: fp32 logits (the reference's forward, default) or uint8 argmax masks "
                         "taken inside the final upsample kernel (HyperGen.segment; test_fps.py:194's epilogue fused)")
    ap.add_argument('--gather', default='logits', choices=['logits', 'masks', 'auto'],
                    help="what the N>1 collective moves (north star: logits); 'auto': logits unless no logits schedule fits the xGMI "
                         'links at the measured step rate (--collective fit) or the masks candidate calibrates faster (--collective auto)')
    ap.add_argument('--collective', default=None, choices=['fit', 'auto', 'ingraph', 'allgather', 'direct', 'gather', 'none'],
                    help="N>1: the all-gather of every rank's logits.  'fit' (default at N>1): the schedule whose busiest xGMI link keeps up "
                         "with the measured single-GPU step rate (hyperseg_amd.distributed.fitting_policy): the RCCL ring all-gather if "
                         "(N-1) x payload x steps/s fits one link, else the all-pairs schedule ('direct': one shard per link and direction); "
                         "'allgather': RCCL all_gather_into_tensor, in place, zero copy, on RCCL's own stream; 'ingraph': the same collective as "
                         "a parallel branch INSIDE the step's HIP graph; 'auto': ingraph, allgather, direct (and the masks payload) timed in a "
                         "short calibration, the fastest kept; 'direct': grouped RCCL point-to-point sends / receives, all pairs; 'gather': "
                         "onto rank 0 (nn.DataParallel semantics); 'none'.  Given explicitly at N=1 it runs the collective on a one-rank "
                         "group and reports its per-step overhead ('collective.overhead_pct')")
    ap.add_argument('--link-gbs', type=float, default=None,
                    help='xGMI bandwidth of one link in one direction, GB/s (default: hyperseg_amd.distributed.XGMI_LINK_GBS_PER_DIRECTION = 76.5)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the other_configs side objects (HyperSeg-S 1536x768 and the config-5 training step) of the default run')
    ap.add_argument('--probe-load', type=int, default=0,
                    help='N=1 collective probe only: that many extra out-of-place all-gathers of the payload per step (one RCCL copy '
                         'kernel each at world 1), so that RCCL kernels really run beside the forward')
    ap.add_argument('--calib-steps', type=int, default=40, help="steps per candidate of --collective auto's calibration")
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of HIP graph replay')
    ap.add_argument('--stock-encoder', action='store_true',
                    help='leave the encoder entirely on stock PyTorch-ROCm/MIOpen (no fused depthwise HIP kernel)')
    ap.add_argument('--library-gemm', dest='split_gemm', action='store_false',
                    help="the encoder's 1x1 convolutions through the library f32 GEMM instead of hs_gemm_split_fwd (the default since "
                         'round 3: parity-green inside the model and 8 %% faster, profiles/round3_first_visit.txt)')
    ap.add_argument('--split-gemm', dest='split_gemm', action='store_true', help='(default) 1x1 convolutions through hs_gemm_split_fwd')
    ap.set_defaults(split_gemm=True)
    ap.add_argument('--ir-math', choices=['auto', 'f32', 'split'], default='auto',
                    help='arithmetic of the fused inverted-residual decoder levels (include/hyperseg_hip.h hs_ir_math).  Default auto '
                         '(what serving runs): the level-4 block multiplies 3-term f16 splits of its f32 operands on the f16 matrix cores '
                         'with f32 accumulation -- `dtype` says so in words; the `exact_f32` object re-times the step with every decoder '
                         'product on the exact-f32 matrix cores (--ir-math f32 makes that the headline)')
    ap.add_argument('--chain-k1', dest='chain_k1', action='store_true',
                    help="(default) the decoder's three coarse k = 1 levels as ONE launch with in-launch neighbour hand-offs (hs_k1_chain_fwd)")
    ap.add_argument('--no-chain-k1', dest='chain_k1', action='store_false', help='one launch per k = 1 level')
    ap.set_defaults(chain_k1=True)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the roofline / parity / protocol passes (timing only)')
    ap.add_argument('--cpu-budget', type=float, default=12.0)
    ap.add_argument('--traffic', default='auto', choices=['auto', 'off'],
                    help="roofline.traffic: 'auto' = spawn the two rocprofv3 --pmc passes of this command before the timing (N=1, rank 0, "
                         "when rocprofv3 is on PATH; ~1 minute), 'off' = null unless --traffic-dir is given")
    ap.add_argument('--traffic-dir', default=None,
                    help='directory with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (same session)')
    args = ap.parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N` (the shape of the driver's N = 1 command; the reference's counterpart is one command for N
        # GPUs too: nn.DataParallel, test_fps.py:155-156): launch the N ranks ourselves, one process per GPU, under torch.distributed.run;
        # rank 0's JSON line is this process' stdout
        return launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv))
    # The ONE JSON line goes to the process' real stdout; everything else that writes to fd 1 (RCCL prints a version banner there
    # when its first communicator comes up, MIOpen may chat) is sent to stderr, so that the line is the only thing a reader sees.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    legs = Legs()
    stub = os.environ.get('HS_BENCH_STUB') == '1'
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        args.gpus = world                                   # the launcher's world size is the truth (torchrun --nproc-per-node)
    dev = select_device(local_rank, stub)
    traffic_note = None
    if (not stub and world == 1 and not args.no_extras and args.traffic == 'auto' and args.traffic_dir is None
            and os.environ.get('HS_BENCH_CHILD') != '1'):
        args.traffic_dir, traffic_note = self_traffic_passes(args.model)      # before this process initialises the GPU
        legs.mark('pmc_traffic_passes')
    if not stub:
        torch.cuda.set_device(dev)
    collective_probe = world == 1 and args.collective not in (None, 'none')      # N=1: measure the collective's own cost
    if args.collective is None:
        # N > 1 default: 'fit' -- the logits all-gather on the schedule whose busiest xGMI link keeps up with the step rate this very run
        # measures on its own GPU (round 4's default, the ring all-gather, needs ~300 GB/s per link at 8 x HyperSeg-M against ~77 per
        # direction; the all-pairs schedule needs 51).  No calibration of collectives, one decision from one all-reduced number: the
        # FIRST multi-GPU run this code ever gets cannot diverge between ranks.  'auto' times every candidate instead.
        args.collective = 'fit' if world > 1 else 'none'
    if args.collective == 'fit' and world == 1:
        args.collective = 'allgather'                       # N = 1 probe: nothing to fit
    if world > 1 or collective_probe:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if collective_probe:
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ.setdefault('MASTER_PORT', str(sk.getsockname()[1]))
        if stub:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from hyperseg_amd import configs
    from hyperseg_amd.distributed import InGraphAllGather, LogitsGatherer
    from hyperseg_amd.utils.synthetic import fill_by_name
    import copy

    plan = plan_workload(args.model, rank, world)
    cfg, spec, h, w = plan['cfg'], plan['spec'], plan['h'], plan['w']
    batch, global_batch, scaling = plan['batch'], plan['global_batch'], plan['scaling']
    stock = stock_cpu = None
    if stub:
        model = StubModel(plan, rank)
        x = torch.zeros(1)
        args.no_graph = args.no_extras = True
    else:
        from hyperseg_amd.utils.inference import prepare_for_inference
        model = fill_by_name(configs.build(cfg).eval(), seed=0)       # synthetic, non-denormal, same on every rank
        if rank == 0 and not args.no_extras:
            stock, stock_cpu = copy.deepcopy(model), (copy.deepcopy(model) if world == 1 and not args.stock_encoder else None)
        if not args.stock_encoder:
            prepare_for_inference(model, fold_bn=False, fused_depthwise=True, split_gemm=args.split_gemm, ir_math=args.ir_math,
                                  chain_k1=args.chain_k1)
        else:
            from hyperseg_amd.utils.inference import set_ir_math
            set_ir_math(model, args.ir_math)
        model = model.to(dev)
        torch.manual_seed(1234 + rank)
        x = torch.rand(batch, 3, h, w, device=dev)                    # resident synthetic batch
    torch.set_grad_enabled(False)

    # ---- the step: a HIP graph of the whole forward ---------------------------------------------------------------
    forward = model.segment if args.output == 'masks' else model
    if args.output == 'masks':
        args.gather = 'masks'
    graph = None
    if stub:
        y = forward(x)
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                y = forward(x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                y = forward(x)

    legs.mark('build_prepare_capture')

    def run_forward():
        nonlocal y
        if graph is not None:
            graph.replay()
        else:
            y = forward(x)
        return y

    def forward_into(out):
        """The forward with the decoder's last kernel writing the logits into ``out`` (zero copy into a ring slot)."""
        model.decoder.output_buffer = out
        try:
            return forward(x)
        finally:
            model.decoder.output_buffer = None

    # ---- the collective: candidates, calibration, choice --------------------------------------------------------------
    from hyperseg_amd.distributed import XGMI_LINK_GBS_PER_DIRECTION, LINK_HEADROOM, fitting_policy, link_gbs_needed, link_schedule
    link_gbs = args.link_gbs if args.link_gbs is not None else XGMI_LINK_GBS_PER_DIRECTION

    def payload_spec(gather):
        """(shape, dtype, to_payload, bytes) of what one rank contributes per step."""
        if gather == 'logits':
            shp, dt, conv = tuple(y.shape), torch.float32, None
        else:
            shp, dt = (y.shape[0],) + tuple(y.shape[-2:]), torch.uint8
            conv = lambda t: t if t.dtype == torch.uint8 else t.argmax(1).to(torch.uint8)   # noqa: E731
        return shp, dt, conv, int(torch.empty((), dtype=dt).element_size()) * int(torch.Size(shp).numel())
    if args.output == 'masks' and args.gather != 'masks':
        args.gather = 'masks'
    gather = 'logits' if args.gather == 'auto' else args.gather
    notes = {}

    def make_loop(policy, gather=None):
        """StepLoop for one collective policy (None = no collective); returns (loop, comm, zero_copy)."""
        if policy in (None, 'none'):
            return StepLoop(run_forward, None, None, world, dev), None, False
        shape, dtype, to_payload, _ = payload_spec(gather)
        zero_copy_ok = (graph is not None and gather == 'logits' and args.output != 'masks'
                        and hasattr(getattr(model, 'decoder', None), 'forward'))
        mode = 'allgather' if policy == 'ingraph' else policy
        comm = LogitsGatherer(world, shape, dtype, dev, mode=mode, probe_load=args.probe_load if world == 1 else 0)
        if policy == 'ingraph':
            if not zero_copy_ok:
                raise RuntimeError('ingraph needs graph replay and a logits payload')
            ing = InGraphAllGather(comm, forward_into, probe_load=args.probe_load if world == 1 else 0)
            loop = StepLoop(ing.step, None, None, world, dev, forward_takes_step=True, on_drain=ing.drain)
            return loop, ing, True
        graphs = None
        if zero_copy_ok and mode in ('allgather', 'direct'):
            # zero copy: one HIP graph per ring slot, the decoder's last kernel writing the logits straight into the slot the
            # collective sends from (LogitsGatherer.slot; VERDICT r2 #9: submit used to copy 39.8 MB per step)
            from hyperseg_amd.distributed import RING
            graphs = []
            for k in range(RING):
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk):
                    yk = forward_into(comm.slot(k))
                graphs.append((gk, yk))
            if not all(yk.data_ptr() == comm.slot(k).data_ptr() for k, (_, yk) in enumerate(graphs)):
                graphs = None

        def run_forward_slot(i):
            gk, yk = graphs[i % len(graphs)]
            gk.replay()
            return yk
        loop = StepLoop(run_forward_slot if graphs else run_forward, comm, to_payload, world, dev, forward_takes_step=bool(graphs))
        return loop, comm, bool(graphs)

    def agree(ok):
        """True only if every rank says so (a policy is usable only if it came up everywhere)."""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if not stub else None)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    base_ms = None
    if collective_probe:                                  # the same step without the collective, for the overhead figure
        base = run_timed(StepLoop(run_forward, None, None, world, dev), args.steps, args.warmup, max(1, args.repeats))
        base_ms = 1e3 * statistics.median(base) / args.steps
    calibration = None
    policy = args.collective
    fit = None
    if policy == 'fit':
        # this rank's own step rate (no collective), MAX-reduced like every timed region: every rank computes the same decision
        t = run_timed(StepLoop(run_forward, None, None, world, dev), args.calib_steps, min(10, args.warmup), 1)[0]
        rate = args.calib_steps / t
        nbytes = payload_spec(gather)[3]
        policy, fits = fitting_policy(world, nbytes, rate, link_gbs)
        if not fits and args.gather == 'auto':
            gather = 'masks'
            policy, fits = fitting_policy(world, payload_spec('masks')[3], rate, link_gbs)
        fit = {'steps_per_s_without_collective': round(rate, 1), 'fits': fits,
               'link_gbs_needed': {c: round(link_gbs_needed(c, world, payload_spec(gather)[3], rate), 2) for c in ('allgather', 'direct')}}
        loop, comm, zero_copy = make_loop(policy, gather)
    elif policy == 'auto':
        graphable = graph is not None and args.output != 'masks' and hasattr(getattr(model, 'decoder', None), 'forward')
        candidates = [(c, gather) for c in ((['ingraph'] if graphable and gather == 'logits' else []) + ['allgather', 'direct'])]
        if gather == 'logits':
            candidates.append(('allgather', 'masks'))       # timed and reported; eligible only under --gather auto
        calibration, built = {}, {}
        for cand, gth in candidates:
            label = cand if gth == gather else f'{cand}:{gth}'
            try:
                trio = make_loop(cand, gth)
                ok = True
            except Exception as e:                        # noqa: BLE001  (e.g. a stack that cannot capture RCCL)
                notes[label] = f'{type(e).__name__}: {e}'[:300]
                trio, ok = None, False
                if not stub:
                    torch.cuda.synchronize()
            if not agree(ok):
                calibration[label] = None
                continue
            t = run_timed(trio[0], args.calib_steps, min(10, args.warmup), 1)[0]
            calibration[label] = round(1e3 * t / args.calib_steps, 4)
            built[label] = (trio, cand, gth)
        usable = {c: v for c, v in calibration.items() if v is not None and (args.gather == 'auto' or built[c][2] == gather)}
        if not usable:
            raise SystemExit(f'no collective policy came up: {notes}')
        best = min(usable, key=usable.get)                  # the calibration time is MAX-reduced: every rank picks the same
        (loop, comm, zero_copy), policy, gather = built[best]
        for c in list(built):
            if c != best:
                del built[c]
    else:
        loop, comm, zero_copy = make_loop(policy, gather)
    times = run_timed(loop, args.steps, args.warmup, max(1, args.repeats))
    med = statistics.median(times)
    fps = args.steps * global_batch / med
    rank_fps = torch.tensor([args.steps * batch / med], dtype=torch.float64, device=None if stub else dev)
    if world > 1:
        all_fps = [torch.zeros_like(rank_fps) for _ in range(world)]
        dist.all_gather(all_fps, rank_fps)
        per_rank = [round(float(t.item()), 1) for t in all_fps]
    else:
        per_rank = [round(float(rank_fps.item()), 1)]

    out = None
    if rank == 0:
        import hyperseg_amd.functional as HFm
        math_name = HFm.get_ir_math(args.ir_math) if not stub else args.ir_math
        out = {
            'metric': f'frames/sec @ bs={spec["batch"]} {LABELS[args.model].split(" / ")[0]} {w}x{h}',
            'value': round(fps, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * med / args.steps, 4), 'higher_is_better': True, 'scaling': scaling,
            'vs_baseline': None,
            'dtype': 'f32' if math_name == 'f32' else 'f32 storage and accumulation; level-4 inverted residual products as 3-term f16 splits '
                                                      '(f32-class: 1.3e-7 of sum|a||b|; exact_f32 beside it)',
            'data': 'synthetic' if not stub else 'stub (HS_BENCH_STUB=1: CPU / gloo plumbing test, no model)',
            'repeats': {'n': len(times), 'ms_per_step': [round(1e3 * t / args.steps, 4) for t in times], 'value_from': 'median'},
            'config': {'workload': f'{LABELS[args.model]}, batch {batch} per GPU (global {global_batch}), whole model forward '
                                   '(encoder + context head as per "encoder", HIP decoder), resident input',
                       'encoder': 'stock PyTorch-ROCm / MIOpen' if args.stock_encoder else
                                  'hyperseg_amd.utils.inference.prepare_for_inference: stem + block 0 depthwise = hs_stem_dw_fwd; MBConv blocks = '
                                  'hs_mbconv_expand_dw_fwd | 1x1 GEMM + hs_depthwise_conv_fwd, hs_se_gate_fwd, 1x1 GEMM; '
                                  'context head = 4 launches of hs_gemm_split_*; 1x1 GEMMs of the MBConv blocks = ' +
                                  ('hs_gemm_split_fwd (f16 matrix cores, split operands, f32 accumulation)' if args.split_gemm
                                   else 'library f32 GEMM (--library-gemm)'),
                       'output': f'fp32 logits {tuple(y.shape)}' if args.output == 'logits' else f'uint8 argmax masks {tuple(y.shape)}',
                       'launch': 'eager' if args.no_graph else 'hipGraph replay',
                       'ir_math': math_name + ' (include/hyperseg_hip.h hs_ir_math)',
                       'decoder_launches': decoder_launches_text(model),
                       'arithmetic': 'f32 storage and f32 accumulation everywhere.  Decoder (the hot path): ' +
                                     ('every product exact f32 (v_mfma_f32_16x16x4_f32 / v_fma_f32)' if math_name == 'f32' else
                                      'level-4 inverted residual products as 3-term f16 splits (1.3e-7 * sum|a||b|), the rest exact f32') +
                                     '.  Encoder / context head (outside the path, inside the metric): 1x1 convolutions ' +
                                     ('as 3-term f16 split products with f32 accumulation (hs_gemm_split_fwd; error below an fp32 fmaf '
                                      "chain's) -- `library_gemm_f32` re-times the frame with IEEE f32 library GEMMs instead"
                                      if args.split_gemm and not args.stock_encoder else 'IEEE f32 (library GEMM)'),
                       'parallelism': f'batch-sharded x{world}' + (f', RCCL {policy} of {gather}' if comm is not None else '')},
            'per_rank_frames_per_s': per_rank,
            'collective': None if comm is None else {
                'policy': policy, 'requested': args.collective,
                'calibration_ms_per_step': calibration, 'calibration_steps': args.calib_steps if calibration else None,
                'notes': notes or None,
                'op': {'allgather': 'all_gather_into_tensor (in place) on the RCCL stream, one submit per step',
                       'ingraph': "all_gather_into_tensor (in place) captured into the step's HIP graph, parallel to the forward",
                       'direct': 'batch_isend_irecv, all pairs (one shard per link and direction)',
                       'gather': 'gather(dst=0)'}[policy], 'payload': gather,
                'fit': fit,
                # bytes across the busiest xGMI link in one direction per step under this schedule (hyperseg_amd.distributed.link_schedule),
                # what that asks of the link at the measured step rate, and what a link has
                **link_schedule(policy, world, comm.bytes_per_step),
                'link_gbs_needed_at_this_rate': round(link_gbs_needed(policy, world, comm.bytes_per_step, args.steps / med), 2),
                'link_gbs_per_direction': link_gbs, 'link_headroom': LINK_HEADROOM,
                'zero_copy': zero_copy, 'copies_into_the_ring': getattr(comm, 'copies', 0),
                'probe_load': args.probe_load if world == 1 else None,
                'ms_per_step_without': None if base_ms is None else round(base_ms, 4),
                'overhead_pct': None if base_ms is None else round(100.0 * (1e3 * med / args.steps - base_ms) / base_ms, 2),
                'bytes_sent_per_rank_per_step': comm.bytes_per_step,
                'bytes_received_per_rank_per_step': comm.bytes_per_step * (world - 1),
                'completed': comm.completed},
        }
        legs.mark('timed_regions_and_collective')
        if not args.no_extras:
            # ---- the benched configuration against the eager stock model, outside the timed regions -----------------
            stock = stock.to(dev)
            y_bench = run_forward()
            torch.cuda.synchronize()
            ys = stock(x)
            if args.output == 'masks':
                flips = int((y_bench.long() != ys.argmax(1)).sum())
                out['parity'] = {'vs': 'eager stock-encoder model, same batch', 'mask_mismatches': flips, 'pixels': int(y_bench.numel())}
            else:
                err = float((y_bench.double() - ys.double()).abs().max() / ys.double().abs().max())
                top2 = ys.topk(2, dim=1).values
                clear = (top2[:, 0] - top2[:, 1]) > 1e-4
                flips = int(((y_bench.argmax(1) != ys.argmax(1)) & clear).sum())
                out['parity'] = {'vs': 'eager stock-encoder model, same batch', 'max_rel_err': err,
                                 'argmax_flips': flips, 'pixels_with_margin_gt_1e-4': int(clear.sum()), 'pixels': int(clear.numel())}
            del stock, ys
            legs.mark('parity_vs_stock')
            # ---- roofline of the dominant decoder launch -------------------------------------------------------------
            launches, dec_us, ev_overhead = instrumented_decoder(model, x, max(10, min(args.steps, 50)))
            alg_bytes, levels = decoder_levels(model, h, w, batch)
            out['roofline'] = roofline_of(launches, levels, h, w, batch, args.traffic_dir)
            out['roofline']['traffic_source'] = traffic_note if traffic_note else \
                ('--traffic-dir' if args.traffic_dir else 'none (--traffic off)')
            out['config']['decoder_launches'] = decoder_launches_text(model, launches)
            out['decoder'] = {'us_per_batch_eager': round(dec_us, 1), 'event_pair_overhead_us': round(ev_overhead, 2),
                              'algorithmic_bytes': alg_bytes,
                              'hbm_frac_of_8TBs': round(alg_bytes / (dec_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                              'launches': launches}
            legs.mark('roofline_launch_table')
            if world == 1 and graph is not None:
                # ---- the same step under the other arithmetic of the fused inverted residual (hs_ir_math), and with IEEE-f32
                # library GEMMs in the encoder: side numbers, one timed region each ------------------------------------------
                headline = {'value': out['value'], 'unit': 'frames/s', 'ms_per_step': out['ms_per_step'], 'is_headline': True}
                for key, mode, what in (('exact_f32', 'f32', 'hs_ir_math = f32: v_mfma_f32_16x16x4_f32 everywhere in the decoder'),
                                        ('split_f16', 'auto', 'hs_ir_math = auto: level-4 products as 3-term f16 splits on '
                                                              'v_mfma_f32_16x16x32_f16, f32 accumulation')):
                    if HFm.get_ir_math(mode) == math_name:
                        out[key] = dict(headline, note=what + ' -- this IS the headline configuration')
                        continue
                    prev_math = HFm.set_ir_math(mode)
                    try:
                        v, ms, y2, g2 = time_replayed(forward, x, args.steps, args.warmup, batch)
                        launches2, _, _ = instrumented_decoder(model, x, max(10, min(args.steps, 50)))
                        dom2 = max([l for l in launches2 if l['in_decoder']], key=lambda l: l['avg_us'])
                        out[key] = {'value': v, 'unit': 'frames/s', 'ms_per_step': ms, 'regions': 1, 'is_headline': False,
                                    'dominant_launch_us': dom2['avg_us'],
                                    'max_abs_diff_vs_benched': float((y2 - y_bench).abs().max()) if args.output != 'masks' else None,
                                    'note': what}
                        del g2
                    finally:
                        HFm.set_ir_math(prev_math)
                if stock_cpu is not None and args.split_gemm:
                    try:
                        prepare_for_inference(stock_cpu, fold_bn=False, fused_depthwise=True, split_gemm=False, ir_math='f32')
                        lib = stock_cpu.to(dev)
                        v, ms, y3, g3 = time_replayed(lib.segment if args.output == 'masks' else lib, x, args.steps, args.warmup, batch)
                        out['library_gemm_f32'] = {'value': v, 'unit': 'frames/s', 'ms_per_step': ms, 'regions': 1,
                                                   'max_abs_diff_vs_benched': float((y3.float() - y_bench.float()).abs().max()),
                                                   'note': 'every product of the frame in IEEE f32: library f32 GEMMs for the 1x1 '
                                                           'convolutions (--library-gemm) + hs_ir_math = f32'}
                        del g3, lib
                    except Exception as e:            # noqa: BLE001  (a side number must never cost the line)
                        out['library_gemm_f32'] = {'error': f'{type(e).__name__}: {e}'[:300]}
                        torch.cuda.synchronize()
            legs.mark('arithmetic_legs')
            if world == 1 and graph is not None:
                dec_mod = getattr(model, 'decoder', None)
                chain_was = getattr(dec_mod, 'chain_k1', False)
                try:                                  # a side number must never cost the line
                    # two launches of the chained levels must not share the chain's workspace concurrently (one frame in flight per
                    # decoder: functional.K1Chain): the two request graphs are captured with one launch per level
                    if dec_mod is not None:
                        dec_mod.chain_k1 = False
                    out['two_frames_in_flight'] = two_in_flight(forward, x, y_bench, args.steps, args.warmup, batch)
                    out['two_frames_in_flight']['note'] += '; captured with chain_k1 off (one launch per decoder level)'
                except Exception as e:                # noqa: BLE001
                    out['two_frames_in_flight'] = {'error': f'{type(e).__name__}: {e}'[:300]}
                    torch.cuda.synchronize()
                finally:
                    if dec_mod is not None:
                        dec_mod.chain_k1 = chain_was
            legs.mark('two_frames_in_flight')
            if world == 1:
                # ---- the reference harness' own protocol (sync + pinned H2D + eager forward per iteration) -------------
                from hyperseg_amd.fps import measure_fps, synthetic_batches
                uniq = synthetic_batches(4, batch, (h, w), spec['num_classes'], dev)
                res = measure_fps(model, [uniq[i % 4] for i in range(max(8, 64 // batch))], dev, spec['num_classes'])
                out['fps_reference_protocol'] = {'value': round(res['fps'], 1), 'unit': 'frames/s', 'frames': res['frames'],
                                                 'protocol': 'test_fps.py:163-191: per iteration sync, pinned H2D, eager forward, sync'}
                try:                                  # the same protocol with one HIP-graph replay per frame (GraphedModel)
                    from hyperseg_amd.utils.inference import GraphedModel
                    res_g = measure_fps(GraphedModel(model), [uniq[i % 4] for i in range(max(8, 64 // batch))], dev,
                                        spec['num_classes'])
                    out['fps_reference_protocol']['graphed'] = {
                        'value': round(res_g['fps'], 1), 'unit': 'frames/s', 'mean_iou_equal_to_eager':
                        abs(res_g['mean_iou'] - res['mean_iou']) < 1e-6,
                        'protocol': 'as above with forward = H2D into the static input + one HIP-graph replay '
                                    '(hyperseg_amd.utils.inference.GraphedModel)'}
                except Exception as e:                # noqa: BLE001
                    out['fps_reference_protocol']['graphed'] = {'error': f'{type(e).__name__}: {e}'[:300]}
                    torch.cuda.synchronize()
                legs.mark('fps_reference_protocol')
                out['cpu_baseline'] = None
                if args.model == 'm' and not args.no_cpu_baseline:
                    out['cpu_baseline'] = cpu_baseline(fill_by_name(configs.build(cfg).eval(), seed=0), (h, w), args.cpu_budget)
                legs.mark('cpu_baseline')
                if args.model == 'm' and graph is not None and not args.no_other_configs and not args.stock_encoder:
                    # ---- the other BASELINE configs on the driver's line: config 3 (1536x768) and config 5 (training
                    # step, fp32 + bf16), one short region each, after and outside every headline region; never `value` ------------
                    graph = None                          # free the headline's graph pool before the side models are built
                    torch.cuda.empty_cache()
                    other = {}
                    for name, fn in (('s', lambda: side_model('s', dev, 60, 10, args.ir_math, args.split_gemm)),
                                     ('train_sc', lambda: side_train_step(dev, 40))):
                        t_side = time.perf_counter()
                        try:
                            other[name] = fn()
                        except Exception as e:            # noqa: BLE001  (a side number must never cost the line)
                            other[name] = {'error': f'{type(e).__name__}: {e}'[:300]}
                            torch.cuda.synchronize()
                        other[name]['wall_s'] = round(time.perf_counter() - t_side, 1)
                    out['other_configs'] = other
                    legs.mark('other_configs')
        out['legs_s'] = legs.out
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if args.traffic_dir and traffic_note:
        import shutil
        shutil.rmtree(args.traffic_dir, ignore_errors=True)
    os.dup2(json_fd, 1)                                  # a caller that imported main() gets its stdout back
    os.close(json_fd)
    return out


if __name__ == '__main__':
    main()

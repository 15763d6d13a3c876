#!/usr/bin/env python
"""bench.py -- frames/sec of HyperSeg-M (EfficientNet-B1, 1024x512, bs=1 per GPU) on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one forward of the whole model on one resident synthetic 1024x512 frame per rank:
encoder + context head (hyperseg_amd.utils.inference.prepare_for_inference; --stock-encoder: plain PyTorch-ROCm), then
the HIP decoder (hot path), captured once in a HIP graph and replayed.  N > 1 is batch-sharded inference (one process
per GPU, weak scaling) with an RCCL gather of the logits onto rank 0 (nn.DataParallel's semantics; --collective
allgather for an all-gather) on RCCL's stream, overlapped with the next frame.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant decoder kernel (hs_patch_ir_fwd at level 4): algorithmic FLOPs (and bytes) per
                launch / its average duration measured with HIP events on the launch stream over `steps`
                instrumented eager steps run right after the timed region (a graph replay cannot host events).
  cpu_baseline  the CPU oracle ("port": stock encoder on CPU + oracle/hyperseg_oracle.py decoder) timed on the
                host cores of this box on a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL peer-to-peer needs it on this driver

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy peak 6290
FP32_PEAK_TFLOPS = 157.3       # f32 vector == f32-input MFMA dense peak
MODEL = 'hyperseg-m'


def decoder_algorithmic(model, h, w, batch=1):
    """Algorithmic HBM bytes and FLOPs of the decoder per frame (definition: SURVEY.md section 8d):
    skips read once, each level output written once and read once at its own resolution, banks read
    once, final logits written once; intermediates 0 B.  Returns (total_bytes, per_level list)."""
    dec = model.decoder
    fh, fw = h // 32, w // 32
    p = batch * fh * fw
    feat = [3] + model.backbone.feat_channels[:-1]
    levels, total = [], 0
    prev_c = 0
    for l in range(dec.levels):
        blk = getattr(dec, f'level_{l}')[0]
        blk = blk[0] if hasattr(blk, '__getitem__') else blk
        stride = 32 >> l
        hl, wl = h // stride, w // stride
        skip_c = feat[::-1][l]
        if hasattr(blk, 'hidden_dim'):
            cin, cout, hid = blk.in_nc, blk.out_nc, blk.hidden_dim
            ph, pw = hl // fh, wl // fw
            macs = p * ((ph + 2) * (pw + 2) * cin * hid + ph * pw * (9 * hid + hid * cout))
        else:
            cin, cout, hid = blk.in_channels, blk.out_channels, 0
            macs = batch * hl * wl * cin * cout
        in_b = 4 * batch * (skip_c * hl * wl + prev_c * (hl // 2) * (wl // 2))
        bank_b = 4 * p * int(blk.hyper_params)
        out_b = 4 * batch * cout * hl * wl
        levels.append(dict(level=l, cin=cin, cout=cout, hidden=hid, macs=macs, in_bytes=in_b, bank_bytes=bank_b,
                           out_bytes=out_b))
        total += in_b + bank_b + out_b
        prev_c = cout
    total += 4 * batch * prev_c * ((h // 2) * (w // 2) + h * w)      # final upsample: read low-res, write full-res
    return total, levels


def cpu_baseline(model_cpu, size, budget_s=20.0):
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
            frame()
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
            'sample': f'{n} frames of HyperSeg-M 1024x512 bs1 ({total:.1f} s) on {best[1]} of {ncpu} host threads: '
                      f'stock encoder + context head on CPU + oracle/cpu_port.py decoder',
            'decoder_ms': round(1e3 * dec / n, 2), 'encoder_ms': round(1e3 * enc / n, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--output', default='logits', choices=['logits', 'masks'],
                    help="what a step produces: fp32 logits (the reference's forward, default) or uint8 argmax masks "
                         "taken inside the final upsample kernel (HyperGen.segment; test_fps.py:194's epilogue fused)")
    ap.add_argument('--gather', default='logits', choices=['logits', 'masks', 'none'],
                    help='what the N>1 collective moves (north star: logits)')
    ap.add_argument('--collective', default='gather', choices=['gather', 'allgather'],
                    help='N>1: gather onto rank 0 (nn.DataParallel semantics, default) or all_gather to every rank')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of HIP graph replay')
    ap.add_argument('--stock-encoder', action='store_true',
                    help='leave the encoder entirely on stock PyTorch-ROCm/MIOpen (no fused depthwise HIP kernel)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
        args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from hyperseg_amd import configs
    from hyperseg_amd.utils.synthetic import fill_by_name
    import hyperseg_amd.functional as HF

# (the BLAS behind each bare GEMM is chosen per call in hyperseg_amd.utils.inference.gemm_library)
    spec = configs.MODELS[MODEL]
    h, w = spec['size']
    from hyperseg_amd.utils.inference import prepare_for_inference
    model = fill_by_name(configs.build(MODEL).eval(), seed=0)       # synthetic, non-denormal, same on every rank
    cpu_model = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import copy
        cpu_model = copy.deepcopy(model)
    # inference preparation of the encoder: depthwise conv + BN + swish of every MBConv block as one
    # HIP launch (MIOpen's fp32 depthwise path costs half of the frame); everything else of the encoder is stock PyTorch
    if not args.stock_encoder:
        prepare_for_inference(model, fold_bn=False, fused_depthwise=True)
    model = model.to(dev)
    torch.manual_seed(1234 + rank)
    x = torch.rand(spec['batch'], 3, h, w, device=dev)              # resident synthetic frame
    torch.set_grad_enabled(False)

    # ---- build the step (HIP graph of the whole forward) -------------------------------------
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    forward = model.segment if args.output == 'masks' else model
    if args.output == 'masks' and args.gather == 'logits':
        args.gather = 'masks'
    with torch.cuda.stream(side):
        for _ in range(3):
            y = forward(x)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y = forward(x)

    comm = None
    if world > 1 and args.gather != 'none':
        from hyperseg_amd.distributed import LogitsGatherer
        shape = tuple(y.shape) if args.gather == 'logits' else (y.shape[0],) + tuple(y.shape[-2:])
        dtype = torch.float32 if args.gather == 'logits' else torch.uint8
        comm = LogitsGatherer(world, shape, dtype, dev, mode=args.collective)

    def step(i):
        nonlocal y
        if graph is not None:
            graph.replay()
        else:
            y = forward(x)
        if comm is not None:
            # RCCL gather over xGMI on RCCL's own stream: overlaps the next frame's compute
            comm.submit(i, y if (args.gather == 'logits' or y.dtype == torch.uint8) else y.argmax(1).to(torch.uint8))

    def drain():
        if comm is not None:
            comm.drain()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = args.steps * spec['batch'] * world
    fps = frames / elapsed

    # ---- instrumented eager pass: per-launch durations of the decoder kernels -----------------
    out = None
    if rank == 0:
        names = ['signal2weights', 'signal2weights_multi', 'bank_pack', 'patch_conv', 'patch_ir', 'upsample_bilinear']
        orig = {n: getattr(HF, n) for n in names}
        recs, counter = {}, [0]

        def wrap(n):
            def f(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig[n](*a, **k)
                e1.record()
                recs.setdefault((counter[0], n), []).append((e0, e1))
                counter[0] += 1
                return r
            return f
        for n in names:
            setattr(HF, n, wrap(n))
        feats = model.backbone(x)
        sig = model.weight_mapper(feats[-1]).contiguous()
        pyr = [t.contiguous() for t in [x] + feats[:-1]]
        d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_inst = max(10, min(args.steps, 100))
        dec_evs = []
        for _ in range(n_inst):
            counter[0] = 0
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # park the GPU for ~0.4 ms so that the host enqueues the whole decoder before the first launch starts:
            # the events then bracket back-to-back kernels (device time) instead of host launch gaps
            torch.cuda._sleep(1_000_000)
            d0.record()
            model.decoder(pyr, sig)
            d1.record()
            dec_evs.append((d0, d1))
        torch.cuda.synchronize()
        for n in names:
            setattr(HF, n, orig[n])
        # an empty event pair on a busy stream is not 0: calibrate that overhead and report both
        cal = []
        for _ in range(50):
            torch.cuda._sleep(200_000)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            cal.append((e0, e1))
        torch.cuda.synchronize()
        cal = sorted(a.elapsed_time(b) * 1e3 for a, b in cal)
        ev_overhead = cal[len(cal) // 2]
        launches = []
        for (i, n), evs in sorted(recs.items()):
            ts = [a.elapsed_time(b) * 1e3 for a, b in evs]
            avg = sum(ts) / len(ts)
            # avg_us = raw event-pair time (conservative: includes the ~ev_overhead/2 of event bookkeeping on each side;
            # rocprofv3's kernel duration lies between avg_us - ev_overhead and avg_us)
            launches.append(dict(idx=i, kernel='hs_' + n + '_fwd', avg_us=round(avg, 2),
                                 minus_event_overhead_us=round(max(avg - ev_overhead, 0.0), 2)))
        dec_us = sum(a.elapsed_time(b) for a, b in dec_evs) * 1e3 / len(dec_evs)

        alg_bytes, levels = decoder_algorithmic(model, h, w, spec['batch'])

        def pmc_traffic(kernel_substr):
            """HBM bytes per launch from the committed PMC passes (profiles/round1_pmc_hbm.json: separate rocprofv3 --pmc
            FETCH_SIZE / WRITE_SIZE runs of tools/prof_decoder.py; KB units; gfx950 FETCH_SIZE x2 correction)."""
            try:
                doc = json.load(open(os.path.join(REPO, 'profiles', 'round1_pmc_hbm.json')))
                for k, v in doc['kernels'].items():
                    if kernel_substr in k:
                        return int((2 * v['FETCH_SIZE_KB_avg_per_launch'] + v['WRITE_SIZE_KB_avg_per_launch']) * 1024)
            except (OSError, KeyError, ValueError):
                pass
            return None
        ir = [l for l in launches if l['kernel'] == 'hs_patch_ir_fwd']
        dom = max(launches, key=lambda l: l['avg_us'])
        lv4 = levels[-1]
        if dom['kernel'] == 'hs_patch_ir_fwd' and dom is ir[-1]:
            flops = 2.0 * lv4['macs']
            kbytes = lv4['in_bytes'] + lv4['bank_bytes'] + lv4['out_bytes']
            t_s = dom['avg_us'] * 1e-6
            # the kernel's binding roof: fp32 FLOPs (8.0 us at peak) > HBM bytes (3.6 us at peak)
            roof = {'bound': 'mfma', 'kernel': 'hs_patch_ir_fwd (level 4: 34->68->19 ch, 16x16 patches)',
                    'achieved': round(flops / t_s / 1e12, 3), 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(flops / t_s / 1e12 / FP32_PEAK_TFLOPS, 4),
                    'traffic': pmc_traffic('patch_ir_mfma_kernel<34, 16, 19, 16>'),
                    'avg_launch_us': dom['avg_us'], 'algorithmic_flops': flops, 'algorithmic_bytes': kbytes,
                    'hbm_frac': round(kbytes / t_s / 1e9 / HBM_PEAK_GBS, 4),
                    'note': 'fp32 math: f32 vector peak == f32-input MFMA dense peak (157.3 TF/s)'}
        else:
            # some other launch dominates: report it against the HBM roof with its own algorithmic bytes
            per = {}
            li = 0
            for l in launches:
                if l['kernel'] in ('hs_patch_conv_fwd', 'hs_patch_ir_fwd'):
                    lv = levels[li]
                    per[l['idx']] = lv['in_bytes'] + lv['bank_bytes'] + lv['out_bytes']
                    li += 1
            li = 0
            for l in launches:
                if l['kernel'] == 'hs_signal2weights_multi_fwd':
                    per[l['idx']] = sum(lv['bank_bytes'] for lv in levels)
                if l['kernel'] == 'hs_signal2weights_fwd':
                    per[l['idx']] = levels[li]['bank_bytes']
                    li += 1
                if l['kernel'] == 'hs_upsample_bilinear_fwd':
                    per[l['idx']] = 4 * levels[-1]['cout'] * ((h // 2) * (w // 2) + h * w)
            kbytes = per.get(dom['idx'], 0)
            t_s = dom['avg_us'] * 1e-6
            roof = {'bound': 'hbm', 'kernel': dom['kernel'], 'achieved': round(kbytes / t_s / 1e9, 1),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(kbytes / t_s / 1e9 / HBM_PEAK_GBS, 4),
                    'traffic': None, 'avg_launch_us': dom['avg_us'], 'algorithmic_bytes': kbytes}

        out = {
            'metric': 'frames/sec @ bs=1 HyperSeg-M 1024x512',
            'value': round(fps, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'HyperSeg-M / EfficientNet-B1 / 1024x512 bs=1 per GPU, whole model forward '
                                   '(encoder + context head as per "encoder", HIP decoder), resident input',
                       'encoder': 'stock PyTorch-ROCm / MIOpen' if args.stock_encoder else
                                  'hyperseg_amd.utils.inference.prepare_for_inference: MBConv blocks = hs_mbconv_expand_dw_fwd | '
                                  'library GEMM + hs_depthwise_conv_fwd, hs_se_gate_fwd, bare library GEMM; hs_stem_conv_fwd; '
                                  'context head = library GEMMs + hs_affine_act_fwd',
                       'output': 'fp32 logits (B,19,512,1024)' if args.output == 'logits' else 'uint8 argmax masks (B,512,1024)',
                       'launch': 'eager' if args.no_graph else 'hipGraph replay',
                       'parallelism': f'batch-sharded x{world}' + (f', RCCL {args.collective} of {args.gather}'
                                                                   if comm is not None else '')},
            'roofline': roof,
            'decoder': {'us_per_frame_eager': round(dec_us, 1), 'event_pair_overhead_us': round(ev_overhead, 2),
                        'algorithmic_bytes': alg_bytes,
                        'hbm_frac_of_8TBs': round(alg_bytes / (dec_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        'launches': launches},
        }
        if cpu_model is not None:
            out['cpu_baseline'] = cpu_baseline(cpu_model, (h, w), args.cpu_budget)
        elif world == 1:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""The other BASELINE configurations as side objects of the default run: HyperSeg-S 1536x768 (config 3) and the config-5 training step.

Part of bench.py's measurement harness (round 6: bench.py was one 1 100-line file running ten legs; the legs live here, bench.py is the
driver entry).  Nothing in this package imports oracle/: the CPU-baseline leg, the only one that may, stays in bench.py."""
import time

import torch

from .constants import HBM_PEAK_GBS, LABELS
from .decoder_probe import decoder_levels, instrumented_decoder, roofline_of
from .launch import plan_workload
from .timing import time_replayed


def side_model(key, dev, steps, warmup, ir_math, split_gemm):
    """BASELINE config 3 (and any other --model key) as a SIDE object of the default run: whole-model frames/s of one timed region of
    HIP-graph replays, the decoder's eager launch table and the roofline of ITS dominant launch (event-timed like the headline's;
    `traffic` null: no PMC pass is spent on side objects).  Built, measured and freed outside every headline region."""
    from hyperseg_amd import configs
    from hyperseg_amd.utils.inference import prepare_for_inference
    from hyperseg_amd.utils.synthetic import fill_by_name
    plan = plan_workload(key, 0, 1)
    h, w, batch = plan['h'], plan['w'], plan['batch']
    model = fill_by_name(configs.build(plan['cfg']).eval(), seed=0)
    prepare_for_inference(model, fold_bn=False, fused_depthwise=True, split_gemm=split_gemm, ir_math=ir_math)
    model = model.to(dev)
    x = torch.rand(batch, 3, h, w, generator=torch.Generator().manual_seed(4321)).to(dev)
    v, ms, y, g = time_replayed(model, x, steps, warmup, batch)
    launches, dec_us, _ = instrumented_decoder(model, x, 6)
    alg_bytes, levels = decoder_levels(model, h, w, batch)
    roof = roofline_of(launches, levels, h, w, batch, None)
    roof['traffic_source'] = 'none (side object: no PMC pass)'
    out = {'workload': f'{LABELS[key]}, batch {batch}, whole model forward, resident input, hipGraph replay', 'value': v, 'unit': 'frames/s',
           'ms_per_step': ms, 'steps': steps, 'regions': 1, 'ir_math': ir_math, 'finite': bool(torch.isfinite(y).all()),
           'decoder': {'us_per_batch_eager': round(dec_us, 1), 'algorithmic_bytes': alg_bytes,
                       'hbm_frac_of_8TBs': round(alg_bytes / (dec_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                       'launches': [(l['kernel'], l['avg_us']) for l in launches if l['in_decoder']]},
           'roofline': roof}
    del g, model
    return out


def side_train_step(dev, iters):
    """BASELINE config 5 as a SIDE object: one training step of the CamVid-S decoder (576x576 crops, bs 2: forward + bootstrapped cross
    entropy + backward + Adam) replayed as ONE HIP graph (hyperseg_amd.training.GraphedTrainStep), fp32 and under bf16 autocast
    (bf16 activation storage, f32 accumulation; banks, statistics and the optimizer fp32).  Roofline at STEP level -- the step is a
    chain of small launches, none of which dominates: algorithmic bytes of the step (forward bytes of SURVEY 8d x 3: the forward pass,
    the input-gradient pass and the weight-gradient pass each touch the forward's tensors once) / the replayed step time; the longest
    kernel of an eager step from torch.profiler beside it when the profiler is available."""
    from hyperseg_amd import configs
    from hyperseg_amd.training import BootstrappedCrossEntropyLoss, GraphedTrainStep
    from hyperseg_amd.utils.synthetic import fill_by_name
    torch.set_grad_enabled(True)
    try:
        model = fill_by_name(configs.build('hyperseg-s-camvid'), seed=0).to(dev)
        gen = torch.Generator().manual_seed(99)
        x = torch.rand(2, 3, 576, 576, generator=gen).to(dev)
        with torch.no_grad():
            model.eval()
            feats = model.backbone(x)
            sig = model.weight_mapper(feats[-1]).contiguous()
            pyr = [t.contiguous() for t in [x] + feats[:-1]]
        alg_fwd, _ = decoder_levels(model, 576, 576, 2)
        dec = model.decoder.train()
        target = torch.randint(0, 12, (2, 576, 576), generator=gen).to(dev)
        crit = BootstrappedCrossEntropyLoss(k=4096, thresh=0.3, ignore_index=255)
        from hyperseg_amd.training import Adam as OneLaunchAdam
        res = {'workload': 'HyperSeg-S / CamVid decoder training step, 576x576 crops, batch 2: forward + bootstrapped CE + backward + Adam, '
                           'one HIP graph per step (encoder features and signal resident, as tools/train_step_time.py)',
               'optimizer': 'hyperseg_amd.training.Adam (torch.optim.Adam arithmetic, the whole parameter list in one launch: hs_adam_step); '
                            'the same step with torch.optim.Adam(capturable, fused) is timed beside it as fp32_torch_adam'}
        state0 = {k: v.clone() for k, v in dec.state_dict().items()}
        for mode in ('fp32', 'bf16', 'fp32_torch_adam'):
            dec.load_state_dict(state0)
            if mode == 'fp32_torch_adam':
                opt = torch.optim.Adam(dec.parameters(), lr=torch.tensor(1e-3, device=dev), betas=(0.5, 0.999), capturable=True, fused=True)
            else:
                opt = OneLaunchAdam(dec.parameters(), lr=torch.tensor(1e-3, device=dev), betas=(0.5, 0.999))

            def fwd(p, s_, half=(mode == 'bf16')):
                with torch.autocast('cuda', dtype=torch.bfloat16, enabled=half):
                    return dec(p, s_)
            gs = GraphedTrainStep(fwd, crit, opt, (pyr, sig), target)
            for _ in range(3):
                gs.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                loss, _ = gs.step()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / iters
            res[mode] = {'ms_per_step': round(ms, 4), 'steps_per_s': round(1e3 / ms, 1), 'steps': iters, 'loss_after': round(float(loss), 4),
                         'finite': bool(torch.isfinite(loss))}
            del gs, opt, loss
        res['bf16_speedup_over_fp32'] = round(res['fp32']['ms_per_step'] / res['bf16']['ms_per_step'], 3)
        step_bytes = 3 * alg_fwd
        t_s = res['fp32']['ms_per_step'] * 1e-3
        res['roofline'] = {'bound': 'hbm', 'kernel': 'whole replayed fp32 step (launch-latency-bound chain; no single dominant kernel)',
                           'achieved': round(step_bytes / t_s / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': round(step_bytes / t_s / 1e9 / HBM_PEAK_GBS, 4), 'traffic': None, 'algorithmic_bytes': step_bytes,
                           'note': f'algorithmic bytes = 3 x the forward pass\' {alg_fwd} B (SURVEY 8d definition)'}
        try:                                                  # the longest kernel of one eager fp32 step (kineto / roctracer)
            from torch.profiler import ProfilerActivity, profile
            dec.load_state_dict(state0)
            opt = OneLaunchAdam(dec.parameters(), lr=1e-3, betas=(0.5, 0.999))

            def eager():
                opt.zero_grad(set_to_none=True)
                loss = crit(dec(pyr, sig), target)
                loss.backward()
                opt.step()
            eager()
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(3):
                    eager()
                torch.cuda.synchronize()
            rows = []
            for e in prof.key_averages():
                tot = getattr(e, 'device_time_total', None)
                tot = getattr(e, 'cuda_time_total', 0.0) if tot is None else tot
                if tot and e.count:
                    rows.append((tot / e.count, e.count / 3.0, e.key))
            total = sum(a * c for a, c, _ in rows)
            top = max(rows)
            res['dominant_kernel'] = {'name': top[2][:120], 'avg_us': round(top[0], 2), 'launches_per_step': round(top[1], 1),
                                      'kernel_time_per_step_us': round(total, 1), 'kernels_per_step': round(sum(c for _, c, _ in rows), 1),
                                      'source': 'torch.profiler (device activities), 3 eager fp32 steps'}
        except Exception as e:                                # noqa: BLE001
            res['dominant_kernel'] = {'error': f'{type(e).__name__}: {e}'[:200]}
        return res
    finally:
        torch.set_grad_enabled(False)

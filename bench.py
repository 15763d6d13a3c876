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

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy peak 6290
FP32_PEAK_TFLOPS = 157.3       # f32 vector == f32-input MFMA dense peak
F16_PEAK_TFLOPS = 2500.0       # dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md)
MODELS = {'m': 'hyperseg-m', 's': 'hyperseg-s', 'sc': 'hyperseg-s-camvid', 'l': 'hyperseg-l', 'lc': 'hyperseg-l-camvid'}
LABELS = {'m': 'HyperSeg-M / EfficientNet-B1 / 1024x512', 's': 'HyperSeg-S / EfficientNet-B1 / 1536x768',
          'sc': 'HyperSeg-S / EfficientNet-B1 / CamVid 768x576', 'l': 'HyperSeg-L / EfficientNet-B3 / 512x512',
          'lc': 'HyperSeg-L / EfficientNet-B1 / CamVid 1024x768 (six-level v1_0 decoder)'}


# --------------------------------------------------------------------------------------------- the timed loop
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


# --------------------------------------------------------------------------------------------- decoder accounting
def decoder_levels(model, h, w, batch):
    """Algorithmic HBM bytes and MACs of every decoder level (definition: SURVEY.md section 8d): skips read once, each
    level output written once and read once at its own resolution, banks read once, intermediates 0 B."""
    dec = model.decoder
    fh, fw = h // 32, w // 32
    p = batch * fh * fw
    feat = [3] + model.backbone.feat_channels[:-1]
    levels, prev_c = [], 0
    for l in range(dec.levels):
        blk = getattr(dec, f'level_{l}', None)
        if blk is None:
            blk = dec.level_blocks[l]                       # unify variant
        blk = blk[0]
        first = blk[0] if isinstance(blk, torch.nn.Sequential) else blk
        stride = 32 >> l
        hl, wl = h // stride, w // stride
        skip_c = feat[::-1][l]
        hid = getattr(first, 'hidden_dim', 0)
        if not hid and hasattr(first, 'conv'):              # v0_1 inverted residual: three blocks
            c1, c3 = first.conv[0][0], first.conv[-1][0]
            cin, hid, cout = c1.in_channels, c1.out_channels, c3.out_channels
        elif hid:
            cin, cout = first.in_nc, first.out_nc
        else:
            cin, cout = first.in_channels, first.out_channels
        if hid:
            ph, pw = hl // fh, wl // fw
            halo = (ph + 2) * (pw + 2) if hasattr(first, 'hidden_dim') else ph * pw     # Op C runs pw1 on the halo tile
            macs = p * (halo * cin * hid + ph * pw * (9 * hid + hid * cout))
            hp = cin * hid + 9 * hid + hid * cout
        else:
            macs = batch * hl * wl * cin * cout
            hp = cin * cout
        route = None
        if hid and hasattr(first, 'hidden_dim'):            # Op C: which kernel the level gets under the module's math mode
            import hyperseg_amd.functional as HF
            route = HF.patch_ir_route((batch, hl, wl), skip_c, prev_c, (fh, fw), hid, cout, math=getattr(first, 'ir_math', None))
        levels.append(dict(level=l, cin=cin, cout=cout, hidden=hid, macs=macs, route=route,
                           in_bytes=4 * batch * (skip_c * hl * wl + prev_c * (hl // 2) * (wl // 2)),
                           bank_bytes=4 * p * hp, out_bytes=4 * batch * cout * hl * wl))
        prev_c = cout
    total = sum(lv['in_bytes'] + lv['bank_bytes'] + lv['out_bytes'] for lv in levels)
    if (32 >> (dec.levels - 1)) > 1:                       # v1_0 / unify stop at stride 2: final 2x upsample of the logits
        total += 4 * batch * levels[-1]['cout'] * ((h // 2) * (w // 2) + h * w)
    return total, levels


EVENT_REPS = 8      # identical back-to-back launches per event pair (instrumented_decoder)


def instrumented_decoder(model, x, n_inst):
    """Per-launch durations of the decoder's HIP launches: HIP events on the launch stream around every hyperseg_amd
    functional entry point, n_inst eager decoder passes with the GPU parked so that the host enqueues a whole pass before
    its first launch starts (device time, not host launch gaps).  An event pair costs ~5 us of its own on this stack
    (`event_pair_overhead_us`: a fifth of the dominant launch), so every launch is issued EVENT_REPS times back to back
    between its two events -- same arguments, same result -- and the average is reported: the pair's cost is amortised and
    the figure agrees with rocprofv3's kernel duration to about an inter-kernel gap (profiles/).  The decoder's total
    (`decoder_us`) is taken from separate passes with single launches.  Returns (launches, decoder_us, event_overhead_us)."""
    import hyperseg_amd.functional as HF
    names = ['signal2weights', 'signal2weights_multi', 'bank_pack', 'patch_conv', 'patch_ir', 'patch_ir_v0', 'upsample_bilinear']
    orig = {n: getattr(HF, n) for n in names}
    recs, counter = {}, [0]

    reps = [1]

    def wrap(n):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig[n](*a, **k)
            for _ in range(reps[0] - 1):
                orig[n](*a, **k)
            e1.record()
            if reps[0] > 1:
                recs.setdefault((counter[0], n), []).append((e0, e1))
            counter[0] += 1
            return r
        return f
    feats = model.backbone(x)
    # the chained levels (functional.K1Chain.run -> hs_k1_chain_fwd / hs_decoder_chain_fwd) are one launch of three / four levels
    chain_run = HF.K1Chain.run

    def chain_wrap(self, *a, **k):
        name = 'decoder_chain' if k.get('ir') is not None else 'k1_chain'
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = chain_run(self, *a, **k)
        for _ in range(reps[0] - 1):
            chain_run(self, *a, **k)
        e1.record()
        if r is not None:                         # a refused shape launched nothing: the per-level launches follow and are recorded
            if reps[0] > 1:
                recs.setdefault((counter[0], name), []).append((e0, e1))
            counter[0] += 1
        return r
    try:
        for n in names:
            setattr(HF, n, wrap(n))
        HF.K1Chain.run = chain_wrap
        dec_evs = []
        for it in range(2 * n_inst):
            reps[0] = 1 if it < n_inst else EVENT_REPS          # first half: the decoder's own duration; second half: per-launch averages
            counter[0] = 0
            head = model.weight_mapper(feats[-1])
            head = head.contiguous() if isinstance(head, torch.Tensor) else head
            pyr = [t.contiguous() for t in [x] + feats[:-1]]
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(1_000_000 * reps[0])              # long enough for the host to enqueue the whole pass behind it
            first = counter[0]
            d0.record()
            model.decoder(pyr, head)
            d1.record()
            if reps[0] == 1:
                dec_evs.append((d0, d1, first))
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(HF, n, orig[n])
        HF.K1Chain.run = chain_run
    cal = []
    for _ in range(50):                      # an empty event pair on a busy stream is not 0: calibrate and report it
        torch.cuda._sleep(200_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        cal.append((e0, e1))
    torch.cuda.synchronize()
    cal = sorted(a.elapsed_time(b) * 1e3 for a, b in cal)
    ev_overhead = cal[len(cal) // 2]
    first_dec = dec_evs[0][2]
    launches = []
    for (i, n), evs in sorted(recs.items()):
        ts = [a.elapsed_time(b) * 1e3 / EVENT_REPS for a, b in evs]
        avg = sum(ts) / len(ts)
        launches.append(dict(idx=i, kernel='hs_' + n + '_fwd', in_decoder=i >= first_dec, avg_us=round(avg, 2),
                             minus_event_overhead_us=round(max(avg - ev_overhead / EVENT_REPS, 0.0), 2), launches_per_event_pair=EVENT_REPS))
    dec_us = sum(a.elapsed_time(b) for a, b, _ in dec_evs) * 1e3 / len(dec_evs)
    return launches, dec_us, ev_overhead


def roofline_of(launches, levels, h, w, batch, traffic_dir):
    """The dominant decoder launch against the roof that binds it."""
    spans = {'hs_patch_conv_fwd': 1, 'hs_patch_ir_fwd': 1, 'hs_patch_ir_v0_fwd': 1, 'hs_k1_chain_fwd': 3, 'hs_decoder_chain_fwd': 4}
    conv = [l for l in launches if l['kernel'] in spans and l['in_decoder']]
    per = {}
    # every level fused into a launch of its own or into the chain launch (levels 0-2 / 0-3): attribute levels to launches in order;
    # otherwise (a level split over several launches) no per-level attribution
    if sum(spans[l['kernel']] for l in conv) == len(levels):
        at = 0
        for l in conv:
            n = spans[l['kernel']]
            if n == 1:
                per[l['idx']] = levels[at]
            else:                                   # the chain: the levels' bytes and multiply-adds together
                grp = levels[at:at + n]
                per[l['idx']] = dict(level='-'.join(str(g['level']) for g in grp), cin=grp[0]['cin'], cout=grp[-1]['cout'], hidden=0, route=None,
                                     macs=sum(g['macs'] for g in grp), in_bytes=sum(g['in_bytes'] for g in grp),
                                     bank_bytes=sum(g['bank_bytes'] for g in grp), out_bytes=sum(g['out_bytes'] for g in grp))
            at += n
    dom = max([l for l in launches if l['in_decoder']], key=lambda l: l['avg_us'])
    t_s = dom['avg_us'] * 1e-6
    traffic = pmc_traffic(traffic_dir, dom['kernel'])
    lv = per.get(dom['idx'])
    if lv is not None and lv['hidden']:
        flops = 2.0 * lv['macs']
        kbytes = lv['in_bytes'] + lv['bank_bytes'] + lv['out_bytes']
        t_fl, t_by = flops / (FP32_PEAK_TFLOPS * 1e12), kbytes / (HBM_PEAK_GBS * 1e9)
        if lv.get('route') == 'split_mfma':
            # The f16-split kernel issues 3 f16 products per f32 product on v_mfma_f32_16x16x32_f16: its matrix-core roof is
            # 3 x flops at the f16 peak, which the launch's HBM time exceeds -- the roof that binds it is HBM.
            t_f16 = 3.0 * flops / (F16_PEAK_TFLOPS * 1e12)
            if t_by >= t_f16:
                return {'bound': 'hbm', 'kernel': f"{dom['kernel']} (level {lv['level']}: {lv['cin']}->{lv['hidden']}->{lv['cout']} ch, "
                                                  'f16-split matrix-core form hs_patch_irc.hip)',
                        'achieved': round(kbytes / t_s / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(kbytes / t_s / 1e9 / HBM_PEAK_GBS, 4), 'traffic': traffic, 'avg_launch_us': dom['avg_us'],
                        'algorithmic_bytes': kbytes, 'algorithmic_flops': flops,
                        'f32_mfma_frac': round(flops / t_s / 1e12 / FP32_PEAK_TFLOPS, 4),
                        'f16_mfma_frac': round(3.0 * flops / t_s / 1e12 / F16_PEAK_TFLOPS, 4),
                        'note': f'roofs of this launch: HBM {t_by * 1e6:.1f} us at 8 TB/s (binding), f16 matrix cores {t_f16 * 1e6:.1f} us '
                                f'(3 products per f32 product), f32 matrix cores {t_fl * 1e6:.1f} us (what the exact-f32 form would '
                                'need; reported as f32_mfma_frac for comparison with rounds 1-2)'}
        if t_fl >= t_by:
            return {'bound': 'mfma', 'kernel': f"{dom['kernel']} (level {lv['level']}: {lv['cin']}->{lv['hidden']}->{lv['cout']} ch)",
                    'achieved': round(flops / t_s / 1e12, 3), 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(flops / t_s / 1e12 / FP32_PEAK_TFLOPS, 4), 'traffic': traffic,
                    'avg_launch_us': dom['avg_us'], 'algorithmic_flops': flops, 'algorithmic_bytes': kbytes,
                    'hbm_frac': round(kbytes / t_s / 1e9 / HBM_PEAK_GBS, 4),
                    'note': 'priced against the f32 peak: f32 vector peak == f32-input MFMA dense peak (157.3 TF/s); the launch needs '
                            f'{t_fl * 1e6:.1f} us at that peak and {t_by * 1e6:.1f} us at the 8 TB/s HBM peak'}
    if lv is not None:
        kbytes = lv['in_bytes'] + lv['bank_bytes'] + lv['out_bytes']
    elif dom['kernel'] in ('hs_signal2weights_multi_fwd', 'hs_signal2weights_fwd'):
        kbytes = sum(x['bank_bytes'] for x in levels)
    elif dom['kernel'] == 'hs_upsample_bilinear_fwd':
        kbytes = 4 * batch * levels[-1]['cout'] * ((h // 2) * (w // 2) + h * w)
    else:
        kbytes = 0
    return {'bound': 'hbm', 'kernel': dom['kernel'], 'achieved': round(kbytes / t_s / 1e9, 1), 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': round(kbytes / t_s / 1e9 / HBM_PEAK_GBS, 4), 'traffic': traffic,
            'avg_launch_us': dom['avg_us'], 'algorithmic_bytes': kbytes}


def pmc_traffic(traffic_dir, kernel):
    """HBM bytes per launch of the dominant kernel from rocprofv3 --pmc passes of THIS command made in the same session
    (FETCH_SIZE and WRITE_SIZE in separate passes; KB units; FETCH_SIZE doubled: gfx950 tallies 128-B reads at 64 B --
    MI355X_MICROARCH.md, HBM section).  None when no such passes were handed over: never a stored constant."""
    if not traffic_dir:
        return None
    import csv
    import glob
    stem = {'hs_patch_ir_fwd': 'patch_ir', 'hs_patch_ir_v0_fwd': 'patch_ir',
            'hs_patch_conv_fwd': 'patch_conv', 'hs_upsample_bilinear_fwd': 'upsample2x_kernel',
            'hs_signal2weights_multi_fwd': 'signal2weights'}.get(kernel)
    if stem is None:
        return None
    acc = {}
    for f in glob.glob(os.path.join(traffic_dir, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if stem in r['Kernel_Name'] and r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                a = acc.setdefault((r['Kernel_Name'], r['Counter_Name']), [0, 0.0])
                a[0] += 1
                a[1] += float(r['Counter_Value'])
    best = None
    for (kname, cname), (n, v) in acc.items():          # the instantiation with the most bytes = the dominant level
        other = acc.get((kname, 'WRITE_SIZE' if cname == 'FETCH_SIZE' else 'FETCH_SIZE'))
        if cname == 'FETCH_SIZE' and other:
            tot = int((2 * v / n + other[1] / other[0]) * 1024)
            best = tot if best is None else max(best, tot)
    return best


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


# --------------------------------------------------------------------------------------------- other BASELINE configs, side objects
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


# --------------------------------------------------------------------------------------------- main
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


def self_traffic_passes(model_key, timeout_s=170):
    """--traffic auto: the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) of THIS command,
    spawned before this process touches the GPU and outside every timed region; returns (directory | None, note).
    Counter collection only -- '--pmc X --kernel-trace', never with a sys / hip / hsa trace domain."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None, 'rocprofv3 not on PATH'
    root = tempfile.mkdtemp(prefix='hs_traffic_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp', HS_BENCH_CHILD='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        cmd = [exe, '--pmc', c, '--kernel-trace', '--output-format', 'csv', '-d', os.path.join(root, c), '--',
               sys.executable, os.path.join(REPO, 'bench.py'), '--model', model_key, '--no-extras', '--steps', '10', '--warmup', '3',
               '--repeats', '1', '--no-graph', '--traffic', 'off']
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return None, f'rocprofv3 --pmc {c} pass timed out after {timeout_s} s'
        if r.returncode != 0:
            return None, f'rocprofv3 --pmc {c} pass exited {r.returncode}: ' + r.stderr.decode(errors='replace')[-200:]
    return root, 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (eager launches, 10 steps), made by this run'


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


class Legs:
    """Wall seconds per leg of a run (`legs_s` on the line: where a default run's minutes go -- the timed regions are milliseconds)."""

    def __init__(self):
        self.t, self.out = time.perf_counter(), {}

    def mark(self, name):
        now = time.perf_counter()
        self.out[name] = round(self.out.get(name, 0.0) + now - self.t, 1)
        self.t = now


def decoder_launches_text(model, launches=None):
    """What the decoder's launches WERE: from the instrumented table when there is one, else from what the warm-up forwards left behind
    (a K1Chain that has launched = levels 0-2 went out as one launch) -- never from a flag (HyperSeg-L's line said "chain")."""
    dec = getattr(model, 'decoder', None)
    if dec is None:
        return None
    if launches:
        names = [l['kernel'] for l in launches if l['in_decoder']]
        return ' | '.join(names) + f' ({len(names)} launches, in issue order)'
    kc = getattr(dec, '_k1_chain', None)
    chained = kc is not None and bool(kc._ws)
    n = dec.levels
    head = 'signal2weights (one launch for every level) | ' if type(dec).__module__.split('.')[-1] != 'hyperseg_v0_1' else ''
    tail = ' | final 2x upsample' if (32 >> (n - 1)) > 1 else ''
    if chained:
        return head + f'levels 0-2 as one launch (hs_k1_chain_fwd: in-launch neighbour hand-offs) | one launch per level 3..{n - 1}' + tail
    return head + f'one launch per level 0..{n - 1}' + tail


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
           '--master-port', str(port), os.path.abspath(__file__), *argv]
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
                                  'hyperseg_amd.utils.inference.prepare_for_inference: MBConv blocks = hs_mbconv_expand_dw_fwd | '
                                  '1x1 GEMM + hs_depthwise_conv_fwd, hs_se_gate_fwd, 1x1 GEMM; hs_stem_conv_fwd; '
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

"""The decoder under the bench's instruments: algorithmic bytes / MACs per level, event-timed launch table, the dominant launch's roofline, the rocprofv3 --pmc traffic passes.

Part of bench.py's measurement harness (round 6: bench.py was one 1 100-line file running ten legs; the legs live here, bench.py is the
driver entry).  Nothing in this package imports oracle/: the CPU-baseline leg, the only one that may, stays in bench.py."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # the repository root: bench.py lives there

from .constants import F16_PEAK_TFLOPS, FP32_PEAK_TFLOPS, HBM_PEAK_GBS


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

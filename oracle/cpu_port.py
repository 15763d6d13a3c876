"""CPU *port* of the reference decoder used ONLY as the timed ``cpu_baseline`` of bench.py.  TEST INFRASTRUCTURE.

oracle/hyperseg_oracle.py is written for legibility (explicit gathers + einsums) and is ~10x slower on a CPU
than what the reference actually executes.  To time "the reference's CPU path" fairly next to the GPU number,
this file restates the reference's own ATen op sequence for HyperSeg v1.0 -- grouped 1x1 conv for
signal2weights, batch-of-patches folded into conv groups, reflect pad + unfold, F.batch_norm, ReLU/ReLU6,
F.interpolate + torch.cat glue (hyperseg_v1_0.py:221-253, 328-370, 486-498) -- as plain functions over a
state dict.  tests/test_oracle_golden.py::test_cpu_port_matches_oracle pins it to the oracle.
Parity status: pinned through the oracle (itself pinned to reference-made fixtures).
"""
import torch
import torch.nn.functional as F

from .hyperseg_oracle import BN_EPS, image_coords


def _bn(x, params, prefix):
    return F.batch_norm(x, params[f'{prefix}.running_mean'], params[f'{prefix}.running_var'],
                        params[f'{prefix}.weight'], params[f'{prefix}.bias'], False, 0.0, BN_EPS)


def _weights(s, w, sw, hp):
    sl = s[:, sw['signal_index']:sw['signal_index'] + sw['signal_channels']]
    return F.conv2d(sl, w, None, groups=sw['groups'])[:, :hp]


def _level_k1(x, wt, cout):
    b, c, h, w = x.shape
    fh, fw = wt.shape[-2:]
    ph, pw = h // fh, w // fw
    wt = wt.permute(0, 2, 3, 1).reshape(b * fh * fw * cout, c, 1, 1)
    x = x.view(b, c, fh, ph, fw, pw).permute(0, 2, 4, 1, 3, 5).reshape(1, -1, ph, pw)
    x = F.conv2d(x, wt, None, groups=b * fh * fw)
    return x.view(b, fh, fw, -1, ph, pw).permute(0, 3, 1, 4, 2, 5).reshape(b, -1, h, w)


def _level_ir(x, wt, hid, cout, params, prefix):
    b, c, h, w = x.shape
    fh, fw = wt.shape[-2:]
    ph, pw = h // fh, w // fw
    p = b * fh * fw
    x = F.pad(x, (1, 1, 1, 1), mode='reflect')
    x = x.permute(0, 2, 3, 1).unfold(1, ph + 2, ph).unfold(2, pw + 2, pw).reshape(1, -1, ph + 2, pw + 2)
    wt = wt.permute(0, 2, 3, 1).reshape(p, -1)
    r1, r2 = c * hid, c * hid + 9 * hid
    x = F.conv2d(x, wt[:, :r1].reshape(p * hid, c, 1, 1), None, groups=p)
    x = F.relu6(_bn(x.view(p, hid, ph + 2, pw + 2), params, f'{prefix}.bn1')).view(1, -1, ph + 2, pw + 2)
    x = F.conv2d(x, wt[:, r1:r2].reshape(p * hid, 1, 3, 3), None, groups=p * hid)
    x = F.relu6(_bn(x.view(p, hid, ph, pw), params, f'{prefix}.bn2')).view(1, -1, ph, pw)
    x = F.conv2d(x, wt[:, r2:].reshape(p * cout, hid, 1, 1), None, groups=p)
    x = _bn(x.view(p, cout, ph, pw), params, f'{prefix}.bn3')
    return x.view(b, fh, fw, cout, ph, pw).permute(0, 3, 1, 4, 2, 5).reshape(b, cout, h, w)


def decoder_v1_0(plan, params, x, s):
    p = None
    for l, lv in enumerate(plan['levels']):
        skip = x[-l - 1]
        if p is not None:
            p = F.interpolate(p, skip.shape[2:], mode='bilinear', align_corners=False)
            skip = torch.cat((skip, p), dim=1)
        b, _, h, w = skip.shape
        inp = torch.cat([image_coords(h, w).unsqueeze(0).expand(b, -1, -1, -1), skip], dim=1)
        sw = plan['s2w'][l]
        if lv['k'] == 1:
            wt = _weights(s, params[f'level_{l}.0.0.signal2weights.weight'], sw, lv['hp'])
            p = F.relu(_bn(_level_k1(inp, wt, lv['cout']), params, f'level_{l}.0.1'))
        else:
            wt = _weights(s, params[f'level_{l}.0.signal2weights.weight'], sw, lv['hp'])
            p = _level_ir(inp, wt, lv['hidden'], lv['cout'], params, f'level_{l}.0')
            if lv['cin'] == lv['cout']:
                p = p + inp
    if p.shape[2:] != x[0].shape[2:]:
        p = F.interpolate(p, x[0].shape[2:], mode='bilinear', align_corners=False)
    return p

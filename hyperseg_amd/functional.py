"""Tensor-level entry points of the HIP decoder kernels (thin wrappers over the C ABI).

Vocabulary: a *bank* is the patch-major per-patch filter bank
``bank[p, m]`` (p = (b*fh + i)*fw + j); a *stage input* is the lazy channel concatenation
[coords | skip | resized previous level] that the reference materialises with torch.cat
(hyperseg_v1_0.py:231-240) and the kernels generate on the fly.
"""
import ctypes as C
import functools
import os
import threading

import torch

from . import _hip
from ._hip import ACT_NONE, ACT_RELU, ACT_RELU6, PAD_MODES  # noqa: F401  (re-exported)

BN_EPS_DEFAULT = 1e-5
S2W_TRAIN_MAX_LAYERS = 8     # S2W_MAX_LAYERS of csrc/hs_s2w_blocked.h: layers one hs_s2w_train_* / hs_signal2weights_multi_fwd launch takes
# Late-level banks on a second stream: saves ~14 us of decoder time in isolation, but a forked/joined capture makes
# the whole-model HIP-graph replay 0.37 ms SLOWER on ROCm 7.2 (measured: 3.62 -> 3.99 ms/frame), so it is off by default.
USE_SIDE_STREAM = os.environ.get('HS_SIDE_STREAM', '0') == '1'
# Round 4: the finer fork -- level 0's bank on the current stream, every later level's bank as its own launch on the side stream
# with one event each, so level l waits for ITS bank only and signal2weights overlaps the latency-bound k = 1 levels
# (HS_SIDE_STREAM=2; measured by tools/gpu_r4c.sh, decision in DESIGN section 3.1).
PIPELINE_BANKS = os.environ.get('HS_SIDE_STREAM', '0') == '2'
# Round 4: the banks of the later levels produced INSIDE the k = 1 levels' launches (CoScheduledBanks / hs_patch_conv_s2w_fwd).
# MEASURED AND OFF (visit r4e, profiles/round4_coscheduled_banks_ab.txt): the carrying launches grow by what the riders take -- level 2
# with bank 4 aboard 7.3 -> 15.2 us, kernel sum of (signal2weights + levels 0-2) 37.5 -> 39.4 us, replayed decoder 0.086 -> 0.087 ms.
# signal2weights is a per-workgroup latency chain too, not idle-CU filler.  HS_COSCHEDULE_BANKS=1 switches it on.
COSCHEDULE_BANKS = os.environ.get('HS_COSCHEDULE_BANKS', '0') == '1'


def _round_up(n, m):
    return (n + m - 1) // m * m


def _device_of(a):
    if isinstance(a, torch.Tensor):
        return a.device
    if isinstance(a, (StageInput, BankRef, SignalRef)):
        return a.device
    return None


def _on_operand_device(fn):
    """Every launch runs with the HIP device of its first tensor operand current: the stream handed to the library
    (``torch.cuda.current_stream()``) then belongs to the device that owns the pointers, also when the caller's current
    device is another one (nn.DataParallel replicas, a model placed on cuda:1 from a cuda:0 context)."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = next((d for d in map(_device_of, args) if d is not None), None)
        if dev is None or dev.type != 'cuda' or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


class StageInput:
    """Lazy ``cat([coords(2)?, skip, resize(prev)?], dim=1)`` consumed by the stage kernels' prologue."""

    def __init__(self, skip, prev=None, coords=False):
        if skip.dim() != 4:
            raise ValueError('skip must be (B, C, H, W)')
        if prev is not None and (prev.dim() != 4 or prev.shape[0] != skip.shape[0]):
            raise ValueError('prev must be (B, C, h, w) with the same batch as skip')
        if prev is not None and prev.device != skip.device:
            raise ValueError(f'stage input spans two devices ({skip.device}, {prev.device})')
        self.skip, self.prev, self.coords = skip, prev, bool(coords)

    @property
    def channels(self):
        return 2 * self.coords + self.skip.shape[1] + (self.prev.shape[1] if self.prev is not None else 0)

    @property
    def shape(self):
        b, _, h, w = self.skip.shape
        return torch.Size((b, self.channels, h, w))

    @property
    def device(self):
        return self.skip.device

    def c_struct(self, prev_dtype=torch.float32):
        b, cs, h, w = self.skip.shape
        st = _hip.StageInputC()
        st.skip = _hip.dev_ptr(self.skip, 'stage input (skip)')
        st.batch, st.H, st.W, st.c_skip = b, h, w, cs
        st.coords = int(self.coords)
        if self.prev is not None:
            st.prev = _hip.dev_ptr(self.prev, 'stage input (prev)', prev_dtype)
            st.c_prev, st.Hp, st.Wp = self.prev.shape[1], self.prev.shape[2], self.prev.shape[3]
            st.prev_mode = _hip.PREV_SAME if self.prev.shape[2:] == self.skip.shape[2:] else _hip.PREV_BILINEAR
        else:
            st.prev, st.c_prev, st.Hp, st.Wp, st.prev_mode = None, 0, 0, 0, _hip.PREV_NONE
        return st

    def materialize(self, dtype=torch.float32):
        """The concatenated tensor itself (diagnostics / modules that cannot consume a lazy input; the training path, where
        ``dtype`` = torch.bfloat16 under autocast and the previous level may be stored as bf16: hs_stage_input_typed_fwd)."""
        codes = {torch.float32: 0, torch.bfloat16: 1}
        pdt = self.prev.dtype if self.prev is not None else torch.float32
        if dtype not in codes or pdt not in codes:
            raise NotImplementedError(f'stage input: storage types {pdt} -> {dtype} (supported: float32, bfloat16)')
        st = self.c_struct(pdt)
        y = torch.empty(self.shape, device=self.device, dtype=dtype)
        with _hip.device_scope(self.device):
            _hip.check(_hip.lib.hs_stage_input_typed_fwd(C.byref(st), codes[pdt], codes[dtype], y.data_ptr(), _hip.stream_ptr()),
                       'hs_stage_input_typed_fwd')
        return y


def _channel_view(t, name):
    """(device pointer, channels of the underlying buffer) of a (B, C, h, w) tensor that is contiguous
    or a channel-range view ``base[:, a:b]`` of a contiguous tensor (what MetaSequential passes)."""
    if t.dim() != 4:
        raise ValueError(f'{name} must be 4-D (B, C, h, w), got {tuple(t.shape)}')
    if not t.is_cuda or t.dtype != torch.float32:
        _hip.dev_ptr(t, name)          # raises with the proper message
    b, c, h, w = t.shape
    sb, sc, sh, sw = t.stride()
    plane = h * w
    ok = (sw == 1 or w == 1) and (sh == w or h == 1) and (sc == plane or c == 1) and \
         (b == 1 or (sb % plane == 0 and sb // plane >= c))
    if not ok:
        t = t.contiguous()
        return t, t.data_ptr(), c
    return t, t.data_ptr(), (sb // plane if b > 1 else c)


def as_stage(x):
    return x if isinstance(x, StageInput) else StageInput(x)


def _epilogue(scale=None, shift=None, act=ACT_NONE):
    ep = _hip.EpilogueC()
    ep.scale = _hip.dev_ptr(scale, 'epilogue scale') if scale is not None else None
    ep.shift = _hip.dev_ptr(shift, 'epilogue shift') if shift is not None else None
    ep.act = int(act)
    return ep


@_on_operand_device
def signal2weights(signal, wsw_t, signal_index, signal_channels, groups, rows, out=None):
    """Grouped 1x1 conv of the signal slice -> patch-major bank (B*fh*fw, ld).  ``wsw_t`` is the
    Conv2d weight transposed to (Cs/G, Wc).  Replaces signal2weights(...)[:, :hp] + permute copy."""
    b, c_view, fh, fw = signal.shape
    if signal_index + signal_channels > c_view:
        raise ValueError(f'signal slice [{signal_index}, {signal_index + signal_channels}) exceeds the '
                         f'{c_view} channels handed to the module')
    signal, sig_ptr, c_signal = _channel_view(signal, 'signal')
    wc = wsw_t.shape[1]
    ld = _round_up(rows, 4)
    if out is None:
        out = torch.empty(b * fh * fw, ld, device=signal.device, dtype=torch.float32)
    st = _hip.lib.hs_signal2weights_fwd(
        sig_ptr, b, c_signal, fh, fw, signal_index, signal_channels, groups,
        _hip.dev_ptr(wsw_t, 'wsw_t'), wc, rows,
        _hip.dev_ptr(out, 'bank'), out.stride(0), _hip.stream_ptr())
    _hip.check(st, 'hs_signal2weights_fwd')
    return out


class BankRef:
    """A bank that already exists (produced by :func:`signal2weights_multi` for a whole decoder / context head).  It
    stands in for the signal / weight tensor on its way through MetaSequential to the module that consumes it:
    ``ref[:, a:b]`` (MetaSequential's channel range) is the bank's column range [a, b), clamped like a tensor slice."""

    def __init__(self, bank, batch, rows, grid):
        self.bank, self.rows, self.grid = bank, rows, tuple(grid)
        self.shape = torch.Size((batch, rows) + self.grid)
        self.requires_grad = False

    @property
    def device(self):
        return self.bank.device

    def __getitem__(self, idx):
        if not (isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None) and isinstance(idx[1], slice)
                and idx[1].step in (None, 1)):
            raise IndexError('a filter bank supports only the channel-range slice ref[:, a:b]')
        a, b, _ = idx[1].indices(self.rows)
        if a == 0 and b == self.rows:
            return self
        return BankRef(self.bank[:, a:max(a, b)], self.shape[0], max(b - a, 0), self.grid)

    def dim(self):
        return 4


class TrainBank:
    """A level's bank on the TRAINING path: the patch-major (P, ld) tensor autograd.S2WBanksTrain produced (it carries the graph
    back to the signal and to the level's signal2weights weight).  Travels through MetaSequential like a BankRef."""

    def __init__(self, bank, batch, rows, grid):
        self.bank, self.rows, self.grid = bank, rows, tuple(grid)
        self.shape = torch.Size((batch, rows) + self.grid)
        self.requires_grad = True

    @property
    def device(self):
        return self.bank.device

    def __getitem__(self, idx):
        if not (isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None) and isinstance(idx[1], slice) and idx[1].step in (None, 1)):
            raise IndexError('a filter bank supports only the channel-range slice ref[:, a:b]')
        a, b, _ = idx[1].indices(self.rows)
        if a == 0 and b == self.rows:
            return self
        return TrainBank(self.bank[:, a:max(a, b)], self.shape[0], max(b - a, 0), self.grid)

    def dim(self):
        return 4


# Levels whose patches are at most this many pixels generate their bank inside the consumer (hs_patch_conv_gen_fwd) instead
# of reading one that hs_signal2weights_multi_fwd wrote.  OFF by default (0): measured on MI355X (profiles/round2_bank_in_
# consumer_ab.txt) it removes 2 x 18.4 MB of HBM traffic per HyperSeg-M frame but each fused launch takes 21.8 us against
# 8.8 us + its ~2 us share of the bank launch -- these levels are latency-, not bandwidth-bound, and the fused kernel
# serialises weight / signal / input round trips that the two-launch route overlaps across its 512 workgroups.
# Set HS_BANK_IN_CONSUMER_MAX_PIXELS=64 to switch it on.
BANK_IN_CONSUMER_MAX_PIXELS = int(os.environ.get('HS_BANK_IN_CONSUMER_MAX_PIXELS', '0'))


class SignalRef:
    """Stands in for a filter bank that is NOT materialised: the signal plus the signal2weights layer that would produce
    the bank (wsw_t, signal_index, signal_channels, groups, rows).  The consuming module hands both to
    :func:`patch_conv_gen`, which generates the bank rows it needs in LDS."""

    def __init__(self, signal, layer):
        self.signal, self.layer = signal, layer
        b, _, fh, fw = signal.shape
        self.shape = torch.Size((b, layer['rows'], fh, fw))
        self.requires_grad = False

    @property
    def device(self):
        return self.signal.device

    def __getitem__(self, idx):
        return self            # MetaSequential's channel range of a level with one signal-fed module: the whole bank

    def dim(self):
        return 4


_S2W_BLK = {}        # (wsw_t data_ptr, version, shape, channels, groups) -> (weakref to wsw_t, packed operand image of the blocked kernel)
# Re-entrancy (SURVEY 8b "Threading": nn.DataParallel.parallel_apply runs one Python thread per replica through these functions):
# every process-global cache is read and written under this lock -- the sweep below iterates the dict, which another thread's
# insert would break ("dictionary changed size during iteration") -- and the per-module caches (FoldedBN, TransposedS2W) keep one
# immutable (key, value) entry per DEVICE, replaced atomically, because DataParallel's replicas share those objects by reference.
_CACHE_LOCK = threading.RLock()


def publish_ready(device):
    """Called after the kernels that FILL a cache entry were launched and before the entry is published: the filling stream is drained
    on the host, so a thread that finds the entry from ANOTHER stream (a second request stream, a replica thread) reads finished data
    -- stream order only protects the stream that launched the fill.  Fills happen once per parameter version, so the wait is paid
    once; under stream capture a host wait is illegal and the entry is only ever consumed by the capturing stream's graph."""
    if device.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(device).synchronize()


def producer_stream(device):
    """Identity of the stream a cache entry is being filled on (kept with the entry; see :func:`adopt`)."""
    return torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0


def adopt(value, device, filled_on):
    """A cache hit from ANOTHER stream than the one the entry was filled on: tell the caching allocator that the entry's tensors are
    in use on the caller's stream (``Tensor.record_stream``).  Without it the memory goes back to the FILLING stream's pool the moment
    the entry is replaced (a new parameter version) and the last Python reference dies -- while kernels the other stream launched on
    it may still be in flight -- and the filling stream's next allocation overwrites what they read.  (Found by
    tests/test_hip_parity.py::test_two_python_threads_two_streams_through_the_decoder: 23 of 40 outputs of one thread differed.)
    Returns ``value``."""
    if device.type == 'cuda':
        cur = torch.cuda.current_stream(device)
        if cur.cuda_stream != filled_on:
            for t in (value if isinstance(value, (tuple, list)) else (value,)):
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
    return value


@_on_operand_device
def s2w_packed(wsw_t, signal_channels, groups):
    """The transposed signal2weights weight re-laid for ``hs_signal2weights_multi_fwd``'s blocked form (hs_s2w_pack_fwd), built
    once per parameter version (the transposed tensor itself is cached per module by version, so its address is the key)."""
    import weakref
    key = (_WEIGHTS_EPOCH[0], wsw_t.data_ptr(), wsw_t._version, tuple(wsw_t.shape), signal_channels, groups, wsw_t.device)
    with _CACHE_LOCK:
        ent = _S2W_BLK.get(key)
        hit = ent[1] if ent is not None and ent[0]() is wsw_t else None       # the address alone could be a dead tensor's, reused
        if hit is not None:
            adopt(hit, wsw_t.device, ent[2])
        if hit is None:
            for k in [k for k, (r, _, _) in list(_S2W_BLK.items()) if r() is None or k[0] != _WEIGHTS_EPOCH[0]]:      # dead sources and earlier weight epochs take their images with them
                del _S2W_BLK[k]
            n = _hip.lib.hs_s2w_pack_floats(signal_channels, groups, wsw_t.shape[1])
            if n < 0:
                _hip.check(int(n), 'hs_s2w_pack_floats')
            hit = torch.empty(n, device=wsw_t.device, dtype=torch.float32)
            _hip.check(_hip.lib.hs_s2w_pack_fwd(_hip.dev_ptr(wsw_t, 'wsw_t'), signal_channels, groups, wsw_t.shape[1], hit.data_ptr(),
                                                _hip.stream_ptr()), 'hs_s2w_pack_fwd')
            publish_ready(wsw_t.device)
            _S2W_BLK[key] = (weakref.ref(wsw_t), hit, producer_stream(wsw_t.device))
    return hit


S2W_BLOCKED = os.environ.get('HS_S2W_BLOCKED', '1') == '1'     # dev A/B switch: 0 = the direct (round-2) kernel


@_on_operand_device
def signal2weights_multi(signal, layers, buf=None):
    """All signal2weights layers of a decoder in ONE launch.  ``layers``: list of dicts with wsw_t, signal_index,
    signal_channels, groups, rows.  Returns one BankRef per layer (views of one buffer; ``buf``: the caller's, at least
    ``bank_floats(signal, layers)`` floats -- for callers that issue the launch on another stream than the one that owns the memory)."""
    b, _, fh, fw = signal.shape
    signal, sig_ptr, c_signal = _channel_view(signal, 'signal')
    arr, refs = _s2w_layer_table(signal, layers, buf)
    st = _hip.lib.hs_signal2weights_multi_fwd(sig_ptr, b, c_signal, fh, fw, arr, len(layers), _hip.stream_ptr())
    _hip.check(st, 'hs_signal2weights_multi_fwd')
    return refs


def _s2w_layer_table(signal, layers, buf=None):
    """The hs_s2w_layer table of ``layers`` over one bank buffer (allocated here unless given) + one BankRef per layer."""
    b, c_view, fh, fw = signal.shape
    p = b * fh * fw
    lds = [_round_up(l['rows'], 4) for l in layers]
    if buf is None:
        buf = torch.empty(p * sum(lds), device=signal.device, dtype=torch.float32)
    elif buf.numel() < p * sum(lds) or buf.dtype != torch.float32 or not buf.is_contiguous():
        raise ValueError('signal2weights_multi: buf is too small / not contiguous fp32')
    arr = (_hip.S2wLayerC * len(layers))()
    refs, off = [], 0
    # The packed weight images are owned by the cache, and the cache drops an entry as soon as the weight epoch moves (another
    # thread's BatchNorm update / mode switch, between two layers of this very loop): every image whose ADDRESS goes into the table
    # is kept alive by the table's first BankRef until the banks themselves die -- i.e. past the launch.  (Round 5's thread test: an
    # image freed between two s2w_packed calls was overwritten by the next layer's pack; wrong banks, 50 % of the logits' scale.)
    keep = []
    for i, (l, ld) in enumerate(zip(layers, lds)):
        if l['signal_index'] + l['signal_channels'] > c_view:
            raise ValueError('signal slice out of range')
        bank = buf[off:off + p * ld].view(p, ld)
        off += p * ld
        a = arr[i]
        a.signal_index, a.signal_channels, a.groups = l['signal_index'], l['signal_channels'], l['groups']
        a.wsw_t, a.wc = _hip.dev_ptr(l['wsw_t'], 'wsw_t'), l['wsw_t'].shape[1]
        a.rows, a.bank, a.ld = l['rows'], bank.data_ptr(), ld
        if S2W_BLOCKED:
            keep.append(s2w_packed(l['wsw_t'], l['signal_channels'], l['groups']))
            a.wsw_blk = keep[-1].data_ptr()
        else:
            a.wsw_blk = None
        keep.append(l['wsw_t'])
        refs.append(BankRef(bank, b, l['rows'], (fh, fw)))
    for r in refs:
        r._operands = keep
    return arr, refs


class CoScheduledBanks:
    """``with CoScheduledBanks(signal, layers) as co:`` -- the NEXT eligible k = 1 :func:`patch_conv` issued inside the block carries
    the signal2weights work of ``layers`` in its own launch (hs_patch_conv_s2w_fwd: a heterogeneous launch, the bank producer's
    blocks filling the idle CUs of a latency-bound k = 1 level).  On exit ``co.refs`` holds one BankRef per layer either way:
    if no launch took the work (or the library said HS_ERR_UNSUPPORTED) it is issued as its own launch then."""
    _active = None

    def __init__(self, signal, layers):
        self.signal, self.layers, self.refs, self.carried = signal, list(layers), None, False

    def __enter__(self):
        if CoScheduledBanks._active is not None:
            raise RuntimeError('CoScheduledBanks blocks do not nest')
        CoScheduledBanks._active = self if self.layers else None
        return self

    def __exit__(self, *exc):
        CoScheduledBanks._active = None
        if self.refs is None and exc[0] is None:
            self.refs = signal2weights_multi(self.signal, self.layers) if self.layers else []
        return False


def bank_floats(signal, layers):
    """Floats of the buffer :func:`signal2weights_multi` needs for ``layers``."""
    b, _, fh, fw = signal.shape
    return b * fh * fw * sum(_round_up(l['rows'], 4) for l in layers)


class SideStream:
    """A second HIP stream per device for work that is independent of the decoder's level-to-level chain (the banks
    of the late levels): forked from / joined to the current stream with events, so it is captured as a parallel
    branch when the forward runs under HIP-graph capture."""
    _streams = {}

    @classmethod
    def get(cls, device):
        key = (device.type, device.index)
        with _CACHE_LOCK:
            if key not in cls._streams:
                cls._streams[key] = torch.cuda.Stream(device=device)
            return cls._streams[key]


@_on_operand_device
def bank_pack(w, ch_offset, rows, out=None):
    """(B, hp_total, fh, fw) channel-major weights -> patch-major bank (B*fh*fw, ld)."""
    b, c_view, fh, fw = w.shape
    if ch_offset + rows > c_view:
        raise ValueError(f'weight has {c_view} channels, the module needs {ch_offset + rows}')
    w, w_ptr, hp_total = _channel_view(w, 'weight')
    ld = _round_up(rows, 4)
    if out is None:
        out = torch.empty(b * fh * fw, ld, device=w.device, dtype=torch.float32)
    st = _hip.lib.hs_bank_pack_fwd(w_ptr, b, hp_total, fh, fw, ch_offset, rows,
                                   _hip.dev_ptr(out, 'bank'), out.stride(0), _hip.stream_ptr())
    _hip.check(st, 'hs_bank_pack_fwd')
    return out


@_on_operand_device
def bn_fold(gamma, beta, mean, var, eps=BN_EPS_DEFAULT):
    n = mean.numel()
    out = torch.empty(2, n, device=mean.device, dtype=torch.float32)
    st = _hip.lib.hs_bn_fold_fwd(_hip.dev_ptr(gamma, 'bn weight'), _hip.dev_ptr(beta, 'bn bias'),
                                 _hip.dev_ptr(mean, 'bn running_mean'), _hip.dev_ptr(var, 'bn running_var'),
                                 float(eps), n, out[0].data_ptr(), out[1].data_ptr(), _hip.stream_ptr())
    _hip.check(st, 'hs_bn_fold_fwd')
    return out[0], out[1]


def _bank_ptr(bank):
    """A bank is any 2-D fp32 CUDA tensor whose rows are contiguous (row stride = ld)."""
    if bank.dim() != 2 or bank.stride(1) != 1 and bank.shape[1] != 1:
        raise ValueError('bank must be 2-D (patches, rows) with contiguous rows')
    if not bank.is_cuda or bank.dtype != torch.float32:
        _hip.dev_ptr(bank, 'bank')
    return bank.data_ptr(), bank.stride(0)


@_on_operand_device
def patch_conv(x, grid, bank, c_out, k=1, padding=0, padding_mode='reflect', groups=1,
               scale=None, shift=None, act=ACT_NONE):
    """Op A / Op B with fused prologue (x may be a StageInput) and BN-affine + activation epilogue."""
    stage = as_stage(x)
    fh, fw = grid
    b, _, h, w = stage.shape
    st_in = stage.c_struct()
    ep = _epilogue(scale, shift, act)
    y = torch.empty(b, c_out, h, w, device=stage.device, dtype=torch.float32)
    bank_ptr, ld = _bank_ptr(bank)
    co = CoScheduledBanks._active
    if co is not None and k == 1 and padding == 0 and co.signal.device == stage.device:
        signal, sig_ptr, c_signal = _channel_view(co.signal, 'signal')
        arr, refs = _s2w_layer_table(signal, co.layers)
        sb, _, sfh, sfw = signal.shape
        st = _hip.lib.hs_patch_conv_s2w_fwd(C.byref(st_in), fh, fw, bank_ptr, ld, c_out, groups, C.byref(ep), y.data_ptr(),
                                            sig_ptr, sb, c_signal, sfh, sfw, arr, len(co.layers), _hip.stream_ptr())
        if st != -3:                                   # HS_ERR_UNSUPPORTED: nothing was launched, the separate calls follow
            _hip.check(st, 'hs_patch_conv_s2w_fwd')
            co.refs, co.carried = refs, True
            CoScheduledBanks._active = None
            return y
    st = _hip.lib.hs_patch_conv_fwd(C.byref(st_in), fh, fw, bank_ptr, ld, c_out, k,
                                    padding, PAD_MODES[padding_mode], groups, C.byref(ep), y.data_ptr(),
                                    _hip.stream_ptr())
    _hip.check(st, 'hs_patch_conv_fwd')
    return y


class ChainGate:
    """One chained launch at a time per device.  hs_decoder_chain_fwd's workgroups spin on their neighbours: the launch needs its WHOLE
    grid resident, which the host checks against an EMPTY chip -- two chained launches running at once (two streams / threads of one
    model, two prepared models) can starve each other of residency, the bounded spins then give up and the logits are wrong.  The gate
    orders chained launches ACROSS streams with events (a launch on stream B waits for the previous chained launch on stream A; launches
    of one stream are ordered anyway); utils.inference.GraphedModel brackets the replay of a graph that contains a chained launch with the
    same ``enter`` / ``leave`` pair.  Process-wide, per device; the lock spans "wait on the previous event -> launch -> record"."""
    _gates, _glock = {}, threading.Lock()

    def __init__(self):
        self.lock = threading.Lock()
        self.last_stream, self.last_event = None, None
        self._events, self._turn = None, 0

    @classmethod
    def of(cls, device):
        with cls._glock:
            g = cls._gates.get(device.index)
            if g is None:
                g = cls._gates[device.index] = cls()
            return g

    def enter(self, device):
        """Call with ``lock`` held, outside a capture: the current stream waits for the previous chained launch of another stream."""
        cur = torch.cuda.current_stream(device)
        if self.last_event is not None and self.last_stream != cur.cuda_stream:
            cur.wait_event(self.last_event)
        return cur

    def leave(self, cur):
        if self._events is None:
            self._events = [torch.cuda.Event(), torch.cuda.Event()]
        self._turn ^= 1
        ev = self._events[self._turn]
        ev.record(cur)
        self.last_stream, self.last_event = cur.cuda_stream, ev


# counts the chained launches issued while a stream capture is open: a capture site compares it before / after to learn whether its
# graph contains one (GraphedModel then replays through ChainGate)
CHAIN_LAUNCHES_CAPTURED = [0]


class K1Chain:
    """hs_decoder_chain_fwd for one decoder: the three coarse k = 1 levels -- and the first inverted-residual level behind them when
    ``ir`` is given -- as ONE launch whose workgroups hand their level outputs to the neighbouring cells inside the launch
    (csrc/hs_k1_chain.hip).  ``run`` returns the last level's output, or None when the library refuses the shape / the grid is not
    resident at once (the caller then issues the per-level launches).

    Safe by construction (SURVEY 8b "re-entrant"): the launch's workspace -- the generation counter the kernel keeps
    between calls lives there -- belongs to ONE stream (eager) or ONE capture (ExclusiveWorkspaces), so two streams through one
    decoder never share hand-off granules; chained launches of different streams are ordered by ChainGate, so two of them never
    compete for residency; and the kernel's error word (a bounded spin that gave up) is copied to pinned memory every ``POLL_EVERY``
    calls and checked on every call -- a frame computed from an abandoned wait raises instead of being returned silently."""
    POLL_EVERY = 16

    def __init__(self):
        self._ws = {}                        # key -> workspace signature (what has been launched at least once)
        self._refused = set()
        self._pool = ExclusiveWorkspaces()
        self._pool.SPARES = 4                # a zero-copy collective ring captures one graph per slot (three): no zero-fill node in any of them
        self._taken = {}                     # id(tensor) -> (tensor, pinned mirror of its error word)
        self._calls = 0
        self._lock = threading.Lock()

    @property
    def refuses_everything(self):
        return bool(self._refused) and not self._ws

    def _workspace(self, dev, key, nbytes):
        ws = self._pool.take(dev, key[1:], nbytes)
        with self._lock:
            if id(ws) not in self._taken:
                self._taken[id(ws)] = [ws, None]     # the pinned mirror is made by request_error_copy: no host allocation under capture
        return ws

    def check_errors(self):
        """Raises if a launch on any workspace of this object abandoned a wait (reads the pinned mirrors: no synchronisation)."""
        with self._lock:
            bad = [int(m.item()) for _, m in self._taken.values() if m is not None and int(m.item()) != 0]
        if bad:
            raise RuntimeError(f'hs_decoder_chain_fwd abandoned a wait (error word {bad[0]:#x}): its grid was not resident at once -- '
                               'another kernel held the chip; the frame is wrong.  Keep chained launches behind ChainGate '
                               '(GraphedModel does) or turn the chain off (prepare_for_inference(chain_k1=False))')

    def request_error_copy(self, device):
        """Asynchronous copies of every workspace's error word into its pinned mirror, on the current stream (not under capture)."""
        if torch.cuda.is_current_stream_capturing():
            return
        with self._lock:
            for ent in self._taken.values():
                if ent[1] is None:
                    ent[1] = torch.zeros(1, dtype=torch.int32).pin_memory()
            pairs = [(ws, m) for ws, m in self._taken.values() if ws.device == device]
        for ws, m in pairs:
            m.copy_(ws[:1].view(torch.int32)[:1], non_blocking=True)

    def run(self, skips, banks, couts, affines, acts, ir=None):
        """``skips``: the three skip features (B, c, fh << l, fw << l); ``banks``: (P, ld) fp32 tensors; ``affines``: (scale, shift) or
        None per level; ``acts``: activation codes.  ``ir``: dict(skip, bank, hidden, c_out, bn=[(s, b) | None] * 3) of the inverted
        residual on the 8 x 8-pixel patches."""
        dev = skips[0].device
        b, _, fh, fw = skips[0].shape
        key = (dev, b, fh, fw, tuple(couts), tuple(s.shape[1] for s in skips),
               None if ir is None else (ir['skip'].shape[1], ir['hidden'], ir['c_out']))
        if key in self._refused:
            return None
        arr = (_hip.K1LevelC * 3)()
        for l in range(3):
            sk = skips[l]
            if tuple(sk.shape[2:]) != (fh << l, fw << l) or sk.shape[0] != b:
                return None
            a = arr[l]
            a.skip, a.c_skip = _hip.dev_ptr(sk, f'skip feature of level {l}'), sk.shape[1]
            a.bank, a.ld = _bank_ptr(banks[l])
            a.c_out = couts[l]
            if affines[l] is not None:
                a.scale, a.shift = _hip.dev_ptr(affines[l][0], 'scale'), _hip.dev_ptr(affines[l][1], 'shift')
            else:
                a.scale, a.shift = None, None
            a.act = acts[l]
        irc = None
        if ir is not None:
            sk = ir['skip']
            if tuple(sk.shape[2:]) != (fh << 3, fw << 3) or sk.shape[0] != b:
                return None
            irc = _hip.ChainIrLevelC()
            irc.skip, irc.c_skip = _hip.dev_ptr(sk, 'skip feature of the inverted residual'), sk.shape[1]
            irc.bank, irc.ld = _bank_ptr(ir['bank'])
            irc.hidden, irc.c_out = ir['hidden'], ir['c_out']
            for k, bn in enumerate(ir['bn']):
                sc, sh = (None, None) if bn is None else (_hip.dev_ptr(bn[0], 'bn scale'), _hip.dev_ptr(bn[1], 'bn shift'))
                setattr(irc, f's{k + 1}', sc)
                setattr(irc, f'b{k + 1}', sh)
        irp = C.byref(irc) if irc is not None else None
        self.check_errors()
        capturing = torch.cuda.is_current_stream_capturing()
        with _hip.device_scope(dev):
            n = self._ws.get(key)
            if n is None:
                n = int(_hip.lib.hs_decoder_chain_workspace(b, fh, fw, arr, 3, irp))
                if n < 0:
                    self._refused.add(key)
                    return None
            # zero-filled ONCE per (stream | capture): generation 0.  A fill inside a capture would become a node of the graph and every
            # replay would restart at generation 1 -- legal, but a memset node per frame: ExclusiveWorkspaces hands out prepared spares
            ws = self._workspace(dev, key, n)
            scale = 8 if ir is not None else 4
            y = torch.empty(b, ir['c_out'] if ir is not None else couts[2], scale * fh, scale * fw, device=dev, dtype=torch.float32)
            gate = ChainGate.of(dev)
            if capturing:
                st = _hip.lib.hs_decoder_chain_fwd(b, fh, fw, arr, 3, irp, ws.data_ptr(), y.data_ptr(), _hip.stream_ptr())
                CHAIN_LAUNCHES_CAPTURED[0] += 1
            else:
                with gate.lock:
                    cur = gate.enter(dev)
                    st = _hip.lib.hs_decoder_chain_fwd(b, fh, fw, arr, 3, irp, ws.data_ptr(), y.data_ptr(), _hip.stream_ptr())
                    if st == 0:
                        gate.leave(cur)
        if st == -3:                              # HS_ERR_UNSUPPORTED: shape or residency -- nothing was launched
            self._refused.add(key)
            return None
        _hip.check(st, 'hs_decoder_chain_fwd')
        self._ws[key] = n
        self._calls += 1
        if not capturing and self._calls % self.POLL_EVERY == 0:
            self.request_error_copy(dev)
        return y

    def reset(self):
        """Drops every workspace (and a raised error with them): the next call starts at generation 0 on fresh buffers.  Graphs captured
        through this object keep their own references alive and must be re-captured."""
        with self._lock:
            self._taken.clear()
        self._pool = ExclusiveWorkspaces()
        self._pool.SPARES = 4

    def error_word(self):
        """Non-zero if a launch on any of this object's workspaces abandoned a wait (synchronising host read: diagnostics / tests)."""
        with self._lock:
            wss = [ws for ws, _ in self._taken.values()]
        return max([int(ws[:1].view(torch.int32)[0].item()) for ws in wss] or [0])


# The three coarse k = 1 levels (+ the first inverted-residual level) as one launch: opt-in per decoder (``decoder.chain_k1 = True``;
# prepare_for_inference(chain_k1=True) sets it), or process-wide with HS_K1_CHAIN=1 (dev A/B switch).  HS_K1_CHAIN_IR=0 keeps the
# inverted residual out of the chain (A/B).
K1_CHAIN = os.environ.get('HS_K1_CHAIN', '0') == '1'
K1_CHAIN_IR = os.environ.get('HS_K1_CHAIN_IR', '1') == '1'
# Whether a chained decoder carries its first inverted-residual level in the chain launch unless ``decoder.chain_ir`` says otherwise
K1_CHAIN_IR_DEFAULT = os.environ.get('HS_K1_CHAIN_IR_DEFAULT', '0') == '1'


@_on_operand_device
def meta_conv(x, w, c_out, kernel_size, stride=(1, 1), padding=(0, 0), dilation=(1, 1), padding_mode='zeros', groups=1,
              scale=None, shift=None, act=ACT_NONE):
    """MetaConv2d.forward for ANY of the reference's arguments (meta_conv.py:141-186): x (B, Cin, H, W), per-sample weights
    w (B, >= Cout * Cin/groups * kh * kw); non-square kernels, stride, dilation, any padding.  The "same"-padded stride-1
    case runs faster through :func:`patch_conv` with a (1, 1) grid."""
    if isinstance(x, StageInput):
        x = x.materialize()
    b, cin, h, wd = x.shape
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = kernel_size, stride, padding, dilation
    # zero padding: ph rows / pw columns on both sides.  The other modes: the reference hands (ph, pw, ph, pw) to F.pad, whose
    # order is (left, right, top, bottom) -- left = top = ph, right = bottom = pw (meta_conv.py:159, 175-176); reproduced.
    pt, pb, pl, pr = (ph, ph, pw, pw) if padding_mode == 'zeros' else (ph, pw, ph, pw)
    ho = (h + pt + pb - dh * (kh - 1) - 1) // sh + 1
    wo = (wd + pl + pr - dw * (kw - 1) - 1) // sw + 1
    if ho <= 0 or wo <= 0:
        raise ValueError(f'kernel {kernel_size} (dilation {dilation}) does not fit the padded {h}x{wd} input')
    if w.dim() != 2 or w.shape[0] != b or w.stride(1) != 1:
        raise ValueError(f'w must be (B, rows) with contiguous rows, got {tuple(w.shape)}')
    ep = _epilogue(scale, shift, act)
    y = torch.empty(b, c_out, ho, wo, device=x.device, dtype=torch.float32)
    w_ptr, ldw = _bank_ptr(w)            # a column range w[:, a:b] of a wider tensor (MetaSequential's slice) is read in place: ldw = its row stride
    st = _hip.lib.hs_meta_conv_fwd(_hip.dev_ptr(x, 'x'), b, cin, h, wd, w_ptr, ldw, c_out, kh, kw,
                                   sh, sw, pt, pb, pl, pr, dh, dw, PAD_MODES[padding_mode], groups, C.byref(ep), y.data_ptr(),
                                   _hip.stream_ptr())
    _hip.check(st, 'hs_meta_conv_fwd')
    return y


@_on_operand_device
def patch_conv_gen(x, sref, c_out, scale=None, shift=None, act=ACT_NONE):
    """Op A with the bank generated inside the kernel from ``sref`` (a :class:`SignalRef`): signal2weights + k = 1 patch
    conv + BN affine + activation, one launch, no bank in HBM.  Returns None when the shape is outside what the kernel
    covers (the caller then materialises the bank)."""
    stage = as_stage(x)
    b, _, h, w = stage.shape
    signal, sig_ptr, c_signal = _channel_view(sref.signal, 'signal')
    fh, fw = signal.shape[-2:]
    l = sref.layer
    lay = _hip.S2wLayerC()
    lay.signal_index, lay.signal_channels, lay.groups = l['signal_index'], l['signal_channels'], l['groups']
    lay.wsw_t, lay.wc, lay.rows = _hip.dev_ptr(l['wsw_t'], 'wsw_t'), l['wsw_t'].shape[1], l['rows']
    lay.bank, lay.ld = None, 0
    st_in = stage.c_struct()
    ep = _epilogue(scale, shift, act)
    y = torch.empty(b, c_out, h, w, device=stage.device, dtype=torch.float32)
    st = _hip.lib.hs_patch_conv_gen_fwd(C.byref(st_in), fh, fw, sig_ptr, c_signal, C.byref(lay), c_out, C.byref(ep),
                                        y.data_ptr(), _hip.stream_ptr())
    if st == -3:
        return None
    _hip.check(st, 'hs_patch_conv_gen_fwd')
    return y


@_on_operand_device
def patch_ir(x, grid, bank, hidden, c_out, bn1, bn2, bn3, residual=False, math=None):
    """Op C: fused per-patch inverted residual.  bn* are (scale, shift) pairs; bank rows in the reference's flat order.
    ``math``: 'f32' | 'split' | 'auto' (include/hyperseg_hip.h, hs_ir_math); None = :func:`get_ir_math`."""
    stage = as_stage(x)
    fh, fw = grid
    b, _, h, w = stage.shape
    st_in = stage.c_struct()
    e1, e2, e3 = _epilogue(*bn1), _epilogue(*bn2), _epilogue(*bn3)
    y = torch.empty(b, c_out, h, w, device=stage.device, dtype=torch.float32)
    bank_ptr, ld = _bank_ptr(bank)
    st = _hip.lib.hs_patch_ir_fwd(C.byref(st_in), fh, fw, bank_ptr, ld, hidden, c_out,
                                  C.byref(e1), C.byref(e2), C.byref(e3), int(bool(residual)), ir_math_code(math),
                                  y.data_ptr(), _hip.stream_ptr())
    _hip.check(st, 'hs_patch_ir_fwd')
    return y


D2_ROUTE = os.environ.get('HS_IR_D2', '1') != '0'      # A/B switch: 0 = small-patch Op D levels on the single-launch tiled kernel


@_on_operand_device
def patch_ir_v0(x, grid, bank, hidden, c_out, bn1, bn2, bn3, math=None):
    """Op D (hyperseg_v0_1.py:205-237) as one launch -- two through a scratch hidden map where patches are 4 x 4 / 8 x 8 pixels
    (hs_patch_ir_v0_ws_fwd; the scratch is a torch allocation of this call: the caching allocator hands the same block back every
    step, and under graph capture it belongs to the graph's pool); returns None when the shape has no fused instantiation (the
    caller then runs the block as three patch convolutions).  ``math`` as for :func:`patch_ir` (Op D has the exact-f32 form only)."""
    stage = as_stage(x)
    fh, fw = grid
    b, _, h, w = stage.shape
    st_in = stage.c_struct()
    e1, e2, e3 = _epilogue(*bn1), _epilogue(*bn2), _epilogue(*bn3)
    y = torch.empty(b, c_out, h, w, device=stage.device, dtype=torch.float32)
    bank_ptr, ld = _bank_ptr(bank)
    need = int(_hip.lib.hs_patch_ir_v0_workspace(C.byref(st_in), fh, fw, hidden, c_out)) if D2_ROUTE else 0
    ws = torch.empty(need // 4, device=stage.device, dtype=torch.float32) if need > 0 else None
    st = _hip.lib.hs_patch_ir_v0_ws_fwd(C.byref(st_in), fh, fw, bank_ptr, ld, hidden, c_out, C.byref(e1), C.byref(e2), C.byref(e3),
                                        ir_math_code(math), ws.data_ptr() if ws is not None else None, need, y.data_ptr(),
                                        _hip.stream_ptr())
    if st == -3:
        return None
    _hip.check(st, 'hs_patch_ir_v0_ws_fwd')
    return y


IR_MATH = {'auto': 0, 'f32': 1, 'split': 2}
# Arithmetic of the fused inverted-residual levels (include/hyperseg_hip.h, hs_ir_math) is an ARGUMENT of every launch: the
# library keeps no mode.  Who chooses: (1) the ``math`` argument / the calling module's ``ir_math`` attribute (the nn.Module
# mirror defaults to 'f32', the reference's arithmetic; ``prepare_for_inference(ir_math='auto')`` opts a model into the
# f16-split form for serving), unless (2) an override is set here -- ``set_ir_math`` (tests, A/B runs) or the environment
# variable HS_IR_MATH at import time.
_ir_math_override = os.environ.get('HS_IR_MATH') or None
if _ir_math_override is not None and _ir_math_override not in IR_MATH:
    raise ValueError(f'HS_IR_MATH={_ir_math_override!r}: expected one of {sorted(IR_MATH)}')
DEFAULT_IR_MATH = 'f32'


def set_ir_math(mode):
    """Process-wide OVERRIDE of the math mode every fused inverted-residual launch is asked for ('split' | 'f32' | 'auto'),
    or None to hand the choice back to the modules.  A Python-side switch for tests and A/B runs -- the C library itself is
    stateless.  Returns the previous override.  One word, swapped under the cache lock; a thread that wants its OWN mode without
    touching the others' (a replica thread of nn.DataParallel) uses :func:`ir_math_scope`."""
    global _ir_math_override
    if mode is not None and mode not in IR_MATH:
        raise ValueError(f'ir math {mode!r}: expected one of {sorted(IR_MATH)} or None')
    with _CACHE_LOCK:
        prev, _ir_math_override = _ir_math_override, mode
    return prev


_ir_math_local = threading.local()


class ir_math_scope:
    """``with ir_math_scope('split'): ...`` -- a THREAD-LOCAL override (outranks the process-wide one) for the calling thread only."""

    def __init__(self, mode):
        if mode is not None and mode not in IR_MATH:
            raise ValueError(f'ir math {mode!r}: expected one of {sorted(IR_MATH)} or None')
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_ir_math_local, 'mode', None)
        _ir_math_local.mode = self.mode
        return self

    def __exit__(self, *exc):
        _ir_math_local.mode = self.prev


def get_ir_math(requested=None):
    """The mode a launch uses: the calling thread's scope if one is open, else the process-wide override if one is set, else
    ``requested`` (a module's attribute), else the default."""
    return getattr(_ir_math_local, 'mode', None) or _ir_math_override or requested or DEFAULT_IR_MATH


def ir_math_code(requested=None):
    return IR_MATH[get_ir_math(requested)]


IR_ROUTES = {0: 'generic', 1: 'f32_mfma', 2: 'split_mfma'}


def patch_ir_route(shape, c_skip, c_prev, grid, hidden, c_out, math=None, residual=False, coords=True):
    """Which kernel ``patch_ir`` would run for a stage of ``shape`` = (B, H, W) with ``c_skip`` skip and ``c_prev`` half-resolution
    previous-level channels: 'split_mfma' | 'f32_mfma' | 'generic' (include/hyperseg_hip.h, hs_patch_ir_route).  Host only."""
    b, h, w = shape
    si = _hip.StageInputC()
    si.skip, si.prev, si.batch, si.H, si.W = None, None, b, h, w
    si.c_skip, si.c_prev, si.Hp, si.Wp = c_skip, c_prev, h // 2, w // 2
    si.coords, si.prev_mode = int(bool(coords)), (2 if c_prev > 0 else 0)      # HS_PREV_BILINEAR: the half-resolution previous level
    r = _hip.lib.hs_patch_ir_route(C.byref(si), grid[0], grid[1], hidden, c_out, int(bool(residual)), ir_math_code(math))
    if r < 0:
        _hip.check(r, 'hs_patch_ir_route')
    return IR_ROUTES[r]


def ir_tile_map(reg, mode, pwr):
    """(n_pixel_tiles, array (n_tiles, 16, 3) of (u, v, live)) -- the fused inverted-residual kernel's tile map
    (host-side introspection, no GPU needed)."""
    import numpy as np
    nt, nt3 = C.c_int32(0), C.c_int32(0)
    _hip.check(_hip.lib.hs_ir_tile_map(reg, mode, pwr, C.byref(nt), C.byref(nt3), None, 0), 'hs_ir_tile_map')
    buf = (C.c_int32 * (nt.value * 48))()
    _hip.check(_hip.lib.hs_ir_tile_map(reg, mode, pwr, C.byref(nt), C.byref(nt3), buf, nt.value * 48), 'hs_ir_tile_map')
    return nt3.value, np.ctypeslib.as_array(buf).reshape(nt.value, 16, 3).copy()


class ExclusiveWorkspaces:
    """Zero-initialised device buffers that belong to ONE launch at a time, for kernels that keep protocol state in memory across
    launches (csrc/hs_se_tail.h: generation words + tagged granules; the state a launch leaves is the state the next one expects,
    so a buffer may serve any number of launches IN SEQUENCE but never two at once).

    ``take(device, signature, nbytes)``: eager calls get the buffer of (current stream, signature) -- launches of one stream are
    ordered.  Calls made while a HIP graph is being captured get a buffer that belongs to THAT capture (and is kept alive for as long
    as this object lives: the graph's kernel arguments point at it): prepared, zeroed spares are handed out, because an allocation
    inside a capture would put its zero-fill into the graph and wipe the state at every replay -- correct, but a memset node per use.
    Spares are topped up by every eager call, so "run eagerly once, then capture" (what every capture site of this repository does:
    lazily built buffers have to exist before a capture anyway) never allocates inside a capture; ``captured_zero_fills`` counts the
    times it had to.  A capture is recognised by the eager -> capturing transition seen here: two captures with no eager call
    between them share buffers, which is only wrong if their graphs are then replayed concurrently."""
    SPARES = 2

    def __init__(self):
        self._lock = threading.Lock()
        self._eager, self._spares, self._captured = {}, {}, {}
        self._epoch, self._was_capturing = 0, False
        self.captured_zero_fills = 0

    @staticmethod
    def _zeros(device, nbytes):
        return torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=device)

    def take(self, device, signature, nbytes):
        capturing = torch.cuda.is_current_stream_capturing()
        stream = torch.cuda.current_stream(device).cuda_stream
        with self._lock:
            if capturing and not self._was_capturing:
                self._epoch += 1
            self._was_capturing = capturing
            skey = (device.index, signature, nbytes)
            if capturing:
                key = (stream, self._epoch) + skey
                t = self._captured.get(key)
                if t is None:
                    pool = self._spares.get(skey)
                    if pool:
                        t = pool.pop()
                    else:
                        self.captured_zero_fills += 1
                        t = self._zeros(device, nbytes)
                    self._captured[key] = t
                return t
            key = (stream,) + skey
            t = self._eager.get(key)
            if t is None:
                t = self._eager[key] = self._zeros(device, nbytes)
            pool = self._spares.setdefault(skey, [])
            while len(pool) < self.SPARES:
                pool.append(self._zeros(device, nbytes))
            return t


SE_WORKSPACES = ExclusiveWorkspaces()
# Opt-in (HS_SE_TAIL=1): measured on the MI355X (visits r5v7 / r5v8, profiles/round5_se_tail_negative.txt) the tails are correct
# (tests/test_hip_encoder.py runs them whatever this says) but cost 5-9 us per block -- two memory-side hand-offs of ~2.5 us each --
# against the 5.3 / 9.6 us of the launches they remove: 0.796 ms per HyperSeg-M frame against 0.776.
SE_TAIL = os.environ.get('HS_SE_TAIL', '0') != '0'


def se_tail_descriptor(device, batch, channels, nblk, wgs_per_batch, w_reduce, b_reduce, w_expand_t, b_expand):
    """(hs_se_tail struct, gate tensor, keep-alive tuple) for a pooling launch that finishes the squeeze-excite gate itself
    (include/hyperseg_hip.h hs_se_tail), or None when the shape is not covered (the caller then pools and calls :func:`se_gate`)."""
    csq = w_reduce.shape[0]
    nbytes = int(_hip.lib.hs_se_tail_workspace(batch, channels, csq, nblk, wgs_per_batch))
    if nbytes <= 0:
        return None
    ws = SE_WORKSPACES.take(device, (batch, channels, csq, nblk, int(wgs_per_batch)), nbytes)
    gate = torch.empty(batch, channels, device=device, dtype=torch.float32)
    d = _hip.SeTailC()
    d.w_reduce, d.b_reduce = _hip.dev_ptr(w_reduce, 'w_reduce'), _hip.dev_ptr(b_reduce, 'b_reduce')
    d.w_expand_t, d.b_expand = _hip.dev_ptr(w_expand_t, 'w_expand_t'), _hip.dev_ptr(b_expand, 'b_expand')
    d.c_squeezed = csq
    d.gate, d.squeezed, d.workspace = gate.data_ptr(), None, ws.data_ptr()
    return d, gate, ws


def se_tail_error(ws, batch):
    """The error word of a tail workspace (nonzero: a bounded wait gave up and the gate of that launch is NaN); synchronises."""
    return int(ws[batch].item())


@_on_operand_device
def depthwise_conv_bn_act(x, weight, stride, pad_top, pad_left, out_size, scale=None, shift=None, act=0, pool=False,
                          in_scale=None, in_shift=None, se=None):
    """Depthwise conv (k 3|5, stride 1|2, TF-"SAME" zero padding given as top/left offsets) + affine + activation
    (3 = swish) in one launch.  ``pool=True`` also returns the per-workgroup partial sums of the outputs (B*C, nblk)
    for :func:`se_gate`.  ``in_scale``/``in_shift`` (C): the taps are swish(in_scale*x + in_shift) -- x is then the RAW
    output of the 1x1 expand GEMM.  ``se=(w_reduce, b_reduce, w_expand_t, b_expand)`` with ``pool=True``: the launch finishes the
    squeeze-excite gate itself where it can (hs_depthwise_conv_se_fwd) and returns (y, gate (B, C), True) instead of
    (y, partial, False).  Encoder-side helper, opt-in (utils/inference.py)."""
    b, c, h, w = x.shape
    k = weight.shape[-1]
    ho, wo = out_size
    y = torch.empty(b, c, ho, wo, device=x.device, dtype=torch.float32)
    if se is not None and pool and SE_TAIL:
        nblk = _hip.lib.hs_depthwise_pool_blocks(ho, wo)
        desc = se_tail_descriptor(x.device, b, c, nblk, c * nblk, *se)
        if desc is not None:
            d, gate, _ws = desc
            st = _hip.lib.hs_depthwise_conv_se_fwd(_hip.dev_ptr(x, 'x'), b, c, h, w, _hip.dev_ptr(weight, 'weight'), k, stride,
                                                   pad_top, pad_left, ho, wo,
                                                   _hip.dev_ptr(scale, 'scale') if scale is not None else None,
                                                   _hip.dev_ptr(shift, 'shift') if shift is not None else None, int(act), y.data_ptr(),
                                                   _hip.dev_ptr(in_scale, 'in_scale') if in_scale is not None else None,
                                                   _hip.dev_ptr(in_shift, 'in_shift') if in_shift is not None else None,
                                                   C.byref(d), _hip.stream_ptr())
            _hip.check(st, 'hs_depthwise_conv_se_fwd')
            return y, gate, True
    partial = None
    if pool:
        partial = torch.empty(b * c, _hip.lib.hs_depthwise_pool_blocks(ho, wo), device=x.device, dtype=torch.float32)
    st = _hip.lib.hs_depthwise_conv_fwd(_hip.dev_ptr(x, 'x'), b, c, h, w, _hip.dev_ptr(weight, 'weight'), k, stride,
                                        pad_top, pad_left, ho, wo,
                                        _hip.dev_ptr(scale, 'scale') if scale is not None else None,
                                        _hip.dev_ptr(shift, 'shift') if shift is not None else None, int(act),
                                        y.data_ptr(), partial.data_ptr() if pool else None,
                                        _hip.dev_ptr(in_scale, 'in_scale') if in_scale is not None else None,
                                        _hip.dev_ptr(in_shift, 'in_shift') if in_shift is not None else None,
                                        _hip.stream_ptr())
    _hip.check(st, 'hs_depthwise_conv_fwd')
    if se is not None and pool:
        return y, partial, False
    return (y, partial) if pool else y


@_on_operand_device
def stem_conv_bn_swish(x, weight, pad_top, pad_left, out_size, scale, shift):
    """3x3 stride-2 conv of the 3-channel image + folded BN + swish, one launch (encoder stem).  Opt-in helper."""
    b, cin, h, w = x.shape
    cout = weight.shape[0]
    ho, wo = out_size
    y = torch.empty(b, cout, ho, wo, device=x.device, dtype=torch.float32)
    st = _hip.lib.hs_stem_conv_fwd(_hip.dev_ptr(x, 'x'), b, cin, h, w, _hip.dev_ptr(weight, 'weight'), cout, pad_top, pad_left,
                                   ho, wo, _hip.dev_ptr(scale, 'scale'), _hip.dev_ptr(shift, 'shift'), y.data_ptr(),
                                   _hip.stream_ptr())
    _hip.check(st, 'hs_stem_conv_fwd')
    return y


@_on_operand_device
def stem_dw(x, w_stem28, scale0, shift0, stem_pad_top, stem_pad_left, stem_out_size, w_dw, pad_top, pad_left, scale1, shift1, pool=True):
    """The encoder's stem (3x3 stride-2 conv of the image + BN + swish) and the first block's depthwise 3x3 + BN + swish (+ SE
    pooling partial sums) in ONE launch: the stem's output map never reaches HBM (hs_stem_dw_fwd).  ``w_stem28``: the stem weight
    flattened to (Cmid, 27) plus one zero column.  Returns ``(y, partial)`` / ``y``, or ``None`` when the launch does not cover the
    shape (the caller then runs the two launches).  Encoder-side helper, opt-in."""
    b, _, h, w = x.shape
    cmid, k = w_dw.shape[0], w_dw.shape[-1]
    hs_, ws_ = stem_out_size
    y = torch.empty(b, cmid, hs_, ws_, device=x.device, dtype=torch.float32)
    partial = torch.empty(b * cmid, _hip.lib.hs_mbconv_tiles(k, 1, hs_, ws_), device=x.device, dtype=torch.float32) if pool else None
    st = _hip.lib.hs_stem_dw_fwd(_hip.dev_ptr(x, 'x'), b, h, w, _hip.dev_ptr(w_stem28, 'w_stem28'), cmid, _hip.dev_ptr(scale0, 'scale0'),
                                 _hip.dev_ptr(shift0, 'shift0'), stem_pad_top, stem_pad_left, hs_, ws_, _hip.dev_ptr(w_dw, 'w_dw'), k,
                                 pad_top, pad_left, _hip.dev_ptr(scale1, 'scale1'), _hip.dev_ptr(shift1, 'shift1'), y.data_ptr(),
                                 partial.data_ptr() if pool else None, _hip.stream_ptr())
    if st == -3:                       # HS_ERR_UNSUPPORTED: nothing was launched
        return None
    _hip.check(st, 'hs_stem_dw_fwd')
    return (y, partial) if pool else y


@_on_operand_device
def mbconv_expand_dw(x, w_expand, scale0, shift0, w_dw, stride, pad_top, pad_left, out_size, scale1, shift1, pool=True, se=None):
    """1x1 expand + BN + swish + depthwise k x k (TF-"SAME" zero padding of the ACTIVATION) + BN + swish in one launch
    (+ SE pooling partial sums): the expanded tensor never reaches HBM.  x (B,Cin,H,W), w_expand (Cmid,Cin[,1,1]),
    w_dw (Cmid,1,k,k) -> y (B,Cmid,Ho,Wo)[, partial (B*Cmid, ntiles)].  ``se``: as in :func:`depthwise_conv_bn_act`
    (hs_mbconv_expand_dw_se_fwd; returns (y, gate | partial, bool)).  Encoder-side helper, opt-in."""
    b, cin, h, w = x.shape
    cmid, k = w_dw.shape[0], w_dw.shape[-1]
    ho, wo = out_size
    y = torch.empty(b, cmid, ho, wo, device=x.device, dtype=torch.float32)
    if se is not None and pool and SE_TAIL:
        nblk = _hip.lib.hs_mbconv_tiles(k, stride, ho, wo)
        desc = se_tail_descriptor(x.device, b, cmid, nblk, int(_hip.lib.hs_mbconv_se_workgroups(b, cmid, k, stride, ho, wo)), *se)
        if desc is not None:
            d, gate, _ws = desc
            st = _hip.lib.hs_mbconv_expand_dw_se_fwd(_hip.dev_ptr(x, 'x'), b, cin, h, w, _hip.dev_ptr(w_expand, 'w_expand'), cmid,
                                                     _hip.dev_ptr(scale0, 'scale0'), _hip.dev_ptr(shift0, 'shift0'),
                                                     _hip.dev_ptr(w_dw, 'w_dw'), k, stride, pad_top, pad_left, ho, wo,
                                                     _hip.dev_ptr(scale1, 'scale1'), _hip.dev_ptr(shift1, 'shift1'), y.data_ptr(),
                                                     C.byref(d), _hip.stream_ptr())
            _hip.check(st, 'hs_mbconv_expand_dw_se_fwd')
            return y, gate, True
    partial = None
    if pool:
        partial = torch.empty(b * cmid, _hip.lib.hs_mbconv_tiles(k, stride, ho, wo), device=x.device, dtype=torch.float32)
    st = _hip.lib.hs_mbconv_expand_dw_fwd(_hip.dev_ptr(x, 'x'), b, cin, h, w, _hip.dev_ptr(w_expand, 'w_expand'), cmid,
                                          _hip.dev_ptr(scale0, 'scale0'), _hip.dev_ptr(shift0, 'shift0'),
                                          _hip.dev_ptr(w_dw, 'w_dw'), k, stride, pad_top, pad_left, ho, wo,
                                          _hip.dev_ptr(scale1, 'scale1'), _hip.dev_ptr(shift1, 'shift1'), y.data_ptr(),
                                          partial.data_ptr() if pool else None, _hip.stream_ptr())
    _hip.check(st, 'hs_mbconv_expand_dw_fwd')
    if se is not None and pool:
        return y, partial, False
    return (y, partial) if pool else y


@_on_operand_device
def pointwise_conv(x, weight, gate=None, scale=None, shift=None, act=0, residual=None):
    """1x1 conv (fp32 MFMA GEMM) + optional input gate (B,Cin) + affine + activation (3 = swish) + residual, one launch."""
    b, cin, h, w = x.shape
    cout = weight.shape[0]
    y = torch.empty(b, cout, h, w, device=x.device, dtype=torch.float32)
    opt = lambda t, n: _hip.dev_ptr(t, n) if t is not None else None   # noqa: E731
    st = _hip.lib.hs_pointwise_conv_fwd(_hip.dev_ptr(x, 'x'), b, cin, h * w, _hip.dev_ptr(weight, 'weight'), cout,
                                        opt(gate, 'gate'), opt(scale, 'scale'), opt(shift, 'shift'), int(act),
                                        opt(residual, 'residual'), y.data_ptr(), _hip.stream_ptr())
    _hip.check(st, 'hs_pointwise_conv_fwd')
    return y


@_on_operand_device
def affine_act_(x, scale, shift, act=0, residual=None):
    """In place: x = act(scale[c]*x + shift[c]) + residual  (folded BN + activation + skip add, one launch);
    ``scale=None`` means 1."""
    b, c, h, w = x.shape
    st = _hip.lib.hs_affine_act_fwd(_hip.dev_ptr(x, 'x'), b, c, h * w,
                                    _hip.dev_ptr(scale, 'scale') if scale is not None else None,
                                    _hip.dev_ptr(shift, 'shift'), int(act),
                                    _hip.dev_ptr(residual, 'residual') if residual is not None else None, x.data_ptr(),
                                    _hip.stream_ptr())
    _hip.check(st, 'hs_affine_act_fwd')
    return x


@_on_operand_device
def se_gate(partial, batch, hw, w_reduce, b_reduce, w_expand, b_expand, w_proj=None, out_scale=None):
    """Squeeze-excite gate from pooled partial sums; with ``w_proj`` (Cout, C[,1,1]) returns the project weights scaled by
    the gate (and by ``out_scale`` (Cout), the project conv's folded BN scale), (B, Cout, C, 1, 1); otherwise the gate
    (B, C).  One launch for blocks with small reduce weights (C <= 768, Csq <= 32), else two (squeeze, excite)."""
    c = partial.shape[0] // batch
    csq = w_reduce.shape[0]
    dev = partial.device
    w_scaled = None
    cout = 0
    work = torch.empty(batch, c + csq, device=dev, dtype=torch.float32)      # gate | squeezed activations
    gate, squeezed = work[:, :c], work[:, c:]
    if batch > 1:
        gate, squeezed = torch.empty(batch, c, device=dev), torch.empty(batch, csq, device=dev)
    if w_proj is not None:
        cout = w_proj.shape[0]
        w_scaled = torch.empty(batch, cout, c, 1, 1, device=dev, dtype=torch.float32)
    st = _hip.lib.hs_se_gate_fwd(_hip.dev_ptr(partial, 'partial'), batch, c, partial.shape[1], 1.0 / float(hw),
                                 _hip.dev_ptr(w_reduce, 'w_reduce'), _hip.dev_ptr(b_reduce, 'b_reduce'), csq,
                                 _hip.dev_ptr(w_expand, 'w_expand'), _hip.dev_ptr(b_expand, 'b_expand'),
                                 squeezed.data_ptr(), gate.data_ptr(),
                                 _hip.dev_ptr(w_proj, 'w_proj') if w_proj is not None else None, cout,
                                 _hip.dev_ptr(out_scale, 'out_scale') if out_scale is not None else None,
                                 w_scaled.data_ptr() if w_scaled is not None else None, _hip.stream_ptr())
    _hip.check(st, 'hs_se_gate_fwd')
    return w_scaled if w_proj is not None else gate


class SplitWeights:
    """A static 1x1-conv weight prepared for ``hs_gemm_split_fwd``: f16 hi / lo pieces of the power-of-two row-scaled weight in
    MFMA-fragment order + the inverse row scales (include/hyperseg_hip.h)."""
    __slots__ = ('frag', 'inv', 'c_out', 'c_in', 'kp')

    def __init__(self, frag, inv, c_out, c_in, kp):
        self.frag, self.inv, self.c_out, self.c_in, self.kp = frag, inv, c_out, c_in, kp


@torch.no_grad()
def gemm_split_weights(w, row_scale=None, max_k=1280):
    """``w`` (Cout, Cin[, 1, 1]) f32 -> SplitWeights, or None when the kernel does not cover Cin (> 1280; the 2x2 / stride-2
    form of ``gemm_split_conv2x2`` takes K = 4 Cin up to 2560: ``max_k=2560`` with the (Cout, Cin, 2, 2) weight).  ``row_scale``
    (Cout): folded into the rows first (a BatchNorm scale).  Torch tensor ops only; runs once per weight."""
    w = w.detach().flatten(1).float()
    m, k = w.shape
    kp = _hip.lib.hs_gemm_split_kp(k) if k <= max_k else -1
    if kp < 0:
        return None
    if row_scale is not None:
        w = w * row_scale.detach().float()[:, None]
    # 2^(141 - eb), eb = biased exponent of the row maximum clamped to [27, 254]: row * scale < 2^15 (gs_exp_of / gs_scale_of)
    eb = (w.abs().amax(1).contiguous().view(torch.int32) >> 23).clamp(27, 254)
    one = torch.ones(m, device=w.device)
    ws = w * torch.ldexp(one, 141 - eb)[:, None]
    hi = ws.half()
    lo = (ws - hi.float()).half()
    rt = -(-m // 16)
    pad = (0, kp - k, 0, 16 * rt - m)
    hi, lo = torch.nn.functional.pad(hi, pad), torch.nn.functional.pad(lo, pad)
    inv = torch.nn.functional.pad(torch.ldexp(one, eb - 141), (0, 16 * rt - m), value=1.0)

    def fragment_order(t):      # [RT][16 rows][KST][4 kgroups][8] -> [RT][KST][kgroup][row][8]: lane = row + 16 * kgroup
        return t.view(rt, 16, kp // 32, 4, 8).permute(0, 2, 3, 1, 4)
    frag = torch.stack([fragment_order(hi), fragment_order(lo)], dim=2).contiguous()
    return SplitWeights(frag, inv.contiguous(), m, k, kp)


@_on_operand_device
def gemm_split(sw, x, gate=None, shift=None, act=ACT_NONE, residual=None, out=None):
    """y (B, Cout, H, W) = act(W @ (gate[:, :, None] * x) + shift[:, None]) + residual per frame, on the f16 matrix cores with
    split operands (hs_gemm_split_fwd).  ``out``: written in place; ``residual`` may be ``out`` itself (accumulation onto a
    skip tensor).  ``act``: ACT_* or 3 = swish."""
    b, cin, h, w = x.shape
    if cin != sw.c_in:
        raise ValueError(f'input has {cin} channels, the weight {sw.c_in}')
    shape = (b, sw.c_out, h, w)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != shape:
        raise ValueError(f'out has shape {tuple(out.shape)}, expected {shape}')
    if residual is not None and tuple(residual.shape) != shape:
        raise ValueError(f'residual has shape {tuple(residual.shape)}, expected {shape}')
    if gate is not None and tuple(gate.shape) != (b, cin):
        raise ValueError(f'gate has shape {tuple(gate.shape)}, expected {(b, cin)}')
    if shift is not None and shift.numel() != sw.c_out:
        raise ValueError(f'shift has {shift.numel()} entries, expected {sw.c_out}')
    st = _hip.lib.hs_gemm_split_fwd(_hip.dev_ptr(sw.frag, 'w_frag', torch.float16), _hip.dev_ptr(sw.inv, 'w_inv'),
                                    _hip.dev_ptr(gate, 'gate') if gate is not None else None, _hip.dev_ptr(x, 'x'),
                                    _hip.dev_ptr(shift, 'shift') if shift is not None else None, int(act),
                                    _hip.dev_ptr(residual, 'residual') if residual is not None else None,
                                    _hip.dev_ptr(out, 'out'), b, sw.c_out, cin, sw.kp, h * w, _hip.stream_ptr())
    _hip.check(st, 'hs_gemm_split_fwd')
    return out


@_on_operand_device
def gemm_split_conv2x2(sw, x, shift=None, act=ACT_NONE, out=None, pool_partial=None):
    """y (B, Cout, H/2, W/2) = act(Conv2d(kernel 2, stride 2)(x) + shift[:, None, None]) on the f16 matrix cores with split
    operands, the window read on load (hs_gemm_split_conv2x2_fwd; no im2col copy).  ``sw`` = gemm_split_weights(conv.weight
    (Cout, Cin, 2, 2), scale, max_k=2560).  ``pool_partial``: a (B, Cout, ceil(H W / 64)) f32 tensor that receives the sums of y
    over blocks of 16 pixels (``pooled_shift`` turns them into the global average's contribution)."""
    b, cin, h, w = x.shape
    if 4 * cin != sw.c_in:
        raise ValueError(f'input has {cin} channels, the weight {sw.c_in} / 4')
    if h % 2 or w % 4:
        raise ValueError(f'map {h}x{w}: height must be even and width a multiple of 4')
    shape = (b, sw.c_out, h // 2, w // 2)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != shape:
        raise ValueError(f'out has shape {tuple(out.shape)}, expected {shape}')
    if shift is not None and shift.numel() != sw.c_out:
        raise ValueError(f'shift has {shift.numel()} entries, expected {sw.c_out}')
    nblk = -(-(h // 2) * (w // 2) // 16)
    if pool_partial is not None and tuple(pool_partial.shape) != (b, sw.c_out, nblk):
        raise ValueError(f'pool_partial has shape {tuple(pool_partial.shape)}, expected {(b, sw.c_out, nblk)}')
    st = _hip.lib.hs_gemm_split_conv2x2_fwd(_hip.dev_ptr(sw.frag, 'w_frag', torch.float16), _hip.dev_ptr(sw.inv, 'w_inv'),
                                            _hip.dev_ptr(x, 'x'), _hip.dev_ptr(shift, 'shift') if shift is not None else None,
                                            int(act), _hip.dev_ptr(out, 'out'),
                                            _hip.dev_ptr(pool_partial, 'pool_partial') if pool_partial is not None else None,
                                            b, sw.c_out, cin, sw.kp, h // 2, w // 2, _hip.stream_ptr())
    _hip.check(st, 'hs_gemm_split_conv2x2_fwd')
    return out


@_on_operand_device
def gemm_split_up2(sw, x, shift=None, act=ACT_NONE, out=None):
    """nearest-2x upsample of act(W @ x + shift): (B, Cin, H, W) -> (B, Cout, 2H, 2W), the small map never stored
    (hs_gemm_split_up2_fwd).  ``out``: e.g. the right half of the context head's signal."""
    b, cin, h, w = x.shape
    if cin != sw.c_in:
        raise ValueError(f'input has {cin} channels, the weight {sw.c_in}')
    shape = (b, sw.c_out, 2 * h, 2 * w)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != shape:
        raise ValueError(f'out has shape {tuple(out.shape)}, expected {shape}')
    if shift is not None and shift.numel() != sw.c_out:
        raise ValueError(f'shift has {shift.numel()} entries, expected {sw.c_out}')
    st = _hip.lib.hs_gemm_split_up2_fwd(_hip.dev_ptr(sw.frag, 'w_frag', torch.float16), _hip.dev_ptr(sw.inv, 'w_inv'),
                                        _hip.dev_ptr(x, 'x'), _hip.dev_ptr(shift, 'shift') if shift is not None else None,
                                        int(act), _hip.dev_ptr(out, 'out'), b, sw.c_out, cin, sw.kp, h, w, _hip.stream_ptr())
    _hip.check(st, 'hs_gemm_split_up2_fwd')
    return out


@_on_operand_device
def pooled_shift(pool_partial, pixels, wb, shift):
    """shift + wb @ mean, mean = the global average whose 16-pixel block sums ``pool_partial`` (1, C, nblk) holds
    (hs_pooled_shift_fwd): the pooled half of ``cat(feat, pooled.expand_as(feat))`` in front of a 1x1 conv, as a row constant."""
    _, c, nblk = pool_partial.shape
    m = shift.numel()
    if pool_partial.shape[0] != 1 or tuple(wb.shape) != (m, c):
        raise ValueError(f'pool_partial {tuple(pool_partial.shape)} / wb {tuple(wb.shape)} / shift {m}: expected (1, C, nblk), (M, C), M')
    out = torch.empty(m, device=shift.device, dtype=torch.float32)
    st = _hip.lib.hs_pooled_shift_fwd(_hip.dev_ptr(pool_partial, 'pool_partial'), nblk, 1.0 / pixels, _hip.dev_ptr(wb, 'wb'),
                                      _hip.dev_ptr(shift, 'shift'), _hip.dev_ptr(out, 'shift_out'), m, c, _hip.stream_ptr())
    _hip.check(st, 'hs_pooled_shift_fwd')
    return out


@_on_operand_device
def upsample_bilinear(x, size, out=None):
    """F.interpolate(x, size, mode='bilinear', align_corners=False).  ``out``: a contiguous (B, C, *size) f32 tensor on x's device
    to write into (a collective's send slot: hyperseg_amd.distributed.LogitsGatherer.slot); ignored if it does not fit."""
    b, c, hi, wi = x.shape
    ho, wo = size
    if out is not None and tuple(out.shape) == (b, c, ho, wo) and out.dtype == torch.float32 and out.device == x.device and out.is_contiguous():
        y = out
    else:
        y = torch.empty(b, c, ho, wo, device=x.device, dtype=torch.float32)
    st = _hip.lib.hs_upsample_bilinear_fwd(_hip.dev_ptr(x, 'x'), b, c, hi, wi, ho, wo, y.data_ptr(),
                                           _hip.stream_ptr())
    _hip.check(st, 'hs_upsample_bilinear_fwd')
    return y


@_on_operand_device
def upsample_argmax(x, size):
    """``F.interpolate(x, size, 'bilinear', align_corners=False).argmax(1)`` as uint8 masks (B, Ho, Wo), one launch; the
    upsampled logits never exist in memory.  Bit-identical to ``upsample_bilinear(x, size).argmax(1)``."""
    b, c, hi, wi = x.shape
    ho, wo = size
    mask = torch.empty(b, ho, wo, device=x.device, dtype=torch.uint8)
    st = _hip.lib.hs_upsample_argmax_fwd(_hip.dev_ptr(x, 'x'), b, c, hi, wi, ho, wo, mask.data_ptr(), _hip.stream_ptr())
    _hip.check(st, 'hs_upsample_argmax_fwd')
    return mask


# ------------------------------------------------------------------------------------------
# small caches keyed on parameter versions (host-side only; used by the fused inference route --
# tensors that require grad take the hyperseg_amd.autograd route, which folds nothing)
# ------------------------------------------------------------------------------------------
_WEIGHTS_EPOCH = [0]


def bump_weights_epoch():
    """Invalidates every host-side cache derived from parameters or BatchNorm statistics (folded BN affines, transposed /
    packed signal2weights weights, the prepared encoder's folded and split weights).  The caches are keyed on
    (address, tensor._version, shape) -- but a HIP-graph replay of a training step, this package's raw-pointer BatchNorm
    kernels and stock BatchNorm's running-statistics update all change values WITHOUT bumping ``_version``.
    Called by ``training.GraphedTrainStep.step``, by ``autograd.bn_act`` and by every train() / eval() switch of the
    package's modules, so a validation pass after any number of training steps re-derives everything once."""
    with _CACHE_LOCK:
        _WEIGHTS_EPOCH[0] += 1


def _key(*tensors):
    return (_WEIGHTS_EPOCH[0],) + tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)


class FoldedBN:
    """Per-module cache of the folded (scale, shift) of an eval-mode nn.BatchNorm2d."""

    def __init__(self):
        self._ent = {}          # device -> (key, value): one immutable pair per device, replaced in ONE assignment (see _CACHE_LOCK)

    def get(self, bn):
        if bn.training or not bn.track_running_stats:
            raise NotImplementedError('hyperseg_amd: a train-mode BatchNorm cannot be folded into a kernel epilogue; '
                                      'the training route (hyperseg_amd.autograd) keeps BatchNorm as a module')
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        k = _key(*ts)
        dev = bn.weight.device
        ent = self._ent.get(dev)
        if ent is None or ent[0] != k:
            with torch.no_grad():
                ent = (k, bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps), producer_stream(dev))
            publish_ready(dev)
            self._ent[dev] = ent
            return ent[1]
        return adopt(ent[1], dev, ent[2])


class TransposedS2W:
    """Per-module cache of the signal2weights Conv2d weight transposed to (Cs/G, Wc)."""

    def __init__(self):
        self._ent = {}          # device -> (key, value), as FoldedBN

    def get(self, conv):
        w = conv.weight
        k = _key(w)
        ent = self._ent.get(w.device)
        if ent is None or ent[0] != k:
            with torch.no_grad():
                ent = (k, w.detach().reshape(w.shape[0], w.shape[1]).t().contiguous(), producer_stream(w.device))
            publish_ready(w.device)
            self._ent[w.device] = ent
            return ent[1]
        return adopt(ent[1], w.device, ent[2])

"""Train-step counterparts of the reference's training harness for config 5 ("harness counterparts
(ii)"): the loss, the LR policy and the step that drive the decoder's autograd path.  Plain PyTorch -- the HIP work happens
inside ``model(x)`` / ``backward()`` through hyperseg_amd.autograd.

* :class:`BootstrappedCrossEntropyLoss` -- hyperseg/losses/bootstrapped_ce_loss.py:8-40: per IMAGE (not per batch), the
  pixel-wise cross entropy is sorted in descending order; if the (k+1)-th largest loss exceeds ``thresh`` every pixel above
  ``thresh`` is kept, otherwise the k largest; the image's loss is the mean of what is kept; the batch loss is the mean of
  the image losses.  ``ignore_index`` pixels contribute a loss of exactly 0 and stay in the ranking.
* :class:`PolyLR` -- hyperseg/utils/polylr.py:4-22: lr = base_lr * (1 - step/max_step)**power, stepped per batch.
* :func:`train_step` -- hyperseg/train.py:118-136: forward, resize the prediction to the target if needed, loss,
  zero_grad / backward / optimizer.step / scheduler.step.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim.lr_scheduler import LRScheduler


USE_HIP_BOOTSTRAP = True       # tests switch it off to reach the torch statements of the rule below
USE_FUSED_LOSS = True          # False: cross entropy and the bootstrapped mean as two Functions (two launches backward; the fused one's equality test)


def bootstrap_mean_reference(per_pixel, k, thresh):
    """The reference's own statement (hyperseg/losses/bootstrapped_ce_loss.py:19-25): two host reads and a full sort per image."""
    ranked = per_pixel.sort(descending=True).values
    kept = ranked[ranked > thresh] if ranked[k] > thresh else ranked[:k]
    return kept.mean()


def bootstrap_mean_on_device(per_pixel, k, thresh):
    """The same rule with ONE host read and, in the common case, no sort: the (k+1)-th largest loss exceeds ``thresh`` exactly when
    more than k losses do, so the count decides the branch -- the mean of everything above ``thresh`` is a masked sum, and only the
    other branch needs the k largest (``topk``).  The reference reads the host twice per image (the comparison, then a boolean
    index) around a full sort of 3e5 losses (18 merge launches, ~0.2 ms per image at config 5).  Values within 1e-6 and gradients
    equal to the reference statement (tests/test_oracle_golden.py)."""
    over = per_pixel > thresh
    count = over.sum()
    if int(count) > k:
        return (per_pixel * over).sum() / count
    return per_pixel.topk(k, sorted=False).values.mean()


def bootstrap_mean_capturable(per_pixel, k, thresh):
    """The rule with NO host read, for a step that is being captured into a HIP graph (GraphedTrainStep): both branches are formed
    (``topk(k + 1)`` -- on ROCm a full merge sort -- and the masked mean) and ``where`` selects; the gradient reaches the selected
    branch only, as in the reference."""
    top = per_pixel.topk(k + 1, sorted=True).values
    over = per_pixel > thresh
    mean_over = (per_pixel * over).sum() / over.sum().clamp(min=1)
    return torch.where(top[k] > thresh, mean_over, top[:k].mean())


def bootstrapped_cross_entropy(pred, target, k=4096, thresh=0.3, weight=None, ignore_index=-100):
    """pred (N, C, H, W) logits, target (N, H, W) int64 -> scalar."""
    capturing = pred.is_cuda and torch.cuda.is_current_stream_capturing()
    # the per-pixel losses of the WHOLE batch in one pass over (N, C, H, W) (the reference permutes every image to (HW, C) first,
    # bootstrapped_ce_loss.py:20-23: same values, a transposed copy + a softmax + a gather per image and direction)
    hip_ce = (USE_HIP_BOOTSTRAP and weight is None and pred.is_cuda and pred.dtype in (torch.float32, torch.bfloat16) and pred.dim() == 4
              and target.dtype == torch.int64 and target.device == pred.device)
    if hip_ce and USE_FUSED_LOSS and pred.shape[2] * pred.shape[3] > k and pred.shape[0] <= 65535 and pred.shape[2] * pred.shape[3] < 2 ** 31:
        from .autograd import BootstrappedCrossEntropy                     # the whole loss as one Function: its adjoint is one launch (round 6)
        return BootstrappedCrossEntropy.apply(pred, target, ignore_index, k, thresh)
    if hip_ce:
        from .autograd import PixelCrossEntropy                            # one launch per direction (hs_cross_entropy_typed_fwd / _bwd)
        per_all = PixelCrossEntropy.apply(pred, target, ignore_index)
    else:
        per_all = F.cross_entropy(pred.float(), target, weight=weight, ignore_index=ignore_index, reduction='none')
    per_rows = per_all.flatten(1)
    if (USE_HIP_BOOTSTRAP and per_rows.is_cuda and per_rows.dtype == torch.float32 and per_rows.shape[1] > k and per_rows.shape[0] <= 65535):
        from .autograd import BootstrapMeanOfBatch                         # every image in one set of launches, no sort, no host read; the
        return BootstrapMeanOfBatch.apply(per_rows, k, thresh)             # batch mean from the same launches (no sum / div launches)
    total = pred.new_zeros((), dtype=torch.float32)
    for per_pixel in per_rows:
        on_device = per_pixel.is_cuda and per_pixel.numel() > k            # (numel <= k: the reference raises; so does its restatement)
        if on_device and per_pixel.dtype == torch.float32 and USE_HIP_BOOTSTRAP:
            from .autograd import BootstrapMean                            # no sort, no host read, 7 small launches: eager and captured alike
            total = total + BootstrapMean.apply(per_pixel, k, thresh)
            continue
        fn = bootstrap_mean_capturable if capturing and on_device else bootstrap_mean_on_device if on_device else bootstrap_mean_reference
        total = total + fn(per_pixel, k, thresh)
    return total / float(pred.shape[0])


class BootstrappedCrossEntropyLoss(nn.Module):
    def __init__(self, k=4096, thresh=0.3, weight=None, ignore_index=-100, reduction='mean'):
        super().__init__()
        if reduction != 'mean':
            raise ValueError("only reduction='mean' is meaningful: the reference stores the argument and always averages")
        self.k, self.thresh, self.ignore_index = k, thresh, ignore_index
        self.register_buffer('weight', weight)

    def forward(self, input, target):
        return bootstrapped_cross_entropy(input, target, self.k, self.thresh, self.weight, self.ignore_index)


class PolyLR(LRScheduler):
    def __init__(self, optimizer, max_epoch, power=0.9, last_epoch=-1):
        self.max_epoch, self.power = max_epoch, power
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        decay = (1.0 - float(self.last_epoch) / float(self.max_epoch)) ** self.power
        return [base * decay for base in self.base_lrs]


class Adam(torch.optim.Optimizer):
    """``torch.optim.Adam`` / ``AdamW`` arithmetic (no amsgrad) with the whole parameter list of a group updated by ONE launch of
    1024-element workgroups (``hs_adam_step``; round 5).  The reference trains with ``optim.Adam(lr, betas=(0.5, 0.999))``
    (hyperseg/train.py:185-188, configs/train/*); torch's fused form is one launch too, but it cuts the list into 65 536-element
    chunks -- a dozen workgroups for the decoder's ~0.7 M parameters, 25 us of a 0.88 ms config-5 step.

    Capturable by construction: the step count lives on the device (one word per workgroup, incremented by the kernel), ``lr`` may be a
    float or a one-element CUDA tensor (what a scheduler updates in place under a captured step, ``GraphedTrainStep``).  fp32 CUDA
    parameters with dense fp32 gradients only; every parameter of a group that has a gradient takes part, and the SET of those
    parameters must not change between steps (the per-workgroup step words are laid out for it; a change raises)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled_weight_decay=False, maximize=False):
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0.0 or weight_decay < 0.0:
            raise ValueError('hyperseg_amd.training.Adam: invalid hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                      decoupled_weight_decay=bool(decoupled_weight_decay), maximize=bool(maximize)))

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes as C
        from . import _hip
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32
                        and not p.grad.is_sparse and p.grad.device == p.device):
                    raise NotImplementedError('hyperseg_amd.training.Adam: contiguous fp32 CUDA parameters with dense fp32 gradients only')
            dev = ps[0].device
            for lo in range(0, len(ps), 48):                                   # hs_adam_step takes up to 48 tensors per launch
                chunk = ps[lo:lo + 48]
                for p in chunk:
                    st = self.state[p]
                    if not st:
                        st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                n = len(chunk)
                numel = (C.c_int64 * n)(*[p.numel() for p in chunk])
                key = f'_hs_steps_{lo}'
                sig = tuple(p.numel() for p in chunk)
                held = group.get(key)
                if held is None:
                    blocks = int(_hip.lib.hs_adam_blocks(numel, n))
                    if blocks <= 0:
                        raise NotImplementedError('hyperseg_amd.training.Adam: parameter list not covered by hs_adam_step')
                    # a loaded state (ours or a torch.optim.Adam checkpoint, train.py:227) carries the steps taken as state[p]['step']:
                    # the bias corrections must continue from there, not restart at t = 1 on warm moments
                    seeds = {float(self.state[p]['step']) for p in chunk if 'step' in self.state[p]}
                    if len(seeds) > 1:
                        raise NotImplementedError('hyperseg_amd.training.Adam: the parameters of one launch must share one step count')
                    held = group[key] = (sig, torch.full((blocks,), seeds.pop() if seeds else 0.0, device=dev, dtype=torch.float32))
                elif held[0] != sig:
                    raise RuntimeError('hyperseg_amd.training.Adam: the set of parameters with gradients changed between steps')
                elif held[1].device != dev:
                    raise RuntimeError('hyperseg_amd.training.Adam: the step words live on another device than the parameters')
                chunks = self.__dict__.setdefault('_hs_chunks', {})
                for p in chunk:
                    chunks[p] = (id(group), key)
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in chunk]
                arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])      # noqa: E731
                lr = group['lr']
                lr_dev = lr if isinstance(lr, torch.Tensor) and lr.is_cuda else None
                if lr_dev is not None and (lr_dev.dtype != torch.float32 or lr_dev.numel() != 1):
                    raise NotImplementedError('hyperseg_amd.training.Adam: a device learning rate must be one fp32 element')
                with _hip.device_scope(dev):
                    status = _hip.lib.hs_adam_step(arr(chunk), arr(grads), arr([self.state[p]['exp_avg'] for p in chunk]),
                                                   arr([self.state[p]['exp_avg_sq'] for p in chunk]), numel, n,
                                                   lr_dev.data_ptr() if lr_dev is not None else None, 0.0 if lr_dev is not None else float(lr),
                                                   float(group['betas'][0]), float(group['betas'][1]), float(group['eps']), float(group['weight_decay']),
                                                   int(group['decoupled_weight_decay']), int(group['maximize']), held[1].data_ptr(), _hip.stream_ptr(dev))
                _hip.check(status, 'hs_adam_step')
        return loss

    def _detached_step_words(self):
        """Removes the per-workgroup step words from the param groups (they are launch-layout state, not checkpoint state) and returns
        what it removed plus the count each one holds."""
        removed, counts = [], {}
        for g in self.param_groups:
            for k in [k for k in g if isinstance(k, str) and k.startswith('_hs_steps_')]:
                held = g.pop(k)
                removed.append((g, k, held))
                counts[(id(g), k)] = float(held[1][0].item())
        return removed, counts

    def state_dict(self):
        """torch.optim.Adam's checkpoint format: per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``, param groups without the device
        step words -- loadable by torch.optim.Adam and by this class on any device (synchronises: reads the step count)."""
        removed, counts = self._detached_step_words()
        try:
            for p, where in self.__dict__.get('_hs_chunks', {}).items():
                if where in counts and self.state.get(p):
                    self.state[p]['step'] = torch.tensor(counts[where], dtype=torch.float32)
            return super().state_dict()
        finally:
            for g, k, held in removed:
                g[k] = held

    def load_state_dict(self, state_dict):
        """Accepts this class' checkpoints and torch.optim.Adam's (hyperseg/train.py:227 resumes from one): moments go to the
        parameters' device (torch does that), the step count is re-seeded onto the device by the next step() from ``state[p]['step']``."""
        super().load_state_dict(state_dict)
        for g in self.param_groups:                                              # step words of an older checkpoint format: never trusted,
            for k in [k for k in g if isinstance(k, str) and k.startswith('_hs_steps_')]:      # they may sit on the wrong device
                held = g.pop(k)
                try:
                    count = float(held[1].reshape(-1)[0])
                except Exception:                                                # noqa: BLE001
                    continue
                for p in g['params']:
                    if self.state.get(p) and 'step' not in self.state[p]:
                        self.state[p]['step'] = torch.tensor(count, dtype=torch.float32)
        self.__dict__['_hs_chunks'] = {}

    def steps_taken(self, group=0):
        """The step count of a group as the device holds it (synchronises)."""
        held = self.param_groups[group].get('_hs_steps_0')
        return 0 if held is None else int(held[1][0].item())


def train_step(model, criterion, optimizer, scheduler, x, target):
    """One optimisation step; returns (loss, prediction).  ``model`` is any callable producing (N, C, h, w) logits."""
    pred = model(x)
    if pred.shape[2:] != target.shape[1:]:
        pred = F.interpolate(pred, size=target.shape[1:], mode='bilinear')
    loss = criterion(pred, target)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach(), pred.detach()


def _flat_tensors(obj):
    if isinstance(obj, torch.Tensor):
        return [obj]
    if isinstance(obj, (list, tuple)):
        return [t for o in obj for t in _flat_tensors(o)]
    return []


class GraphedTrainStep:
    """forward + loss + backward + optimizer step of a fixed-shape batch as ONE HIP graph (the training-side counterpart of
    utils.inference.GraphedModel).  An eager config-5 decoder step is ~300 launches whose host side (4.8 ms) is longer than their GPU
    time (3.4 ms): replaying the captured step removes the difference.

    ``model(*inputs)`` -> logits, ``criterion(logits, target)`` -> scalar.  ``inputs`` (tensors or nested lists of tensors) and
    ``target`` given here are the STATIC buffers the graph reads: ``step(inputs, target)`` copies new data into them and replays.
    The optimizer must be capturable (``torch.optim.Adam(..., capturable=True)``: its step count lives on the device); a learning-rate
    schedule changes ``param_group['lr']`` in place if it is a tensor (``lr=torch.tensor(1e-3, device=...)``).  BatchNorm statistics,
    parameters and optimizer state are updated by every replay exactly as by an eager step.

    One requirement beyond PyTorch's usual ones (static shapes, warm-up before capture): NO autograd graph of an earlier eager step
    may still be alive when this object is built -- e.g. a ``loss`` variable that still carries its ``grad_fn``.  Such a graph keeps the
    parameters' gradient accumulators alive, and those stay bound to the stream of the eager steps; the captured backward would then
    hop to that (non-capturing) stream and hipStreamEndCapture dies on ROCm 7.2 (bisected in round 3: ``del loss`` is the whole fix).
    Keep ``loss.detach()`` / ``float(loss)`` instead."""

    def __init__(self, model, criterion, optimizer, inputs, target, warmup=3):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.inputs, self.target = inputs, target
        self._unit = None
        dev = target.device
        if dev.type != 'cuda':
            raise ValueError('GraphedTrainStep needs CUDA tensors')
        self._static_in = _flat_tensors(inputs) + [target]
        if warmup < 1 and not optimizer.state:
            # the optimizer creates its state (moments, step count) on first use: inside a capture those zero fills become graph nodes
            # and EVERY replay would reset the moments
            raise ValueError('GraphedTrainStep: warmup >= 1 is required while the optimizer has no state yet')
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # first calls allocate, tune and set kernel attributes: not capturable
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        # gradients left by earlier eager steps (the caller's or the warm-up's) would be FREED inside the capture by zero_grad -- blocks
        # of another stream's pool released while a global-mode capture is open bring hipStreamEndCapture down on ROCm 7.2: drop them now
        self.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.pred = self._eager()

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        pred = self.model(*self.inputs) if isinstance(self.inputs, (list, tuple)) else self.model(self.inputs)
        if pred.shape[2:] != self.target.shape[1:]:
            pred = F.interpolate(pred, size=self.target.shape[1:], mode='bilinear')
        loss = self.criterion(pred, self.target)
        # the root gradient is a tensor this object owns (made in the warm-up, outside the capture): `loss.backward()` alone makes autograd
        # fill a fresh ones_like(loss) -- one more launch in every replay (round 6: 82 -> 81 per config-5 step)
        if self._unit is None or self._unit.shape != loss.shape or self._unit.dtype != loss.dtype:
            self._unit = torch.ones_like(loss)
        loss.backward(self._unit)
        self.optimizer.step()
        return loss.detach(), pred.detach()

    def step(self, inputs=None, target=None):
        """Copies ``inputs`` / ``target`` (same structure and shapes; None = keep what is in the static buffers) and replays.
        Returns the static (loss, prediction) tensors: valid until the next call."""
        if inputs is not None or target is not None:
            new = (_flat_tensors(inputs) if inputs is not None else self._static_in[:-1]) + [target if target is not None else self.target]
            if len(new) != len(self._static_in):
                raise ValueError('inputs do not match the captured structure')
            for dst, src in zip(self._static_in, new):
                if dst is not src:
                    dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # the replay moved every parameter and BatchNorm buffer WITHOUT bumping a tensor version: the eval-route caches keyed on
        # versions (folded BN, transposed / packed signal2weights weights, split GEMM weights) must not survive it
        from .functional import bump_weights_epoch
        bump_weights_epoch()
        return self.loss, self.pred

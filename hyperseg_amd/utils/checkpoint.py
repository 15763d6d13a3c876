"""Checkpoint tooling with the reference's on-disk contract (hyperseg/utils/utils.py:61-181, train.py:264-274,
test.py:67-101): a checkpoint is a dict ``{'epoch', 'state_dict', 'optimizer', 'scheduler', 'best_iou', 'arch'}`` where
``arch`` is the python expression that rebuilds the model through ``obj_factory`` and ``state_dict`` carries no
``module.`` prefixes.  Reference checkpoints load into this package's models unchanged (same state-dict keys, the arch
string's ``hyperseg.`` module paths are redirected by :mod:`hyperseg_amd.utils.obj_factory`), and checkpoints written
here load in the reference.
"""
import os
import shutil
from collections import OrderedDict
from functools import partial

import torch

from .obj_factory import _split, obj_factory, reference_name


def remove_data_parallel_from_state_dict(state_dict):
    """Keys of a model that was wrapped in nn.DataParallel, without the wrapper's ``module.`` (utils.py:76-82)."""
    return OrderedDict((k.replace('module.', ''), v) for k, v in state_dict.items())


def save_checkpoint(exp_dir, base_name, state, is_best=False):
    """``<exp_dir>/<base_name>_latest.pth`` (+ a copy as ``_best.pth``); DataParallel prefixes are stripped from
    ``state['state_dict']`` (utils.py:61-73)."""
    path = os.path.join(exp_dir, base_name + '_latest.pth')
    if 'state_dict' in state:
        state = dict(state, state_dict=remove_data_parallel_from_state_dict(state['state_dict']))
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(exp_dir, base_name + '_best.pth'))
    return path


def get_arch(obj, *args, eval_partial=True, **kwargs):
    """The ``arch`` string of an object given as an expression string or a ``functools.partial``, with extra
    (keyword) arguments appended: ``'pkg.mod.fn(arg,...,key=value,...)'`` without spaces -- what train.py stores in
    every checkpoint and ``obj_factory`` turns back into the object (utils.py:96-144).  Nested partials are rendered as
    ``functools.partial('pkg.mod.fn',...)`` like the reference does; anything else returns None.  Partials of this
    package's functions are written under the REFERENCE's module names (``hyperseg.models...``), so the strings equal the
    reference's for the same config (tests/golden/checkpoint_ref.npz, made by the reference's own get_arch).  String
    inputs: the reference raises for them (utils.py:116 evals a name it never imports; utils.py:126 adds a tuple to a
    list) although train.py's ``model`` argument may be a string -- here they work as the docstring there promises."""
    if isinstance(obj, str):
        if '(' in obj and ')' in obj:
            _, own_args, own_kwargs = _split(obj)
            func = obj[:obj.find('(')]
        else:
            func, own_args, own_kwargs = obj, (), {}
    elif isinstance(obj, partial):
        func = reference_name(obj.func.__module__, obj.func.__name__)       # always the reference's namespace
        own_args, own_kwargs = obj.args, obj.keywords
    else:
        return None
    pos = [get_arch(o, eval_partial=False) if isinstance(o, partial) else o for o in tuple(own_args) + args]
    named = {k: get_arch(v, eval_partial=False) if isinstance(v, partial) else v for k, v in {**own_kwargs, **kwargs}.items()}
    if not eval_partial:
        pos.insert(0, func)
        func = 'functools.partial'
    parts = [repr(o) for o in pos] + [f'{k}={v!r}' for k, v in named.items()]
    return f"{func}({','.join(parts)})".replace(' ', '')


def _build_without_download(arch):
    """``obj_factory(arch)`` with ``pretrained`` forced off.  train.py stores the config's partial verbatim, so every
    released checkpoint says ``pretrained=True``; the reference then downloads the ImageNet backbone and overwrites it
    with the checkpoint's state dict one line later (utils.py:174-177).  The state dict supplies every weight, so the
    download is skipped (there may be no network)."""
    if not isinstance(arch, str) or '(' not in arch:
        return obj_factory(arch)
    fn, args, kwargs = _split(arch)
    if 'pretrained' in kwargs:
        kwargs['pretrained'] = False
    elif len(args) >= 2 and isinstance(args[1], bool) and getattr(fn, '__name__', '').endswith('efficientnet'):
        args = (args[0], False) + tuple(args[2:])
    return fn(*args, **kwargs)


def load_model(model_path, name='', device=None, arch=None, return_checkpoint=False, train=False, trusted=False):
    """Model from a checkpoint: ``obj_factory(checkpoint['arch'])`` + ``load_state_dict`` (strict), eval mode unless
    ``train`` (utils.py:147-181).  The file is read with ``torch.load(weights_only=True)``: tensors, containers and plain
    Python scalars / strings only.  A checkpoint that pickles other objects (an optimizer or scheduler state saved by an
    old torch, say) is refused with the unpickler's own message unless the caller vouches for the file with
    ``trusted=True`` -- a full pickle load executes code from the file, so it is never taken silently."""
    if model_path is None:
        raise AssertionError(f'{name} model must be specified!')
    if not os.path.exists(model_path):
        raise AssertionError(f"Couldn't find {name} model in path: {model_path}")
    checkpoint = torch.load(model_path, map_location='cpu', weights_only=not trusted)
    if arch is None and 'arch' not in checkpoint:
        raise AssertionError(f"Couldn't determine {name} model architecture!")
    arch = checkpoint['arch'] if arch is None else arch
    model = _build_without_download(arch)
    if device is not None:
        model.to(device)
    model.load_state_dict(remove_data_parallel_from_state_dict(checkpoint['state_dict']))
    model.train(train)
    return (model, checkpoint) if return_checkpoint else model

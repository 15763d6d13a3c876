"""Checkpoint tooling (reference utils.py:61-181, train.py:264-274): arch strings round-trip through
get_arch / obj_factory, save_checkpoint strips DataParallel prefixes and writes the reference's file names, load_model
rebuilds the model from the stored arch and loads it strictly."""
import os
from functools import partial

import pytest
import torch

from hyperseg_amd import configs
from hyperseg_amd.utils.checkpoint import get_arch, load_model, remove_data_parallel_from_state_dict, save_checkpoint
from hyperseg_amd.utils.obj_factory import obj_factory
from hyperseg_amd.utils.synthetic import fill_by_name


def test_get_arch_matches_the_reference_format():
    from hyperseg_amd.models.hyperseg_v1_0 import hyperseg_efficientnet
    p = partial(hyperseg_efficientnet, 'efficientnet-b1', False, levels=2, kernel_sizes=[1, 1, 1, 3, 3],
                level_channels=[64, 32, 16, 16, 16], expand_ratio=2, weight_groups=[32, 16, 8, 16, 4],
                coords_res=[(512, 512), (512, 1024)])
    arch = get_arch(p, num_classes=19)
    assert arch == ("hyperseg_amd.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1',False,levels=2,"
                    "kernel_sizes=[1,1,1,3,3],level_channels=[64,32,16,16,16],expand_ratio=2,weight_groups=[32,16,8,16,4],"
                    "coords_res=[(512,512),(512,1024)],num_classes=19)")
    # a string expression keeps its own arguments first, extra ones are appended (utils.py:131-133)
    s = get_arch("hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1', levels=2)", num_classes=3)
    assert s == "hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1',levels=2,num_classes=3)"
    assert get_arch('torch.nn.ReLU') == 'torch.nn.ReLU()'
    assert get_arch(partial(torch.optim.Adam, lr=1e-3, betas=(0.5, 0.999))) == 'torch.optim.adam.Adam(lr=0.001,betas=(0.5,0.999))'
    assert get_arch(42) is None
    # nested partials are rendered like the reference does: as the repr of their own arch string (utils.py:135-143)
    assert get_arch(partial(max, partial(min, 1))) == "builtins.max(\"functools.partial('builtins.min',1)\")"


@pytest.mark.parametrize('name', ['hyperseg-m', 'hyperseg-l'])
def test_checkpoint_round_trip_cpu(tmp_path, name):
    """A reference-style checkpoint (DataParallel-prefixed keys, ``arch`` with the reference's module paths) -> load_model:
    same class, identical state dict, eval mode; file names as train.py writes them."""
    spec = configs.MODELS[name]
    arch = get_arch(spec['arch'], num_classes=spec['num_classes'])
    assert arch.startswith('hyperseg.models.')
    src = fill_by_name(obj_factory(arch).eval(), seed=5)
    state = {'epoch': 3, 'state_dict': {'module.' + k: v for k, v in src.state_dict().items()},
             'optimizer': torch.optim.Adam(src.parameters(), lr=1e-3, betas=(0.5, 0.999)).state_dict(),
             'scheduler': None, 'best_iou': 0.5, 'arch': arch}
    path = save_checkpoint(str(tmp_path), 'model', state, is_best=True)
    assert os.path.basename(path) == 'model_latest.pth' and os.path.exists(tmp_path / 'model_best.pth')
    assert all(k.startswith('module.') for k in state['state_dict'])          # the caller's dict is left alone
    model, ckpt = load_model(path, 'test', return_checkpoint=True)
    assert type(model) is type(src) and not model.training
    assert ckpt['epoch'] == 3 and ckpt['arch'] == arch and not any(k.startswith('module.') for k in ckpt['state_dict'])
    a, b = model.state_dict(), src.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert load_model(path, train=True).training
    with pytest.raises(AssertionError):
        load_model(str(tmp_path / 'missing.pth'), 'test')
    assert list(remove_data_parallel_from_state_dict({'module.a.b': 1, 'c': 2})) == ['a.b', 'c']


class _Marker:                      # a class the safe unpickler has never heard of
    pass


def test_load_model_never_unpickles_silently(tmp_path):
    """a checkpoint that needs the full pickle loader (an arbitrary object inside) is REFUSED by default -- no silent
    retry with weights_only=False -- and loads only when the caller vouches for the file (trusted=True)."""
    import pickle
    spec = configs.MODELS['hyperseg-m']
    arch = get_arch(spec['arch'], num_classes=spec['num_classes'])
    src = obj_factory(arch).eval()
    path = str(tmp_path / 'with_object.pth')
    torch.save({'state_dict': src.state_dict(), 'arch': arch, 'extra': _Marker()}, path)
    with pytest.raises(pickle.UnpicklingError):
        load_model(path, 'test')
    assert type(load_model(path, 'test', trusted=True)) is type(src)


@pytest.mark.gpu
def test_loaded_model_reproduces_the_logits(tmp_path):
    dev = torch.device('cuda:0')
    spec = configs.MODELS['hyperseg-m']
    arch = get_arch(spec['arch'], num_classes=spec['num_classes'])
    src = fill_by_name(obj_factory(arch).eval(), seed=6)
    path = save_checkpoint(str(tmp_path), 'model', {'state_dict': {'module.' + k: v for k, v in src.state_dict().items()}, 'arch': arch})
    model = load_model(path, 'hyperseg-m', device=dev)
    x = torch.rand(1, 3, 256, 512, device=dev)
    with torch.no_grad():
        a, b = model(x), src.to(dev)(x)
    # same weights, same kernels; MIOpen may still pick different convolution algorithms for the two instances
    assert float((a - b).abs().max() / b.abs().max()) < 1e-5

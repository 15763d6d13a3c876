"""Checkpoint tooling (reference utils.py:61-181, train.py:203, 264-274) held to the REFERENCE's own
outputs: ``tests/golden/checkpoint_ref.npz`` carries the strings the reference's ``get_arch`` returned for the config
files' partials and ``tests/golden/ref_ckpt_{latest,best}.pth`` is a checkpoint FILE the reference's ``save_checkpoint``
wrote (``make_golden.py gen_checkpoint``).  Here: the build's get_arch gives the same strings, its save_checkpoint the same
file names / dict layout / key list, its load_model loads the reference-written file (DataParallel keys stripped by the
writer, arch in the reference's namespace) and -- on the GPU -- the loaded decoder reproduces the reference's logits."""
import os
from functools import partial

import pytest
import torch

from conftest import G

from hyperseg_amd import configs
from hyperseg_amd.utils.checkpoint import get_arch, load_model, remove_data_parallel_from_state_dict, save_checkpoint
from hyperseg_amd.utils.obj_factory import obj_factory
from hyperseg_amd.utils.synthetic import fill_by_name

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _config_partials():
    """The model partials of configs/train/*.py, kwargs verbatim (incl. ``pretrained=True`` by keyword), on THIS package's
    factory functions."""
    from hyperseg_amd.models import hyperseg_v0_1 as v0, hyperseg_v1_0 as v1, hyperseg_v1_0_unify as vu
    return {
        'M': (partial(v1.hyperseg_efficientnet, 'efficientnet-b1', pretrained=True, levels=2,
                      out_feat_scale=[1., 0.25, 0.25, 0.25, 0.25], kernel_sizes=[1, 1, 1, 3, 3],
                      level_channels=[64, 32, 16, 16, 16], expand_ratio=2, with_out_fc=False, decoder_dropout=None,
                      weight_groups=[32, 16, 8, 16, 4], decoder_groups=1, inference_hflip=True,
                      coords_res=[(512, 512), (512, 1024)]), 19),
        'S': (partial(vu.hyperseg_efficientnet, 'efficientnet-b1', pretrained=True, levels=2,
                      out_feat_scale=[1., 0.166, 0.2, 0.25, 0.4], kernel_sizes=[1, 1, 1, 3, 3],
                      level_channels=[32, 16, 8, 8, 8], expand_ratio=2, with_out_fc=False, decoder_dropout=None,
                      weight_groups=[32, 16, 8, 16, 4], decoder_groups=1, inference_hflip=True, unify_level=4,
                      coords_res=[(768, 768), (768, 1536)]), 19),
        'Sc': (partial(v1.hyperseg_efficientnet, 'efficientnet-b1', pretrained=True, levels=2,
                       kernel_sizes=(1, 1, 1, 3, 3), level_channels=[64, 32, 16, 16, 16], expand_ratio=2,
                       with_out_fc=False, decoder_dropout=None, weight_groups=[64, 32, 32, 16, 8], decoder_groups=1,
                       inference_hflip=True, coords_res=[(576, 576), (576, 768)]), 12),
        'L': (partial(v0.hyperseg_efficientnet, 'efficientnet-b3', pretrained=True, levels=3,
                      kernel_sizes=(1, 1, 3, 3, 3, 3), expand_ratio=2, inference_hflip=True, with_out_fc=False,
                      decoder_dropout=None, weight_groups=16), 21),
    }


def test_get_arch_equals_the_reference(golden):
    """get_arch on the four config partials + num_classes (train.py:203), on the configs' optimizer / scheduler partials
    and on nested / unevaluated partials: string-identical to what the reference's get_arch returned."""
    from hyperseg_amd.training import PolyLR
    g = golden('checkpoint_ref')
    for tag, (p, ncls) in _config_partials().items():
        assert int(g[f'classes.{tag}']) == ncls
        assert get_arch(p, num_classes=ncls) == str(g[f'arch.{tag}']), tag
    assert get_arch(partial(torch.optim.Adam, lr=1e-3, betas=(0.5, 0.999))) == str(g['arch.adam'])
    assert get_arch(partial(PolyLR, power=0.9, max_epoch=90000)) == str(g['arch.polylr'])
    assert get_arch(partial(max, partial(min, 1))) == str(g['arch.nested'])
    assert get_arch(partial(torch.nn.ReLU6, True), eval_partial=False) == str(g['arch.not_eval'])
    assert bool(g['arch.none_is_none']) and get_arch(42) is None
    # string inputs: the reference RAISES for both forms (recorded facts, see get_arch's docstring); the build serves them
    assert str(g['arch.str_args_raises']) == 'NameError' and str(g['arch.str_plain_raises']) == 'TypeError'
    s = get_arch("hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1', levels=2)", num_classes=3)
    assert s == "hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1',levels=2,num_classes=3)"
    assert get_arch('torch.nn.ReLU') == 'torch.nn.ReLU()'


def test_reference_arch_strings_build_the_models(golden):
    """The reference-made arch strings (``pretrained=True`` as every released checkpoint has it) rebuild the models here:
    obj_factory itself refuses to download, load_model's builder switches the download off (the state dict supplies every
    weight) -- and the built model has the state-dict keys the reference's model has (model_{M,S,L}.npz hashes)."""
    import hashlib
    from hyperseg_amd.utils.checkpoint import _build_without_download
    g = golden('checkpoint_ref')
    with pytest.raises(RuntimeError):
        obj_factory(str(g['arch.M']))
    for tag in ('M', 'S', 'L'):
        model = _build_without_download(str(g[f'arch.{tag}']))
        keys = [k for k in model.state_dict() if 'num_batches' not in k]
        h = int.from_bytes(hashlib.sha256(' '.join(keys).encode()).digest()[:7], 'little')
        assert h == int(golden(f'model_{tag}')['key_hash'][0]), tag


def test_loads_a_checkpoint_the_reference_wrote(golden, tmp_path):
    """ref_ckpt_latest.pth was written by the reference's save_checkpoint from a DataParallel-wrapped tiny v1_0 decoder with
    the dict train.py:267-274 stores.  load_model: right class, strict load, eval mode, the stored extras intact; and the
    build's own save_checkpoint writes the same file names, top-level keys and state-dict key list."""
    from hyperseg_amd.models.hyperseg_v1_0 import MultiScaleDecoder
    g = golden('checkpoint_ref')
    path = os.path.join(GOLDEN, 'ref_ckpt_latest.pth')
    model, ck = load_model(path, 'reference-written', return_checkpoint=True)
    assert type(model) is MultiScaleDecoder and not model.training
    assert ck['arch'] == str(g['ckpt.arch']) and ck['epoch'] == 4 and ck['best_iou'] == 0.625
    assert list(ck.keys()) == [str(k) for k in g['ckpt.top_keys']]
    assert list(ck['state_dict'].keys()) == [str(k) for k in g['ckpt.state_keys']]
    assert list(model.state_dict().keys()) == list(ck['state_dict'].keys())
    assert all(torch.equal(model.state_dict()[k], v) for k, v in ck['state_dict'].items())
    assert open(path, 'rb').read() == open(os.path.join(GOLDEN, 'ref_ckpt_best.pth'), 'rb').read()
    # optimizer / scheduler state as torch wrote it: restorable
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.5, 0.999))
    opt.load_state_dict(ck['optimizer'])
    assert ck['scheduler']['max_epoch'] == 10 and ck['scheduler']['last_epoch'] == 1
    # the build's writer: same file names, same layout
    wrapped = torch.nn.DataParallel(model)
    state = {'epoch': 4, 'state_dict': wrapped.state_dict(), 'optimizer': ck['optimizer'], 'scheduler': ck['scheduler'],
             'best_iou': 0.625, 'arch': get_arch(ck['arch'])}
    assert all(k.startswith('module.') for k in state['state_dict'])
    save_checkpoint(str(tmp_path), 'model', state, is_best=True)
    assert sorted(os.listdir(tmp_path)) == [str(f) for f in g['ckpt.files']]
    mine = torch.load(tmp_path / 'model_latest.pth', weights_only=True)
    assert list(mine.keys()) == list(ck.keys()) and mine['arch'] == ck['arch']
    assert list(mine['state_dict'].keys()) == list(ck['state_dict'].keys())
    assert all(torch.equal(mine['state_dict'][k], v) for k, v in ck['state_dict'].items())


@pytest.mark.gpu
def test_reference_written_checkpoint_reproduces_the_reference_logits(golden):
    """The decoder loaded from the reference-written file, run on the HIP path, against the logits the REFERENCE computed
    with those weights on the stored inputs."""
    g = golden('checkpoint_ref')
    dev = torch.device('cuda:0')
    model = load_model(os.path.join(GOLDEN, 'ref_ckpt_latest.pth'), 'reference-written', device=dev)
    x = [g[f'ckpt.x{i}'].to(dev) for i in range(sum(1 for k in g if k.startswith('ckpt.x')))]
    with torch.no_grad():
        y = model(x, g['ckpt.s'].to(dev)).cpu()
    ref = g['ckpt.y']
    assert y.shape == ref.shape
    assert float((y - ref).abs().max() / ref.abs().max()) < 2e-5


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
    x = torch.rand(1, 3, 256, 512, generator=G(1001)).to(dev)
    with torch.no_grad():
        a, b = model(x), src.to(dev)(x)
    # same weights, same kernels; MIOpen may still pick different convolution algorithms for the two instances
    assert float((a - b).abs().max() / b.abs().max()) < 1e-5

"""String / partial -> object factory with the reference's contract (hyperseg/utils/obj_factory.py:39-127).

Checkpoints store the model as an ``arch`` expression such as
``"hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1', levels=2, ...)"`` and the config
files pass ``functools.partial`` objects; both resolve here.  Module paths of the reference package are
redirected to this package, so reference arch strings build the MI355X models unchanged.
"""
import importlib
from functools import partial

KNOWN_MODULES = {
    'hyperseg.models.hyperseg_v1_0': 'hyperseg_amd.models.hyperseg_v1_0',
    'hyperseg.models.hyperseg_v1_0_unify': 'hyperseg_amd.models.hyperseg_v1_0_unify',
    'hyperseg.models.hyperseg_v0_1': 'hyperseg_amd.models.hyperseg_v0_1',
    'hyperseg.models.layers.meta_conv': 'hyperseg_amd.models.layers.meta_conv',
    'hyperseg.models.layers.meta_patch': 'hyperseg_amd.models.layers.meta_patch',
    'hyperseg.models.layers.meta_sequential': 'hyperseg_amd.models.layers.meta_sequential',
    'hyperseg.models.backbones.efficientnet': 'hyperseg_amd.models.backbones.efficientnet',
    'hyperseg.utils.polylr': 'hyperseg_amd.training',
    'hyperseg.losses.bootstrapped_ce_loss': 'hyperseg_amd.training',
    'nn': 'torch.nn',
    'optim': 'torch.optim',
    'lr_scheduler': 'torch.optim.lr_scheduler',
}


# our module (, attribute) -> the reference's module: arch strings are always WRITTEN in the reference's namespace, so a
# checkpoint saved here loads in the reference and vice versa (get_arch in utils/checkpoint.py)
REFERENCE_MODULES = {v: k for k, v in KNOWN_MODULES.items() if k.startswith('hyperseg.') and v != 'hyperseg_amd.training'}
REFERENCE_ATTRS = {('hyperseg_amd.training', 'PolyLR'): 'hyperseg.utils.polylr',
                   ('hyperseg_amd.training', 'BootstrappedCrossEntropyLoss'): 'hyperseg.losses.bootstrapped_ce_loss'}


def reference_name(module_name, attr):
    """'hyperseg_amd.models.hyperseg_v1_0', 'hyperseg_efficientnet' -> 'hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet'."""
    return f"{REFERENCE_ATTRS.get((module_name, attr), REFERENCE_MODULES.get(module_name, module_name))}.{attr}"


def _collect(*args, **kwargs):
    return args, kwargs


def _split(expr):
    """'pkg.mod.Name(a, b=1)' -> (callable, args, kwargs)."""
    args, kwargs = (), {}
    if '(' in expr and expr.rstrip().endswith(')'):
        head, tail = expr[:expr.find('(')], expr[expr.find('('):]
        args, kwargs = eval('_collect' + tail, {'_collect': _collect, '__builtins__': {}})  # literals only
        expr = head
    module_name, _, attr = expr.strip().rpartition('.')
    if not module_name:
        raise ValueError(f'"{expr}" is not of the form package.module.Name')
    module = importlib.import_module(KNOWN_MODULES.get(module_name, module_name))
    return getattr(module, attr), args, kwargs


def obj_factory(obj_exp, *args, **kwargs):
    """Build an object from a string expression or a partial (recursively for sequences); anything else is
    returned untouched.  Extra ``args`` follow the expression's own positional arguments."""
    if isinstance(obj_exp, (list, tuple)):
        return [obj_factory(o, *args, **kwargs) for o in obj_exp]
    if isinstance(obj_exp, partial):
        return obj_exp(*args, **kwargs)
    if not isinstance(obj_exp, str):
        return obj_exp
    fn, a, k = _split(obj_exp)
    return fn(*(a + args), **{**kwargs, **k})


def partial_obj_factory(obj_exp, *args, **kwargs):
    """Like :func:`obj_factory` but returns a ``functools.partial`` instead of calling."""
    if isinstance(obj_exp, (list, tuple)):
        return [partial_obj_factory(o, *args, **kwargs) for o in obj_exp]
    if isinstance(obj_exp, partial):
        return partial(obj_exp.func, *(obj_exp.args + args), **{**obj_exp.keywords, **kwargs})
    if not isinstance(obj_exp, str):
        return partial(obj_exp)
    fn, a, k = _split(obj_exp)
    return partial(fn, *(a + args), **{**kwargs, **k})

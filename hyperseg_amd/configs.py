"""The model-defining factory calls of the reference's config files (configs/train/*.py), verbatim
arguments.  ``build(name)`` returns the HyperGen model for a BASELINE config."""
from .utils.obj_factory import obj_factory

MODELS = {
    # configs/train/cityscapes_efficientnet_b1_hyperseg-m.py:36-40
    'hyperseg-m': dict(
        arch="hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1', False, levels=2, "
             "out_feat_scale=[1., .25, .25, .25, .25], kernel_sizes=[1, 1, 1, 3, 3], "
             "level_channels=[64, 32, 16, 16, 16], expand_ratio=2, with_out_fc=False, decoder_dropout=None, "
             "weight_groups=[32, 16, 8, 16, 4], decoder_groups=1, inference_hflip=True, "
             "coords_res=[(512, 512), (512, 1024)])",
        num_classes=19, size=(512, 1024), batch=1),
    # configs/train/camvid_efficientnet_b1_hyperseg-s.py:35-38
    'hyperseg-s-camvid': dict(
        arch="hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1', False, levels=2, "
             "kernel_sizes=(1, 1, 1, 3, 3), level_channels=[64, 32, 16, 16, 16], expand_ratio=2, "
             "with_out_fc=False, decoder_dropout=None, weight_groups=[64, 32, 32, 16, 8], decoder_groups=1, "
             "inference_hflip=True, coords_res=[(576, 576), (576, 768)])",
        num_classes=12, size=(576, 768), batch=1),
    # configs/train/camvid_efficientnet_b1_hyperseg-l.py:35-38 (evaluated at 1024 x 768: val_img_transforms, :21); README.md:31
    'hyperseg-l-camvid': dict(
        arch="hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1', False, levels=2, "
             "kernel_sizes=(1, 1, 1, 3, 3, 3), level_channels=[64, 32, 16, 16, 16, 16], expand_ratio=2, "
             "inference_hflip=True, with_out_fc=False, decoder_dropout=None, weight_groups=[64, 32, 32, 16, 8, 8], "
             "coords_res=[(768, 768), (768, 1024)])",
        num_classes=12, size=(768, 1024), batch=1),
    # configs/train/cityscapes_efficientnet_b1_hyperseg-s.py:36-40
    'hyperseg-s': dict(
        arch="hyperseg.models.hyperseg_v1_0_unify.hyperseg_efficientnet('efficientnet-b1', False, levels=2, "
             "out_feat_scale=[1., .166, .2, .25, .4], kernel_sizes=[1, 1, 1, 3, 3], level_channels=[32, 16, 8, 8, 8], "
             "expand_ratio=2, with_out_fc=False, decoder_dropout=None, weight_groups=[32, 16, 8, 16, 4], "
             "decoder_groups=1, inference_hflip=True, unify_level=4, coords_res=[(768, 768), (768, 1536)])",
        num_classes=19, size=(768, 1536), batch=1),
    # configs/train/vocsbd_efficientnet_b3_hyperseg-l.py:32-34
    'hyperseg-l': dict(
        arch="hyperseg.models.hyperseg_v0_1.hyperseg_efficientnet('efficientnet-b3', False, levels=3, "
             "kernel_sizes=(1, 1, 3, 3, 3, 3), expand_ratio=2, inference_hflip=True, with_out_fc=False, "
             "decoder_dropout=None, weight_groups=16)",
        num_classes=21, size=(512, 512), batch=32),
}


def build(name):
    spec = MODELS[name]
    return obj_factory(spec['arch'], num_classes=spec['num_classes'])

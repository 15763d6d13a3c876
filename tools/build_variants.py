"""Dev-only variant builds of the HIP library for on-GPU A/B runs (selected with HS_HIP_LIB=<path>); the product build
(python -m hyperseg_amd.build) contains none of this.
    oldir     round-1 Op C kernel (hs_patch_ir_mfma.hip) behind hs_patch_ir_fwd
    nointer   fused inverted-residual kernel with the pw1 / depthwise stages NOT interleaved (HS_IRF_INTERLEAVE=0)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperseg_amd import build as B

VARIANTS = {
    'oldir': dict(flags=['-DHS_IR_USE_OLD'], extra=['hs_patch_ir_mfma.hip']),
    'nointer': dict(flags=['-DHS_IRF_INTERLEAVE=0'], extra=[]),
}

if __name__ == '__main__':
    for name in (sys.argv[1:] or VARIANTS):
        v = VARIANTS[name]
        path = os.path.join(B.LIB_DIR, f'libhyperseg_hip_{name}.so')
        print(B.build(force=True, extra_flags=v['flags'], sources=B.SOURCES + v['extra'], lib_path=path, obj_suffix='_' + name))

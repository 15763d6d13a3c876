"""Builds libhyperseg_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m hyperseg_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
snapshot.  No JIT, no torch.utils.cpp_extension: the library has no torch types in its ABI.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB_DIR = os.path.join(PKG, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libhyperseg_hip.so')
SOURCES = ['hs_weights.hip', 'hs_patch_conv.hip', 'hs_patch_conv_gen.hip', 'hs_patch_conv_k1m.hip', 'hs_k1_chain.hip', 'hs_meta_conv.hip', 'hs_patch_ir.hip', 'hs_patch_ir_fused.hip', 'hs_patch_irc.hip', 'hs_patch_ir_px.hip', 'hs_patch_ir_d2.hip', 'hs_encoder.hip', 'hs_mbconv.hip', 'hs_mbconv_lean.hip', 'hs_gemm_split.hip', 'hs_patch_conv_bwd.hip', 'hs_patch_conv_train.hip', 'hs_train_aux.hip', 'hs_s2w_train.hip']
HEADERS = [os.path.join(CSRC, 'hs_common.h'), os.path.join(CSRC, 'hs_ir_tiles.h'), os.path.join(CSRC, 'hs_ir_common.h'), os.path.join(CSRC, 'hs_s2w_blocked.h'), os.path.join(CSRC, 'hs_se_tail.h'), os.path.join(REPO, 'include', 'hyperseg_hip.h')]
# -amdgpu-kernarg-preload-count=16: gfx950 preloads the first 16 kernel-argument dwords into SGPRs at wave launch, so the
# first address computations do not wait for a scalar load (round 3, visit r5a: 0.9168 -> 0.9019 ms per HyperSeg-M frame,
# encoder tests green on it; the whole GPU suite runs on this build since).
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-mllvm', '-amdgpu-kernarg-preload-count=16',
         '-Wall', '-Wno-unused-function', '-I', os.path.join(REPO, 'include'), '-I', CSRC]


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm >= 7.0 with gfx950 support)')


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), sources=None, lib_path=None, obj_suffix=''):
    """``sources`` / ``lib_path`` / ``extra_flags``: dev builds of variant libraries (tools/); the product build takes none."""
    if sources is None and lib_path is None and not force and not needs_build():
        return LIB_PATH
    sources = list(sources or SOURCES)
    lib_path = lib_path or LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for s in sources:
        obj = os.path.join(LIB_DIR, os.path.basename(s).replace('.hip', obj_suffix + '.o'))
        cmd = [_hipcc(), *FLAGS, *extra_flags, '-c', os.path.join(CSRC, s), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n{out}')
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', lib_path]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return lib_path


if __name__ == '__main__':
    path = build(force='--force' in sys.argv, verbose='--verbose' in sys.argv)
    print(path)

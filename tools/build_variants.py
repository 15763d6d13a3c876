"""Dev-only variant builds of the HIP library for on-GPU A/B runs (selected with HS_HIP_LIB=<path>); the product build
(python -m hyperseg_amd.build) contains none of this, and the product sources carry no dev hooks: the 'stamps' variant
is made by patching a COPY of hs_patch_ir_fused.hip ('stamps_irc': of hs_patch_irc.hip, whose '// @stamp' comments mark the phases).
    stamps    s_memtime stamps of wave 0 of every workgroup at the phase boundaries (tools/ir_phase_times.py reads them)
    kpreload  the whole library with -amdgpu-kernarg-preload-count=16 (not yet measured)
"""
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperseg_amd import build as B

STAMP_DECL = '''
__device__ long long hs_irf_stamps[8192 * 32];
#define HS_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) { hs_irf_stamps[blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); \
    if ((k) == 0) { hs_irf_stamps[blockIdx.x * 32 + 31] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); \
                    hs_irf_stamps[blockIdx.x * 32 + 30] = __builtin_amdgcn_s_memrealtime(); } \
    if ((k) == 24) hs_irf_stamps[blockIdx.x * 32 + 29] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int hs_debug_read_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hs_irf_stamps), sizeof(long long) * n);
}
'''


def stamped_source(fname='hs_patch_ir_fused.hip'):
    src = open(os.path.join(B.CSRC, fname)).read()

    def after(anchor, code, count=1):
        nonlocal src
        assert src.count(anchor) >= 1, anchor
        src = src.replace(anchor, anchor + code, count)

    def before(anchor, code):
        nonlocal src
        assert src.count(anchor) == 1, anchor
        src = src.replace(anchor, code + anchor)
    after('namespace hs {\n', STAMP_DECL)
    after('    const int lrow = lane & 15, lk = lane >> 4;\n', '    HS_STAMP(0);\n')
    src, n = re.subn(r'(    __syncthreads\(\); +// window \+ BN rows[^\n]*\n)', r'    HS_STAMP(1);\n\1    HS_STAMP(2);\n', src, count=1)
    assert n == 1
    src, n = re.subn(r'(    __syncthreads\(\); +// the window is dead)', r'    HS_STAMP(3);\n\1', src, count=1)
    assert n == 1
    src = re.sub(r'(    stage_pw1\(0\);\n    __syncthreads\(\);\n)', r'\1    HS_STAMP(4);\n', src, count=1)
    after('        stage_dw(h0);\n', '        HS_STAMP(5 + 4 * (h0 < 48 ? h0 / 16 : 3));\n')
    after('        stage_dw(h0);\n        HS_STAMP(5 + 4 * (h0 < 48 ? h0 / 16 : 3));\n        __syncthreads();\n',
          '        HS_STAMP(6 + 4 * (h0 < 48 ? h0 / 16 : 3));\n')
    after('        stage_pw3(h0);\n', '        HS_STAMP(7 + 4 * (h0 < 48 ? h0 / 16 : 3));\n')
    src, n = re.subn(r'(            stage_pw1\(h0 \+ 16\);[^\n]*\n            __syncthreads\(\);[^\n]*\n)',
                     r'\1            HS_STAMP(8 + 4 * (h0 < 48 ? h0 / 16 : 3));\n', src, count=1)
    assert n == 1
    # end of the kernel body: the closing brace that precedes launch_irf's template header
    i = re.search(r'\ntemplate <[^>]*>\nstatic int launch_ir[fs]\(', src).start()
    j = src.rindex('}', 0, i)
    src = src[:j] + '    HS_STAMP(24);\n' + src[j:]
    os.makedirs(os.path.join(B.LIB_DIR, 'dev_src'), exist_ok=True)
    path = os.path.join(B.LIB_DIR, 'dev_src', fname.replace('.hip', '_stamps.hip'))
    open(path, 'w').write(src)
    return path


def stamped_irc_source():
    """hs_patch_irc.hip carries '// @stamp <expr>' comments at its phase boundaries: the dev build turns them into stamps."""
    src = open(os.path.join(B.CSRC, 'hs_patch_irc.hip')).read()
    src = src.replace('namespace hs {\n', 'namespace hs {\n' + STAMP_DECL, 1)
    src, n = re.subn(r'// @stamp ([^\n]+)\n', r'HS_STAMP(\1);\n', src)
    assert n >= 12, n
    os.makedirs(os.path.join(B.LIB_DIR, 'dev_src'), exist_ok=True)
    path = os.path.join(B.LIB_DIR, 'dev_src', 'hs_patch_irc_stamps.hip')
    open(path, 'w').write(src)
    return path


def stamped_kc_source():
    """hs_k1_chain.hip carries '// @stamp <n>' comments at its phase boundaries: the dev build turns them into stamps (tools/kc_phase_times.py)."""
    src = open(os.path.join(B.CSRC, 'hs_k1_chain.hip')).read()
    src = src.replace('namespace hs {\n', 'namespace hs {\n' + STAMP_DECL, 1)
    src, n = re.subn(r'// @stamp ([^\n]+)\n', r'HS_STAMP(\1);\n', src)
    assert n >= 10, n
    os.makedirs(os.path.join(B.LIB_DIR, 'dev_src'), exist_ok=True)
    path = os.path.join(B.LIB_DIR, 'dev_src', 'hs_k1_chain_stamps.hip')
    open(path, 'w').write(src)
    return path


def stamped_k1m_source():
    """hs_patch_conv_bwd.hip's k1m_pixel_stream carries '// @stamp <expr>' comments: the dev build turns them into stamps (wave 0 of a workgroup)."""
    src = open(os.path.join(B.CSRC, 'hs_patch_conv_bwd.hip')).read()
    src = src.replace('namespace hs {\n', 'namespace hs {\n' + STAMP_DECL, 1)
    src, n = re.subn(r'// @stamp ([^\n]+)\n', r'HS_STAMP(\1);\n', src)
    assert n >= 5, n
    os.makedirs(os.path.join(B.LIB_DIR, 'dev_src'), exist_ok=True)
    path = os.path.join(B.LIB_DIR, 'dev_src', 'hs_patch_conv_bwd_stamps.hip')
    open(path, 'w').write(src)
    return path


def patched_source(name, replacements, fname='hs_patch_ir_fused.hip'):
    src = open(os.path.join(B.CSRC, fname)).read()
    for old, new in replacements:
        assert src.count(old) == 1, old
        src = src.replace(old, new)
    os.makedirs(os.path.join(B.LIB_DIR, 'dev_src'), exist_ok=True)
    path = os.path.join(B.LIB_DIR, 'dev_src', fname.replace('.hip', f'_{name}.hip'))
    open(path, 'w').write(src)
    return path


STORE_LINE = '                for (int jt = 0; jt < J3; ++jt) yo[yoff[jt]] = fmaf(acc3[m][jt][r], sc, sh);'
PATCHES = {
    # dev experiments on the epilogue of the fused inverted-residual kernel (is HyperSeg-L level 5 store-bound?)
    'nostore': [(STORE_LINE, '                for (int jt = 0; jt < J3; ++jt) if (o == 0 && jt == 0) yo[yoff[jt]] = fmaf(acc3[m][jt][r], sc, sh);'
                             ' else asm volatile("" :: "v"(acc3[m][jt][r]));')],
    # occupancy probe of the lane-per-pixel kernel: 28 KB of unused LDS -> 2 workgroups per CU instead of 3 at HyperSeg-L level 5
    'px2wg': [('    constexpr size_t lds = (size_t)G::FLOATS * sizeof(float);', '    constexpr size_t lds = (size_t)G::FLOATS * sizeof(float) + 28 * 1024;')],
    # the LDS-tiled depthwise form (BN0 + swish once per input element) at batch 1: only the 5x5 launches / every launch that
    # qualifies.  The late 5x5 launches are vector-ALU bound (40 swishes per thread on the taps, ~5 waves per SIMD): DESIGN 7.1
    'dwtile_k5': [('    if (in_scale && (nplanes >= 8192 || k == 5) && threads == 256', '    if (in_scale && nplanes >= 8192 && threads == 256')],     # round 3: the product has it; this variant turns it OFF
    'dwtile_k5_128': [('    if (in_scale && (nplanes >= 8192 || k == 5) && threads == 256', '    if (in_scale && (nplanes >= 8192 || k == 5) && threads >= 128')],
    'dwtile_all': [('    if (in_scale && (nplanes >= 8192 || k == 5) && threads == 256', '    if (in_scale && nplanes >= 1 && threads == 256')],
    'up_plainstore': [('        __builtin_nontemporal_store(v4f{o0[0], o0[1], o0[2], o0[3]}, reinterpret_cast<v4f*>(dst));\n        __builtin_nontemporal_store(v4f{o1[0], o1[1], o1[2], o1[3]}, reinterpret_cast<v4f*>(dst + Wo));',
                       '        *reinterpret_cast<v4f*>(dst) = v4f{o0[0], o0[1], o0[2], o0[3]};\n        *reinterpret_cast<v4f*>(dst + Wo) = v4f{o1[0], o1[1], o1[2], o1[3]};')],
    # round 4: where do the 60 us of s2w_train_bwd_kernel<0> go?  one phase removed per variant (results are wrong: timing only)
    's2wt_noA': [('av16[q] = dbank[(unsigned)(min(p, P - 1) * (int)ld + n)];', 'av16[q] = (float)n;')],
    's2wt_noB': [('for (int q = 0; q < 20; ++q) bv20[q] = signal[so + (unsigned)(min(hi + 4 * q, K - 1) * grid_sz)];', 'for (int q = 0; q < 20; ++q) bv20[q] = (float)(so + q);')],
    's2wt_nomfma': [('''                    acc[t] = MODE == 0 ? __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[t], 0, 0, 0);''', '                    acc[t][0] += av * bv;')],
    's2wt_nostore': [('if (t < KT && k < K) dw[(size_t)n * K + k] = acc[t][r];', 'if (t < KT && k < K && acc[t][r] == 12345.0f) dw[(size_t)n * K + k] = acc[t][r];')],
    # round 5 (VERDICT r4 #1a): the two level-4 workgroups that share a CU start in phase and meet at every barrier together; delay one of
    # them by a fraction of a chunk so that one's waits fall under the other's vector work.  Which blocks share a CU is not specified:
    # 'hi' assumes blocks b and b + 256 (the dispatcher fills every CU once before it doubles up), 'lo' assumes b and b + 1.
    'irc_stag_hi40': [('    // @stamp 0\n', '    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(40);\n')],
    'irc_stag_hi100': [('    // @stamp 0\n', '    if ((blockIdx.x >> 8) & 1) { __builtin_amdgcn_s_sleep(100); }\n')],
    'irc_stag_lo40': [('    // @stamp 0\n', '    if (blockIdx.x & 1) __builtin_amdgcn_s_sleep(40);\n')],
    'irc_stag_xcd40': [('    // @stamp 0\n', '    if ((blockIdx.x >> 3) & 1) __builtin_amdgcn_s_sleep(40);\n')],
    # round 6: TIMING-ONLY builds of the level-4 kernel (results are wrong) -- what a bank that arrives already split
    # by its producer would buy: the split phase + its barrier removed; the f32 landing zone removed (the DMA lands on the f16 images);
    # the 16 x 8 regions at 4 workgroups per CU that the smaller LDS footprint then allows
    'irc_nosplit': [('    if ((cin & 1) == 0 && (hid & 3) == 0) {\n        // quads of 4 consecutive weights', '    if (cin < 0) {\n        // quads of 4 consecutive weights'),
                    ('    } else {\n        // odd channel counts: one JOB per thread', '    } else if (cin < 0) {\n        // odd channel counts: one JOB per thread')],
    'irc_inplace': [('    if ((cin & 1) == 0 && (hid & 3) == 0) {\n        // quads of 4 consecutive weights', '    if (cin < 0) {\n        // quads of 4 consecutive weights'),
                    ('    } else {\n        // odd channel counts: one JOB per thread', '    } else if (cin < 0) {\n        // odd channel counts: one JOB per thread'),
                    ('    m.raw = o; o += m.raw_chunks * 1024;', '    m.raw = 0;'),
                    ('            tv[i] = raw[off_kd + e]; ts[i] = raw_s2[(unsigned)e / 9u];', '            tv[i] = 1.0f; ts[i] = 1.0f;'),
                    ('            if (e < 9 * hid) taps[e] = tv[i] * (ts[i] * IRC_H2_SCALE);', '            if (e < 0) taps[e] = tv[i] * (ts[i] * IRC_H2_SCALE);')],
    'irc_rh8': [('    const bool tall = a.ph % 16 == 0;', '    const bool tall = false;')],
    # round 6: TIMING-ONLY bounds for the "swap the matrix-core operand roles" idea (a lane then owns 4 consecutive pixels of ONE channel:
    # one 16-byte store where it now issues four 4-byte stores): a quarter of the epilogue's global stores / of pw1's LDS stores
    'irc_store_quarter': [('                    for (int jt = 0; jt < J3; ++jt)\n                        yb[(unsigned)o * plane + (unsigned)((y0 + wave + NW * jt) * W + x0 + lrow)] = fmaf(acc3[m][jt][r], sc, sh);',
                           '                    for (int jt = 0; jt < J3; ++jt)\n                        if (r == 0) yb[(unsigned)o * plane + (unsigned)((y0 + wave + NW * jt) * W + x0 + lrow)] = fmaf(acc3[m][jt][r], sc, sh) + acc3[m][jt][1] + acc3[m][jt][2] + acc3[m][jt][3];')],
    'irc_h1_quarter': [('                for (int r = 0; r < 4; ++r)\n                    dst[4 * r * PS1] = __builtin_amdgcn_fmed3f(fmaf(acc[r], sc1[r] * ib, sh1[r]), 0.0f, 6.0f);',
                        '                for (int r = 0; r < 1; ++r) dst[0] = __builtin_amdgcn_fmed3f(fmaf(acc[0], sc1[0] * ib, sh1[0]), 0.0f, 6.0f) + __builtin_amdgcn_fmed3f(fmaf(acc[1], sc1[1] * ib, sh1[1]), 0.0f, 6.0f) + __builtin_amdgcn_fmed3f(fmaf(acc[2], sc1[2] * ib, sh1[2]), 0.0f, 6.0f) + __builtin_amdgcn_fmed3f(fmaf(acc[3], sc1[3] * ib, sh1[3]), 0.0f, 6.0f);')],
    # round 6: TIMING-ONLY bound of a barrier-free chunk loop (every wave owning a strip of the region end to end): the chunk loop's
    # three workgroup barriers replaced by the wave's own LDS wait -- results are wrong (the waves still depend on each other's data)
    'irc_nobarrier': [('        stage_dw(h0);\n        // @stamp 10 + 4 * (h0 < 32 ? h0 / 16 : 2)\n        __syncthreads();',
                       '        stage_dw(h0);\n        // @stamp 10 + 4 * (h0 < 32 ? h0 / 16 : 2)\n        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");'),
                      ('            stage_pw1(h0 + 16);\n            __syncthreads();',
                       '            stage_pw1(h0 + 16);\n            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");')],
    # round 6: TIMING-ONLY -- is the lane-per-pixel Op D kernel (HyperSeg-L level 5: 705 MB of logits per bs-32 batch) bound by its stores?
    'px_nostore': [('            if (o < COUT) yo[(size_t)o * plane] = fmaf(acc3[og][r], bnl[16 * HQ + o], bnl[16 * HQ + 4 * OQ + o]);',
                    '            if (o == 0) yo[(size_t)o * plane] = fmaf(acc3[og][r], bnl[16 * HQ + o], bnl[16 * HQ + 4 * OQ + o]); else asm volatile("" :: "v"(acc3[og][r]));')],
    # round 6: the wave-slot priority of the level-4 kernel tried on the lane-per-pixel Op D kernel (HyperSeg-L level 5: 32 generations of 4 workgroups per CU)
    'px_prio_young': [('    int blk = blockIdx.x;                 // XCD-contiguous region ranges (as the tiled kernels)',
                       '    { const unsigned sl = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 4) & 3u; if (sl == 3) __builtin_amdgcn_s_setprio(3); else if (sl == 2) __builtin_amdgcn_s_setprio(2); else if (sl == 1) __builtin_amdgcn_s_setprio(1); }\n    int blk = blockIdx.x;                 // XCD-contiguous region ranges (as the tiled kernels)')],
    'px_prio_odd': [('    int blk = blockIdx.x;                 // XCD-contiguous region ranges (as the tiled kernels)',
                     '    { const unsigned sl = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 4) & 1u; if (sl) __builtin_amdgcn_s_setprio(2); }\n    int blk = blockIdx.x;                 // XCD-contiguous region ranges (as the tiled kernels)')],
    # round 6: wave-slot priority in the chain kernel (two workgroups per CU that WAIT for their neighbours: the slowest cell sets the pace)
    'kc_prio_young': [('    const int cell = (int)blockIdx.x;\n    const int pib = cell / fw',
                       '    if (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 4) & 1u) __builtin_amdgcn_s_setprio(2);\n    const int cell = (int)blockIdx.x;\n    const int pib = cell / fw')],
    'kc_prio_old': [('    const int cell = (int)blockIdx.x;\n    const int pib = cell / fw',
                     '    if (!(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 4) & 1u)) __builtin_amdgcn_s_setprio(2);\n    const int cell = (int)blockIdx.x;\n    const int pib = cell / fw')],
    # round 6: wave-slot priority in the exact-f32 fused kernel (CamVid-L levels 4-5: several generations of two workgroups per CU; level 3 of M: one)
    'irf_prio_young': [('    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n',
                        '    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n    if (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 4) & 1u) __builtin_amdgcn_s_setprio(2);\n')],
    'irf_prio_old': [('    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n',
                      '    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n    if (!(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 4) & 1u)) __builtin_amdgcn_s_setprio(2);\n')],
    # round 6 (second half): TIMING-ONLY bound of the depthwise stage on packed FMAs (two hidden channels per lane: v_pk_fma_f32): half of
    # the stage's 72 v_fma_f32 per row block removed -- results are wrong
    'irc_dw_half': [('                    for (int v = 0; v < 8; ++v) o[v] = fmaf(k[ky * 3 + kx], rowv[v + kx], o[v]);',
                     '                    for (int v = 0; v < 8; v += 2) o[v] = fmaf(k[ky * 3 + kx], rowv[v + kx], o[v]);')],
    # round 6: TIMING-ONLY phase removal in the lean fused expand + depthwise kernel (hs_mbconv_lean.hip): results are wrong
    'mbl_noswish': [('                        const float sv = swishf(fmaf(acc[r], sc0[r], sh0[r]));', '                        const float sv = fmaf(acc[r], sc0[r], sh0[r]);'),
                    ('            for (int v = 0; v < G::NOUT; ++v) o[v] = swishf(fmaf(o[v], sc1, sh1));', '            for (int v = 0; v < G::NOUT; ++v) o[v] = fmaf(o[v], sc1, sh1);')],
    'mbl_nostore': [('                *reinterpret_cast<f32x4*>(dst + 4 * qd) = f32x4{o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]};',
                     '                if (o[4 * qd] == 12345.0f) *reinterpret_cast<f32x4*>(dst + 4 * qd) = f32x4{o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]};')],
    'mbl_noload': [('        for (int ks = 0; ks < KS; ++ks) bf[jt][ks] = mbl_ld(xb + (size_t)(4 * ks) * plane, off);',
                    '        for (int ks = 0; ks < KS; ++ks) bf[jt][ks] = __uint_as_float(off + ks) * 1e-9f;')],
    'mbl_nomfma': [('                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[jt][ks], acc, 0, 0, 0);',
                    '                for (int ks = 0; ks < KS; ++ks) acc[ks & 3] += af[ks] * bf[jt][ks];')],
    'mbl_nodw': [('                    for (int v = 0; v < G::NOUT; ++v) o[v] = fmaf(kd[ky * K + kx], rowv[v * S + kx], o[v]);',
                  '                    for (int v = 0; v < G::NOUT; ++v) if (kx == 0) o[v] = fmaf(kd[ky * K + kx], rowv[v * S + kx], o[v]);')],
    'ntstore': [(STORE_LINE, '                for (int jt = 0; jt < J3; ++jt) __builtin_nontemporal_store(fmaf(acc3[m][jt][r], sc, sh), &yo[yoff[jt]]);')],
}

PATCHES['irc_rh8_inplace'] = PATCHES['irc_inplace'] + PATCHES['irc_rh8']

PATCHES['irc_both_quarter'] = PATCHES['irc_store_quarter'] + PATCHES['irc_h1_quarter']

VARIANTS = {
    'stamps': dict(flags=[], extra=[], patch=True),
    'stamps_irc': dict(flags=[], extra=[], patch='irc', file='hs_patch_irc.hip'),
    'stamps_k1m': dict(flags=[], extra=[], patch='k1m', file='hs_patch_conv_bwd.hip'),
    'stamps_kc': dict(flags=[], extra=[], patch='kc', file='hs_k1_chain.hip'),              # round 5: phase stamps of the k = 1 chain kernel (tools/kc_phase_times.py)      # round 4: phase stamps of k1m_pixel_stream (tools/k1m_phase_times.py)
    'nostore': dict(flags=[], extra=[], patch='nostore'),
    'ntstore': dict(flags=[], extra=[], patch='ntstore'),
    'px2wg': dict(flags=[], extra=[], patch='px2wg', file='hs_patch_ir_px.hip'),
    # untried: gfx950 can preload the first kernel arguments into SGPRs at wave launch -- one scalar
    # round trip less at the top of every kernel.  A/B with HS_HIP_LIB=hyperseg_amd/lib/libhyperseg_hip_kpreload.so
    'dwtile_k5': dict(flags=[], extra=[], patch='dwtile_k5', file='hs_encoder.hip'),
    'dwtile_k5_128': dict(flags=[], extra=[], patch='dwtile_k5_128', file='hs_encoder.hip'),   # + the 16x32 maps (128 threads)
    'dwtile_all': dict(flags=[], extra=[], patch='dwtile_all', file='hs_encoder.hip'),
    'up_plainstore': dict(flags=[], extra=[], patch='up_plainstore', file='hs_patch_conv.hip'),   # final 2x upsample with ordinary stores
    'kpreload': dict(flags=['-mllvm', '-amdgpu-kernarg-preload-count=16'], extra=[], patch=None),
    's2b_kc20': dict(flags=['-DHS_S2B_KC=20'], extra=[], patch=None),          # blocked signal2weights: one LDS fill for K = 80 (40 KB)
    's2b_kc5': dict(flags=['-DHS_S2B_KC=5'], extra=[], patch=None),
    # split GEMM: 16-pixel workgroups when the 32-pixel grid has at most this many workgroups (product: 256)
    'gs_narrow_0': dict(flags=['-DHS_GS_NARROW_MAX_WG=0'], extra=[], patch=None),
    'gs_narrow_192': dict(flags=['-DHS_GS_NARROW_MAX_WG=192'], extra=[], patch=None),
    'gs_narrow_384': dict(flags=['-DHS_GS_NARROW_MAX_WG=384'], extra=[], patch=None),
    # split GEMM: 64-row workgroups when the 32-row grid has at least this many workgroups (product: 257)
    'gs_tall_never': dict(flags=['-DHS_GS_TALL_MIN_WG=100000000'], extra=[], patch=None),
    'gs_tall_129': dict(flags=['-DHS_GS_TALL_MIN_WG=129'], extra=[], patch=None),
    'gs_tall_513': dict(flags=['-DHS_GS_TALL_MIN_WG=513'], extra=[], patch=None),
    'k1m_256': dict(flags=['-DHS_K1M_MIN_PATCHES=256'], extra=[], patch=None),            # ... from 256 patches (HyperSeg-M level 1: 512)
    'k1m_off': dict(flags=['-DHS_K1M_MIN_PATCHES=2000000000'], extra=[], patch=None),   # batched k = 1 levels on the LDS-staged kernel
    's2b_nt': dict(flags=['-DHS_S2B_NT=1'], extra=[], patch=None),                    # round 6: non-temporal bank stores of the blocked signal2weights
    's2b_light_first': dict(flags=['-DHS_S2B_LIGHT_FIRST'], extra=[], patch=None),
    'irc_wide_store': dict(flags=['-DHS_IRC_WIDE_STORE'], extra=[], patch='git:9d8dcac', file='hs_patch_irc.hip'),   # (round 5's source: the flag left the product in round 6)
    'irc_r5': dict(flags=[], extra=[], patch='git:9d8dcac', file='hs_patch_irc.hip'),                      # round 5's level-4 kernel, for same-box A/Bs
         # round 4: the level-4 epilogue re-laid through LDS into 16-byte stores (measured slower)
    'irc_nw8': dict(flags=['-DHS_IRC_NW=8'], extra=[], patch=None),                      # round 4: eight waves per 16 x 16 region (four per SIMD at two workgroups per CU)
    'irc_r3': dict(flags=[], extra=[], patch='irc_r3', file='hs_patch_irc.hip'),          # round 3's level-4 kernel
    'irc_stag_hi40': dict(flags=[], extra=[], patch='irc_stag_hi40', file='hs_patch_irc.hip'),      # round 5: staggered co-resident level-4 workgroups
    'irc_stag_hi100': dict(flags=[], extra=[], patch='irc_stag_hi100', file='hs_patch_irc.hip'),
    'irc_stag_lo40': dict(flags=[], extra=[], patch='irc_stag_lo40', file='hs_patch_irc.hip'),
    'irc_stag_xcd40': dict(flags=[], extra=[], patch='irc_stag_xcd40', file='hs_patch_irc.hip'),
    'irc_nosplit': dict(flags=[], extra=[], patch='irc_nosplit', file='hs_patch_irc.hip'),       # round 6: timing only
    'irc_inplace': dict(flags=[], extra=[], patch='irc_inplace', file='hs_patch_irc.hip'),
    'irc_rh8': dict(flags=[], extra=[], patch='irc_rh8', file='hs_patch_irc.hip'),
    'irc_rh8_inplace': dict(flags=[], extra=[], patch='irc_rh8_inplace', file='hs_patch_irc.hip'),
    'irc_store_quarter': dict(flags=[], extra=[], patch='irc_store_quarter', file='hs_patch_irc.hip'),
    'irc_h1_quarter': dict(flags=[], extra=[], patch='irc_h1_quarter', file='hs_patch_irc.hip'),
    'irc_both_quarter': dict(flags=[], extra=[], patch='irc_both_quarter', file='hs_patch_irc.hip'),
    'irc_prio0': dict(flags=['-DHS_IRC_PRIO=0'], extra=[], patch=None),             # round 6: s_setprio turns of the CU's two level-4 workgroups: off / younger always first / per stage
    'irc_prio2': dict(flags=['-DHS_IRC_PRIO=2'], extra=[], patch=None),
    'irc_prio3': dict(flags=['-DHS_IRC_PRIO=3'], extra=[], patch=None),
    'irc_dw_half': dict(flags=[], extra=[], patch='irc_dw_half', file='hs_patch_irc.hip'),
    'mbl_noswish': dict(flags=[], extra=[], patch='mbl_noswish', file='hs_mbconv_lean.hip'),
    'mbl_nostore': dict(flags=[], extra=[], patch='mbl_nostore', file='hs_mbconv_lean.hip'),
    'mbl_noload': dict(flags=[], extra=[], patch='mbl_noload', file='hs_mbconv_lean.hip'),
    'mbl_nomfma': dict(flags=[], extra=[], patch='mbl_nomfma', file='hs_mbconv_lean.hip'),
    'mbl_nodw': dict(flags=[], extra=[], patch='mbl_nodw', file='hs_mbconv_lean.hip'),
    'mbl_nostage': dict(flags=['-DHS_MBL_STAGE=0'], extra=[], patch=None),      # lean fused expand + depthwise kernel: direct output stores (no LDS staging)
    'enc_se_in': dict(flags=[], extra=[], patch='git:64a8791', file='hs_encoder.hip'),      # round 6: depthwise kernel with the squeeze-excite tail compiled into every instantiation (same-box A/B)
    'irc_nobarrier': dict(flags=[], extra=[], patch='irc_nobarrier', file='hs_patch_irc.hip'),
    'px_nostore': dict(flags=[], extra=[], patch='px_nostore', file='hs_patch_ir_px.hip'),
    'px_prio_young': dict(flags=[], extra=[], patch='px_prio_young', file='hs_patch_ir_px.hip'),
    'px_prio_odd': dict(flags=[], extra=[], patch='px_prio_odd', file='hs_patch_ir_px.hip'),
    'kc_prio_young': dict(flags=[], extra=[], patch='kc_prio_young', file='hs_k1_chain.hip'),
    'kc_prio_old': dict(flags=[], extra=[], patch='kc_prio_old', file='hs_k1_chain.hip'),
    'irf_prio_young': dict(flags=[], extra=[], patch='irf_prio_young'),
    'irf_prio_old': dict(flags=[], extra=[], patch='irf_prio_old'),
    'gs_tail_switch': dict(flags=[], extra=[], patch='git:b89aa7e', file='hs_gemm_split.hip'),   # round 4: the split GEMM with apply_act(v, a.act) inside the unrolled tail
    'st_slice64': dict(flags=['-DHS_ST_SLICE=64'], extra=[], patch=None),                 # round 4: patches per dW slice of the s2w backward (product: 256)
    'st_slice128': dict(flags=['-DHS_ST_SLICE=128'], extra=[], patch=None),
    's2wt_noA': dict(flags=[], extra=[], patch='s2wt_noA', file='hs_s2w_train.hip'),
    's2wt_noB': dict(flags=[], extra=[], patch='s2wt_noB', file='hs_s2w_train.hip'),
    's2wt_nomfma': dict(flags=[], extra=[], patch='s2wt_nomfma', file='hs_s2w_train.hip'),
    's2wt_nostore': dict(flags=[], extra=[], patch='s2wt_nostore', file='hs_s2w_train.hip'),
    'stem_cg8': dict(flags=['-DHS_STEM_CG=8'], extra=[], patch=None),                      # round 5: output channels per thread of the stem (product: 4; rounds 1-4: 8)
    'stem_cg16': dict(flags=['-DHS_STEM_CG=16'], extra=[], patch=None),
}

def git_source(rev, fname, tag):
    """<rev>:hyperseg_amd/csrc/<fname> as a dev source (same-box A/B of a kernel file against an earlier commit)."""
    import subprocess
    os.makedirs(os.path.join(B.LIB_DIR, 'dev_src'), exist_ok=True)
    path = os.path.join(B.LIB_DIR, 'dev_src', fname.replace('.hip', f'_{tag}.hip'))
    src = subprocess.run(['git', 'show', f'{rev}:hyperseg_amd/csrc/{fname}'], cwd=B.REPO, capture_output=True, text=True, check=True).stdout
    open(path, 'w').write(src)
    return path


def r3_irc_source():
    """Round 3's hs_patch_irc.hip (git show e068d5a:...), for same-box A/B runs against the round-4 prologue."""
    import subprocess
    os.makedirs(os.path.join(B.LIB_DIR, 'dev_src'), exist_ok=True)
    path = os.path.join(B.LIB_DIR, 'dev_src', 'hs_patch_irc_r3.hip')
    src = subprocess.run(['git', 'show', 'e068d5a:hyperseg_amd/csrc/hs_patch_irc.hip'], cwd=B.REPO, capture_output=True, text=True, check=True).stdout
    # the route query of round 4 passes a null y; round 3's launcher is otherwise ABI-compatible
    open(path, 'w').write(src)
    return path


if __name__ == '__main__':
    for name in (sys.argv[1:] or VARIANTS):
        v = VARIANTS[name]
        path = os.path.join(B.LIB_DIR, f'libhyperseg_hip_{name}.so')
        sources = list(B.SOURCES) + v['extra']
        if v.get('patch'):
            fname = v.get('file', 'hs_patch_ir_fused.hip')
            src_path = stamped_kc_source() if v['patch'] == 'kc' else stamped_k1m_source() if v['patch'] == 'k1m' else git_source(v['patch'][4:], fname, name) if str(v['patch']).startswith('git:') else r3_irc_source() if v['patch'] == 'irc_r3' else stamped_irc_source() if v['patch'] == 'irc' else stamped_source(fname) if v['patch'] is True else \
                patched_source(v['patch'], PATCHES[v['patch']], fname)
            rel = os.path.relpath(src_path, B.CSRC)
            sources = [rel if s == fname else s for s in sources]
        print(B.build(force=True, extra_flags=v['flags'], sources=sources, lib_path=path, obj_suffix='_' + name))

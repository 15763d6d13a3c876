// The blocked form of signal2weights (hs_signal2weights_multi_fwd) as a workgroup BODY, shared by its own kernel (hs_weights.hip)
// and by the heterogeneous launch of hs_patch_conv.hip, which runs these workgroups beside the k = 1 patch convolution's
// (round 4: the k = 1 levels are latency-bound launches that leave most of the chip idle).
#pragma once
#include "hs_common.h"

namespace hs {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int S2W_MAX_LAYERS = 8;

#ifndef HS_S2B_NT
#define HS_S2B_NT 0
#endif
#ifndef HS_S2B_KC
#define HS_S2B_KC 10
#endif
constexpr int S2B_ROWS = 64, S2B_PATCHES = 64, S2B_KC = HS_S2B_KC;      // HS_S2B_KC: dev A/B knob (tools/build_variants.py)
constexpr int S2B_LDS_FLOATS = 2 * S2B_KC * 256;
constexpr int S2B_DS = 68;                                              // floats per patch of the staged output block: 64 rows + the <= 3-float alignment shift, == 4 (mod 32)

struct S2bLayer {
    const float* __restrict__ blk;          // packed weights (hs_s2w_pack_fwd)
    float* __restrict__ bank;
    long ld;
    int signal_index, cs_g, rpg, rows, ks, rb, groups;
    int wg_begin;                           // first workgroup of the layer
};
struct S2bArgs {
    const float* __restrict__ signal;
    int c_signal, grid_sz, n_patches, n_layers, pb, n_wg;
    unsigned m_grid, m_pb;                  // magic multipliers: / grid_sz, / pb
    S2bLayer layer[S2W_MAX_LAYERS];
};

__device__ __forceinline__ unsigned s2b_div(unsigned x, unsigned m) { return m ? __umulhi(x, m) : x; }
static inline unsigned s2b_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / d + 1); }

// One workgroup (256 threads) of the blocked form.  ``ka``: the arguments in the KERNEL-ARGUMENT segment (scalar loads, the layer
// table indexed at run time), ``wg``: the workgroup's index among the blocked workgroups, ``lds``: S2B_LDS_FLOATS floats.
__device__ __forceinline__ void s2b_body(const __attribute__((address_space(4))) S2bArgs* ka, const int wg, float* __restrict__ lds) {
    int li = 0;
    for (int q = 1; q < ka->n_layers; ++q)
        if (wg >= ka->layer[q].wg_begin) li = q;
    const float* __restrict__ blk = ka->layer[li].blk;
    float* __restrict__ bank = ka->layer[li].bank;
    const long ld = ka->layer[li].ld;
    const int signal_index = ka->layer[li].signal_index, cs_g = ka->layer[li].cs_g, rpg = ka->layer[li].rpg;
    const int rows = ka->layer[li].rows, KS = ka->layer[li].ks, RB = ka->layer[li].rb;
    const int grid_sz = ka->grid_sz, n_patches = ka->n_patches, c_signal = ka->c_signal, PB = ka->pb;
    const float* __restrict__ signal = ka->signal;
    const int local = wg - ka->layer[li].wg_begin;
    const int grb = (int)s2b_div((unsigned)local, ka->m_pb), pb = local - grb * PB;      // patch block fastest: neighbours share A
    const int g = grb / RB, rb = grb - g * RB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* A = lds;                                                                      // A fill | B fill; the output block aliases both
    float* Bm = lds + S2B_KC * 256;
    static_assert(S2B_PATCHES * S2B_DS <= 2 * S2B_KC * 256, "the output block fits the operand fills");

    // B source of this lane: (patch tile, k mod 4, 4 consecutive patches)
    const int pt = lane >> 4, kq = (lane >> 2) & 3, j4 = lane & 3;
    const int p4 = min(pb * S2B_PATCHES + 16 * pt + 4 * j4, n_patches - 4);                // whole 16-byte groups stay inside the signal
    const int bb = (int)s2b_div((unsigned)p4, ka->m_grid), ij = p4 - bb * grid_sz;
    const unsigned sig0 = (unsigned)((bb * c_signal + signal_index + g * cs_g) * grid_sz + ij);
    const float* ablk = blk + (size_t)((g * RB + rb) * KS) * 256;

    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KS; k0 += S2B_KC) {
        const int kn = min(S2B_KC, KS - k0);
        if (k0 > 0) __syncthreads();                                                       // the previous fill has been consumed
        for (int c = wave; c < kn; c += 4) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ablk + (size_t)(k0 + c) * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(A + c * 256), 16, 0, 0);
            const unsigned k = (unsigned)min(4 * (k0 + c) + kq, cs_g - 1);                 // k past the group reads a finite neighbour: A is zero there
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(signal + sig0 + k * (unsigned)grid_sz),
                                             (__attribute__((address_space(3))) void*)(Bm + c * 256), 16, 0, 0);
        }
        __syncthreads();                                                                   // both fills have landed
        for (int c = 0; c < kn; ++c) {
            const float av = A[c * 256 + wave * 64 + lane];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bm[c * 256 + t * 64 + lane], acc[t], 0, 0, 0);
        }
    }
    __syncthreads();                                                                       // operands dead: the output block takes their place
    // D leaves through LDS as [patch][S2B_DS floats]: the patch's 64 bank rows, SHIFTED by al = n0 & 3 so that 16-byte aligned quads of
    // the bank row (row start p * ld: ld is a multiple of 4) are 16-byte aligned in LDS too.  Round 6: the stores were 16 dword
    // instructions per lane (256-byte runs) -- a store-issue-bound tail (3.6 of the launch's 16 us at HyperSeg-M, 790 MB at HyperSeg-L
    // bs 32: profiles/round5_s2w_phase_removal.txt); now <= 5 dwordx4 instructions per lane, single floats only at the two ends of
    // a block whose first row is not a multiple of 4 (rows per group 147, 281, 1054 ...)
    const int r0 = rb * S2B_ROWS, n0 = g * rpg + r0;
    const int al = n0 & 3;                                                                 // uniform
    {
        const int j = lane & 15, q4 = lane >> 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float* d = lds + (16 * t + j) * S2B_DS + 16 * wave + 4 * q4 + al;
            if (al == 0) *reinterpret_cast<f32x4*>(d) = acc[t];                            // conflict-free: 8 lanes x 4 dwords cover the 32 banks (S2B_DS == 4 mod 32)
            else { d[0] = acc[t][0]; d[1] = acc[t][1]; d[2] = acc[t][2]; d[3] = acc[t][3]; }
        }
    }
    __syncthreads();
    const int nv = min(min(rpg - r0, rows - n0), S2B_ROWS);                                // valid rows of the block (uniform; <= 0: a block of padding rows)
    float* __restrict__ brow = bank + (n0 - al);
#pragma unroll
    for (int it = 0; it < 5; ++it) {                                                       // 16 patches x 17 quads per wave
        const int item = it * 64 + lane;
        const int pq = item / 17, quad = item - 17 * pq;
        const int pl = wave * 16 + pq, p = pb * S2B_PATCHES + pl;
        const int e0 = 4 * quad - al;                                                      // block row of the quad's first float
        if (pq < 16 && p < n_patches && e0 + 3 >= 0 && e0 < nv) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(lds + pl * S2B_DS + 4 * quad);
            float* dst = brow + (size_t)p * ld + 4 * quad;
#if HS_S2B_NT
            if (e0 >= 0 && e0 + 3 < nv) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));      // dev A/B (tools/build_variants.py s2b_nt)
#else
            if (e0 >= 0 && e0 + 3 < nv) *reinterpret_cast<f32x4*>(dst) = v;
#endif
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e0 + e >= 0 && e0 + e < nv) dst[e] = v[e];
            }
        }
    }
}


// Host side: the launch arguments of the blocked form for ``layers`` (in the order ``order``); returns 1 when it does not apply
// (a layer without packed weights, patches not in whole 16-byte groups, a misaligned operand), 0 otherwise.
inline int s2b_fill_args(S2bArgs& a, const float* signal, int batch, int c_signal, int fh, int fw, const hs_s2w_layer* layers,
                         const int* order, int n_layers) {
    const int grid_sz = fh * fw, n_patches = batch * grid_sz;
    if ((grid_sz & 3) != 0 || n_patches < 4) return 1;
    if (((size_t)signal & 15) != 0) return 1;            // 16-byte global_load_lds pieces: a view with an odd storage offset takes the direct kernel
    a.signal = signal; a.c_signal = c_signal; a.grid_sz = grid_sz; a.n_patches = n_patches; a.n_layers = n_layers;
    a.pb = (n_patches + S2B_PATCHES - 1) / S2B_PATCHES;
    a.m_grid = s2b_magic((unsigned)grid_sz); a.m_pb = s2b_magic((unsigned)a.pb);
    int wgs = 0;
    for (int i = 0; i < n_layers; ++i) {
        const hs_s2w_layer& l = layers[order ? order[i] : i];
        if (!l.wsw_blk || ((size_t)l.wsw_blk & 15) != 0) return 1;
        if (((size_t)l.bank & 15) != 0 || (l.ld & 3) != 0) return 1;      // 16-byte stores into 16-byte aligned bank rows (the direct kernel takes anything)
        S2bLayer& d = a.layer[i];
        d.blk = l.wsw_blk; d.bank = l.bank; d.ld = (long)l.ld; d.signal_index = l.signal_index;
        d.cs_g = l.signal_channels / l.groups; d.rpg = l.wc / l.groups; d.rows = l.rows; d.groups = l.groups;
        d.ks = (d.cs_g + 3) / 4; d.rb = (d.rpg + S2B_ROWS - 1) / S2B_ROWS;
        d.wg_begin = wgs;
        wgs += l.groups * d.rb * a.pb;
    }
    for (int i = n_layers; i < S2W_MAX_LAYERS; ++i) { a.layer[i] = a.layer[0]; a.layer[i].wg_begin = 0x7fffffff; }
    a.n_wg = wgs;
    return 0;
}

// hs_weights.hip: the all-layers launch behind hs_signal2weights_multi_fwd (native = the weights in the conv's own layout: training)
int s2w_multi_launch(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                     const hs_s2w_layer* layers, int32_t n_layers, bool native, void* stream);

// argument checks shared by the entry points that take hs_s2w_layer tables
inline int s2w_check_layer(const hs_s2w_layer& l, int c_signal) {
    if (!l.wsw_t || !l.bank || l.groups <= 0 || l.rows <= 0 || l.wc <= 0 || l.ld < l.rows || l.rows > l.wc) return HS_ERR_BAD_ARG;
    if (l.signal_index < 0 || l.signal_channels <= 0 || l.signal_index + l.signal_channels > c_signal) return HS_ERR_BAD_ARG;
    if (l.signal_channels % l.groups != 0 || l.wc % l.groups != 0) return HS_ERR_BAD_ARG;
    if (l.signal_channels / l.groups > 80) return HS_ERR_UNSUPPORTED;   // K-step register buckets stop at 20
    return HS_OK;
}

}  // namespace hs

// signal2weights on the TRAINING path (BASELINE config 5): every level's bank and both gradients, one launch each.
//
// The reference trains through  w = signal2weights(s[:, idx:idx+Cs])[:, :hp]  -- a grouped 1x1 Conv2d per level
// (hyperseg_v1_0.py:473-484) -- followed by the permute / reshape into per-patch weights (:334-337, 491); autograd gives the
// adjoints.  Rounds 2-3 ran that as one strided-batched GEMM per level and direction (rocBLAS: 15 launches of 10-25 us per
// step for 81 M multiply-adds each way -- the launches ARE the cost) plus a re-layout kernel per level and direction.  Here:
//   forward   hs_s2w_train_fwd   the inference kernel of hs_weights.hip reading the conv weight in its OWN (wc, K) layout
//                                (weights change every step: no transposed / packed copy exists), bank written patch-major;
//   backward  hs_s2w_train_bwd   dW[n, k] = sum_p dBank[p, n] S[p, k]   and   dS_l[p, k] = sum_n dBank[p, n] W[n, k]
//                                as 64 x K output tiles with the reduction staged through LDS in chunks of 64 (one kernel,
//                                two modes, every level in the same launch), then the levels' dS summed into d signal
//                                (v1_0's levels all read signal channels from 0 up): 3 launches.
// fp32, deterministic (no atomics).  Patch-major dBank (P, ld) is what the patch-convolution adjoints produce, so no re-layout.
#include "hs_common.h"
#include "hs_s2w_blocked.h"

namespace hs {

constexpr int ST_TILE = 64;                 // output rows (dW) / patches (dS) per workgroup, and the reduction chunk
constexpr int ST_KMAX = 80;                 // signal channels per group
constexpr int ST_APAD = ST_TILE + 1;
#ifndef HS_ST_SLICE
#define HS_ST_SLICE 256
#endif
constexpr int ST_SLICE = HS_ST_SLICE;         // patches per dW slice (a multiple of ST_TILE): one workgroup reduces ST_SLICE / ST_TILE chunks into one partial tile

struct StLayer {
    const float* __restrict__ w;            // (wc, K)
    const float* __restrict__ dbank;        // (P, ld) or null
    float* __restrict__ dw;                 // (wc, K) or null
    float* __restrict__ ds;                 // (B, Cs, grid) level-private, or null
    long ld;
    int signal_index, K, rpg, wc, rows, groups;
    int blocks_per_group;                   // row blocks (mode 0) / patch blocks (mode 1)
    int wg_begin;
    long dw_off;                            // mode 0: where this layer's (wc, K) partial sums start inside one slice of the workspace
};
struct StArgs {
    const float* __restrict__ signal;
    int c_signal, grid_sz, n_patches, n_layers;
    float* __restrict__ dw_partial;         // mode 0: [patch slice][sum over layers of wc K] partial sums, or null (one slice: straight into dw)
    long dw_slice_floats;
    StLayer layer[S2W_MAX_LAYERS];
};

// acc[t] += A^T B over one staged chunk of 64 reduction steps, KTV 16-wide tiles along K: operands of four steps requested from LDS together
template <int MODE, int KTV>
__device__ __forceinline__ void st_products(const float* __restrict__ A, const float* __restrict__ Bm, int hi, int l16, int kk, f32x4 (&acc)[5]) {
#pragma unroll
    for (int q0 = 0; q0 < ST_TILE / 4; q0 += 4) {
        float av[4], bv[4][KTV];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            av[j] = A[(4 * (q0 + j) + kk) * ST_APAD + 16 * hi + l16];
#pragma unroll
            for (int t = 0; t < KTV; ++t) bv[j][t] = Bm[(4 * (q0 + j) + kk) * ST_KMAX + 16 * t + l16];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < KTV; ++t)
                acc[t] = MODE == 0 ? __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j][t], acc[t], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j][t], av[j], acc[t], 0, 0, 0);
    }
}

// MODE 0: dW tile = rows [n0, n0 + 64) of group g, reduction over the patches.   A[i][j] = dBank[patch i][row j], B[i][k] = S[patch i][k]
// MODE 1: dS tile = patches [p0, p0 + 64) of group g, reduction over the group's rows.  A[i][j] = dBank[patch j][row i], B[i][k] = W[row i][k]
template <int MODE>
__global__ __launch_bounds__(256)         // (a minimum-occupancy hint of 3 or 4 makes the register allocator spill the 36 loads in flight)
void s2w_train_bwd_kernel(StArgs a) {
    const __attribute__((address_space(4))) StArgs* ka = (const __attribute__((address_space(4))) StArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int wg = (int)blockIdx.x;
    int li = 0;
    for (int q = 1; q < ka->n_layers; ++q)
        if (wg >= ka->layer[q].wg_begin) li = q;
    const float* __restrict__ w = ka->layer[li].w;
    const float* __restrict__ dbank = ka->layer[li].dbank;
    const long ld = ka->layer[li].ld;
    const int K = ka->layer[li].K, rpg = ka->layer[li].rpg, wc = ka->layer[li].wc, rows = ka->layer[li].rows;
    const int sidx = ka->layer[li].signal_index, bpg = ka->layer[li].blocks_per_group;
    const int grid_sz = ka->grid_sz, P = ka->n_patches, c_signal = ka->c_signal;
    const float* __restrict__ signal = ka->signal;
    const int local = wg - ka->layer[li].wg_begin;
    const int g = local / bpg, blk = local - g * bpg;
    const int KP = (K + 3) & ~3;

    __shared__ __attribute__((aligned(16))) float A[ST_TILE * ST_APAD];
    __shared__ __attribute__((aligned(16))) float Bm[ST_TILE * ST_KMAX];
    const int tid = threadIdx.x, lo = tid & 63, hi = tid >> 6;        // hi = wave index (uniform)
    const int l16 = lo & 15, kk = lo >> 4;
    const int KT = (KP + 15) >> 4;                                    // 16-wide tiles along K
    f32x4 acc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int out0 = blk * ST_TILE;                                   // first row of the group (mode 0) / first patch (mode 1) of the tile
    // mode 0 with a workspace: the patches are cut into slices of ST_SLICE (blockIdx.y; 64 / 128 / 256 patches per slice measured 36.7 + 6.1 /
    // 31.8 + 5 / 29.9 + 4.4 us for this launch + the slice sum at config 5, visit r5a), every workgroup reduces ONE slice and leaves a partial
    // tile; s2w_train_dwsum_kernel adds the slices in order.  (One workgroup walking all 648 patches was a 122 us latency chain on 304
    // workgroups -- the longest launch of the config-5 step, visit r4m.)
    const int red_lo = (MODE == 0 && ka->dw_partial) ? (int)blockIdx.y * ST_SLICE : 0;
    const int red_hi = MODE == 0 ? (ka->dw_partial ? min(red_lo + ST_SLICE, P) : P) : rpg;
    for (int c0 = red_lo; c0 < red_hi; c0 += ST_TILE) {
        if (c0 > red_lo) __syncthreads();
        // ---- A: dBank, coalesced along the bank's rows.  Every load is UNCONDITIONAL from a clamped address and masked afterwards: behind
        //      `ok ? load : 0` the compiler branches around each load and waits for it at the join -- 16 serialised round trips per
        //      thread and chunk, which is what the first version of this kernel spent its 54-80 us on (visit r4o).
        float av16[16];
        if (dbank) {                                                  // uniform
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int p = MODE == 0 ? c0 + hi + 4 * q : out0 + hi + 4 * q;
                const int r = MODE == 0 ? out0 + lo : c0 + lo;
                const int n = min(g * rpg + min(r, rpg - 1), rows - 1);
                // 32-bit element offsets from the layer's (uniform) base: one address register per load in flight instead of two
                // (P ld < 2^31 is checked on the host), which is what keeps 36 loads in flight under 128 VGPRs
                av16[q] = dbank[(unsigned)(min(p, P - 1) * (int)ld + n)];
            }
        }
        // ---- B
        float bv20[20];
        if (MODE == 0) {                                              // S[patch c0 + lo][k]: consecutive lanes = consecutive patches
            const int p = min(c0 + lo, P - 1);
            const int bb = p / grid_sz, ij = p - bb * grid_sz;
            const unsigned so = (unsigned)((bb * c_signal + sidx + g * K) * grid_sz + ij);          // (B c_signal grid < 2^31: st_check)
#pragma unroll
            for (int q = 0; q < 20; ++q) bv20[q] = signal[so + (unsigned)(min(hi + 4 * q, K - 1) * grid_sz)];
        }
        __builtin_amdgcn_sched_barrier(0);                            // all of the chunk's global loads are in flight before the first LDS store
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int p = MODE == 0 ? c0 + hi + 4 * q : out0 + hi + 4 * q;
            const int r = MODE == 0 ? out0 + lo : c0 + lo;
            const bool ok = dbank && p < P && r < rpg && g * rpg + r < rows;
            if (MODE == 0) A[(hi + 4 * q) * ST_APAD + lo] = ok ? av16[q] : 0.0f;     // i = patch of the chunk, j = lo = row of the tile
            else A[lo * ST_APAD + hi + 4 * q] = ok ? av16[q] : 0.0f;                 // j = patch of the tile, i = lo = row of the chunk
        }
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 20; ++q) {
                const int k = hi + 4 * q;
                if (k < 16 * KT) Bm[lo * ST_KMAX + k] = (k < K && c0 + lo < P) ? bv20[q] : 0.0f;      // columns past K: zeros (16-wide tiles)
            }
        } else {                                                      // W[row c0 + i][k]: one linear run of 64 K floats
            const int n_first = g * rpg + c0;
            const int KW = 16 * KT;
            for (int e = tid; e < ST_TILE * KW; e += 256) {
                const int i = e / KW, k = e - i * KW;
                const int n = n_first + i;
                const float wv = w[(size_t)min(n, wc - 1) * K + min(k, K - 1)];
                Bm[i * ST_KMAX + k] = (k < K && c0 + i < rpg && n < wc) ? wv : 0.0f;
            }
        }
        __syncthreads();
        // ---- acc[j][k] += sum_i A[i][j] B[i][k] on the f32 matrix cores (v_mfma_f32_16x16x4_f32: the exact fma chain, i ascending):
        //      wave hi owns j in [16 hi, 16 hi + 16), KT tiles of 16 k; a lane fetches ONE element of each operand per 4 reduction steps
        //      (the vector form read B as broadcast float4s: 6 LDS instructions per step, and the launch was LDS-bound -- 54 / 80 us).
        //      Mode 0: D[row j][k] (lanes along k: dW's rows are contiguous in k); mode 1: the operands swapped, D[k][patch j] (lanes along
        //      the patches: d signal is contiguous in them).
        // a wave whose 16 rows (mode 0) / patches (mode 1) lie past the group's rows / the last patch has nothing to accumulate.
        // (phase-removal variants of tools/build_variants.py, visit r4w: 59.7 us with, 25.8 us without the products; loads and stores each
        // within 1 us of nothing)
        if (out0 + 16 * hi >= (MODE == 0 ? rpg : P)) continue;          // (wave-uniform; the barriers are at the top of the chunk loop)
        // (KT is a property of the LAYER, so it is only known per workgroup: one straight-line body per value, chosen by a uniform switch.
        //  With `if (t < KT)` inside the loops the compiler emitted 80 branch blocks that shuffled the accumulators between AGPRs and VGPRs
        //  around every matrix instruction -- 5100 lines of ISA, 34 of the launch's 60 us.)
        switch (KT) {
            case 1: st_products<MODE, 1>(A, Bm, hi, l16, kk, acc); break;
            case 2: st_products<MODE, 2>(A, Bm, hi, l16, kk, acc); break;
            case 3: st_products<MODE, 3>(A, Bm, hi, l16, kk, acc); break;
            case 4: st_products<MODE, 4>(A, Bm, hi, l16, kk, acc); break;
            default: st_products<MODE, 5>(A, Bm, hi, l16, kk, acc); break;
        }
    }
    // ---- store (D: the lane holds rows 4 kk + r, column l16 of every tile)
    if (MODE == 0) {
        float* __restrict__ dw = ka->dw_partial ? ka->dw_partial + (size_t)blockIdx.y * ka->dw_slice_floats + ka->layer[li].dw_off : ka->layer[li].dw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = out0 + 16 * hi + 4 * kk + r, n = g * rpg + rr;
            if (dw && rr < rpg && n < wc) {
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const int k = 16 * t + l16;
                    if (t < KT && k < K) dw[(size_t)n * K + k] = acc[t][r];   // rows past `rows` accumulated zeros only
                }
            }
        }
    } else {
        float* __restrict__ ds = ka->layer[li].ds;
        const int p = out0 + 16 * hi + l16;
        if (ds && p < P) {
            const int bb = p / grid_sz, ij = p - bb * grid_sz;
            const int Cs = K * ka->layer[li].groups;
            float* dp = ds + ((size_t)bb * Cs + g * K) * grid_sz + ij;
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * t + 4 * kk + r;
                    if (t < KT && k < K) dp[(size_t)k * grid_sz] = acc[t][r];
                }
        }
    }
}

// dw[layer][n][k] = sum over the patch slices of their partial tiles, in slice order (deterministic)
__global__ __launch_bounds__(256)
void s2w_train_dwsum_kernel(StArgs a, int n_slices) {
    const __attribute__((address_space(4))) StArgs* ka = (const __attribute__((address_space(4))) StArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const long total = ka->dw_slice_floats;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        int li = 0;
        for (int q = 1; q < ka->n_layers; ++q)
            if (e >= ka->layer[q].dw_off) li = q;
        float* __restrict__ dw = ka->layer[li].dw;
        if (!dw) continue;
        float v = 0.0f;
        for (int sl = 0; sl < n_slices; ++sl) v += ka->dw_partial[(size_t)sl * total + e];
        dw[e - ka->layer[li].dw_off] = v;
    }
}

// d signal[b][c][ij] = sum over the layers whose channel range holds c of their private d signal; channels nobody read get zeros
__global__ __launch_bounds__(256)
void s2w_train_dsum_kernel(StArgs a, float* __restrict__ dsignal, long total) {
    const __attribute__((address_space(4))) StArgs* ka = (const __attribute__((address_space(4))) StArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int grid_sz = ka->grid_sz, C = ka->c_signal;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ij = (int)(e % grid_sz);
        const long bc = e / grid_sz;
        const int c = (int)(bc % C), bb = (int)(bc / C);
        float v = 0.0f;
        for (int l = 0; l < ka->n_layers; ++l) {
            const int Cs = ka->layer[l].K * ka->layer[l].groups, cl = c - ka->layer[l].signal_index;
            if (ka->layer[l].ds && cl >= 0 && cl < Cs) v += ka->layer[l].ds[((size_t)bb * Cs + cl) * grid_sz + ij];
        }
        dsignal[e] = v;
    }
}

static int st_fill(StArgs& a, const float* signal, int batch, int c_signal, int fh, int fw, const hs_s2w_train_layer* layers, int n_layers,
                   int mode, int* n_wg) {
    a.signal = signal; a.c_signal = c_signal; a.grid_sz = fh * fw; a.n_patches = batch * fh * fw; a.n_layers = n_layers;
    int wgs = 0;
    long dw_floats = 0;
    a.dw_partial = nullptr; a.dw_slice_floats = 0;
    for (int i = 0; i < n_layers; ++i) {
        const hs_s2w_train_layer& l = layers[i];
        StLayer& d = a.layer[i];
        d.dw_off = dw_floats;
        dw_floats += (long)l.wc * (l.signal_channels / l.groups);
        d.w = l.w; d.dbank = l.dbank; d.dw = l.dw; d.ds = l.ds; d.ld = (long)l.ld;
        d.signal_index = l.signal_index; d.K = l.signal_channels / l.groups; d.rpg = l.wc / l.groups; d.wc = l.wc; d.rows = l.rows;
        d.groups = l.groups;
        d.blocks_per_group = mode == 0 ? (d.rpg + ST_TILE - 1) / ST_TILE : (a.n_patches + ST_TILE - 1) / ST_TILE;
        d.wg_begin = wgs;
        const bool wanted = mode == 0 ? (l.dw != nullptr) : (l.ds != nullptr);
        wgs += wanted ? l.groups * d.blocks_per_group : 0;
    }
    for (int i = n_layers; i < S2W_MAX_LAYERS; ++i) { a.layer[i] = a.layer[0]; a.layer[i].wg_begin = 0x7fffffff; a.layer[i].dw_off = 0x7fffffffffffffffL; }
    a.dw_slice_floats = dw_floats;
    *n_wg = wgs;
    return HS_OK;
}

static int st_check(const float* signal, int batch, int c_signal, int fh, int fw, const hs_s2w_train_layer* layers, int n_layers) {
    if (!signal || !layers || n_layers <= 0 || n_layers > S2W_MAX_LAYERS || batch <= 0 || fh <= 0 || fw <= 0 || c_signal <= 0) return HS_ERR_BAD_ARG;
    if ((size_t)batch * c_signal * fh * fw >= (1ull << 31)) return HS_ERR_UNSUPPORTED;
    for (int i = 0; i < n_layers; ++i) {
        const hs_s2w_train_layer& l = layers[i];
        if (!l.w || l.groups <= 0 || l.rows <= 0 || l.wc <= 0 || l.ld < l.rows || l.rows > l.wc) return HS_ERR_BAD_ARG;
        if (l.signal_index < 0 || l.signal_channels <= 0 || l.signal_index + l.signal_channels > c_signal) return HS_ERR_BAD_ARG;
        if (l.signal_channels % l.groups != 0 || l.wc % l.groups != 0) return HS_ERR_BAD_ARG;
        if (l.signal_channels / l.groups > ST_KMAX) return HS_ERR_UNSUPPORTED;
        if ((size_t)batch * fh * fw * (size_t)l.ld >= (1ull << 31)) return HS_ERR_UNSUPPORTED;      // 32-bit element offsets into dBank
    }
    return HS_OK;
}

}  // namespace hs

using namespace hs;

extern "C" int hs_s2w_train_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                const hs_s2w_train_layer* layers, int32_t n_layers, void* stream) {
    const int chk = st_check(signal, batch, c_signal, fh, fw, layers, n_layers);
    if (chk != HS_OK) return chk;
    hs_s2w_layer tab[S2W_MAX_LAYERS];
    for (int i = 0; i < n_layers; ++i) {
        if (!layers[i].bank) return HS_ERR_BAD_ARG;
        tab[i].signal_index = layers[i].signal_index; tab[i].signal_channels = layers[i].signal_channels; tab[i].groups = layers[i].groups;
        tab[i].wsw_t = layers[i].w; tab[i].wc = layers[i].wc; tab[i].rows = layers[i].rows;
        tab[i].bank = layers[i].bank; tab[i].ld = layers[i].ld; tab[i].wsw_blk = nullptr;
    }
    return s2w_multi_launch(signal, batch, c_signal, fh, fw, tab, n_layers, true, stream);
}

extern "C" int64_t hs_s2w_train_workspace(int32_t batch, int32_t fh, int32_t fw, const hs_s2w_train_layer* layers, int32_t n_layers) {
    if (!layers || n_layers <= 0 || n_layers > S2W_MAX_LAYERS || batch <= 0 || fh <= 0 || fw <= 0) return 0;
    long dw_floats = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (layers[i].groups <= 0) return 0;
        dw_floats += (long)layers[i].wc * (layers[i].signal_channels / layers[i].groups);
    }
    const long slices = ((long)batch * fh * fw + ST_SLICE - 1) / ST_SLICE;
    return (int64_t)(slices * dw_floats * 4);
}

extern "C" int hs_s2w_train_bwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                const hs_s2w_train_layer* layers, int32_t n_layers, float* dsignal, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    const int chk = st_check(signal, batch, c_signal, fh, fw, layers, n_layers);
    if (chk != HS_OK) return chk;
    hipStream_t s = (hipStream_t)stream;
    StArgs a;
    int n_wg = 0;
    st_fill(a, signal, batch, c_signal, fh, fw, layers, n_layers, 0, &n_wg);
    if (n_wg > 0) {
        const int n_slices = (a.n_patches + ST_SLICE - 1) / ST_SLICE;
        const bool sliced = workspace && n_slices > 1 && workspace_bytes >= hs_s2w_train_workspace(batch, fh, fw, layers, n_layers) && n_slices <= 65535;
        a.dw_partial = sliced ? (float*)workspace : nullptr;
        hipLaunchKernelGGL(s2w_train_bwd_kernel<0>, dim3((unsigned)n_wg, sliced ? (unsigned)n_slices : 1u), dim3(256), 0, s, a);
        int st = launch_status();
        if (st != HS_OK) return st;
        if (sliced) {
            const unsigned blocks = (unsigned)((a.dw_slice_floats + 255) / 256 > 2048 ? 2048 : (a.dw_slice_floats + 255) / 256);
            hipLaunchKernelGGL(s2w_train_dwsum_kernel, dim3(blocks), dim3(256), 0, s, a, n_slices);
            st = launch_status();
            if (st != HS_OK) return st;
        }
        a.dw_partial = nullptr;
    }
    if (dsignal) {
        for (int i = 0; i < n_layers; ++i)
            if (!layers[i].ds) return HS_ERR_BAD_ARG;                 // d signal needs every layer's private buffer
        st_fill(a, signal, batch, c_signal, fh, fw, layers, n_layers, 1, &n_wg);
        hipLaunchKernelGGL(s2w_train_bwd_kernel<1>, dim3((unsigned)n_wg), dim3(256), 0, s, a);
        int st = launch_status();
        if (st != HS_OK) return st;
        const long total = (long)batch * c_signal * fh * fw;
        const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(s2w_train_dsum_kernel, dim3(blocks), dim3(256), 0, s, a, dsignal, total);
        st = launch_status();
        if (st != HS_OK) return st;
    }
    return HS_OK;
}

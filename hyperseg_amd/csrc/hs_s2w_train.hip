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

struct StLayer {
    const float* __restrict__ w;            // (wc, K)
    const float* __restrict__ dbank;        // (P, ld) or null
    float* __restrict__ dw;                 // (wc, K) or null
    float* __restrict__ ds;                 // (B, Cs, grid) level-private, or null
    long ld;
    int signal_index, K, rpg, wc, rows, groups;
    int blocks_per_group;                   // row blocks (mode 0) / patch blocks (mode 1)
    int wg_begin;
};
struct StArgs {
    const float* __restrict__ signal;
    int c_signal, grid_sz, n_patches, n_layers;
    StLayer layer[S2W_MAX_LAYERS];
};

// MODE 0: dW tile = rows [n0, n0 + 64) of group g, reduction over the patches.   A[i][j] = dBank[patch i][row j], B[i][k] = S[patch i][k]
// MODE 1: dS tile = patches [p0, p0 + 64) of group g, reduction over the group's rows.  A[i][j] = dBank[patch j][row i], B[i][k] = W[row i][k]
template <int MODE>
__global__ __launch_bounds__(256)
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
    f32x4 acc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int out0 = blk * ST_TILE;                                   // first row of the group (mode 0) / first patch (mode 1) of the tile
    const int red_n = MODE == 0 ? P : rpg;
    for (int c0 = 0; c0 < red_n; c0 += ST_TILE) {
        if (c0 > 0) __syncthreads();
        // ---- A: dBank, coalesced along the bank's rows
        if (MODE == 0) {
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {                            // i = patch of the chunk, j = lo = row of the tile
                const int i = hi + 4 * q, p = c0 + i, r = out0 + lo, n = g * rpg + r;
                const bool ok = dbank && p < P && r < rpg && n < rows;
                A[i * ST_APAD + lo] = ok ? dbank[(size_t)p * ld + n] : 0.0f;
            }
        } else {
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {                            // j = patch of the tile, i = lo = row of the chunk
                const int j = hi + 4 * q, p = out0 + j, r = c0 + lo, n = g * rpg + r;
                const bool ok = dbank && p < P && r < rpg && n < rows;
                A[lo * ST_APAD + j] = ok ? dbank[(size_t)p * ld + n] : 0.0f;
            }
        }
        // ---- B
        if (MODE == 0) {                                              // S[patch c0 + lo][k]: consecutive lanes = consecutive patches
            const int p = min(c0 + lo, P - 1);
            const int bb = p / grid_sz, ij = p - bb * grid_sz;
            const float* sp = signal + ((size_t)bb * c_signal + sidx + g * K) * grid_sz + ij;
            for (int k = hi; k < KP; k += 4)
                Bm[lo * ST_KMAX + k] = (k < K && c0 + lo < P) ? sp[(size_t)k * grid_sz] : 0.0f;
        } else {                                                      // W[row c0 + i][k]: one linear run of 64 K floats
            const int n_first = g * rpg + c0;
            for (int e = tid; e < ST_TILE * KP; e += 256) {
                const int i = e / KP, k = e - i * KP;
                const int n = n_first + i;
                Bm[i * ST_KMAX + k] = (k < K && c0 + i < rpg && n < wc) ? w[(size_t)n * K + k] : 0.0f;
            }
        }
        __syncthreads();
        // ---- acc[j][k] += A[i][j] B[i][k]: lane = j, the wave owns the k quads hi, hi + 4, ... (broadcast reads of B)
        for (int i = 0; i < ST_TILE; ++i) {
            const float av = A[i * ST_APAD + lo];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int kq = hi + 4 * q;
                if (4 * kq < KP) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bm + i * ST_KMAX + 4 * kq);
                    acc[q][0] = fmaf(av, b4[0], acc[q][0]); acc[q][1] = fmaf(av, b4[1], acc[q][1]);
                    acc[q][2] = fmaf(av, b4[2], acc[q][2]); acc[q][3] = fmaf(av, b4[3], acc[q][3]);
                }
            }
        }
    }
    // ---- store
    if (MODE == 0) {
        float* __restrict__ dw = ka->layer[li].dw;
        const int r = out0 + lo, n = g * rpg + r;
        if (dw && r < rpg && n < wc) {
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * (hi + 4 * q) + e;
                    if (k < K) dw[(size_t)n * K + k] = acc[q][e];        // rows past `rows` accumulated zeros only
                }
        }
    } else {
        float* __restrict__ ds = ka->layer[li].ds;
        const int p = out0 + lo;
        if (ds && p < P) {
            const int bb = p / grid_sz, ij = p - bb * grid_sz;
            const int Cs = K * ka->layer[li].groups;
            float* dp = ds + ((size_t)bb * Cs + g * K) * grid_sz + ij;
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * (hi + 4 * q) + e;
                    if (k < K) dp[(size_t)k * grid_sz] = acc[q][e];
                }
        }
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
    for (int i = 0; i < n_layers; ++i) {
        const hs_s2w_train_layer& l = layers[i];
        StLayer& d = a.layer[i];
        d.w = l.w; d.dbank = l.dbank; d.dw = l.dw; d.ds = l.ds; d.ld = (long)l.ld;
        d.signal_index = l.signal_index; d.K = l.signal_channels / l.groups; d.rpg = l.wc / l.groups; d.wc = l.wc; d.rows = l.rows;
        d.groups = l.groups;
        d.blocks_per_group = mode == 0 ? (d.rpg + ST_TILE - 1) / ST_TILE : (a.n_patches + ST_TILE - 1) / ST_TILE;
        d.wg_begin = wgs;
        const bool wanted = mode == 0 ? (l.dw != nullptr) : (l.ds != nullptr);
        wgs += wanted ? l.groups * d.blocks_per_group : 0;
    }
    for (int i = n_layers; i < S2W_MAX_LAYERS; ++i) { a.layer[i] = a.layer[0]; a.layer[i].wg_begin = 0x7fffffff; }
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

extern "C" int hs_s2w_train_bwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                const hs_s2w_train_layer* layers, int32_t n_layers, float* dsignal, void* stream) {
    const int chk = st_check(signal, batch, c_signal, fh, fw, layers, n_layers);
    if (chk != HS_OK) return chk;
    hipStream_t s = (hipStream_t)stream;
    StArgs a;
    int n_wg = 0;
    st_fill(a, signal, batch, c_signal, fh, fw, layers, n_layers, 0, &n_wg);
    if (n_wg > 0) {
        hipLaunchKernelGGL(s2w_train_bwd_kernel<0>, dim3((unsigned)n_wg), dim3(256), 0, s, a);
        const int st = launch_status();
        if (st != HS_OK) return st;
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

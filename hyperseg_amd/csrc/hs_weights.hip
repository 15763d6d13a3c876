// Filter-bank producers: signal2weights (grouped 1x1 conv -> patch-major bank), bank re-layout
// from the reference's channel-major weight tensor, and inference BatchNorm folding.
//
// HBM layout: bank[p*ld + m], p = (b*fh + i)*fw + j.  One patch's whole bank is one contiguous
// run of ld floats, so the stage kernels read it with full-line coalesced (or scalar-cache)
// loads; the reference instead keeps (B, hp, fh, fw) and pays a permute+reshape copy per level
// (hyperseg_v1_0.py:334-337, 491-492).
#include "hs_common.h"

namespace hs {

// ------------------------------------------------------------------------------------------
// signal2weights, all levels of a decoder in ONE launch, on the f32 MATRIX cores.
//
// Per layer and group this is a dense GEMM  bank[p, n] = sum_k S[p, k] * Wsw[n, k]  (M = rows of the group,
// N = patches, K = Cs/G = 7..80) -- exactly the case the north star reserves MFMA for.  One wave owns a strip of
// 32 bank rows (two 16x16 tiles sharing the signal operand) and walks 16-patch tiles with
// v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fma chain, bit-compatible with the VALU form):
//   A[i = row][k]     lane l holds Wsw_t[k0 + (l>>4)][n0 + (l&15)]      64-byte coalesced runs, L2 resident
//   B[k][j = patch]   lane l holds S[b, idx + g*K + k0 + (l>>4), ij]    16 consecutive patches = 64 bytes
//   D[i][j]           lane l holds rows n0 + 4*(l>>4) + {0..3} of patch p0 + (l&15) -> one 16-byte store
// No LDS, no scalar-cache traffic; operand loads are independent of the accumulator chain so they stream.
// ------------------------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int S2W_THREADS = 256;          // 4 waves
constexpr int S2W_STRIP = 32;             // bank rows per wave-task (2 MFMA row tiles)
constexpr int S2W_PT = 8;                 // 16-patch tiles per wave-task
constexpr int S2W_MAX_LAYERS = 8;

struct S2wLayer {
    const float* __restrict__ wsw_t;
    float* __restrict__ bank;
    long ld;
    int signal_index, cs_g, rows_per_group, wc, rows;
    int strips_per_group, strip_begin;     // strips of this layer start at blockIdx.x == strip_begin
};
struct S2wArgs {
    const float* __restrict__ signal;
    int c_signal, grid_sz, n_patches, n_layers;
    S2wLayer layer[S2W_MAX_LAYERS];
};

__global__ __launch_bounds__(S2W_THREADS)
void signal2weights_kernel(S2wArgs a) {
    // layer descriptor through the kernarg segment with a uniform index -> scalar loads, no register copies
    const __attribute__((address_space(4))) S2wArgs* ka =
        (const __attribute__((address_space(4))) S2wArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    int li = 0;
    for (int q = 1; q < ka->n_layers; ++q)
        if ((int)blockIdx.x >= ka->layer[q].strip_begin) li = q;
    const float* __restrict__ wsw_t = ka->layer[li].wsw_t;
    float* __restrict__ bank = ka->layer[li].bank;
    const long ld = ka->layer[li].ld;
    const int signal_index = ka->layer[li].signal_index, cs_g = ka->layer[li].cs_g;
    const int rpg = ka->layer[li].rows_per_group, wc = ka->layer[li].wc, rows = ka->layer[li].rows;
    const int spg = ka->layer[li].strips_per_group;
    const int strip = (int)blockIdx.x - ka->layer[li].strip_begin;
    const int grid_sz = ka->grid_sz, n_patches = ka->n_patches, c_signal = ka->c_signal;
    const float* __restrict__ signal = ka->signal;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = strip / spg;
    const int r0 = (strip - g * spg) * S2W_STRIP;            // first row of the strip inside the group
    const int n0 = g * rpg + r0;                             // natural row index
    const int lrow = lane & 15, lk = lane >> 4;

    // per-lane row validity of the two A tiles
    const bool a0_ok = (r0 + lrow) < rpg;
    const bool a1_ok = (r0 + 16 + lrow) < rpg;
    // clamped row addresses: masked lanes still read inside the (cs_g, wc) array
    const float* __restrict__ wa0 = wsw_t + min(n0 + lrow, wc - 1);        // + k*wc
    const float* __restrict__ wa1 = wsw_t + min(n0 + 16 + lrow, wc - 1);
    const int ksteps = (cs_g + 3) >> 2;
    const size_t sig_base = (size_t)(signal_index + g * cs_g) * grid_sz;

    const int tile0 = (blockIdx.y * 4 + wave) * S2W_PT;
    for (int t = 0; t < S2W_PT; ++t) {
        const int p0 = (tile0 + t) * 16;
        if (p0 >= n_patches) break;
        const int p = p0 + lrow;
        const bool p_ok = p < n_patches;
        const int bb = p_ok ? p / grid_sz : 0, ij = p_ok ? p - bb * grid_sz : 0;
        const float* __restrict__ sb = signal + (size_t)bb * c_signal * grid_sz + sig_base + ij;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ks = 0; ks < ksteps; ++ks) {
            const int k = ks * 4 + lk;
            const bool k_ok = k < cs_g;
            const int kc = k_ok ? k : 0;
            const float w0 = wa0[(size_t)kc * wc], w1 = wa1[(size_t)kc * wc];
            const float sv = sb[(size_t)kc * grid_sz];
            const float fa0 = (k_ok && a0_ok) ? w0 : 0.0f;
            const float fa1 = (k_ok && a1_ok) ? w1 : 0.0f;
            const float fb = (k_ok && p_ok) ? sv : 0.0f;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1, fb, acc1, 0, 0, 0);
        }
        if (!p_ok) continue;
        // D: this lane holds rows (tile row 4*lk + r) of patch p
        float* __restrict__ dst = bank + (size_t)p * ld + n0 + 4 * lk;
        const int rr0 = r0 + 4 * lk;                          // row inside the group of acc0[0]
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x4 v = half ? acc1 : acc0;
            const int rg = rr0 + 16 * half;                    // row in group
            const int n = n0 + 4 * lk + 16 * half;             // natural row
            float* d = dst + 16 * half;
            if (rg + 3 < rpg && n + 3 < rows && ((((size_t)p * ld + n) & 3) == 0)) {
                *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rg + r < rpg && n + r < rows) d[r] = v[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// bank_pack: (B, hp_total, fh, fw) channel-major -> patch-major, 32x32 LDS transpose tiles.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void bank_pack_kernel(const float* __restrict__ w, int hp_total, int grid_sz, int ch_offset,
                      int rows, float* __restrict__ bank, long ld) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int m_base = blockIdx.x * 32, q_base = blockIdx.y * 32;   // q = i*fw + j
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;         // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int m = m_base + r, q = q_base + tx;
        float v = 0.0f;
        if (m < rows && q < grid_sz) v = w[((size_t)b * hp_total + ch_offset + m) * grid_sz + q];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int q = q_base + r, m = m_base + tx;
        if (m < rows && q < grid_sz)
            bank[((size_t)b * grid_sz + q) * ld + m] = tile[tx][r];
    }
}

__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps,
                               int n, float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sc = gamma[i] / sqrtf(var[i] + eps);
    scale[i] = sc;
    shift[i] = beta[i] - mean[i] * sc;
}

}  // namespace hs

using namespace hs;

extern "C" int hs_signal2weights_multi_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                           const hs_s2w_layer* layers, int32_t n_layers, void* stream) {
    if (!signal || !layers || n_layers <= 0 || n_layers > S2W_MAX_LAYERS) return HS_ERR_BAD_ARG;
    if (batch <= 0 || fh <= 0 || fw <= 0) return HS_ERR_BAD_ARG;
    S2wArgs a;
    a.signal = signal; a.c_signal = c_signal; a.grid_sz = fh * fw; a.n_patches = batch * fh * fw; a.n_layers = n_layers;
    int strips = 0;
    for (int i = 0; i < n_layers; ++i) {
        const hs_s2w_layer& l = layers[i];
        if (!l.wsw_t || !l.bank || l.groups <= 0 || l.rows <= 0 || l.wc <= 0 || l.ld < l.rows || l.rows > l.wc) return HS_ERR_BAD_ARG;
        if (l.signal_index < 0 || l.signal_channels <= 0 || l.signal_index + l.signal_channels > c_signal) return HS_ERR_BAD_ARG;
        if (l.signal_channels % l.groups != 0 || l.wc % l.groups != 0) return HS_ERR_BAD_ARG;
        S2wLayer& d = a.layer[i];
        d.wsw_t = l.wsw_t; d.bank = l.bank; d.ld = (long)l.ld;
        d.signal_index = l.signal_index;
        d.cs_g = l.signal_channels / l.groups; d.rows_per_group = l.wc / l.groups; d.wc = l.wc; d.rows = l.rows;
        d.strips_per_group = (d.rows_per_group + S2W_STRIP - 1) / S2W_STRIP;
        d.strip_begin = strips;
        strips += d.strips_per_group * l.groups;
    }
    for (int i = n_layers; i < S2W_MAX_LAYERS; ++i) { a.layer[i] = a.layer[0]; a.layer[i].strip_begin = 0x7fffffff; }
    const int tiles = (a.n_patches + 15) / 16;
    dim3 grid(strips, (tiles + 4 * S2W_PT - 1) / (4 * S2W_PT));
    hipLaunchKernelGGL(signal2weights_kernel, grid, dim3(S2W_THREADS), 0, (hipStream_t)stream, a);
    return launch_status();
}

extern "C" int hs_signal2weights_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                     int32_t signal_index, int32_t signal_channels, int32_t groups,
                                     const float* wsw_t, int32_t wc, int32_t rows,
                                     float* bank, int64_t ld, void* stream) {
    hs_s2w_layer l;
    l.signal_index = signal_index; l.signal_channels = signal_channels; l.groups = groups;
    l.wsw_t = wsw_t; l.wc = wc; l.rows = rows; l.bank = bank; l.ld = ld;
    return hs_signal2weights_multi_fwd(signal, batch, c_signal, fh, fw, &l, 1, stream);
}

extern "C" int hs_bank_pack_fwd(const float* w, int32_t batch, int32_t hp_total, int32_t fh, int32_t fw,
                                int32_t ch_offset, int32_t rows, float* bank, int64_t ld, void* stream) {
    if (!w || !bank || batch <= 0 || fh <= 0 || fw <= 0 || rows <= 0 || ld < rows || ch_offset < 0) return HS_ERR_BAD_ARG;
    if (ch_offset + rows > hp_total) return HS_ERR_BAD_ARG;
    const int grid_sz = fh * fw;
    dim3 grid((rows + 31) / 32, (grid_sz + 31) / 32, batch);
    hipLaunchKernelGGL(bank_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       w, hp_total, grid_sz, ch_offset, rows, bank, (long)ld);
    return launch_status();
}

extern "C" int hs_bn_fold_fwd(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                              int32_t n, float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !mean || !var || !scale || !shift || n <= 0) return HS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, mean, var, eps, n, scale, shift);
    return launch_status();
}

extern "C" int hs_version(void) { return HS_ABI_VERSION; }
extern "C" const char* hs_build_info(void) { return "libhyperseg_hip gfx950 (CDNA4) fp32; built " __DATE__ " " __TIME__; }

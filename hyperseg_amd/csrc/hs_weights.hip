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
// signal2weights, all levels of a decoder in ONE launch.  Block = 256 threads = (2 adjacent bank
// rows per thread) x 16 patches of one layer.  The signal slice of the 16 patches is staged once in
// LDS as [channel][patch] so that one ds_read_b128 feeds 4 patches x 2 rows = 8 FMAs; the
// (transposed) Conv2d weight is read coalesced from L2, 8 k-steps of loads in flight at a time
// (the grid is only ~4 waves per CU, so latency has to be covered inside the wave).
// ------------------------------------------------------------------------------------------
constexpr int S2W_TP = 16;     // patches per block
constexpr int S2W_THREADS = 256;
constexpr int S2W_ROWS = 2 * S2W_THREADS;
constexpr int S2W_KU = 8;      // k-steps per load batch
constexpr int S2W_MAX_LAYERS = 8;

struct S2wLayer {
    const float* __restrict__ wsw_t;
    const int* __restrict__ row_src;
    float* __restrict__ bank;
    long ld;
    int signal_index, signal_channels, cs_g, rows_per_group, wc, rows;
    int block_begin;           // first blockIdx.x of this layer
};
struct S2wArgs {
    const float* __restrict__ signal;
    int c_signal, grid_sz, n_patches, n_layers;
    S2wLayer layer[S2W_MAX_LAYERS];
};

__global__ __launch_bounds__(S2W_THREADS)
void signal2weights_kernel(S2wArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_lds[];   // [signal_channels][16]
    const int tid = threadIdx.x;
    const int p0 = blockIdx.y * S2W_TP;
    int li = 0;
#pragma unroll
    for (int q = 1; q < S2W_MAX_LAYERS; ++q)
        if (q < a.n_layers && (int)blockIdx.x >= a.layer[q].block_begin) li = q;
    const S2wLayer& L = a.layer[li];
    const int signal_channels = L.signal_channels, cs_g = L.cs_g, wc = L.wc, rows = L.rows;

    // stage signal[b, signal_index + c, ij] for the block's 16 patches, patch index fastest
    {
        const int t = tid % S2W_TP;
        const int p = p0 + t;
        const bool ok = p < a.n_patches;
        const int bb = ok ? p / a.grid_sz : 0, ij = ok ? p - bb * a.grid_sz : 0;
        const float* __restrict__ src = a.signal + ((size_t)bb * a.c_signal + L.signal_index) * a.grid_sz + ij;
        constexpr int CSTEP = S2W_THREADS / S2W_TP;     // 16 channels per pass
        for (int c0 = tid / S2W_TP; c0 < signal_channels; c0 += 4 * CSTEP) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + q * CSTEP;
                v[q] = (ok && c < signal_channels) ? src[(size_t)c * a.grid_sz] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + q * CSTEP;
                if (c < signal_channels) s_lds[c * S2W_TP + t] = v[q];
            }
        }
    }
    __syncthreads();

    const int m0 = ((int)blockIdx.x - L.block_begin) * S2W_ROWS + 2 * tid;
    if (m0 >= rows) return;
    const bool has1 = (m0 + 1) < rows;
    const int n0 = L.row_src ? L.row_src[m0] : m0;
    const int n1 = has1 ? (L.row_src ? L.row_src[m0 + 1] : m0 + 1) : -1;
    const int g0 = n0 >= 0 ? n0 / L.rows_per_group : 0;
    const int g1 = n1 >= 0 ? n1 / L.rows_per_group : g0;
    const bool same = (g0 == g1);

    float acc0[S2W_TP], acc1[S2W_TP];
#pragma unroll
    for (int t = 0; t < S2W_TP; ++t) { acc0[t] = 0.0f; acc1[t] = 0.0f; }

    const float4* s0 = reinterpret_cast<const float4*>(s_lds + (size_t)g0 * cs_g * S2W_TP);
    const float4* s1 = reinterpret_cast<const float4*>(s_lds + (size_t)g1 * cs_g * S2W_TP);
    const float* __restrict__ w0p = L.wsw_t + (n0 >= 0 ? n0 : 0);
    const float* __restrict__ w1p = L.wsw_t + (n1 >= 0 ? n1 : 0);
    for (int k0 = 0; k0 < cs_g; k0 += S2W_KU) {
        float w0[S2W_KU], w1[S2W_KU];
#pragma unroll
        for (int q = 0; q < S2W_KU; ++q) {               // 16 independent L2 loads in flight
            const int k = k0 + q;
            const int kc = k < cs_g ? k : cs_g - 1;
            const float a0 = w0p[(size_t)kc * wc], a1 = w1p[(size_t)kc * wc];
            w0[q] = (n0 >= 0 && k < cs_g) ? a0 : 0.0f;
            w1[q] = (n1 >= 0 && k < cs_g) ? a1 : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < S2W_KU; ++q) {
            const int k = (k0 + q) < cs_g ? (k0 + q) : cs_g - 1;
            float4 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = s0[k * 4 + r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0[4 * r + 0] = fmaf(w0[q], v[r].x, acc0[4 * r + 0]);
                acc0[4 * r + 1] = fmaf(w0[q], v[r].y, acc0[4 * r + 1]);
                acc0[4 * r + 2] = fmaf(w0[q], v[r].z, acc0[4 * r + 2]);
                acc0[4 * r + 3] = fmaf(w0[q], v[r].w, acc0[4 * r + 3]);
            }
            if (!same) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = s1[k * 4 + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc1[4 * r + 0] = fmaf(w1[q], v[r].x, acc1[4 * r + 0]);
                acc1[4 * r + 1] = fmaf(w1[q], v[r].y, acc1[4 * r + 1]);
                acc1[4 * r + 2] = fmaf(w1[q], v[r].z, acc1[4 * r + 2]);
                acc1[4 * r + 3] = fmaf(w1[q], v[r].w, acc1[4 * r + 3]);
            }
        }
    }
    // bank rows m0, m0+1 of 16 patches: 8-byte stores, consecutive lanes -> consecutive rows
    const bool vec = has1 && ((L.ld & 1) == 0);
#pragma unroll
    for (int t = 0; t < S2W_TP; ++t) {
        const int p = p0 + t;
        if (p >= a.n_patches) break;
        float* dst = L.bank + (size_t)p * L.ld + m0;
        if (vec) {
            *reinterpret_cast<float2*>(dst) = make_float2(acc0[t], acc1[t]);
        } else {
            dst[0] = acc0[t];
            if (has1) dst[1] = acc1[t];
        }
    }
}

// ------------------------------------------------------------------------------------------
// bank_pack: (B, hp_total, fh, fw) channel-major -> patch-major, 32x32 LDS transpose tiles.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void bank_pack_kernel(const float* __restrict__ w, int hp_total, int grid_sz, int ch_offset,
                      const int* __restrict__ row_src, int rows, float* __restrict__ bank, long ld) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int m_base = blockIdx.x * 32, q_base = blockIdx.y * 32;   // q = i*fw + j
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;         // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int m = m_base + r, q = q_base + tx;
        float v = 0.0f;
        if (m < rows && q < grid_sz) {
            const int n = row_src ? row_src[m] : m;
            if (n >= 0) v = w[((size_t)b * hp_total + ch_offset + n) * grid_sz + q];
        }
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
    int blocks = 0;
    size_t lds = 0;
    for (int i = 0; i < n_layers; ++i) {
        const hs_s2w_layer& l = layers[i];
        if (!l.wsw_t || !l.bank || l.groups <= 0 || l.rows <= 0 || l.wc <= 0 || l.ld < l.rows) return HS_ERR_BAD_ARG;
        if (l.signal_index < 0 || l.signal_channels <= 0 || l.signal_index + l.signal_channels > c_signal) return HS_ERR_BAD_ARG;
        if (l.signal_channels % l.groups != 0 || l.wc % l.groups != 0) return HS_ERR_BAD_ARG;
        S2wLayer& d = a.layer[i];
        d.wsw_t = l.wsw_t; d.row_src = l.row_src; d.bank = l.bank; d.ld = (long)l.ld;
        d.signal_index = l.signal_index; d.signal_channels = l.signal_channels;
        d.cs_g = l.signal_channels / l.groups; d.rows_per_group = l.wc / l.groups; d.wc = l.wc; d.rows = l.rows;
        d.block_begin = blocks;
        blocks += (l.rows + S2W_ROWS - 1) / S2W_ROWS;
        const size_t need = (size_t)l.signal_channels * S2W_TP * sizeof(float);
        lds = need > lds ? need : lds;
    }
    for (int i = n_layers; i < S2W_MAX_LAYERS; ++i) { a.layer[i] = a.layer[0]; a.layer[i].block_begin = 0x7fffffff; }
    if (lds > 160 * 1024) return HS_ERR_LDS;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)signal2weights_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(blocks, (a.n_patches + S2W_TP - 1) / S2W_TP);
    hipLaunchKernelGGL(signal2weights_kernel, grid, dim3(S2W_THREADS), lds, (hipStream_t)stream, a);
    return launch_status();
}

extern "C" int hs_signal2weights_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                     int32_t signal_index, int32_t signal_channels, int32_t groups,
                                     const float* wsw_t, int32_t wc, const int32_t* row_src, int32_t rows,
                                     float* bank, int64_t ld, void* stream) {
    hs_s2w_layer l;
    l.signal_index = signal_index; l.signal_channels = signal_channels; l.groups = groups;
    l.wsw_t = wsw_t; l.wc = wc; l.row_src = row_src; l.rows = rows; l.bank = bank; l.ld = ld;
    return hs_signal2weights_multi_fwd(signal, batch, c_signal, fh, fw, &l, 1, stream);
}

extern "C" int hs_bank_pack_fwd(const float* w, int32_t batch, int32_t hp_total, int32_t fh, int32_t fw,
                                int32_t ch_offset, const int32_t* row_src, int32_t rows,
                                float* bank, int64_t ld, void* stream) {
    if (!w || !bank || batch <= 0 || fh <= 0 || fw <= 0 || rows <= 0 || ld < rows || ch_offset < 0) return HS_ERR_BAD_ARG;
    if (!row_src && ch_offset + rows > hp_total) return HS_ERR_BAD_ARG;
    const int grid_sz = fh * fw;
    dim3 grid((rows + 31) / 32, (grid_sz + 31) / 32, batch);
    hipLaunchKernelGGL(bank_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       w, hp_total, grid_sz, ch_offset, row_src, rows, bank, (long)ld);
    return launch_status();
}

extern "C" int hs_bn_fold_fwd(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                              int32_t n, float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !mean || !var || !scale || !shift || n <= 0) return HS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, mean, var, eps, n, scale, shift);
    return launch_status();
}

extern "C" int hs_ir_row_map(int32_t cin, int32_t hidden, int32_t c_out, int32_t* row_src) {
    if (cin <= 0 || hidden <= 0 || c_out <= 0) return HS_ERR_BAD_ARG;
    const int r2 = cin * hidden + 9 * hidden;
    const int rows = r2 + hidden * c_out;
    if (row_src) {
        for (int m = 0; m < r2; ++m) row_src[m] = m;
        for (int h = 0; h < hidden; ++h)
            for (int o = 0; o < c_out; ++o) row_src[r2 + h * c_out + o] = r2 + o * hidden + h;
    }
    return rows;
}

extern "C" int hs_version(void) { return HS_ABI_VERSION; }
extern "C" const char* hs_build_info(void) { return "libhyperseg_hip gfx950 (CDNA4) fp32; built " __DATE__ " " __TIME__; }

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
// signal2weights.  Block = 256 threads = (2 adjacent bank rows per thread) x 16 patches.
// The signal slice of the 16 patches is staged once in LDS as [channel][patch] so that one
// ds_read_b128 feeds 4 patches x 2 rows = 8 FMAs; the (transposed) Conv2d weight is read
// coalesced from L2 (it is re-read by every patch group; <= 1.4 MB per level).
// ------------------------------------------------------------------------------------------
constexpr int S2W_TP = 16;     // patches per block
constexpr int S2W_THREADS = 256;
constexpr int S2W_ROWS = 2 * S2W_THREADS;

__global__ __launch_bounds__(S2W_THREADS)
void signal2weights_kernel(const float* __restrict__ signal, int c_signal, int grid_sz, int n_patches,
                           int signal_index, int signal_channels, int cs_g, int rows_per_group,
                           const float* __restrict__ wsw_t, int wc,
                           const int* __restrict__ row_src, int rows,
                           float* __restrict__ bank, long ld) {
    extern __shared__ __attribute__((aligned(16))) float s_lds[];   // [signal_channels][16]
    const int tid = threadIdx.x;
    const int p0 = blockIdx.y * S2W_TP;

    // stage signal[b, signal_index + c, ij] for the block's 16 patches, patch index fastest
    for (int e = tid; e < signal_channels * S2W_TP; e += S2W_THREADS) {
        const int t = e % S2W_TP, c = e / S2W_TP;
        const int p = p0 + t;
        float v = 0.0f;
        if (p < n_patches) {
            const int b = p / grid_sz, ij = p - b * grid_sz;
            v = signal[((size_t)b * c_signal + signal_index + c) * grid_sz + ij];
        }
        s_lds[e] = v;
    }
    __syncthreads();

    const int m0 = blockIdx.x * S2W_ROWS + 2 * tid;
    if (m0 >= rows) return;
    const bool has1 = (m0 + 1) < rows;
    const int n0 = row_src ? row_src[m0] : m0;
    const int n1 = has1 ? (row_src ? row_src[m0 + 1] : m0 + 1) : -1;
    const int g0 = n0 >= 0 ? n0 / rows_per_group : 0;
    const int g1 = n1 >= 0 ? n1 / rows_per_group : g0;
    const bool same = (g0 == g1);

    float acc0[S2W_TP], acc1[S2W_TP];
#pragma unroll
    for (int t = 0; t < S2W_TP; ++t) { acc0[t] = 0.0f; acc1[t] = 0.0f; }

    const float4* s0 = reinterpret_cast<const float4*>(s_lds + (size_t)g0 * cs_g * S2W_TP);
    const float4* s1 = reinterpret_cast<const float4*>(s_lds + (size_t)g1 * cs_g * S2W_TP);
    const float* w0p = wsw_t + (n0 >= 0 ? n0 : 0);
    const float* w1p = wsw_t + (n1 >= 0 ? n1 : 0);
    for (int k = 0; k < cs_g; ++k) {
        const float w0 = n0 >= 0 ? w0p[(size_t)k * wc] : 0.0f;
        const float w1 = n1 >= 0 ? w1p[(size_t)k * wc] : 0.0f;
        float4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = s0[k * 4 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0[4 * q + 0] = fmaf(w0, a[q].x, acc0[4 * q + 0]);
            acc0[4 * q + 1] = fmaf(w0, a[q].y, acc0[4 * q + 1]);
            acc0[4 * q + 2] = fmaf(w0, a[q].z, acc0[4 * q + 2]);
            acc0[4 * q + 3] = fmaf(w0, a[q].w, acc0[4 * q + 3]);
        }
        if (!same) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = s1[k * 4 + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc1[4 * q + 0] = fmaf(w1, a[q].x, acc1[4 * q + 0]);
            acc1[4 * q + 1] = fmaf(w1, a[q].y, acc1[4 * q + 1]);
            acc1[4 * q + 2] = fmaf(w1, a[q].z, acc1[4 * q + 2]);
            acc1[4 * q + 3] = fmaf(w1, a[q].w, acc1[4 * q + 3]);
        }
    }
    // bank rows m0, m0+1 of 16 patches: 8-byte stores, consecutive lanes -> consecutive rows
    const bool vec = has1 && ((ld & 1) == 0);
#pragma unroll
    for (int t = 0; t < S2W_TP; ++t) {
        const int p = p0 + t;
        if (p >= n_patches) break;
        float* dst = bank + (size_t)p * ld + m0;
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

extern "C" int hs_signal2weights_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                     int32_t signal_index, int32_t signal_channels, int32_t groups,
                                     const float* wsw_t, int32_t wc, const int32_t* row_src, int32_t rows,
                                     float* bank, int64_t ld, void* stream) {
    if (!signal || !wsw_t || !bank) return HS_ERR_BAD_ARG;
    if (batch <= 0 || fh <= 0 || fw <= 0 || groups <= 0 || rows <= 0 || wc <= 0 || ld < rows) return HS_ERR_BAD_ARG;
    if (signal_index < 0 || signal_channels <= 0 || signal_index + signal_channels > c_signal) return HS_ERR_BAD_ARG;
    if (signal_channels % groups != 0 || wc % groups != 0) return HS_ERR_BAD_ARG;
    const size_t lds = (size_t)signal_channels * S2W_TP * sizeof(float);
    if (lds > 160 * 1024) return HS_ERR_LDS;
    const int n_patches = batch * fh * fw;
    dim3 grid((rows + S2W_ROWS - 1) / S2W_ROWS, (n_patches + S2W_TP - 1) / S2W_TP);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)signal2weights_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(signal2weights_kernel, grid, dim3(S2W_THREADS), lds, (hipStream_t)stream,
                       signal, c_signal, fh * fw, n_patches, signal_index, signal_channels,
                       signal_channels / groups, wc / groups, wsw_t, wc, row_src, rows, bank, (long)ld);
    return launch_status();
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

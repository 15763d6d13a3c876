// Encoder-side helper (SURVEY.md section 8f, rank 2 -- "next" row, opt-in through prepare_for_inference):
// depthwise k x k convolution + folded BatchNorm + swish in ONE launch for EfficientNet's MBConv blocks.
//
// Why it exists: on ROCm 7.2 MIOpen has no tuned fp32 depthwise solver for these shapes -- a steady-state profile of
// the stock encoder shows ~1.1 ms/frame in Winograd kernels run per group and ~0.7 ms/frame in
// naive_conv_ab_nonpacked_fwd_nchw (profiles/round1_model_graph_replay_kernels.txt), i.e. half of the frame, for an
// operation that moves ~0.1 GB.  The kernel below is a plain HBM-streaming stencil: one thread = 4 consecutive
// outputs of one row of one (batch, channel) plane, the K*K filter taps are wave-uniform (scalar loads), input
// rows are re-used through L1/L2, the BN affine and x*sigmoid(x) are applied before the single 16-byte store.
// Zero padding is TensorFlow-"SAME" style: arbitrary (top, left) offsets, so the asymmetric stride-2 case needs no
// separate pad kernel.  Replaces: F.pad + F.conv2d(groups=C) + BatchNorm2d + SiLU of
// hyperseg/models/backbones/efficientnet.py:59-66, 101-103.
#include "hs_common.h"

namespace hs {

template <int K, int S>
__global__ __launch_bounds__(256)
void depthwise_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                           const float* __restrict__ shift, float* __restrict__ y, int C, int H, int W, int Ho, int Wo,
                           int pad_t, int pad_l, int act, float* __restrict__ pool_partial) {
    const int plane = blockIdx.y;                        // b*C + c
    const int c = plane % C;
    const int wq = (Wo + 3) >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < Ho * wq;
    float psum = 0.0f;
    if (live) {
    const int yo = q / wq, xo = (q - yo * wq) * 4;
    const float* __restrict__ xp = x + (size_t)plane * H * W;
    const float* __restrict__ wc = w + (size_t)c * K * K;          // uniform -> scalar loads
    constexpr int NCOL = 3 * S + K;                                // input columns feeding 4 outputs
    const int xi0 = xo * S - pad_l, yi0 = yo * S - pad_t;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int yi = yi0 + ky;
        const bool row_ok = yi >= 0 && yi < H;
        const float* __restrict__ row = xp + (size_t)(row_ok ? yi : 0) * W;
        float v[NCOL];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            const int xi = xi0 + j;
            const bool ok = row_ok && xi >= 0 && xi < W;
            const float t = row[ok ? xi : 0];
            v[j] = ok ? t : 0.0f;
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const float wv = wc[ky * K + kx];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = fmaf(wv, v[t * S + kx], acc[t]);
        }
    }
    const float sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
    float o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float r = fmaf(acc[t], sc, sh);
        if (act == 3) r = r / (1.0f + expf(-r));                   // swish / SiLU
        else r = apply_act(r, act);
        o[t] = r;
    }
    float* dst = y + ((size_t)plane * Ho + yo) * Wo + xo;
    if ((Wo & 3) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        psum = (o[0] + o[1]) + (o[2] + o[3]);
    } else {
        for (int t = 0; t < 4 && xo + t < Wo; ++t) { dst[t] = o[t]; psum += o[t]; }
    }
    }
    // squeeze-excite pooling: deterministic per-workgroup partial sums of the outputs (summed by hs_se_gate_fwd)
    if (pool_partial) {
        __shared__ float wsum[4];
        for (int m = 32; m > 0; m >>= 1) psum += __shfl_xor(psum, m, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = psum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += wsum[i];
            pool_partial[(size_t)plane * gridDim.x + blockIdx.x] = t;
        }
    }
}

// Squeeze-excite gate of one MBConv block from the pooled partial sums: pooled -> 1x1 reduce + swish -> 1x1 expand ->
// sigmoid, one workgroup per batch element.  With w_proj != NULL the gate is folded into the block's project
// convolution, w_scaled[b, o, c] = w_proj[o, c] * gate[b, c]: scaling ~1e5 weights replaces a full elementwise pass over
// the (up to 100 MB) activation.  Replaces adaptive_avg_pool2d + 2 convs + swish + sigmoid + mul
// (hyperseg/models/backbones/efficientnet.py:106-111).
__global__ __launch_bounds__(256)
void se_gate_kernel(const float* __restrict__ partial, int nblk, float inv_hw, const float* __restrict__ w1,
                    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, int C, int Csq,
                    float* __restrict__ gate, const float* __restrict__ w_proj, int Cout, float* __restrict__ w_scaled) {
    extern __shared__ float sm[];            // pooled[C] | z[Csq] | gate[C]
    float* pooled = sm; float* z = sm + C; float* g = z + Csq;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += blockDim.x) {
        const float* __restrict__ p = partial + ((size_t)b * C + c) * nblk;
        float t = 0.0f;
        for (int i = 0; i < nblk; ++i) t += p[i];
        pooled[c] = t * inv_hw;
    }
    __syncthreads();
    for (int j = tid >> 6; j < Csq; j += (int)(blockDim.x >> 6)) {       // one wave per squeezed channel
        const float* __restrict__ wr = w1 + (size_t)j * C;
        float t = 0.0f;
        for (int c = tid & 63; c < C; c += 64) t = fmaf(wr[c], pooled[c], t);
        for (int m = 32; m > 0; m >>= 1) t += __shfl_xor(t, m, 64);
        if ((tid & 63) == 0) { t += b1[j]; z[j] = t / (1.0f + expf(-t)); }
    }
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        const float* __restrict__ wr = w2 + (size_t)c * Csq;
        float t = b2[c];
        for (int j = 0; j < Csq; ++j) t = fmaf(wr[j], z[j], t);
        const float gv = 1.0f / (1.0f + expf(-t));
        g[c] = gv;
        if (gate) gate[(size_t)b * C + c] = gv;
    }
    if (!w_proj) return;
    __syncthreads();
    const size_t n = (size_t)Cout * C;
    for (size_t e = tid; e < n; e += blockDim.x) w_scaled[(size_t)b * n + e] = w_proj[e] * g[e % C];
}

}  // namespace hs

using namespace hs;

extern "C" int hs_depthwise_conv_fwd(const float* x, int32_t batch, int32_t channels, int32_t H, int32_t W,
                                     const float* w, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                                     int32_t Ho, int32_t Wo, const float* scale, const float* shift, int32_t act,
                                     float* y, float* pool_partial, void* stream) {
    if (!x || !w || !y || batch <= 0 || channels <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return HS_ERR_BAD_ARG;
    if (pad_t < 0 || pad_l < 0 || (scale && !shift)) return HS_ERR_BAD_ARG;
    if ((long)batch * channels > 65535) return HS_ERR_UNSUPPORTED;
    const int quads = Ho * ((Wo + 3) / 4);
    const int threads = quads >= 256 ? 256 : ((quads + 63) / 64) * 64;
    dim3 grid((quads + threads - 1) / threads, batch * channels);
    hipStream_t s = (hipStream_t)stream;
#define HS_DW(KK, SS) hipLaunchKernelGGL((depthwise_conv_kernel<KK, SS>), grid, dim3(threads), 0, s, x, w, scale, shift, y, \
                                         channels, H, W, Ho, Wo, pad_t, pad_l, act, pool_partial)
    if (k == 3 && stride == 1) HS_DW(3, 1);
    else if (k == 3 && stride == 2) HS_DW(3, 2);
    else if (k == 5 && stride == 1) HS_DW(5, 1);
    else if (k == 5 && stride == 2) HS_DW(5, 2);
    else return HS_ERR_UNSUPPORTED;
#undef HS_DW
    return launch_status();
}

extern "C" int hs_depthwise_pool_blocks(int32_t Ho, int32_t Wo) {
    const int quads = Ho * ((Wo + 3) / 4);
    const int threads = quads >= 256 ? 256 : ((quads + 63) / 64) * 64;
    return (quads + threads - 1) / threads;
}

extern "C" int hs_se_gate_fwd(const float* partial, int32_t batch, int32_t channels, int32_t nblk, float inv_hw,
                              const float* w_reduce, const float* b_reduce, int32_t c_squeezed, const float* w_expand,
                              const float* b_expand, float* gate, const float* w_proj, int32_t c_out, float* w_scaled,
                              void* stream) {
    if (!partial || !w_reduce || !b_reduce || !w_expand || !b_expand || batch <= 0 || channels <= 0 || nblk <= 0 ||
        c_squeezed <= 0) return HS_ERR_BAD_ARG;
    if ((w_proj != nullptr) != (w_scaled != nullptr) || (w_proj && c_out <= 0) || (!gate && !w_proj)) return HS_ERR_BAD_ARG;
    const size_t lds = (size_t)(2 * channels + c_squeezed) * sizeof(float);
    if (lds > 64 * 1024) return HS_ERR_LDS;
    hipLaunchKernelGGL(se_gate_kernel, dim3(batch), dim3(256), lds, (hipStream_t)stream, partial, nblk, inv_hw, w_reduce,
                       b_reduce, w_expand, b_expand, channels, c_squeezed, gate, w_proj, c_out, w_scaled);
    return launch_status();
}

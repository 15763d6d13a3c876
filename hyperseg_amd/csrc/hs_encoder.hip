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
#include "hs_se_tail.h"

namespace hs {

// PL >= 0: the left padding is the compile-time constant PL and W % 4 == 0 -- every tap row is then fetched as aligned
// 16-byte loads (3-4 per row instead of 6-11 dword loads; the kernel is bound by the texture-address path, which spends
// the same cycles on a 4-byte as on a 16-byte per-lane access).  PL = -1: arbitrary pad_l / W, dword loads.
// TILE (only with the BN0 + swish prologue, whole output rows per workgroup): the workgroup's input band goes through LDS
// ONCE, with the prologue applied once per input element -- on load from L1 every tap of every output re-applied it
// (K*(3S+K)/4 = 8.75 swishes per output at K = 5: the batch-32 launches of HyperSeg-L were bound by exactly that: 90 us
// per 5x5 launch) -- zero padding materialised in the tile, tap rows read back as aligned ds_read_b128.
// SET: the squeeze-excite tail of round 5 (hs_se_tail.h, opt-in) compiled in.  The product's launches instantiate SET = false: the
// tail is two thirds of the kernel's code and its descriptor 20 more kernel-argument dwords (round 6: HyperSeg-M frame 0.7436 ->
// 0.7407 ms, profiles/round6_depthwise_without_se_tail_ab_w11.txt).
template <int K, int S, int PL, bool TILE, bool SET>
__global__ __launch_bounds__(256)
void depthwise_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                           const float* __restrict__ shift, float* __restrict__ y, int C, int H, int W, int Ho, int Wo,
                           int pad_t, int pad_l, int act, float* __restrict__ pool_partial,
                           const float* __restrict__ in_scale, const float* __restrict__ in_shift, int nplanes, SeTail se) {
    const int plane = blockIdx.y + blockIdx.z * gridDim.y;       // b*C + c (folded over y/z: more than 65535 planes at bs 32)
    if (plane >= nplanes) return;
    const int c = plane % C;
    unsigned se_gen = 0u;                                        // hs_se_tail.h: requested here, needed when the partial is published
    if constexpr (SET) se_gen = se.ws ? se_tag(se, plane / C) : 0u;
    // optional prologue: the taps are swish(in_scale[c] * x + in_shift[c]) -- the BatchNorm + swish of the 1x1 expand
    // convolution that produced x, applied on load so that the raw GEMM output needs no elementwise pass of its own
    // (fetched AFTER the taps have been requested in the untiled form: the compiler issues a scalar load where the source
    // has it and drains the scalar queue before the next address computation -- at the top of the kernel these two cost a
    // memory round trip before the first tap load went out; tools/isa_phases.py)
    const bool pre = in_scale != nullptr;
    float isc = 1.0f, ish = 0.0f;
    if constexpr (TILE) {
        if (pre) { isc = in_scale[c]; ish = in_shift[c]; }
    }
    const int wq = (Wo + 3) >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < Ho * wq;
    float psum = 0.0f;
    const float* __restrict__ xp = x + (size_t)plane * H * W;
    extern __shared__ __attribute__((aligned(16))) float dw_tile[];
    const int tw = ((Wo - 1) * S + K + 3) & ~3;                    // TILE: tile row = input columns [-pad_l, -pad_l + tw)
    const int yo0 = (int)(blockIdx.x * blockDim.x) / wq;           // TILE: first output row of this workgroup (wq | blockDim.x)
    if constexpr (TILE) {
        const int rows_out = (int)blockDim.x / wq, rows_in = (rows_out - 1) * S + K;
        for (int e = threadIdx.x; e < rows_in * tw; e += blockDim.x) {
            const int r = e / tw, t = e - r * tw;
            const int yi = yo0 * S - pad_t + r, xi = t - pad_l;
            float val = 0.0f;
            if (yi >= 0 && yi < H && xi >= 0 && xi < W) {
                val = xp[(size_t)yi * W + xi];
                if (pre) val = swishf(fmaf(val, isc, ish));
            }
            dw_tile[e] = val;
        }
        __syncthreads();
    }
    if (live) {
    const int yo = q / wq, xo = (q - yo * wq) * 4;
    const float* __restrict__ wc = w + (size_t)c * K * K;          // uniform -> scalar loads
    constexpr int NCOL = 3 * S + K;                                // input columns feeding 4 outputs
    const int xi0 = xo * S - pad_l, yi0 = yo * S - pad_t;
    // every tap of the K x NCOL window is loaded BEFORE the first use (clamped addresses, masks applied afterwards): a
    // row-by-row load/fma interleaving exposes one memory round trip per tap row
    float v[K][NCOL];
    if constexpr (TILE) {
        constexpr int NV = (NCOL + 3) / 4;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const float* row = dw_tile + ((yo - yo0) * S + ky) * tw + xo * S;      // 16-byte aligned: tw, xo multiples of 4
            float win[NV * 4];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(row + 4 * i);   // tw - xo * S is a multiple of 4 and >= NCOL
                win[4 * i] = t.x; win[4 * i + 1] = t.y; win[4 * i + 2] = t.z; win[4 * i + 3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < NCOL; ++j) v[ky][j] = win[j];
        }
    } else if constexpr (PL >= 0) {
        constexpr int OFF = (4 - PL % 4) % 4;                      // xi0 = 4*q*S - PL: misalignment of the window start
        constexpr int NV = (OFF + NCOL + 3) / 4;
        const int a0 = xi0 - OFF;                                  // multiple of 4
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yi = yi0 + ky;
            const float* __restrict__ row = xp + (size_t)min(max(yi, 0), H - 1) * W;
            float win[NV * 4];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int col = a0 + 4 * i;                        // W % 4 == 0: a quad is entirely inside or outside
                const float4 t = *reinterpret_cast<const float4*>(row + ((col >= 0 && col < W) ? col : 0));
                win[4 * i] = t.x; win[4 * i + 1] = t.y; win[4 * i + 2] = t.z; win[4 * i + 3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < NCOL; ++j) v[ky][j] = win[OFF + j];
        }
    } else {
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yi = yi0 + ky;
            const float* __restrict__ row = xp + (size_t)min(max(yi, 0), H - 1) * W;
#pragma unroll
            for (int j = 0; j < NCOL; ++j) v[ky][j] = row[min(max(xi0 + j, 0), W - 1)];
        }
    }
    if constexpr (!TILE) {
        if (pre) { isc = in_scale[c]; ish = in_shift[c]; }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int yi = yi0 + ky;
        const bool row_ok = yi >= 0 && yi < H;
        if constexpr (!TILE) {
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const int xi = xi0 + j;
                const bool ok = row_ok && xi >= 0 && xi < W;
                float t = v[ky][j];
                if (pre) t = swishf(fmaf(t, isc, ish));
                v[ky][j] = ok ? t : 0.0f;
            }
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const float wv = wc[ky * K + kx];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = fmaf(wv, v[ky][t * S + kx], acc[t]);
        }
    }
    const float sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
    float o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float r = fmaf(acc[t], sc, sh);
        if (act == 3) r = swishf(r);                               // swish / SiLU
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
    if (pool_partial || (SET && se.ws)) {
        __shared__ float wsum[4];
        psum = wave_sum64(psum);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = psum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += wsum[i];
            if (SET && se.ws) se_publish(se.ws, se_ws_pg(se) + (size_t)plane * gridDim.x + blockIdx.x, t, se_gen);
            else pool_partial[(size_t)plane * gridDim.x + blockIdx.x] = t;
        }
    }
    // the squeeze-excite gate by the last workgroups of the launch (hs_se_tail.h): the same partial sums, summed in the same order
    // (its LDS: the dynamic segment -- the input tile of the TILE form is dead by now; the host sizes it for both uses)
    if constexpr (SET) {
        if (se.ws) {
            if constexpr (TILE) __syncthreads();
            se_tail_run(se, plane / C, (long)c * gridDim.x + blockIdx.x, (long)C * gridDim.x, se_gen, dw_tile);
        }
    }
}

// Squeeze-excite gate of one MBConv block from the pooled partial sums, as TWO multi-workgroup launches that each
// make a single trip to memory (the first version was one workgroup walking pool -> reduce -> expand serially: three
// dependent load phases and up to 1.2 MB of weights through one CU, 14 us per block):
//
//   se_squeeze_kernel  grid (Csq, B): z[b,j] = swish(b1[j] + (1/HW) * sum_{c,i} w1[j,c] * partial[b,c,i])
//       the average pool and the 1x1 reduce conv are one dot product over the flattened (c, i) partials -- the w1 gather
//       index depends only on the element index, so partials and weights are all in flight together.
//   se_excite_kernel   grid (ceil(C/64), B, ceil(Cout/64)): gate[b,c] = sigmoid(b2[c] + sum_j w2t[j,c] * z[b,j]) for a
//       strip of 64 channels (4 waves each take a quarter of j), then w_scaled[b,o,c] = w_proj[o,c] * gate[b,c]
//       (* out_scale[o]) for 64 rows of the strip: the gate (and optionally the project conv's BatchNorm scale) folded
//       into the project weights -- scaling ~1e5 weights replaces an elementwise pass over the activation.
// (Tried and rejected: (i) both phases in one launch of <= 128 co-resident workgroups around a bounded device-scope barrier.
//  Bit-identical, but 9-19 us per call against 9.5-10.5 us for the two launches: the agent-scope fences and cross-XCD
//  atomics cost more than the ~4.5 us launch they save -- 924 vs 979 frames/s, gpurun round r1i.  (ii) round 2: one launch
//  in which every excite workgroup re-derives the squeezed vector for itself (no cross-workgroup traffic at all): 6.5-14 us
//  for the narrow blocks, 30-170 us for the wide ones -- each workgroup then pulls the whole Csq x C reduce weight (up to
//  221 KB) through dependent L2 round trips; 619 vs 1006 frames/s, gpurun round r2i.)
// Replaces adaptive_avg_pool2d + 2 convs + swish + sigmoid + mul (hyperseg/models/backbones/efficientnet.py:106-111).
// z[b, j] for one squeezed channel j: block-wide dot product over the flattened (channel, partial) index
__device__ __forceinline__ void se_squeeze_body(int j, int b, const float* __restrict__ partial, int nblk, float inv_hw,
                                                const float* __restrict__ w1, const float* __restrict__ b1, int C, int Csq,
                                                float* __restrict__ z, float* ws /* LDS [4] */) {
    const int tid = threadIdx.x;
    const int n = C * nblk;
    const float* __restrict__ p = partial + (size_t)b * n;
    const float* __restrict__ wr = w1 + (size_t)j * C;
    float acc = 0.0f;
    if ((nblk & 3) == 0) {
        // 16-byte loads of the partials: the four elements of a quad belong to the same channel
        const int n4 = n >> 2, q4 = nblk >> 2;
        for (int e0 = tid; e0 < n4; e0 += 256 * 8) {
            float4 pv[8]; float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + 256 * u, n4 - 1);
                pv[u] = reinterpret_cast<const float4*>(p)[e];
                wv[u] = wr[e / q4];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + 256 * u < n4) acc = fmaf((pv[u].x + pv[u].y) + (pv[u].z + pv[u].w), wv[u], acc);
        }
    } else {
        for (int e0 = tid; e0 < n; e0 += 256 * 8) {
            float pv[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + 256 * u, n - 1);
                pv[u] = p[e];
                wv[u] = wr[e / nblk];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + 256 * u < n) acc = fmaf(pv[u], wv[u], acc);
        }
    }
    acc = wave_sum64(acc);
    __syncthreads();                         // ws may still be read by a previous call of this body
    if ((tid & 63) == 0) ws[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        const float t = fmaf((ws[0] + ws[1]) + (ws[2] + ws[3]), inv_hw, b1[j]);
        z[(size_t)b * Csq + j] = swishf(t);
    }
}

// gate for the 64-channel strip cb (+ scaled project-weight rows of row group rg)
__device__ __forceinline__ void se_excite_body(int cb, int rg, int b, const float* __restrict__ z,
                                               const float* __restrict__ w2t, const float* __restrict__ b2, int C, int Csq,
                                               float* __restrict__ gate, const float* __restrict__ w_proj, int Cout,
                                               const float* __restrict__ out_scale, float* __restrict__ w_scaled,
                                               float (*red)[64] /* LDS [4][64] */) {
    const int tid = threadIdx.x, col = tid & 63, part = tid >> 6;
    const int c = cb * 64 + col, cc = min(c, C - 1);
    const int o0 = rg * 64 + part * 16;
    // project-weight rows of this thread: issued before the gate so that both sets of loads share one round trip
    float wp[16], osc[16];
    if (w_proj) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            wp[r] = w_proj[(size_t)min(o0 + r, Cout - 1) * C + cc];
            osc[r] = out_scale ? out_scale[min(o0 + r, Cout - 1)] : 1.0f;     // with the rows: not one round trip per store
        }
    }
    const float b2v = b2[cc];
    const int jq = (Csq + 3) >> 2, j0 = part * jq, j1 = min(j0 + jq, Csq);
    const float* __restrict__ zb = z + (size_t)b * Csq;
    float acc = 0.0f;
    for (int jb = j0; jb < j1; jb += 8) {
        float wv[8], zv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = min(jb + u, Csq - 1);
            wv[u] = w2t[(size_t)j * C + cc];
            zv[u] = zb[j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (jb + u < j1) acc = fmaf(wv[u], zv[u], acc);
    }
    __syncthreads();                         // red may still be read by a previous call of this body
    red[part][col] = acc;
    __syncthreads();
    const float t = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + b2v;
    const float g = sigmoidf_fast(t);
    if (rg == 0 && part == 0 && c < C) gate[(size_t)b * C + c] = g;
    if (w_proj && c < C) {
        float* __restrict__ dst = w_scaled + (size_t)b * Cout * C + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + r;
            if (o < Cout) dst[(size_t)o * C] = wp[r] * g * osc[r];
        }
    }
}

__global__ __launch_bounds__(256)
void se_squeeze_kernel(const float* __restrict__ partial, int nblk, float inv_hw, const float* __restrict__ w1,
                       const float* __restrict__ b1, int C, int Csq, float* __restrict__ z) {
    __shared__ float ws[4];
    se_squeeze_body(blockIdx.x, blockIdx.y, partial, nblk, inv_hw, w1, b1, C, Csq, z, ws);
}

__global__ __launch_bounds__(256)
void se_excite_kernel(const float* __restrict__ z, const float* __restrict__ w2t, const float* __restrict__ b2, int C,
                      int Csq, float* __restrict__ gate, const float* __restrict__ w_proj, int Cout,
                      const float* __restrict__ out_scale, float* __restrict__ w_scaled) {
    __shared__ float red[4][64];
    se_excite_body(blockIdx.x, blockIdx.z, blockIdx.y, z, w2t, b2, C, Csq, gate, w_proj, Cout, out_scale, w_scaled, red);
}

// The EARLY blocks' gate in one launch (round 6): few channels (<= 144), a handful of squeezed channels (<= 8) -- but 64-256 pool
// partials per channel, which is why they took the squeeze + excite pair (10 us for 25 k additions).  Round 5's wide instantiation of
// se_gate_fused_kernel lost to that pair because every thread added a channel's 64-256 partials serially; here a channel's partials
// are ONE 16-byte load per lane of a wave (<= 64 loads x 4 = 256 partials), every load of the workgroup is issued before the first
// use, a channel's sum is four in-lane additions + the DPP wave sum, and one workgroup does the whole gate: partials -> means ->
// squeezed -> gate, two barriers.  Sums are in a fixed order (deterministic); the order differs from the pair's, i.e. gates agree to
// rounding (test_se_gate).  Measured (visit r6w10, same box, interleaved): HyperSeg-M frame 0.7465 -> 0.7444 ms for its four launches,
// CamVid-S unchanged (0.7877 vs 0.7887): the launch is ~9.5 us against the pair's 10 -- ONE compute unit pulling a block's 96-147 KB of
// partials (written a moment ago through other XCDs' L2s) is the cost, not the arithmetic; the first form (a DPP wave sum per
// channel) was slower than the pair (0.7495 ms).  profiles/round6_se_gate_early_ab_w10.txt.
constexpr int SEE_MAX_IT = 36;                       // channels per wave: C <= 144
__global__ __launch_bounds__(256)
void se_gate_early_kernel(const float* __restrict__ partial, int nblk, float inv_hw, const float* __restrict__ w1,
                          const float* __restrict__ b1, const float* __restrict__ w2t, const float* __restrict__ b2, int C,
                          int Csq, float* __restrict__ z_out, float* __restrict__ gate, int fold) {
    // fold (1, 2, 4): a channel with more than 256 partials enters as `fold` pseudo-channels of nblk partials each (C and nblk are the
    // pseudo-channels' counts; the real channel c is pseudo-channels fold c .. fold c + fold - 1): stem + block 0 has 512 tiles
    __shared__ __attribute__((aligned(16))) float part[4 * SEE_MAX_IT * 64];      // [channel][lane]
    __shared__ __attribute__((aligned(16))) float mean_c[4 * SEE_MAX_IT];
    __shared__ float mean_p[4 * SEE_MAX_IT];
    __shared__ float zs[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int Cr = C / fold;                                                    // real channels
    const int nu = nblk >> 2;                                                   // 16-byte units per channel, <= 64
    const float4* __restrict__ pb = reinterpret_cast<const float4*>(partial + (size_t)b * C * nblk);
    const int nit = (C + 3) >> 2;
    const bool lane_ok = lane < nu;
    float4 pv[SEE_MAX_IT];
#pragma unroll
    for (int i = 0; i < SEE_MAX_IT; ++i) {
        const int c = min(4 * i + wave, C - 1);
        pv[i] = i < nit ? pb[(size_t)c * nu + (lane_ok ? lane : 0)] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // squeeze rows of this wave (j = wave, wave + 4) and this thread's gate column: requested with the partials
    const int c4 = min(4 * lane, Cr - 4);                                       // Cr % 4 == 0: lanes past the end re-read the last quad
    float4 w1v[2];
    float b1v[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = min(wave + 4 * r, Csq - 1);
        w1v[r] = *reinterpret_cast<const float4*>(w1 + (size_t)j * Cr + c4);
        b1v[r] = b1[j];
    }
    const int tc = min(tid, Cr - 1);
    float w2v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w2v[j] = w2t[(size_t)min(j, Csq - 1) * Cr + tc];
    const float b2v = b2[tc];
    __builtin_amdgcn_sched_barrier(0);
    // per-channel sums in two steps through LDS: lane sums -> part[channel][lane]; then a thread per (channel, quarter) adds 16 of them
    // and a quad of lanes combines (a DPP wave sum per channel -- 36 dependent readlane chains per wave -- made this launch slower than
    // the pair it replaces: visit r6w10)
#pragma unroll
    for (int i = 0; i < SEE_MAX_IT; ++i)
        if (i < nit) part[min(4 * i + wave, C - 1) * 64 + lane] = lane_ok ? (pv[i].x + pv[i].y) + (pv[i].z + pv[i].w) : 0.0f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int job = tid + 256 * r, c = min(job >> 2, C - 1), q = job & 3;
        const float4* src = reinterpret_cast<const float4*>(part + c * 64 + 16 * q);
        const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
        float t = (((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w))) + (((v2.x + v2.y) + (v2.z + v2.w)) + ((v3.x + v3.y) + (v3.z + v3.w)));
        int x = __float_as_int(t);
        x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false)));      // quad_perm [1,0,3,2]
        x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false)));      // quad_perm [2,3,0,1]
        if (q == 0 && job < 4 * C) (fold > 1 ? mean_p : mean_c)[c] = __int_as_float(x) * inv_hw;
    }
    if (fold > 1) {                                                             // uniform
        __syncthreads();
        if (tid < Cr) {
            float t = mean_p[tid * fold];
            for (int f = 1; f < fold; ++f) t += mean_p[tid * fold + f];
            mean_c[tid] = t;
        }
    }
    __syncthreads();
    {
        const float4 mv = 4 * lane < Cr ? *reinterpret_cast<const float4*>(mean_c + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = wave + 4 * r;
            const float t = wave_sum64(fmaf(w1v[r].w, mv.w, fmaf(w1v[r].z, mv.z, fmaf(w1v[r].y, mv.y, w1v[r].x * mv.x))));
            if (lane == 0 && j < Csq) zs[j] = swishf(t + b1v[r]);
        }
    }
    __syncthreads();
    if (z_out && tid < Csq) z_out[(size_t)b * Csq + tid] = zs[tid];
    float acc = b2v;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < Csq) acc = fmaf(w2v[j], zs[j], acc);
    if (tid < Cr) gate[(size_t)b * Cr + tid] = sigmoidf_fast(acc);
}

// The whole gate in ONE launch, for the blocks whose reduce weights are small enough that every excite workgroup can
// re-derive the squeezed vector for itself (C <= 768, Csq <= 32: 16 of EfficientNet-B1's 23 blocks).  Same grid as
// se_excite_kernel.  Rejected variant (ii) above walked the reduce rows one block-wide reduction at a time; here every
// global load of the workgroup -- partial sums, the whole reduce weight (<= 84 KB), this strip's expand column and project
// rows -- is issued before the first use (one memory round trip), a wave owns whole reduce rows (no block-wide reductions in
// the squeeze), and only two barriers separate the phases: a dependent launch in a replayed graph costs 1.6-2.2 us on this
// box (profiles/round2_graph_launch_floor.txt) while the squeeze launch it removes took 4.6 us.
// Round 5: a second instantiation <24, 1> for the EARLY blocks (C <= 256 but 64-256 pool partials per channel: 2304-6144 units), which
// used to take the two-launch route only because their partial sums did not fit the 8 loads per thread: 24 loads per thread, one
// 256-channel column block of the reduce weights.
// MEASURED NEGATIVE (visit r5v10, profiles/round5_se_tail_negative.txt): 0.7836 / 0.7855 ms per HyperSeg-M frame with it against 0.7760 /
// 0.7768 without, same box, interleaved -- every one of the 2-4 workgroups walks all 18-24 k partial sums and then adds 64-256 of them per
// channel serially, which costs more than the launch it saves.  Kept behind the build switch (tests/test_hip_encoder.py::test_se_gate
// covers the shapes either way); the product build leaves those blocks on the two-launch route.
#ifndef HS_SE_FUSED_WIDE
#define HS_SE_FUSED_WIDE 0
#endif
constexpr int SEF_MAX_CSQ = 32;
template <int PV, int W1U>
__global__ __launch_bounds__(256)
void se_gate_fused_kernel(const float* __restrict__ partial, int nblk, float inv_hw, const float* __restrict__ w1,
                          const float* __restrict__ b1, const float* __restrict__ w2t, const float* __restrict__ b2, int C,
                          int Csq, float* __restrict__ z_out, float* __restrict__ gate, const float* __restrict__ w_proj,
                          int Cout, const float* __restrict__ out_scale, float* __restrict__ w_scaled) {
    constexpr int SEF_MAX_C = 256 * W1U, SEF_MAX_UNITS = 256 * PV;
    __shared__ __attribute__((aligned(16))) float mean_c[SEF_MAX_C];
    __shared__ float se_part[SEF_MAX_UNITS + SEF_MAX_C];     // per-channel runs of partial sums, one pad word per channel
    __shared__ float zs[SEF_MAX_CSQ];
    __shared__ float red[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = blockIdx.x, b = blockIdx.y, rg = blockIdx.z;
    const int c = cb * 64 + lane, cc = min(c, C - 1);
    const int o0 = rg * 64 + wave * 16;
    // ---- every global load of this workgroup, issued up front (the partial sums first: the two-way branch on their
    // vector width makes the compiler drain the load queue where its arms join)
    const bool vec = (nblk & 3) == 0;
    const int qn = vec ? nblk >> 2 : nblk, units = C * qn;                          // <= SEF_MAX_UNITS
    const float* __restrict__ pb = partial + (size_t)b * C * nblk;
    float4 pv[PV];
    if (vec) {
#pragma unroll
        for (int k = 0; k < PV; ++k) pv[k] = reinterpret_cast<const float4*>(pb)[min(tid + 256 * k, units - 1)];
    } else {
#pragma unroll
        for (int k = 0; k < PV; ++k) pv[k] = make_float4(pb[min(tid + 256 * k, units - 1)], 0.0f, 0.0f, 0.0f);
    }
    float wp[16], osc[16];
    if (w_proj) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            wp[r] = w_proj[(size_t)min(o0 + r, Cout - 1) * C + cc];
            osc[r] = out_scale ? out_scale[min(o0 + r, Cout - 1)] : 1.0f;
        }
    }
    const int jq = (Csq + 3) >> 2, j0 = wave * jq, j1 = min(j0 + jq, Csq);          // jq <= 8
    float w2v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w2v[u] = w2t[(size_t)min(j0 + u, Csq - 1) * C + cc];
    const float b2v = b2[cc];
    const int c4 = C >> 2;                                                          // C % 4 == 0, c4 <= 64 W1U
    float4 w1v[8][W1U];
    float b1v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int j = min(wave + 4 * r, Csq - 1);
        b1v[r] = b1[j];
#pragma unroll
        for (int u = 0; u < W1U; ++u) {
            w1v[r][u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (4 * r < Csq && 64 * u < c4)                                         // uniform: rows / columns that exist
                w1v[r][u] = reinterpret_cast<const float4*>(w1 + (size_t)j * C)[min(lane + 64 * u, c4 - 1)];
        }
    }
    // ---- phase A: channel means
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        const int e = tid + 256 * k;
        if (e < units) {
            const int ch = e / qn;
            se_part[ch * (qn + 1) + (e - ch * qn)] = (pv[k].x + pv[k].y) + (pv[k].z + pv[k].w);
        }
    }
    __syncthreads();
    for (int ch = tid; ch < C; ch += 256) {
        const float* run = se_part + ch * (qn + 1);
        float t = 0.0f;
        for (int i = 0; i < qn; ++i) t += run[i];
        mean_c[ch] = t * inv_hw;
    }
    __syncthreads();
    // ---- phase B: squeezed activations, wave w owns rows w, w + 4, ...; sums over the wave on the DPP path
    float4 mv[W1U];
#pragma unroll
    for (int u = 0; u < W1U; ++u) {
        const int k = lane + 64 * u;
        mv[u] = *reinterpret_cast<const float4*>(mean_c + 4 * min(k, c4 - 1));
        if (k >= c4) mv[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (4 * r < Csq) {                                                          // uniform
            float a = 0.0f;
#pragma unroll
            for (int u = 0; u < W1U; ++u) {
                a = fmaf(w1v[r][u].x, mv[u].x, a); a = fmaf(w1v[r][u].y, mv[u].y, a);
                a = fmaf(w1v[r][u].z, mv[u].z, a); a = fmaf(w1v[r][u].w, mv[u].w, a);
            }
            a = wave_sum64(a);
            const int j = wave + 4 * r;
            if (lane == 0 && j < Csq) zs[j] = swishf(a + b1v[r]);
        }
    }
    __syncthreads();
    if (z_out && cb == 0 && rg == 0 && tid < Csq) z_out[(size_t)b * Csq + tid] = zs[tid];
    // ---- phase C: the strip's gate (each wave a quarter of the squeezed channels), then the scaled project rows
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (j0 + u < j1) acc = fmaf(w2v[u], zs[j0 + u], acc);
    red[wave][lane] = acc;
    __syncthreads();
    const float g = sigmoidf_fast(((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) + b2v);
    if (rg == 0 && wave == 0 && c < C) gate[(size_t)b * C + c] = g;
    if (w_proj && c < C) {
        float* __restrict__ dst = w_scaled + (size_t)b * Cout * C + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + r;
            if (o < Cout) dst[(size_t)o * C] = wp[r] * g * osc[r];
        }
    }
}

}  // namespace hs

using namespace hs;

static int depthwise_conv_launch(const float* x, int32_t batch, int32_t channels, int32_t H, int32_t W,
                                 const float* w, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                                 int32_t Ho, int32_t Wo, const float* scale, const float* shift, int32_t act,
                                 float* y, float* pool_partial, const float* in_scale, const float* in_shift,
                                 const hs_se_tail* se_in, void* stream) {
    if ((in_scale != nullptr) != (in_shift != nullptr)) return HS_ERR_BAD_ARG;
    if (!x || !w || !y || batch <= 0 || channels <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return HS_ERR_BAD_ARG;
    if (pad_t < 0 || pad_l < 0 || (scale && !shift)) return HS_ERR_BAD_ARG;
    const long nplanes = (long)batch * channels;
    if (nplanes > 65535L * 65535L) return HS_ERR_UNSUPPORTED;
    const int quads = Ho * ((Wo + 3) / 4);
    const int threads = quads >= 256 ? 256 : ((quads + 63) / 64) * 64;
    const int gy = nplanes > 65535 ? 32768 : (int)nplanes;
    dim3 grid((quads + threads - 1) / threads, gy, (unsigned)((nplanes + gy - 1) / gy));
    hipStream_t s = (hipStream_t)stream;
    SeTail se{};
    if (se_in) {
        const int st = make_se_tail(se_in, batch, channels, (int)grid.x, (long)channels * grid.x, Ho * Wo, se);
        if (st != HS_OK) return st;
    }
    const size_t se_lds = se_in ? (size_t)se_lds_floats(se.Csq) * sizeof(float) : 0;
#define HS_DW_(KK, SS, PP, SETV) hipLaunchKernelGGL((depthwise_conv_kernel<KK, SS, PP, false, SETV>), grid, dim3(threads), se_lds, s, x, w, scale, shift, \
                                             y, channels, H, W, Ho, Wo, pad_t, pad_l, act, pool_partial, in_scale, in_shift, (int)nplanes, se)
#define HS_DW(KK, SS, PP) do { if (se_in) HS_DW_(KK, SS, PP, true); else HS_DW_(KK, SS, PP, false); } while (0)
    // the LDS-tiled form: BN0 + swish prologue, whole output rows per workgroup (same thread -> output map, same partials)
    const int wq = Wo / 4;
    // For batched work (>= 8192 planes) and for every 5 x 5 launch (8.75 swishes per output there), on planes of >= 256 output quads:
    // the 3 x 3 launches of a batch-1 encoder are latency-bound and the extra barrier costs more than the taps' swishes (HyperSeg-M,
    // round 2: 1009 vs 1020 frames/s with it on everywhere; round 3 same-box A/B, profiles/round3_gemm_split_policy_ab.txt section 6:
    // 5 x 5 only 0.7758 ms per frame against 0.7801, everywhere 0.7794, 5 x 5 incl. the 128-thread maps 0.7822).
    if (in_scale && (nplanes >= 8192 || k == 5) && threads == 256 && (Wo & 3) == 0 && threads % wq == 0 && (k == 3 || k == 5) &&
        (stride == 1 || stride == 2)) {
        const int rows_out = threads / wq, rows_in = (rows_out - 1) * stride + k, tw = ((Wo - 1) * stride + k + 3) & ~3;
        const size_t tile_lds = (size_t)rows_in * tw * sizeof(float);
        const size_t lds = tile_lds > se_lds ? tile_lds : se_lds;
        if (tile_lds <= 64 * 1024) {
#define HS_DWT_(KK, SS, SETV) hipLaunchKernelGGL((depthwise_conv_kernel<KK, SS, -1, true, SETV>), grid, dim3(threads), lds, s, x, w, scale, shift, y, \
                                          channels, H, W, Ho, Wo, pad_t, pad_l, act, pool_partial, in_scale, in_shift, (int)nplanes, se)
#define HS_DWT(KK, SS) do { if (se_in) HS_DWT_(KK, SS, true); else HS_DWT_(KK, SS, false); } while (0)
            if (k == 3 && stride == 1) HS_DWT(3, 1); else if (k == 3) HS_DWT(3, 2); else if (stride == 1) HS_DWT(5, 1); else HS_DWT(5, 2);
#undef HS_DWT
#undef HS_DWT_
            return launch_status();
        }
    }
    const bool vec = (W & 3) == 0 && (((size_t)x) & 15) == 0;
    if (k == 3 && stride == 1) { if (vec && pad_l == 1) HS_DW(3, 1, 1); else HS_DW(3, 1, -1); }
    else if (k == 3 && stride == 2) { if (vec && pad_l == 0) HS_DW(3, 2, 0); else if (vec && pad_l == 1) HS_DW(3, 2, 1); else HS_DW(3, 2, -1); }
    else if (k == 5 && stride == 1) { if (vec && pad_l == 2) HS_DW(5, 1, 2); else HS_DW(5, 1, -1); }
    else if (k == 5 && stride == 2) { if (vec && pad_l == 1) HS_DW(5, 2, 1); else if (vec && pad_l == 2) HS_DW(5, 2, 2); else HS_DW(5, 2, -1); }
    else return HS_ERR_UNSUPPORTED;
#undef HS_DW
#undef HS_DW_
    return launch_status();
}

extern "C" int hs_depthwise_conv_fwd(const float* x, int32_t batch, int32_t channels, int32_t H, int32_t W,
                                     const float* w, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                                     int32_t Ho, int32_t Wo, const float* scale, const float* shift, int32_t act,
                                     float* y, float* pool_partial, const float* in_scale, const float* in_shift,
                                     void* stream) {
    return depthwise_conv_launch(x, batch, channels, H, W, w, k, stride, pad_t, pad_l, Ho, Wo, scale, shift, act, y, pool_partial,
                                 in_scale, in_shift, nullptr, stream);
}

// The same launch finishing the block's squeeze-excite gate in its last workgroups (hs_se_tail.h): no pool partials come back,
// se->gate (B, C) does.  HS_ERR_UNSUPPORTED: this shape keeps hs_se_gate_fwd.
extern "C" int hs_depthwise_conv_se_fwd(const float* x, int32_t batch, int32_t channels, int32_t H, int32_t W,
                                        const float* w, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                                        int32_t Ho, int32_t Wo, const float* scale, const float* shift, int32_t act,
                                        float* y, const float* in_scale, const float* in_shift, const hs_se_tail* se,
                                        void* stream) {
    if (!se) return HS_ERR_BAD_ARG;
    return depthwise_conv_launch(x, batch, channels, H, W, w, k, stride, pad_t, pad_l, Ho, Wo, scale, shift, act, y, nullptr,
                                 in_scale, in_shift, se, stream);
}

// bytes of the (zero-initialised, exclusively owned) workspace of a launch with an SE tail; 0: the shape is not covered
extern "C" int64_t hs_se_tail_workspace(int32_t batch, int32_t channels, int32_t c_squeezed, int32_t nblk, int64_t wgs_per_batch) {
    int T, L;
    if (batch <= 0 || !se_tail_plan(channels, c_squeezed, nblk, wgs_per_batch, T, L)) return 0;
    SeTail t{};
    t.B = batch; t.C = channels; t.Csq = c_squeezed; t.nblk = nblk; t.T = T; t.L = L;
    return (int64_t)(se_ws_words(t) * sizeof(se_u64));
}

extern "C" int hs_se_tail_tails(int32_t channels, int32_t c_squeezed, int32_t nblk, int64_t wgs_per_batch) {
    int T, L;
    return se_tail_plan(channels, c_squeezed, nblk, wgs_per_batch, T, L) ? T : 0;
}

extern "C" int hs_depthwise_pool_blocks(int32_t Ho, int32_t Wo) {
    const int quads = Ho * ((Wo + 3) / 4);
    const int threads = quads >= 256 ? 256 : ((quads + 63) / 64) * 64;
    return (quads + threads - 1) / threads;
}

extern "C" int hs_se_gate_fwd(const float* partial, int32_t batch, int32_t channels, int32_t nblk, float inv_hw,
                              const float* w_reduce, const float* b_reduce, int32_t c_squeezed, const float* w_expand,
                              const float* b_expand, float* squeezed, float* gate, const float* w_proj, int32_t c_out,
                              const float* out_scale, float* w_scaled, void* stream) {
    if (!partial || !w_reduce || !b_reduce || !w_expand || !b_expand || !squeezed || !gate || batch <= 0 || channels <= 0 ||
        nblk <= 0 || c_squeezed <= 0) return HS_ERR_BAD_ARG;
    if ((w_proj != nullptr) != (w_scaled != nullptr) || (w_proj && c_out <= 0) || (out_scale && !w_proj)) return HS_ERR_BAD_ARG;
    if (batch > 65535 || c_squeezed > 65535) return HS_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int row_groups = w_proj ? (c_out + 63) / 64 : 1;
    const long units = (long)channels * ((nblk & 3) == 0 ? nblk >> 2 : nblk);
    if ((channels & 3) == 0 && channels <= 768 && c_squeezed <= SEF_MAX_CSQ && units <= 2048) {
        hipLaunchKernelGGL((se_gate_fused_kernel<8, 3>), dim3((channels + 63) / 64, batch, row_groups), dim3(256), 0, s, partial, nblk,
                           inv_hw, w_reduce, b_reduce, w_expand, b_expand, channels, c_squeezed, squeezed, gate, w_proj, c_out,
                           out_scale, w_scaled);
        return launch_status();
    }
    {   // the early blocks (many partials, few channels) as one single-workgroup launch
        static const bool off = [] { const char* e = getenv("HS_SE_EARLY"); return e && atoi(e) == 0; }();      // dev A/B knob
        const int fold = nblk <= 256 ? 1 : (nblk <= 512 ? 2 : 4);                // pseudo-channels per channel (<= 256 partials each)
        if (!off && !w_proj && (channels & 3) == 0 && channels * fold <= 4 * SEE_MAX_IT && c_squeezed <= 8 && (nblk & (4 * fold - 1)) == 0 &&
            nblk <= 1024 && ((size_t)partial & 15) == 0 && ((size_t)w_reduce & 15) == 0) {
            hipLaunchKernelGGL(se_gate_early_kernel, dim3(1, batch), dim3(256), 0, s, partial, nblk / fold, inv_hw, w_reduce, b_reduce, w_expand,
                               b_expand, channels * fold, c_squeezed, squeezed, gate, fold);
            return launch_status();
        }
    }
    if (HS_SE_FUSED_WIDE && (channels & 3) == 0 && channels <= 256 && c_squeezed <= SEF_MAX_CSQ && units <= 6144) {
        hipLaunchKernelGGL((se_gate_fused_kernel<24, 1>), dim3((channels + 63) / 64, batch, row_groups), dim3(256), 0, s, partial, nblk,
                           inv_hw, w_reduce, b_reduce, w_expand, b_expand, channels, c_squeezed, squeezed, gate, w_proj, c_out,
                           out_scale, w_scaled);
        return launch_status();
    }
    hipLaunchKernelGGL(se_squeeze_kernel, dim3(c_squeezed, batch), dim3(256), 0, s, partial, nblk, inv_hw, w_reduce,
                       b_reduce, channels, c_squeezed, squeezed);
    int st = launch_status();
    if (st != HS_OK) return st;
    hipLaunchKernelGGL(se_excite_kernel, dim3((channels + 63) / 64, batch, row_groups), dim3(256), 0, s, squeezed, w_expand,
                       b_expand, channels, c_squeezed, gate, w_proj, c_out, out_scale, w_scaled);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolution as an fp32 MFMA GEMM with everything around it fused: y[b,o,p] = act(scale[o] * sum_c W[o,c] *
// (gate[b,c] * x[b,c,p]) + shift[o]) + residual[b,o,p].  One launch replaces {SE multiply, conv, BatchNorm, swish,
// skip add} of an MBConv block (efficientnet.py:101-103, 110-124).  NCHW: pixels are the contiguous GEMM dimension, so
// the B operand (x) is read in 64-byte runs and D is stored in 64-byte runs; no LDS -- x tiles are re-read by the
// ceil(Cout/32) workgroups of a pixel strip through L1/L2.  Wave tile 32 (o) x 64 (p), v_mfma_f32_16x16x4_f32.
// ---------------------------------------------------------------------------------------------------------------
namespace hs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// NB = number of 16-channel K batches (fully unrolled, double buffered: the loads of batch i+1 are in flight while the
// MFMAs of batch i run), NT = 16-pixel tiles per wave (the host shrinks it until the launch has >= 512 workgroups).
// VEC (round 6; NT = 4, P % 64 == 0, 16-byte aligned maps): the wave's 4 pixel tiles are INTERLEAVED -- tile n holds pixels
// p0 + 4 lrow + n -- so a lane's four B values of a k-row are ONE 16-byte load, its four outputs of a row ONE 16-byte store (and one
// 16-byte residual load): the matrix-core instruction does not care which pixel a column is, and these launches are bound by the
// number of vector-memory instructions (4-byte gathers: 64 of them per lane at NB = 2).  Same products, same sums: bit-identical.
template <int NB, int NT, bool VEC = false>
__global__ __launch_bounds__(256)
void pointwise_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ gate,
                           const float* __restrict__ scale, const float* __restrict__ shift,
                           const float* __restrict__ residual, float* __restrict__ y, int Cin, int Cout, int P, int act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane & 15, lk = lane >> 4;
    const int b = blockIdx.z;
    const int o0 = blockIdx.y * 32;
    const int p0 = (blockIdx.x * 4 + wave) * (16 * NT);
    if (p0 >= P) return;
    const float* __restrict__ xb = x + (size_t)b * Cin * P;
    const float* __restrict__ gb = gate ? gate + (size_t)b * Cin : nullptr;
    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // clamped per-lane rows / pixels (masked lanes read valid addresses; their products are zeroed)
    int orow[2]; bool ook[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { const int o = o0 + 16 * m + lrow; ook[m] = o < Cout; orow[m] = ook[m] ? o : Cout - 1; }
    int pcol[NT]; bool pok[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { const int p = VEC ? p0 + 4 * lrow + n : p0 + 16 * n + lrow; pok[n] = p < P; pcol[n] = pok[n] ? p : P - 1; }
    static_assert(!VEC || NT == 4, "interleaved tiles: four pixels per lane");
    constexpr int KU = 4;                                   // k-steps per batch (16 input channels)
    float av[2][KU][2], bv[2][KU][NT];
    auto load_batch = [&](int i, float (&a)[KU][2], float (&bb)[KU][NT]) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = i * 4 * KU + 4 * u + lk;
            const bool kok = k < Cin;
            const int kc = kok ? k : Cin - 1;
            // unconditional loads from clamped addresses, masked by a multiply: with selects the compiler put the loads
            // behind exec-mask branches and waited for each (tools/isa_phases.py: 3 of 72 loads in flight at the first wait)
            const float g = (gb ? gb[kc] : 1.0f) * (kok ? 1.0f : 0.0f);
#pragma unroll
            for (int m = 0; m < 2; ++m) a[u][m] = w[(size_t)orow[m] * Cin + kc] * (ook[m] ? 1.0f : 0.0f);
            if constexpr (VEC) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)kc * P + pcol[0]);
#pragma unroll
                for (int n = 0; n < NT; ++n) bb[u][n] = v[n] * g;
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n) bb[u][n] = xb[(size_t)kc * P + pcol[n]] * g;
            }
        }
    };
    load_batch(0, av[0], bv[0]);
    // epilogue operands, requested with the first batch: BN rows of this lane's 8 output rows and, for a skip block, the
    // residual values of its 8 x NT outputs (one load per output in the store loop meant one round trip per output)
    float scv[2][4], shv[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = min(o0 + 16 * m + 4 * lk + r, Cout - 1);
            scv[m][r] = scale ? scale[o] : 1.0f;
            shv[m][r] = shift ? shift[o] : 0.0f;
        }
    float rv[2][4][NT];
    if (residual) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = min(o0 + 16 * m + 4 * lk + r, Cout - 1);
                if constexpr (VEC) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(residual + ((size_t)b * Cout + o) * P + pcol[0]);
#pragma unroll
                    for (int n = 0; n < NT; ++n) rv[m][r][n] = v[n];
                } else {
#pragma unroll
                    for (int n = 0; n < NT; ++n) rv[m][r][n] = residual[((size_t)b * Cout + o) * P + pcol[n]];
                }
            }
    } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int n = 0; n < NT; ++n) rv[m][r][n] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (i + 1 < NB) load_batch(i + 1, av[(i + 1) & 1], bv[(i + 1) & 1]);
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][u][m], bv[i & 1][u][n], acc[m][n], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = o0 + 16 * m + 4 * lk + r;
            if (o >= Cout) continue;
            float ov[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float v = fmaf(acc[m][n][r], scv[m][r], shv[m][r]);
                if (act == 3) v = swishf(v);
                else v = apply_act(v, act);
                ov[n] = v + rv[m][r][n];
            }
            if constexpr (VEC) {
                *reinterpret_cast<f32x4*>(y + ((size_t)b * Cout + o) * P + pcol[0]) = f32x4{ov[0], ov[1], ov[2], ov[3]};
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (pok[n]) y[((size_t)b * Cout + o) * P + pcol[n]] = ov[n];
            }
        }
}

template <int NB>
static int launch_pointwise(int nt, dim3 grid, hipStream_t s, const float* x, const float* w, const float* gate,
                            const float* scale, const float* shift, const float* residual, float* y, int c_in, int c_out,
                            int pixels, int act) {
#define HS_PW(NT) hipLaunchKernelGGL((pointwise_conv_kernel<NB, NT>), grid, dim3(256), 0, s, x, w, gate, scale, shift, \
                                     residual, y, c_in, c_out, pixels, act)
    static const bool vec_off = [] { const char* e = getenv("HS_PW_VEC"); return e && atoi(e) == 0; }();      // dev A/B knob
    const bool vec = !vec_off && nt == 4 && (pixels & 63) == 0 && ((((size_t)x | (size_t)y | (size_t)residual)) & 15) == 0;
    if (vec) hipLaunchKernelGGL((pointwise_conv_kernel<NB, 4, true>), grid, dim3(256), 0, s, x, w, gate, scale, shift, residual, y, c_in,
                                c_out, pixels, act);
    else if (nt == 4) HS_PW(4); else if (nt == 2) HS_PW(2); else HS_PW(1);
#undef HS_PW
    return launch_status();
}

}  // namespace hs

extern "C" int hs_pointwise_conv_fwd(const float* x, int32_t batch, int32_t c_in, int32_t pixels, const float* w,
                                     int32_t c_out, const float* gate, const float* scale, const float* shift, int32_t act,
                                     const float* residual, float* y, void* stream) {
    if (!x || !w || !y || batch <= 0 || c_in <= 0 || c_out <= 0 || pixels <= 0 || (scale && !shift)) return HS_ERR_BAD_ARG;
    if (batch > 65535 || (c_out + 31) / 32 > 65535) return HS_ERR_UNSUPPORTED;
    if (c_in > 128) return HS_ERR_UNSUPPORTED;              // larger K: library GEMM (utils/inference.py routes those)
    const int mblocks = (c_out + 31) / 32;
    int nt = 4;
    while (nt > 1 && (long)((pixels + 64 * nt - 1) / (64 * nt)) * mblocks * batch < 512) nt >>= 1;
    dim3 grid((pixels + 64 * nt - 1) / (64 * nt), mblocks, batch);
    hipStream_t s = (hipStream_t)stream;
    switch ((c_in + 15) / 16) {
        case 1: return hs::launch_pointwise<1>(nt, grid, s, x, w, gate, scale, shift, residual, y, c_in, c_out, pixels, act);
        case 2: return hs::launch_pointwise<2>(nt, grid, s, x, w, gate, scale, shift, residual, y, c_in, c_out, pixels, act);
        case 3: return hs::launch_pointwise<3>(nt, grid, s, x, w, gate, scale, shift, residual, y, c_in, c_out, pixels, act);
        case 4: return hs::launch_pointwise<4>(nt, grid, s, x, w, gate, scale, shift, residual, y, c_in, c_out, pixels, act);
        case 5: return hs::launch_pointwise<5>(nt, grid, s, x, w, gate, scale, shift, residual, y, c_in, c_out, pixels, act);
        case 6: return hs::launch_pointwise<6>(nt, grid, s, x, w, gate, scale, shift, residual, y, c_in, c_out, pixels, act);
        case 7: return hs::launch_pointwise<7>(nt, grid, s, x, w, gate, scale, shift, residual, y, c_in, c_out, pixels, act);
        default: return hs::launch_pointwise<8>(nt, grid, s, x, w, gate, scale, shift, residual, y, c_in, c_out, pixels, act);
    }
}

// y = act(scale[c] * x + shift[c]) + residual, elementwise over (B, C, P); y may alias x.  Used after the stock
// (rocBLAS) 1x1 convolutions of the encoder where K is large: one launch instead of BatchNorm + swish (+ skip add).
namespace hs {
__global__ __launch_bounds__(256)
void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                       const float* __restrict__ residual, float* __restrict__ y, int C, int P, size_t n4, int act) {
    const int pq = P >> 2;                     // P % 4 == 0 (checked by the host)
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((e / pq) % C);
        const float sc = scale ? scale[c] : 1.0f, sh = shift[c];
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        float o[4] = {fmaf(v.x, sc, sh), fmaf(v.y, sc, sh), fmaf(v.z, sc, sh), fmaf(v.w, sc, sh)};
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = act == 3 ? swishf(o[t]) : apply_act(o[t], act);
        if (residual) {
            const float4 r = reinterpret_cast<const float4*>(residual)[e];
            o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
        reinterpret_cast<float4*>(y)[e] = make_float4(o[0], o[1], o[2], o[3]);
    }
}
}  // namespace hs

extern "C" int hs_affine_act_fwd(const float* x, int32_t batch, int32_t channels, int32_t pixels, const float* scale,
                                 const float* shift, int32_t act, const float* residual, float* y, void* stream) {
    if (!x || !y || !shift || batch <= 0 || channels <= 0 || pixels <= 0) return HS_ERR_BAD_ARG;
    if (pixels & 3) return HS_ERR_UNSUPPORTED;
    const size_t n4 = (size_t)batch * channels * pixels / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipLaunchKernelGGL(hs::affine_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scale, shift, residual, y,
                       channels, pixels, n4, act);
    return hs::launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Stem: dense 3x3 stride-2 convolution of the (3-channel) image + folded BatchNorm + swish in one launch, TF-"SAME" zero
// padding by (top, left) offsets.  Replaces F.pad (fill + copy) + MIOpen conv + BatchNorm2d + SiLU = 5 launches
// (efficientnet.py:321-322 of the reference).  One thread = 4 consecutive output pixels x 8 output channels: 27 input
// values per channel-plane window (81 loads, all in flight), the 8 x 27 weights are workgroup-uniform (scalar loads),
// 1 KB contiguous per store instruction.
// ---------------------------------------------------------------------------------------------------------------
namespace hs {
#ifndef HS_STEM_CG
#define HS_STEM_CG 4                  // output channels per thread: 4 / 8 / 16 measured 15.0 / 16.1 / 16.5 us, frame 0.7761-0.7734 / 0.7773-0.7755 / 0.7763-0.7769 ms (visit r5v15; tools/build_variants.py stem_cg*)
#endif
template <int CIN, int PL>
__global__ __launch_bounds__(256)
void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                      const float* __restrict__ shift, float* __restrict__ y, int Cout, int H, int W, int Ho, int Wo,
                      int pad_t, int pad_l) {
    constexpr int K = 3, S = 2, NCOL = 3 * S + K, CG = HS_STEM_CG;
    const int wq = (Wo + 3) >> 2;
    const int q0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q0 < Ho * wq;
    const int q = live ? q0 : 0;                                        // dead lanes shadow quad 0 up to the barrier
    const int g = blockIdx.y, b = blockIdx.z;
    const int yo = q / wq, xo = (q - yo * wq) * 4;
    const int xi0 = xo * S - pad_l, yi0 = yo * S - pad_t;
    const float* __restrict__ xb = x + (size_t)b * CIN * H * W;
    float v[CIN][K][NCOL];
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yi = yi0 + ky;
            const bool row_ok = yi >= 0 && yi < H;
            const float* __restrict__ row = xb + ((size_t)c * H + min(max(yi, 0), H - 1)) * W;
            if constexpr (PL >= 0) {                                   // W % 4 == 0: aligned 16-byte loads (see the depthwise kernel)
                constexpr int OFF = (4 - PL % 4) % 4;
                constexpr int NV = (OFF + NCOL + 3) / 4;
                const int a0 = xi0 - OFF;
                float win[NV * 4];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int col = a0 + 4 * i;
                    const bool ok = row_ok && col >= 0 && col < W;
                    const float4 t = *reinterpret_cast<const float4*>(row + ((col >= 0 && col < W) ? col : 0));
                    const float m = ok ? 1.0f : 0.0f;
                    win[4 * i] = t.x * m; win[4 * i + 1] = t.y * m; win[4 * i + 2] = t.z * m; win[4 * i + 3] = t.w * m;
                }
#pragma unroll
                for (int j = 0; j < NCOL; ++j) v[c][ky][j] = win[OFF + j];
            } else {
#pragma unroll
                for (int j = 0; j < NCOL; ++j) {
                    const int xi = xi0 + j;
                    const float t = row[min(max(xi, 0), W - 1)];
                    v[c][ky][j] = t * ((row_ok && xi >= 0 && xi < W) ? 1.0f : 0.0f);
                }
            }
        }
    // the workgroup's 8 x 27 weights, staged in LDS and read back as broadcast ds_read_b128 (all lanes, same address):
    // as SGPR operands they trickle in through the scalar cache one s_load at a time -- 28 us instead of ~10 for this kernel
    constexpr int NW = CIN * K * K;                                    // 27
    constexpr int NWP = (NW + 3) & ~3;                                 // row pitch 28: 16-byte rows
    __shared__ __attribute__((aligned(16))) float wl[CG * NWP];
    for (int e = threadIdx.x; e < CG * NWP; e += blockDim.x) {
        const int oc = e / NWP, r = e - oc * NWP;
        const int o = min(g * CG + oc, Cout - 1);
        wl[e] = r < NW ? w[(size_t)o * NW + r] : 0.0f;
    }
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int oc = 0; oc < CG; ++oc) {
        float wv[NWP];
#pragma unroll
        for (int r4 = 0; r4 < NWP / 4; ++r4) {
            const float4 t = *reinterpret_cast<const float4*>(wl + oc * NWP + 4 * r4);
            wv[4 * r4] = t.x; wv[4 * r4 + 1] = t.y; wv[4 * r4 + 2] = t.z; wv[4 * r4 + 3] = t.w;
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = fmaf(wv[(c * K + ky) * K + kx], v[c][ky][t * S + kx], acc[t]);
                }
        const int o = g * CG + oc;
        if (o < Cout) {
            const float sc = scale[o], sh = shift[o];
            float r[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) r[t] = swishf(fmaf(acc[t], sc, sh));
            float* dst = y + (((size_t)b * Cout + o) * Ho + yo) * Wo + xo;
            if ((Wo & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
            else for (int t = 0; t < 4 && xo + t < Wo; ++t) dst[t] = r[t];
        }
    }
}
}  // namespace hs

extern "C" int hs_stem_conv_fwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W, const float* w,
                                int32_t c_out, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, const float* scale,
                                const float* shift, float* y, void* stream) {
    if (!x || !w || !scale || !shift || !y || batch <= 0 || c_out <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 ||
        pad_t < 0 || pad_l < 0) return HS_ERR_BAD_ARG;
    if (c_in != 3 || batch > 65535) return HS_ERR_UNSUPPORTED;
    const int quads = Ho * ((Wo + 3) / 4);
    dim3 grid((quads + 255) / 256, (c_out + HS_STEM_CG - 1) / HS_STEM_CG, batch);
    const bool vec = (W & 3) == 0 && (((size_t)x) & 15) == 0;
#define HS_STEM(PP) hipLaunchKernelGGL((hs::stem_conv_kernel<3, PP>), grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, \
                                       y, c_out, H, W, Ho, Wo, pad_t, pad_l)
    if (vec && pad_l == 0) HS_STEM(0); else if (vec && pad_l == 1) HS_STEM(1); else HS_STEM(-1);
#undef HS_STEM
    return hs::launch_status();
}

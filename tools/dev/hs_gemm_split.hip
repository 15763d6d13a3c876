// DEV ONLY (not part of libhyperseg_hip.so; built by tools/gemm_split_kernel_probe.py): first cut of DESIGN section 7 item 2,
// the encoder's 1x1 convolutions as our own GEMM on the f16 matrix cores with split operands.
//
//   Y[b][m][n] (= | +=) sum_k W[m][k] * gate[b][k] * X[b][k][n]        m < M (16..320), k < K (96..1920), n < N pixels, NCHW
//
// * W is static: pre-split on the host into f16 pieces Whi / Wlo of  W[m][:] * 2^e(m)  (row scale to < 2^15), winv[m] = 2^-e(m);
//   [M][Kp] row-major, Kp = nwv * KS * 32 >= K, zero padded.  A fragment of v_mfma_f32_16x16x32_f16: lane (row = lane & 15,
//   kg = lane >> 4) holds k = 32 s + 8 kg + j, j < 8: ONE 16-byte load per (row tile, k-step, piece).
// * X is split on the fly.  A workgroup = one 16-pixel strip x up to 16*MT rows x all of K, K split across its nwv waves
//   (blockDim = 64 nwv).  A wave loads its whole K slice of the strip (8 KS dwords per lane, 64-byte runs across the 16 pixel
//   lanes), applies the SE gate, takes the per-PIXEL maximum over its slice (lane-local, then across the 4 lane groups), scales
//   that column to < 2^15 and splits it: a column scale only scales that column of D, so it is undone on the lane's own
//   accumulators.  Three products per k-step (hi*hi, lo*hi, hi*lo), f32 accumulation.
// * The waves' partial strips meet in LDS ([wave][row][16 pixels] f32), are summed in wave order (deterministic), scaled by
//   winv[row], added to Y when beta != 0 (the in-place skip accumulation of FusedMBConv) and stored as 64-byte runs.
#include <hip/hip_runtime.h>
#include <stdint.h>

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ int exp_of(float m) { return min(max(__float_as_int(m) >> 23, 27), 254); }
__device__ __forceinline__ float scale_of(int eb) { return __int_as_float((268 - eb) << 23); }       // m * scale < 2^15
__device__ __forceinline__ float inv_scale_of(int eb) { return __int_as_float((eb - 14) << 23); }

struct GemmArgs {
    const _Float16* __restrict__ whi; const _Float16* __restrict__ wlo; const float* __restrict__ winv;
    const float* __restrict__ gate;        // (B, K) or null
    const float* __restrict__ x;           // (B, K, N)
    float* __restrict__ y;                 // (B, M, N)
    int M, K, Kp, N, beta;
};

template <int MT, int KS>
__global__ __launch_bounds__(512)
void gemm_split_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];           // [nwv][16 MT][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwv = blockDim.x >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MT), b = blockIdx.z;
    const int n = min(n0 + lrow, a.N - 1);
    const int kbase = wave * (KS * 32);
    const float* __restrict__ xb = a.x + (size_t)b * a.K * a.N;
    const float* __restrict__ gb = a.gate ? a.gate + (size_t)b * a.K : nullptr;

    // ---- this wave's K slice of the strip: every load first (clamped addresses, masks by multiplication)
    float xv[KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kbase + 32 * s + 8 * kg + j;
            const int kc = min(k, a.K - 1);
            const float g = (gb ? gb[kc] : 1.0f) * (k < a.K ? 1.0f : 0.0f);
            xv[s][j] = xb[(size_t)kc * a.N + n] * g;
        }
    float mx = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(xv[s][j]));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                             // the pixel's maximum over the wave's slice
    const int eb = exp_of(mx);
    const float sc = scale_of(eb), invb = inv_scale_of(eb);
    half8 bh[KS], bl[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = xv[s][j] * sc;
            const _Float16 hi = (_Float16)v;
            bh[s][j] = hi;
            bl[s][j] = (_Float16)(v - (float)hi);
        }

    // ---- products
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A fragments one row tile ahead of the MFMAs that use them (the L2 latency of tile mt + 1 hides behind tile mt)
    half8 ah[2][KS], al[2][KS];
    auto load_a = [&](int mt, half8 (&h)[KS], half8 (&l)[KS]) {
        const size_t ro = (size_t)min(m0 + 16 * mt + lrow, a.M - 1) * a.Kp + kbase + 8 * kg;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            h[s] = *reinterpret_cast<const half8*>(a.whi + ro + 32 * s);
            l[s] = *reinterpret_cast<const half8*>(a.wlo + ro + 32 * s);
        }
    };
    load_a(0, ah[0], al[0]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt + 1 < MT) load_a(mt + 1, ah[(mt + 1) & 1], al[(mt + 1) & 1]);    // clamped rows: always a valid address
        if (m0 + 16 * mt < a.M) {                                       // uniform
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt & 1][s], bh[s], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt & 1][s], bl[s], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt & 1][s], bh[s], acc[mt], 0, 0, 0);
            }
        }
    }

    // ---- the waves' partial strips meet in LDS: D element r of this lane = row 4 kg + r, column lrow
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            red[(wave * (16 * MT) + 16 * mt + 4 * kg + r) * 16 + lrow] = acc[mt][r] * invb;
    __syncthreads();
    const int rows = min(16 * MT, a.M - m0);
    float* __restrict__ yb = a.y + (size_t)b * a.M * a.N;
    for (int e = tid; e < rows * 16; e += blockDim.x) {
        const int row = e >> 4, col = e & 15;
        float t = 0.0f;
        for (int w = 0; w < nwv; ++w) t += red[(w * (16 * MT) + row) * 16 + col];
        if (n0 + col < a.N) {
            const size_t idx = (size_t)(m0 + row) * a.N + n0 + col;
            const float v = t * a.winv[m0 + row];
            yb[idx] = a.beta ? yb[idx] + v : v;
        }
    }
}

template <int MT>
static int launch_mt(const GemmArgs& a, int ks, dim3 grid, int nwv, hipStream_t s) {
    const size_t lds = (size_t)nwv * 16 * MT * 16 * sizeof(float);
#define HS_G(KSV) hipLaunchKernelGGL((gemm_split_kernel<MT, KSV>), grid, dim3(64 * nwv), lds, s, a)
    switch (ks) {
        case 1: HS_G(1); break; case 2: HS_G(2); break; case 3: HS_G(3); break; case 4: HS_G(4); break;
        case 5: HS_G(5); break; case 6: HS_G(6); break; case 7: HS_G(7); break; case 8: HS_G(8); break;
        default: return -3;
    }
#undef HS_G
    return (int)hipGetLastError();
}

// Kp must equal nwv * ks * 32 with ks in 1..8, nwv in {1, 2, 4, 8}; mt (row tiles per workgroup) in {2, 4, 8}
extern "C" int hs_dev_gemm_split(const void* whi, const void* wlo, const float* winv, const float* gate, const float* x, float* y,
                                 int32_t batch, int32_t M, int32_t K, int32_t Kp, int32_t N, int32_t beta, int32_t nwv, int32_t mt,
                                 void* stream) {
    if (!whi || !wlo || !winv || !x || !y || batch <= 0 || M <= 0 || K <= 0 || N <= 0) return -1;
    if (nwv != 1 && nwv != 2 && nwv != 4 && nwv != 8) return -1;
    if (Kp % (nwv * 32) != 0 || Kp < K) return -1;
    const int ks = Kp / (nwv * 32);
    GemmArgs a{(const _Float16*)whi, (const _Float16*)wlo, winv, gate, x, y, M, K, Kp, N, beta};
    dim3 grid((N + 15) / 16, (M + 16 * mt - 1) / (16 * mt), batch);
    hipStream_t s = (hipStream_t)stream;
    if (mt == 2) return launch_mt<2>(a, ks, grid, nwv, s);
    if (mt == 4) return launch_mt<4>(a, ks, grid, nwv, s);
    if (mt == 8) return launch_mt<8>(a, ks, grid, nwv, s);
    return -1;
}


// ---------------------------------------------------------------------------------------------------------------------------
// v1 (after the first GPU run of v0: correct at 1.6-2.9e-7, but 22 us where the library takes 7-9: every workgroup streamed
// 128 rows of W through its CU in half-used cache lines, and the tail re-loaded winv once per serial loop iteration).
//   * W pre-arranged on the host in FRAGMENT order: block (row tile R, k-step S, piece p) = 64 lanes x 8 halfs contiguous
//     (lane = lrow + 16 kg holds W[16 R + lrow][32 S + 8 kg + j]): one fully used 1 KB run per load instruction.
//   * workgroup tile 32 rows x 32 pixels (two row tiles x two 16-pixel strips per wave): per workgroup W 2 x Kp x 32 x 2 B and
//     X K x 32 x 4 B, ~300 KB at the largest layer instead of 565 KB; 96 workgroups at M = 192, N = 512.
//   * K split over nwv = 2, 4 or 8 waves, KS <= 6 k-steps per wave, K <= 1536 (all of a wave's X and A loads in flight together;
//     the one K = 1920 layer of EfficientNet-B1 would need 16 waves at <= 128 registers: left to the library).
//   * tail operands (winv rows, the old Y for beta) requested BEFORE the barrier.
struct GemmArgsV1 {
    const _Float16* __restrict__ wsw;      // [RT][KST][2][64][8] halfs, RT = ceil(M / 16), KST = Kp / 32
    const float* __restrict__ winv;        // [16 RT]
    const float* __restrict__ gate; const float* __restrict__ x; float* __restrict__ y;
    int M, K, KST, N, beta;
};

template <int KS>
__global__ __launch_bounds__(512)
void gemm_split_v1_kernel(GemmArgsV1 a) {
    extern __shared__ __attribute__((aligned(16))) float red[];           // [nwv][32 rows][32 pixels]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwv = nthr >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 32, r0 = blockIdx.y * 2, b = blockIdx.z;
    const float* __restrict__ xb = a.x + (size_t)b * a.K * a.N;
    const float* __restrict__ gb = a.gate ? a.gate + (size_t)b * a.K : nullptr;
    float* __restrict__ yb = a.y + (size_t)b * a.M * a.N;
    const int rt_max = (a.M + 15) >> 4;

    // ---- every load of this wave: X (two strips), the gate, A fragments of both row tiles, tail operands
    int ncol[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) ncol[t] = min(n0 + 16 * t + lrow, a.N - 1);
    float xv[2][KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = (wave * KS + s) * 32 + 8 * kg + j;
            const int kc = min(k, a.K - 1);
            const float g = (gb ? gb[kc] : 1.0f) * (k < a.K ? 1.0f : 0.0f);
#pragma unroll
            for (int t = 0; t < 2; ++t) xv[t][s][j] = xb[(size_t)kc * a.N + ncol[t]] * g;
        }
    half8 ah[2][KS], al[2][KS];
    auto load_a = [&](int mt) {
        const int rt = min(r0 + mt, rt_max - 1);                        // clamped: a valid block; its results are not stored
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const _Float16* blk = a.wsw + ((size_t)(rt * a.KST + wave * KS + s) * 2) * 512 + lane * 8;
            ah[mt][s] = *reinterpret_cast<const half8*>(blk);
            al[mt][s] = *reinterpret_cast<const half8*>(blk + 512);
        }
    };
    load_a(0);                                                          // tile 1 follows once the f32 strips are split (registers)
    constexpr int TE = 8;                                               // tail elements per thread: 1024 / nthr <= 8 (nwv >= 2)
    float wi[TE], yo[TE];
#pragma unroll
    for (int i = 0; i < TE; ++i) {
        const int e = tid + i * nthr;
        const int row = min(16 * r0 + (e >> 5), a.M - 1), col = min(n0 + (e & 31), a.N - 1);
        wi[i] = a.winv[min(16 * r0 + (e >> 5), 16 * rt_max - 1)];
        yo[i] = a.beta ? yb[(size_t)row * a.N + col] : 0.0f;
    }

    // ---- per-pixel scale and split of the two strips
    float invb[2];
    half8 bh[2][KS], bl[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float mx = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(xv[t][s][j]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const int eb = exp_of(mx);
        const float sc = scale_of(eb);
        invb[t] = inv_scale_of(eb);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = xv[t][s][j] * sc;
                const _Float16 hi = (_Float16)v;
                bh[t][s][j] = hi;
                bl[t][s][j] = (_Float16)(v - (float)hi);
            }
    }
    load_a(1);
    // ---- products, partial tile to LDS
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt][s], bh[t][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][s], bl[t][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][s], bh[t][s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * 32 + 16 * mt + 4 * kg + r) * 32 + 16 * t + lrow] = acc[r] * invb[t];
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TE; ++i) {
        const int e = tid + i * nthr;
        if (e < 1024) {
            const int row = 16 * r0 + (e >> 5), col = n0 + (e & 31);
            float t = 0.0f;
            for (int w = 0; w < nwv; ++w) t += red[w * 1024 + e];
            if (row < a.M && col < a.N) yb[(size_t)row * a.N + col] = yo[i] + t * wi[i];
        }
    }
}

// wsw: fragment-ordered split weights (see GemmArgsV1); Kp = 32 * KST = nwv * ks * 32 with ks in 1..6, nwv in {2, 4, 8}
extern "C" int hs_dev_gemm_split_v1(const void* wsw, const float* winv, const float* gate, const float* x, float* y, int32_t batch,
                                    int32_t M, int32_t K, int32_t Kp, int32_t N, int32_t beta, int32_t nwv, void* stream) {
    if (!wsw || !winv || !x || !y || batch <= 0 || M <= 0 || K <= 0 || N <= 0) return -1;
    if (nwv != 2 && nwv != 4 && nwv != 8) return -1;
    if (Kp % (nwv * 32) != 0 || Kp < K) return -1;
    const int ks = Kp / (nwv * 32);
    GemmArgsV1 a{(const _Float16*)wsw, winv, gate, x, y, M, K, Kp / 32, N, beta};
    dim3 grid((N + 31) / 32, ((M + 15) / 16 + 1) / 2, batch);
    const size_t lds = (size_t)nwv * 1024 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define HS_G1(KSV) hipLaunchKernelGGL((gemm_split_v1_kernel<KSV>), grid, dim3(64 * nwv), lds, s, a)
    switch (ks) {
        case 1: HS_G1(1); break; case 2: HS_G1(2); break; case 3: HS_G1(3); break; case 4: HS_G1(4); break; case 5: HS_G1(5); break;
        case 6: HS_G1(6); break;
        default: return -3;
    }
#undef HS_G1
    return (int)hipGetLastError();
}

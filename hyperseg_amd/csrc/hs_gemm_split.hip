// Encoder-side helper (opt-in: prepare_for_inference(split_gemm=True), OFF by default -- the
// kernel below was measured as a dev probe at the end of round 2, profiles/round2_dev_gemm_split_probe.txt, its wiring into
// the prepared encoder is new): a 1x1 convolution as a GEMM on the f16 matrix cores with split operands,
//
//     Y[b][m][n] = act(sum_k W[m][k] * gate[b][k] * X[b][k][n] + shift[m]) + R[b][m][n]       f32 in / out / accumulation
//     (gate, shift, R optional; R may be Y itself: the in-place skip accumulation of FusedMBConv)
//
// replacing the library f32 GEMMs of the MBConv blocks' expand / project convolutions (efficientnet.py:101, 115) and the
// weight-scaling half of the SE gate (the gate multiplies X's rows on load here, so no per-frame copy of W is written).
// * W is static: split once on the host into f16 pieces hi / lo of W[m][:] * 2^e(m) (row scaled to < 2^15), w_inv[m] = 2^-e(m),
//   stored in MFMA-FRAGMENT order -- block (row tile R, k-step S, piece) = 64 lanes x 8 halfs contiguous, lane = lrow + 16 kg
//   holding W[16 R + lrow][32 S + 8 kg + j] -- so that every A load instruction is one fully used 1 KB run (row-major pieces
//   cost 22 us per launch where this layout costs 6: each workgroup streamed its rows through half-used cache lines).
// * X is split on the fly: workgroup = 32 rows x 32 pixels x all of K, K split across its 2 / 4 / 8 waves (<= 5 k-steps of 32
//   per wave, K <= 1280).  A wave requests its whole slice of both 16-pixel strips at once (64-byte runs across the pixel
//   lanes), applies the gate, scales each PIXEL's column by a power of two to < 2^15 (a column scale only scales that column
//   of D: undone on the lane's own accumulators) and splits it into hi / lo.  Three products per k-step (lo*hi, hi*lo, hi*hi):
//   error 1.6-2.9e-7 of the f64 product at the encoder's shapes, below the library f32 GEMM's 4.5e-7-1.2e-6.
// * the waves' partial tiles meet in LDS, are summed in wave order (deterministic), scaled by w_inv[row], get the BatchNorm
//   shift, the activation and the residual, and are stored as 128-byte runs; the tail's operands are requested before the
//   barrier.  (The probed dev kernel had only the accumulate-onto-Y tail; shift / activation / separate residual are new.)
#include "hs_common.h"

namespace hs {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using gs_f32x4 = __attribute__((ext_vector_type(4))) float;

// biased exponent eb of m (clamped so that 2^(141 - eb) and 2^(eb - 141) are normal floats); m * 2^(141 - eb) < 2^15
__device__ __forceinline__ int gs_exp_of(float m) { return min(max(__float_as_int(m) >> 23, 27), 254); }
__device__ __forceinline__ float gs_scale_of(int eb) { return __int_as_float((268 - eb) << 23); }
__device__ __forceinline__ float gs_inv_scale_of(int eb) { return __int_as_float((eb - 14) << 23); }

struct GemmSplitArgs {
    const _Float16* __restrict__ wf;       // [RT][KST][2][64][8] halfs, RT = ceil(M / 16), KST = Kp / 32
    const float* __restrict__ w_inv;       // [16 RT]
    const float* __restrict__ gate; const float* __restrict__ x;
    const float* shift;                    // [M] or null
    const float* residual;                 // (B, M, N) or null; may alias y
    float* y;
    int M, K, KST, N, act;
};

template <int KS>
__global__ __launch_bounds__(512)
void gemm_split_kernel(GemmSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gs_red[];        // [nwv][32 rows][32 pixels]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwv = nthr >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 32, r0 = blockIdx.y * 2, b = blockIdx.z;
    const float* __restrict__ xb = a.x + (size_t)b * a.K * a.N;
    const float* __restrict__ gb = a.gate ? a.gate + (size_t)b * a.K : nullptr;
    float* yb = a.y + (size_t)b * a.M * a.N;
    const float* rb = a.residual ? a.residual + (size_t)b * a.M * a.N : nullptr;
    const int rt_max = (a.M + 15) >> 4;

    // ---- every load of this wave: X (two strips) with the gate, A fragments of row tile 0, the tail's operands
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
            const float g = (gb ? gb[kc] : 1.0f) * (k < a.K ? 1.0f : 0.0f);      // clamped address, masked by a multiply
#pragma unroll
            for (int t = 0; t < 2; ++t) xv[t][s][j] = xb[(size_t)kc * a.N + ncol[t]] * g;
        }
    half8 ah[2][KS], al[2][KS];
    auto load_a = [&](int mt) {
        const int rt = min(r0 + mt, rt_max - 1);                        // clamped: a valid block; its results are not stored
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const _Float16* blk = a.wf + ((size_t)(rt * a.KST + wave * KS + s) * 2) * 512 + lane * 8;
            ah[mt][s] = *reinterpret_cast<const half8*>(blk);
            al[mt][s] = *reinterpret_cast<const half8*>(blk + 512);
        }
    };
    load_a(0);                                                          // tile 1 follows once the f32 strips are split (registers)
    constexpr int TE = KS == 1 ? 8 : 2;                                 // tail elements per thread, 1024 / nthr: more than one
                                                                        // k-step per wave only occurs with 8 waves (gemm_split_plan)
    float wi[TE], sh[TE], yo[TE];
#pragma unroll
    for (int i = 0; i < TE; ++i) {
        const int e = tid + i * nthr;
        const int row = min(16 * r0 + (e >> 5), a.M - 1), col = min(n0 + (e & 31), a.N - 1);
        wi[i] = a.w_inv[min(16 * r0 + (e >> 5), 16 * rt_max - 1)];
        sh[i] = a.shift ? a.shift[row] : 0.0f;
        yo[i] = rb ? rb[(size_t)row * a.N + col] : 0.0f;
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
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                         // the pixel's maximum over this wave's K slice
        const int eb = gs_exp_of(mx);
        const float sc = gs_scale_of(eb);
        invb[t] = gs_inv_scale_of(eb);
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
    __builtin_amdgcn_sched_barrier(0);                                  // tile 1's fragments go out before tile 0's first MFMA
    // ---- products; D element r of this lane = row 4 kg + r of the tile, column lrow of the strip
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            gs_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt][s], bh[t][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][s], bl[t][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][s], bh[t][s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gs_red[(wave * 32 + 16 * mt + 4 * kg + r) * 32 + 16 * t + lrow] = acc[r] * invb[t];
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TE; ++i) {
        const int e = tid + i * nthr;
        if (e < 1024) {
            const int row = 16 * r0 + (e >> 5), col = n0 + (e & 31);
            float t = 0.0f;
            for (int w = 0; w < nwv; ++w) t += gs_red[w * 1024 + e];
            if (row < a.M && col < a.N) {
                float v = fmaf(t, wi[i], sh[i]);
                v = a.act == 3 ? swishf(v) : apply_act(v, a.act);
                yb[(size_t)row * a.N + col] = v + yo[i];
            }
        }
    }
}

// waves per workgroup and k-steps per wave for an inner dimension K (the fewest k-steps per wave with 2, 4 or 8 waves)
static bool gemm_split_plan(int K, int& nwv, int& ks) {
    const int steps = (K + 31) / 32;
    nwv = 2;
    while (nwv < 8 && nwv < steps) nwv *= 2;
    ks = (steps + nwv - 1) / nwv;
    return ks <= 5;
}

}  // namespace hs

using namespace hs;

extern "C" int hs_gemm_split_kp(int32_t c_in) {
    int nwv, ks;
    if (c_in <= 0 || !gemm_split_plan(c_in, nwv, ks)) return HS_ERR_UNSUPPORTED;
    return nwv * ks * 32;
}

extern "C" int hs_gemm_split_fwd(const void* w_frag, const float* w_inv, const float* gate, const float* x,
                                 const float* shift, int32_t act, const float* residual, float* y,
                                 int32_t batch, int32_t c_out, int32_t c_in, int32_t kp, int32_t pixels, void* stream) {
    if (!w_frag || !w_inv || !x || !y || batch <= 0 || c_out <= 0 || c_in <= 0 || pixels <= 0) return HS_ERR_BAD_ARG;
    if (act < 0 || act > 3) return HS_ERR_BAD_ARG;
    int nwv, ks;
    if (!gemm_split_plan(c_in, nwv, ks)) return HS_ERR_UNSUPPORTED;
    if (kp != nwv * ks * 32 || (ks > 1 && nwv != 8)) return HS_ERR_BAD_ARG;
    if (batch > 65535 || (c_out + 31) / 32 > 65535) return HS_ERR_UNSUPPORTED;
    GemmSplitArgs a{(const _Float16*)w_frag, w_inv, gate, x, shift, residual, y, c_out, c_in, kp / 32, pixels, act};
    dim3 grid((pixels + 31) / 32, ((c_out + 15) / 16 + 1) / 2, batch);
    const size_t lds = (size_t)nwv * 1024 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define HS_GS(KSV) hipLaunchKernelGGL((gemm_split_kernel<KSV>), grid, dim3(64 * nwv), lds, s, a)
    switch (ks) {
        case 1: HS_GS(1); break;
        case 2: HS_GS(2); break;
        case 3: HS_GS(3); break;
        case 4: HS_GS(4); break;
        default: HS_GS(5); break;
    }
#undef HS_GS
    return launch_status();
}

// Encoder-side helper (prepare_for_inference(split_gemm=True): what bench.py runs since round 3,
// profiles/round3_first_visit.txt): a 1x1 convolution as a GEMM on the f16 matrix cores with split operands,
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

// FAST: K % 8 == 0 and N % 2 == 0 (every encoder / context-head layer): paired pixel loads and vector gate loads, decided at
// LAUNCH time -- a run-time branch around a load makes the compiler wait for it where the branches join.
// NS: 16-pixel strips per workgroup.  2 = the 32-pixel block above; 1 = HALF of it, for launches whose 32-pixel grid would leave
// more than half of the 256 CUs idle (the 16x32 and 32x64 maps): these launches are bound by the instructions ONE CU has to
// issue for its workgroup (the on-the-fly split is ~2/3 of them), so half the pixels on twice the CUs is the shorter launch.
#ifndef HS_GS_NARROW_MAX_WG
#define HS_GS_NARROW_MAX_WG 128
#endif

// MT: 16-row tiles per workgroup.  2 = the 32-row block above; 4 where that grid would need more than one round of workgroups
// (M = 640 ... 1920 on the 16x32 map): the split of X is per workgroup, so twice the rows on half the workgroups halves it.
// A fragments roll through two register slots (tile mt + 2 is requested when tile mt's products have been issued).
#ifndef HS_GS_TALL_MIN_WG
#define HS_GS_TALL_MIN_WG 257
#endif

template <int KS, int NWV, bool FAST, int NS, int MT>
__global__ __launch_bounds__(64 * NWV)
void gemm_split_kernel(GemmSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gs_red[];        // [nwv][16 MT rows][16 NS pixels]
    constexpr int nthr = 64 * NWV, nwv = NWV;                              // compile-time: the tail's element count and the reduction unroll
    constexpr int NP = 16 * NS, NR = 16 * MT, NE = NR * NP;                // pixels, rows and outputs per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * NP, r0 = blockIdx.y * MT, b = blockIdx.z;
    const float* __restrict__ xb = a.x + (size_t)b * a.K * a.N;
    const float* __restrict__ gb = a.gate ? a.gate + (size_t)b * a.K : nullptr;
    float* yb = a.y + (size_t)b * a.M * a.N;
    const float* rb = a.residual ? a.residual + (size_t)b * a.M * a.N : nullptr;
    const int rt_max = (a.M + 15) >> 4;

    // ---- every load of this wave: X (two strips) with the gate, A fragments of row tile 0, the tail's operands.
    // Strip t of the 32-pixel block = the pixels of parity t: lane lrow owns pixels n0 + 2 lrow and n0 + 2 lrow + 1, so a k-row's two
    // values are ONE 8-byte load (round 2: two 4-byte gathers), and the lane group's 8 gate values are two 16-byte loads (round 2:
    // eight): 50 -> 22 vector-memory instructions per wave at KS = 1 (tools/isa_phases.py) -- these launches are bound by the
    // per-instruction cost of the CU's address path, not by bytes.  N is even and n0 a multiple of 32, so the pair is aligned.
    using gs_f32x2 = __attribute__((ext_vector_type(2))) float;
    const int ncol0 = min(n0 + NS * lrow, a.N - 1), ncol1 = min(n0 + NS * lrow + 1, a.N - 1);
    float xv[NS][KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = (wave * KS + s) * 32 + 8 * kg + j;
            const int kc = min(k, a.K - 1);
            if constexpr (NS == 1) {
                xv[0][s][j] = xb[(size_t)kc * a.N + ncol0];
            } else if constexpr (FAST) {
                const gs_f32x2 v = *reinterpret_cast<const gs_f32x2*>(xb + (size_t)kc * a.N + ncol0);
                xv[0][s][j] = v[0]; xv[1][s][j] = v[1];
            } else {
                xv[0][s][j] = xb[(size_t)kc * a.N + ncol0]; xv[1][s][j] = xb[(size_t)kc * a.N + ncol1];
            }
        }
    float gv[KS][8];                                                    // raw here; masked after the last load has been issued
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k0 = (wave * KS + s) * 32 + 8 * kg;
        if constexpr (FAST) {
            // a run of 8 gate values either exists entirely or not at all (K % 8 == 0): clamp the address, mask by a multiply
            const float* gp = gb ? gb + min(k0, a.K - 8) : a.w_inv;       // w_inv: any valid 32 bytes when there is no gate
            const gs_f32x4 g0 = *reinterpret_cast<const gs_f32x4*>(gp), g1 = *reinterpret_cast<const gs_f32x4*>(gp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { gv[s][j] = g0[j]; gv[s][4 + j] = g1[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[s][j] = gb ? gb[min(k0 + j, a.K - 1)] : 1.0f;
        }
    }
    half8 ah[2][KS], al[2][KS];
    auto load_a = [&](int mt) {                                         // into slot mt & 1
        const int rt = min(r0 + mt, rt_max - 1);                        // clamped: a valid block; its results are not stored
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const _Float16* blk = a.wf + ((size_t)(rt * a.KST + wave * KS + s) * 2) * 512 + lane * 8;
            ah[mt & 1][s] = *reinterpret_cast<const half8*>(blk);
            al[mt & 1][s] = *reinterpret_cast<const half8*>(blk + 512);
        }
    };
    load_a(0);                                                          // tile 1 follows once the f32 strips are split (registers)
    constexpr int TE = (NE + nthr - 1) / nthr;                          // tail elements per thread
    float wi[TE], sh[TE], yo[TE];
    auto load_tail = [&]() {
#pragma unroll
        for (int i = 0; i < TE; ++i) {
            const int e = min(tid + i * nthr, NE - 1);
            const int row = min(16 * r0 + e / NP, a.M - 1), col = min(n0 + (e & (NP - 1)), a.N - 1);
            wi[i] = a.w_inv[min(16 * r0 + e / NP, 16 * rt_max - 1)];
            sh[i] = a.shift ? a.shift[row] : 0.0f;
            yo[i] = rb ? rb[(size_t)row * a.N + col] : 0.0f;
        }
    };
    constexpr bool LATE_TAIL = KS >= 3;                                 // deep K: the f32 strips fill the registers until they are split
    if constexpr (!LATE_TAIL) load_tail();

    __builtin_amdgcn_sched_barrier(0);                                  // every request above is out before the first wait
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k0 = (wave * KS + s) * 32 + 8 * kg;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float mk = (FAST ? k0 : k0 + j) < a.K ? 1.0f : 0.0f;
            gv[s][j] = gb ? gv[s][j] * mk : mk;
        }
    }
    // ---- per-pixel scale and split of the two strips (2 vector instructions per element: v_fma_mix* converts on the way out)
    float invb[NS];
    half8 bh[NS][KS], bl[NS][KS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        unsigned mx = 0;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xv[t][s][j] *= gv[s][j];
                mx = max(mx, __float_as_uint(xv[t][s][j]) & 0x7fffffffu);
            }
        {   // the pixel's maximum over this wave's K slice: lanes lrow, lrow + 16, + 32, + 48 (no LDS round trip)
            auto r16 = __builtin_amdgcn_permlane16_swap(mx, mx, false, false);
            mx = max(r16[0], r16[1]);
            auto r32 = __builtin_amdgcn_permlane32_swap(mx, mx, false, false);
            mx = max(r32[0], r32[1]);
        }
        const int eb = gs_exp_of(__uint_as_float(mx));
        const float sc = gs_scale_of(eb);
        invb[t] = gs_inv_scale_of(eb);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if constexpr (FAST) {
                unsigned hh[4], ll[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
                        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
                        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
                        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                        : "=&v"(hh[j]), "=&v"(ll[j]) : "v"(xv[t][s][2 * j]), "v"(xv[t][s][2 * j + 1]), "v"(sc));
                }
                using gs_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
                bh[t][s] = __builtin_bit_cast(half8, gs_u32x4{hh[0], hh[1], hh[2], hh[3]});
                bl[t][s] = __builtin_bit_cast(half8, gs_u32x4{ll[0], ll[1], ll[2], ll[3]});
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = xv[t][s][j] * sc;
                    const _Float16 hi = (_Float16)v;
                    bh[t][s][j] = hi;
                    bl[t][s][j] = (_Float16)(v - (float)hi);
                }
            }
        }
    }
    load_a(1);
    if constexpr (LATE_TAIL) load_tail();
    __builtin_amdgcn_sched_barrier(0);                                  // tile 1's fragments go out before tile 0's first MFMA
    // ---- products; D element r of this lane = row 4 kg + r of the tile, column lrow of the strip
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        gs_f32x4 acc[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            acc[t] = gs_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt & 1][s], bh[t][s], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt & 1][s], bl[t][s], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt & 1][s], bh[t][s], acc[t], 0, 0, 0);
            }
        }
        if (mt + 2 < MT) { load_a(mt + 2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int t = 0; t < NS; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gs_red[(wave * NR + 16 * mt + 4 * kg + r) * NP + NS * lrow + t] = acc[t][r] * invb[t];  // pixel n0 + NS lrow + t
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TE; ++i) {
        const int e = tid + i * nthr;
        if (e < NE) {
            const int row = 16 * r0 + e / NP, col = n0 + (e & (NP - 1));
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < nwv; ++w) t += gs_red[w * NE + e];
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
    const size_t wgs = (size_t)grid.x * grid.y * grid.z;
    const bool narrow = wgs <= HS_GS_NARROW_MAX_WG && pixels > 16;
    const bool tall = wgs >= HS_GS_TALL_MIN_WG && c_out > 32;
    if (narrow) grid.x = (pixels + 15) / 16;
    if (tall) grid.y = ((c_out + 15) / 16 + 3) / 4;
    const size_t lds = (size_t)nwv * (narrow ? 512 : tall ? 2048 : 1024) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    // the vector-load / 2-instruction-split form for every K depth: whole frame 0.814 ms against 0.832 (k-steps <= 2 only) and 0.858
    // (never), same box, interleaved (profiles/round3_gemm_split_latency.txt)
    const bool fast = (c_in & 7) == 0 && (pixels & 1) == 0 && c_in >= 8;
#define HS_GS_(KSV, NWV, F) do { if (narrow) hipLaunchKernelGGL((gemm_split_kernel<KSV, NWV, F, 1, 2>), grid, dim3(64 * NWV), lds, s, a); \
                                 else if (tall) hipLaunchKernelGGL((gemm_split_kernel<KSV, NWV, F, 2, 4>), grid, dim3(64 * NWV), lds, s, a); \
                                 else hipLaunchKernelGGL((gemm_split_kernel<KSV, NWV, F, 2, 2>), grid, dim3(64 * NWV), lds, s, a); } while (0)
#define HS_GS(KSV, NWV) do { if (fast) HS_GS_(KSV, NWV, true); else HS_GS_(KSV, NWV, false); } while (0)
    switch (ks) {
        case 1: if (nwv == 2) HS_GS(1, 2); else if (nwv == 4) HS_GS(1, 4); else HS_GS(1, 8); break;
        case 2: HS_GS(2, 8); break;
        case 3: HS_GS(3, 8); break;
        case 4: HS_GS(4, 8); break;
        default: HS_GS(5, 8); break;
    }
#undef HS_GS
#undef HS_GS_
    return launch_status();
}

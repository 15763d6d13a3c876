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
#include <type_traits>

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
    int conv_wo;                           // PATCH form: output width (the input map is 2 Ho x 2 Wo), else 0
    float* pool_partial;                   // [B][M][gridDim.x] sums of the stored values over the workgroup's 16 pixels, or null
    int up2_wo;                            // > 0: the N = Ho x up2_wo outputs are stored nearest-2x upsampled, y (B, M, 2 Ho, 2 up2_wo)
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
// A fragments roll through two register slots (the tile after next is requested when a tile's products have been issued).
#ifndef HS_GS_TALL_MIN_WG
#define HS_GS_TALL_MIN_WG 257
#endif

// NCH: chunks of KS k-steps per wave (K <= 1280 NCH).  Every X request of every chunk is issued up front; a chunk is scaled by
// its OWN per-pixel power of two, split, multiplied, and its accumulators join the running sum with that scale undone.
// PATCH: X is read through a 2x2 / stride-2 window -- the GEMM of a Conv2d(C, M, kernel 2, stride 2) on a (C, 2 Ho, 2 Wo) map,
// k = 4 c + 2 dy + dx (the conv weight's own flatten order), pixel n = oy Wo + ox: the two dx of a (c, dy) are ONE 8-byte load
// (16 bytes for a lane's two pixels), so the im2col copy of the library path is never written.  No gate in this form.
template <int KS, int NWV, bool FAST, int NS, int MT, int NCH, bool PATCH>
__global__ __launch_bounds__(64 * NWV)
void gemm_split_kernel(GemmSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gs_red[];        // [nwv][16 MT rows][16 NS pixels]
    constexpr int nthr = 64 * NWV, nwv = NWV;                              // compile-time: the tail's element count and the reduction unroll
    constexpr int NP = 16 * NS, NR = 16 * MT, NE = NR * NP;                // pixels, rows and outputs per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * NP, r0 = blockIdx.y * MT, b = blockIdx.z;
    const float* __restrict__ xb = a.x + (size_t)b * a.K * a.N;           // PATCH: K N = 4 C Ho Wo = the input map's size
    const float* __restrict__ gb = a.gate ? a.gate + (size_t)b * a.K : nullptr;
    float* yb = a.y + (size_t)b * a.M * a.N * (a.up2_wo > 0 ? 4 : 1);
    const float* rb = a.residual ? a.residual + (size_t)b * a.M * a.N : nullptr;
    const int rt_max = (a.M + 15) >> 4;

    // ---- every load of this wave: X (two strips) with the gate, A fragments of row tile 0, the tail's operands.
    // Strip t of the 32-pixel block = the pixels of parity t: lane lrow owns pixels n0 + 2 lrow and n0 + 2 lrow + 1, so a k-row's two
    // values are ONE 8-byte load (round 2: two 4-byte gathers), and the lane group's 8 gate values are two 16-byte loads (round 2:
    // eight): 50 -> 22 vector-memory instructions per wave at KS = 1 (tools/isa_phases.py) -- these launches are bound by the
    // per-instruction cost of the CU's address path, not by bytes.  N is even and n0 a multiple of 32, so the pair is aligned.
    using gs_f32x2 = __attribute__((ext_vector_type(2))) float;
    const int ncol0 = min(n0 + NS * lrow, a.N - 1), ncol1 = min(n0 + NS * lrow + 1, a.N - 1);
    float xv[NCH][NS][KS][8];
    if constexpr (PATCH) {
        const int oy = ncol0 / a.conv_wo, ox = ncol0 - oy * a.conv_wo;     // NS = 2: Wo even, so ncol1 is the next pixel of the row
        const float* __restrict__ win = xb + (size_t)(2 * oy) * (2 * a.conv_wo) + 2 * ox;
        const int cmax = (a.K >> 2) - 1;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) {                              // (channel, dy) = (ch0 + (q >> 1), q & 1)
                    const int ch = min((((wave * NCH + c) * KS + s) * 32 + 8 * kg) / 4 + (q >> 1), cmax);
                    const float* src = win + (size_t)ch * (4 * a.N) + (q & 1) * (2 * a.conv_wo);
                    if constexpr (NS == 1) {
                        const gs_f32x2 v = *reinterpret_cast<const gs_f32x2*>(src);
                        xv[c][0][s][2 * q] = v[0]; xv[c][0][s][2 * q + 1] = v[1];
                    } else {
                        const gs_f32x4 v = *reinterpret_cast<const gs_f32x4*>(src);
                        xv[c][0][s][2 * q] = v[0]; xv[c][0][s][2 * q + 1] = v[1];
                        xv[c][1][s][2 * q] = v[2]; xv[c][1][s][2 * q + 1] = v[3];
                    }
                }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = ((wave * NCH + c) * KS + s) * 32 + 8 * kg + j;
                    const int kc = min(k, a.K - 1);
                    if constexpr (NS == 1) {
                        xv[c][0][s][j] = xb[(size_t)kc * a.N + ncol0];
                    } else if constexpr (FAST) {
                        const gs_f32x2 v = *reinterpret_cast<const gs_f32x2*>(xb + (size_t)kc * a.N + ncol0);
                        xv[c][0][s][j] = v[0]; xv[c][1][s][j] = v[1];
                    } else {
                        xv[c][0][s][j] = xb[(size_t)kc * a.N + ncol0]; xv[c][1][s][j] = xb[(size_t)kc * a.N + ncol1];
                    }
                }
    }
    constexpr int NG = PATCH ? 1 : NCH;
    float gv[NG][KS][8];                                                // raw here; masked after the last load has been issued
    if constexpr (!PATCH) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k0 = ((wave * NCH + c) * KS + s) * 32 + 8 * kg;
                if constexpr (FAST) {
                    // a run of 8 gate values either exists entirely or not at all (K % 8 == 0): clamp the address, mask by a multiply
                    const float* gp = gb ? gb + min(k0, a.K - 8) : a.w_inv;   // w_inv: any valid 32 bytes when there is no gate
                    const gs_f32x4 g0 = *reinterpret_cast<const gs_f32x4*>(gp), g1 = *reinterpret_cast<const gs_f32x4*>(gp + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { gv[c][s][j] = g0[j]; gv[c][s][4 + j] = g1[j]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) gv[c][s][j] = gb ? gb[min(k0 + j, a.K - 1)] : 1.0f;
                }
            }
    }
    half8 ah[2][KS], al[2][KS];
    auto load_a = [&](int q) {                                          // q = chunk * MT + row tile, into slot q & 1
        const int rt = min(r0 + q % MT, rt_max - 1);                    // clamped: a valid block; its results are not stored
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const _Float16* blk = a.wf + ((size_t)(rt * a.KST + (wave * NCH + q / MT) * KS + s) * 2) * 512 + lane * 8;
            ah[q & 1][s] = *reinterpret_cast<const half8*>(blk);
            al[q & 1][s] = *reinterpret_cast<const half8*>(blk + 512);
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
    constexpr bool LATE_TAIL = KS * NCH >= 3;                           // deep K: the f32 strips fill the registers until they are split
    if constexpr (!LATE_TAIL) load_tail();
    __builtin_amdgcn_sched_barrier(0);                                  // every request above is out before the first wait

    gs_f32x4 tot[MT][NS];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // ---- per-pixel scale and split of this chunk's strips (2 vector instructions per element: v_fma_mix* converts on the way out)
        float invb[NS];
        half8 bh[NS][KS], bl[NS][KS];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            unsigned mx = 0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k0 = ((wave * NCH + c) * KS + s) * 32 + 8 * kg;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float mk = (FAST || PATCH ? k0 : k0 + j) < a.K ? 1.0f : 0.0f;
                    float g = mk;
                    if constexpr (!PATCH) g = gb ? gv[c][s][j] * mk : mk;
                    xv[c][t][s][j] *= g;
                    mx = max(mx, __float_as_uint(xv[c][t][s][j]) & 0x7fffffffu);
                }
            }
            {   // the pixel's maximum over this chunk of the wave's K slice: lanes lrow, lrow + 16, + 32, + 48 (no LDS round trip)
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
                if constexpr (FAST || PATCH) {
                    unsigned hh[4], ll[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
                            "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
                            "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
                            "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                            : "=&v"(hh[j]), "=&v"(ll[j]) : "v"(xv[c][t][s][2 * j]), "v"(xv[c][t][s][2 * j + 1]), "v"(sc));
                    }
                    using gs_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
                    bh[t][s] = __builtin_bit_cast(half8, gs_u32x4{hh[0], hh[1], hh[2], hh[3]});
                    bl[t][s] = __builtin_bit_cast(half8, gs_u32x4{ll[0], ll[1], ll[2], ll[3]});
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = xv[c][t][s][j] * sc;
                        const _Float16 hi = (_Float16)v;
                        bh[t][s][j] = hi;
                        bl[t][s][j] = (_Float16)(v - (float)hi);
                    }
                }
            }
        }
        if (c == 0) {
            load_a(1);
            if constexpr (LATE_TAIL) load_tail();
        }
        __builtin_amdgcn_sched_barrier(0);                              // the next tile's fragments go out before this tile's first MFMA
        // ---- products; D element r of this lane = row 4 kg + r of the tile, column lrow of the strip
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int q = c * MT + mt;
            gs_f32x4 acc[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                acc[t] = gs_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q & 1][s], bh[t][s], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q & 1][s], bl[t][s], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q & 1][s], bh[t][s], acc[t], 0, 0, 0);
                }
            }
            if (q + 2 < NCH * MT) { load_a(q + 2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int t = 0; t < NS; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = c == 0 ? acc[t][r] * invb[t] : fmaf(acc[t][r], invb[t], tot[mt][t][r]);
                    if (c == NCH - 1) gs_red[(wave * NR + 16 * mt + 4 * kg + r) * NP + NS * lrow + t] = v;   // pixel n0 + NS lrow + t
                    else tot[mt][t][r] = v;
                }
            }
        }
    }
    __syncthreads();
    // the tail once per ACTIVATION (a uniform run-time value) instead of `apply_act(v, a.act)` -- three scalar compare-and-branch pairs per
    // output -- inside the unrolled element loop.  Same-box A/B of the whole frame: 0.7760 / 0.7764 vs 0.7765 / 0.7777 ms, i.e. nothing
    // measurable (profiles/round4_gemm_split_tail_unswitch_ab.txt): uniform branches are not what these 5-7 us launches wait for.
    auto tail = [&](auto act_kind) {
        constexpr int ACT = decltype(act_kind)::value;
#pragma unroll
        for (int i = 0; i < TE; ++i) {
            const int e = tid + i * nthr;
            if (e < NE) {
                const int row = 16 * r0 + e / NP, col = n0 + (e & (NP - 1));
                float t = 0.0f;
#pragma unroll
                for (int w = 0; w < nwv; ++w) t += gs_red[w * NE + e];
                const bool live = row < a.M && col < a.N;
                float v = fmaf(t, wi[i], sh[i]);
                if constexpr (ACT == HS_ACT_SWISH) v = swishf(v);
                else if constexpr (ACT == HS_ACT_RELU) v = fmaxf(v, 0.0f);
                else if constexpr (ACT == HS_ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
                v += yo[i];
                if (live) {
                    if (a.up2_wo > 0) {                                 // pixel (oy, ox) -> the 2x2 block at (2 oy, 2 ox) of a 2 Wo wide map
                        const int oy = col / a.up2_wo, ox = col - oy * a.up2_wo;
                        float* dst = yb + (size_t)row * (4 * a.N) + (size_t)(2 * oy) * (2 * a.up2_wo) + 2 * ox;
                        *reinterpret_cast<gs_f32x2*>(dst) = gs_f32x2{v, v};
                        *reinterpret_cast<gs_f32x2*>(dst + 2 * a.up2_wo) = gs_f32x2{v, v};
                    } else {
                        yb[(size_t)row * a.N + col] = v;
                    }
                }
                if constexpr (NP == 16) {                               // 16 consecutive lanes = one output row of this workgroup
                    if (a.pool_partial) {
                        const float rs = rowsum16(live ? v : 0.0f);
                        if ((e & 15) == 0 && row < a.M) a.pool_partial[((size_t)b * a.M + row) * gridDim.x + blockIdx.x] = rs;
                    }
                }
            }
        }
    };
    if (a.act == HS_ACT_SWISH) tail(std::integral_constant<int, HS_ACT_SWISH>{});
    else if (a.act == HS_ACT_NONE) tail(std::integral_constant<int, HS_ACT_NONE>{});
    else if (a.act == HS_ACT_RELU) tail(std::integral_constant<int, HS_ACT_RELU>{});
    else tail(std::integral_constant<int, HS_ACT_RELU6>{});
}

// shift_out[m] = shift[m] + sum_c wb[m][c] * mean[c], mean[c] = inv_p * sum_j partial[c][j]: the context head's deepest merge, where
// the pooled half of cat(feat, pooled.expand_as(feat)) is constant over the pixels and its product with the right half of the 1x1
// conv's weights is a per-row constant (hyperseg_v1_0.py:404-409; utils/inference.py FusedContextHead).  16 rows per workgroup.
__global__ __launch_bounds__(256)
void pooled_shift_kernel(const float* __restrict__ partial, int nblk, float inv_p, const float* __restrict__ wb,
                         const float* __restrict__ shift, float* __restrict__ out, int M, int C) {
    extern __shared__ float ps_mean[];                                  // [C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        float t = 0.0f;
        for (int j = 0; j < nblk; ++j) t += partial[(size_t)c * nblk + j];
        ps_mean[c] = t * inv_p;
    }
    __syncthreads();
    const int r0 = 16 * blockIdx.x + 4 * wave;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int c = lane; c < C; c += 64) {
        const float m = ps_mean[c];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(wb[(size_t)min(r0 + i, M - 1) * C + c], m, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float t = wave_sum64(acc[i]);
        if (lane == 0 && r0 + i < M) out[r0 + i] = shift[r0 + i] + t;
    }
}

// waves per workgroup, chunks per wave and k-steps per chunk for an inner dimension K (the fewest k-steps per wave with 2, 4 or
// 8 waves; more than 5 per wave go in two chunks -- only the PATCH form is instantiated for that)
static bool gemm_split_plan(int K, int& nwv, int& ks, int& nch) {
    const int steps = (K + 31) / 32;
    nwv = 2;
    while (nwv < 8 && nwv < steps) nwv *= 2;
    const int per = (steps + nwv - 1) / nwv;
    nch = per > 5 ? 2 : 1;
    ks = (steps + nwv * nch - 1) / (nwv * nch);
    return ks <= 5;
}

}  // namespace hs

using namespace hs;

extern "C" int hs_gemm_split_kp(int32_t c_in) {
    int nwv, ks, nch;
    if (c_in <= 0 || !gemm_split_plan(c_in, nwv, ks, nch)) return HS_ERR_UNSUPPORTED;
    return nwv * nch * ks * 32;
}

static int launch_gemm_split_plain(const void* w_frag, const float* w_inv, const float* gate, const float* x, const float* shift,
                                   int32_t act, const float* residual, float* y, int32_t batch, int32_t c_out, int32_t c_in,
                                   int32_t kp, int32_t pixels, int up2_wo, void* stream) {
    if (!w_frag || !w_inv || !x || !y || batch <= 0 || c_out <= 0 || c_in <= 0 || pixels <= 0) return HS_ERR_BAD_ARG;
    if (act < 0 || act > 3) return HS_ERR_BAD_ARG;
    int nwv, ks, nch;
    if (!gemm_split_plan(c_in, nwv, ks, nch)) return HS_ERR_UNSUPPORTED;                  // Cin > 2560: the library GEMM
    if (kp != nwv * nch * ks * 32 || (ks > 1 && nwv != 8)) return HS_ERR_BAD_ARG;
    if (batch > 65535 || (c_out + 31) / 32 > 65535) return HS_ERR_UNSUPPORTED;
    GemmSplitArgs a{(const _Float16*)w_frag, w_inv, gate, x, shift, residual, y, c_out, c_in, kp / 32, pixels, act, 0, nullptr, up2_wo};
    dim3 grid((pixels + 31) / 32, ((c_out + 15) / 16 + 1) / 2, batch);
    const size_t wgs = (size_t)grid.x * grid.y * grid.z;
    const bool narrow = wgs <= HS_GS_NARROW_MAX_WG && pixels > 16;
    const bool tall = wgs >= HS_GS_TALL_MIN_WG && c_out > 32;
    if (narrow) grid.x = (pixels + 15) / 16;
    if (tall) grid.y = ((c_out + 15) / 16 + 3) / 4;
    const size_t lds = (size_t)nwv * (narrow ? 512 : tall ? 2048 : 1024) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    // the vector-load / 2-instruction-split form for every K depth: whole frame 0.814 ms against 0.832 (k-steps <= 2 only) and 0.858
    // (never), same box, interleaved (profiles/round3_gemm_split_policy_ab.txt)
    const bool fast = (c_in & 7) == 0 && (pixels & 1) == 0 && c_in >= 8;
#define HS_GS_(KSV, NWV, F) do { \
        if (narrow) hipLaunchKernelGGL((gemm_split_kernel<KSV, NWV, F, 1, 2, 1, false>), grid, dim3(64 * NWV), lds, s, a); \
        else if (tall) hipLaunchKernelGGL((gemm_split_kernel<KSV, NWV, F, 2, 4, 1, false>), grid, dim3(64 * NWV), lds, s, a); \
        else hipLaunchKernelGGL((gemm_split_kernel<KSV, NWV, F, 2, 2, 1, false>), grid, dim3(64 * NWV), lds, s, a); } while (0)
#define HS_GS(KSV, NWV) do { if (fast) HS_GS_(KSV, NWV, true); else HS_GS_(KSV, NWV, false); } while (0)
    if (nch == 2) {
        // 1280 < Cin <= 2560 (round 6: HyperSeg-M's last project conv, K = 1920): two chunks of KS k-steps per wave, each scaled by its own
        // per-pixel power of two; the 16-pixel form only (every launch of this depth is on a 16 x 32 map or smaller), FAST shapes only
        if (!fast || up2_wo > 0) return HS_ERR_UNSUPPORTED;
        grid.x = (pixels + 15) / 16; grid.y = ((c_out + 15) / 16 + 1) / 2;
        const size_t lds2 = (size_t)8 * 512 * sizeof(float);
#define HS_GS2(KSV) hipLaunchKernelGGL((gemm_split_kernel<KSV, 8, true, 1, 2, 2, false>), grid, dim3(512), lds2, s, a)
        switch (ks) { case 3: HS_GS2(3); break; case 4: HS_GS2(4); break; case 5: HS_GS2(5); break; default: return HS_ERR_UNSUPPORTED; }
#undef HS_GS2
        return launch_status();
    }
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

extern "C" int hs_gemm_split_fwd(const void* w_frag, const float* w_inv, const float* gate, const float* x,
                                 const float* shift, int32_t act, const float* residual, float* y,
                                 int32_t batch, int32_t c_out, int32_t c_in, int32_t kp, int32_t pixels, void* stream) {
    return launch_gemm_split_plain(w_frag, w_inv, gate, x, shift, act, residual, y, batch, c_out, c_in, kp, pixels, 0, stream);
}

// The same 1x1 convolution with its (B, c_out, Ho, Wo) result stored nearest-2x upsampled, y (B, c_out, 2 Ho, 2 Wo): the context
// head's last merge writes straight into the right half of the signal (hyperseg_v1_0.py:409-410: upsample + cat never built).
extern "C" int hs_gemm_split_up2_fwd(const void* w_frag, const float* w_inv, const float* x, const float* shift, int32_t act,
                                     float* y, int32_t batch, int32_t c_out, int32_t c_in, int32_t kp, int32_t Ho, int32_t Wo,
                                     void* stream) {
    if (Ho <= 0 || Wo <= 0 || (((size_t)y) & 7) != 0) return HS_ERR_BAD_ARG;
    return launch_gemm_split_plain(w_frag, w_inv, nullptr, x, shift, act, nullptr, y, batch, c_out, c_in, kp, Ho * Wo, Wo, stream);
}

extern "C" int hs_pooled_shift_fwd(const float* partial, int32_t nblk, float inv_pixels, const float* wb, const float* shift,
                                   float* shift_out, int32_t rows, int32_t channels, void* stream) {
    if (!partial || !wb || !shift || !shift_out || nblk <= 0 || rows <= 0 || channels <= 0) return HS_ERR_BAD_ARG;
    if (channels > 8192) return HS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pooled_shift_kernel, dim3((rows + 15) / 16), dim3(256), (size_t)channels * sizeof(float), (hipStream_t)stream,
                       partial, nblk, inv_pixels, wb, shift, shift_out, rows, channels);
    return launch_status();
}

// Conv2d(c_in, c_out, kernel 2, stride 2, no padding) on x (B, c_in, 2 Ho, 2 Wo) -> y (B, c_out, Ho, Wo) as the same GEMM with
// K = 4 c_in read through the window (the context head's down blocks, hyperseg_v1_0.py:396-401): w_frag / w_inv from the conv
// weight flattened to (c_out, 4 c_in), kp = hs_gemm_split_kp(4 c_in).  64 <= c_in <= 640; Wo even.  pool_partial (optional):
// [B][c_out][ceil(Ho Wo / 16)] sums of y over the 16-pixel blocks, for hs_pooled_shift_fwd (the global average without a launch).
extern "C" int hs_gemm_split_conv2x2_fwd(const void* w_frag, const float* w_inv, const float* x, const float* shift, int32_t act,
                                         float* y, float* pool_partial, int32_t batch, int32_t c_out, int32_t c_in, int32_t kp,
                                         int32_t Ho, int32_t Wo, void* stream) {
    if (!w_frag || !w_inv || !x || !y || batch <= 0 || c_out <= 0 || c_in <= 0 || Ho <= 0 || Wo <= 0) return HS_ERR_BAD_ARG;
    if (act < 0 || act > 3) return HS_ERR_BAD_ARG;
    int nwv, ks, nch;
    if (c_in > (1 << 20) || !gemm_split_plan(4 * c_in, nwv, ks, nch) || nwv != 8 || (Wo & 1) || (c_in & 1)) return HS_ERR_UNSUPPORTED;
    if (kp != nwv * nch * ks * 32) return HS_ERR_BAD_ARG;
    if ((((size_t)x) & 15) != 0) return HS_ERR_BAD_ARG;
    const int pixels = Ho * Wo;
    if (batch > 65535 || (c_out + 31) / 32 > 65535) return HS_ERR_UNSUPPORTED;
    GemmSplitArgs a{(const _Float16*)w_frag, w_inv, nullptr, x, shift, nullptr, y, c_out, 4 * c_in, kp / 32, pixels, act, Wo, pool_partial, 0};
    dim3 grid((pixels + 15) / 16, ((c_out + 15) / 16 + 1) / 2, batch);
    const size_t lds = (size_t)8 * 512 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define HS_GP(KSV, NCHV) hipLaunchKernelGGL((gemm_split_kernel<KSV, 8, true, 1, 2, NCHV, true>), grid, dim3(512), lds, s, a)
    switch (ks * 2 + nch - 1) {
        case 2: HS_GP(1, 1); break;
        case 4: HS_GP(2, 1); break;
        case 6: HS_GP(3, 1); break;
        case 8: HS_GP(4, 1); break;
        case 10: HS_GP(5, 1); break;
        case 7: HS_GP(3, 2); break;
        case 9: HS_GP(4, 2); break;
        case 11: HS_GP(5, 2); break;
        default: return HS_ERR_UNSUPPORTED;
    }
#undef HS_GP
    return launch_status();
}

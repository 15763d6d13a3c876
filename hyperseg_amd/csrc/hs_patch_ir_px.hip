// Op D (hyperseg_v0_1.py:205-237) for the NARROW levels of HyperSeg-L -- 11 -> 22 -> 21 channels at full resolution
// (level 5) and 16 -> 32 -> 6 at half (level 4) -- with one LANE per PIXEL and the matrix core used as a broadcast-FMA
// engine.
//
// Why a second form: a 16 x 16 x K tile of these layers is mostly padding (cin 11, hid 22, cout 21), and the tiled kernel
// (hs_patch_ir_fused.hip) spends its time in per-tile fixed costs: 26 k cycles per 256-pixel region for 0.25 M
// multiply-adds, 800 us per bs-32 batch at level 5 (profiles/round2_decoder_L_kernel_stats_mid.csv).  Per pixel the
// block is only ~980 multiply-adds -- but every one of them needs a WEIGHT that is uniform over the region.  Measured
// dead ends for feeding those weights to one-thread-per-pixel vector code (tools/ubench/valu_rate.hip, gpurun r2t):
//   * scalar loads (v_fmac with an SGPR operand): 62 s_load_dwordx16 per wave, each waited for with lgkmcnt(0) a few
//     instructions after its issue, all missing the 16 KB scalar cache that 6 workgroups thrash -> 805 us, no better;
//   * LDS broadcast reads: one ds_read_b128 per 4 FMAs runs 2.8x slower than the bare FMAs (the LDS serves a uniform
//     16-byte read in 4 cycles per wave: 64 FMA lanes per cycle and CU at best);
//   * v_pk_fma_f32 issues at exactly the rate of two v_fma_f32.
// What works is v_mfma_f32_4x4x1_16B_f32 with its A-block broadcast (tools/ubench/mfma4x4_probe.hip checks the lane
// maps): the instruction is 16 independent 4x4 outer products, lane = 4 * block + j,
//     D[lane][r] += A[4 * src(block) + r] * B[lane],   src(block) = (block & ~(2^CBSZ - 1)) + ABID.
// With B = the lane's own input value x[k] and A = four weights W[4g .. 4g+3][k], CBSZ = 4 hands the SAME four weights
// to all 64 lanes: one instruction = 4 output channels x 1 input channel for 64 pixels, at the f32 matrix rate, with no
// per-lane weight traffic at all -- ONE coalesced ds_read_b32 fetches the A operands of 16 instructions (16 values of k:
// ABID picks the block).  pw1 of a wave's 64 pixels is 66 MFMAs + 6 LDS reads, pw3 132 MFMAs + 12 reads.
//
// One workgroup = one 16 x 16 region inside a patch (patch edge % 16 == 0), 256 threads, thread = pixel:
//   inputs    cat(coords, skip, bilinear2x(prev)) of the pixel in registers (skip: coalesced loads; prev: the region's
//             low-resolution window goes through LDS once).  The 68 ring positions around the region (reflect-mapped
//             at the image border) are built by threads 0..67 into LDS.
//   pw1       interior: the region's own W1.  Ring: the hidden activation a depthwise tap reads across the region's edge
//             belongs to the patch that OWNS that position (three image-level patch convolutions in the reference), so
//             the 64 edge positions = the 64 lanes of a wave, 16 per side, use the four side owners' W1: CBSZ = 2 lets
//             every group of 4 blocks (= one side) take its A from its own lanes.  The 4 corners: plain FMAs, weights
//             straight from their owners' banks.
//             -> BN1 -> ReLU6 -> LDS h1[position][hid] (position stride = an odd number of 16-byte granules: the 16
//             lanes a ds_read_b128 serves together hit distinct banks)
//   dw        thread = pixel, 9 x hid/4 ds_read_b128 of h1, taps as uniform ds_read_b128 (the depthwise stage has no
//             outer-product structure: vector FMAs) -> BN2 -> ReLU6 in registers
//   pw3       broadcast MFMAs again (B = the lane's h2) -> BN3 -> cout coalesced stores.
// The hidden channels go through LDS in CHUNKS of GC groups of 4 (pw1(chunk) -> h1 | barrier | dw(chunk) in registers,
// pw3 += W3[:, chunk] . h2 | barrier; the lane's input vector and the pw3 accumulators live in registers across chunks):
// the h1 footprint sets the workgroups per CU, and this kernel is wait-bound -- 2 -> 3 -> 4 workgroups per CU measured
// 565 -> 470 -> 447 us at level 5, and 277 (1) -> 169 (2) -> 121 us (4) at level 4.
// Exact f32 (the f32 matrix cores compute fma chains).
#include "hs_ir_common.h"
#include <type_traits>
#include <utility>

namespace hs {

template <int CIN, int CSKIP, int COUT, int HID, int GC> struct IrPxGeom {
    static constexpr int REG = 16, HW = 18, NPOS = HW * HW;
    static constexpr int CPREV = CIN - 2 - CSKIP;
    static constexpr int CINP = (CIN + 3) & ~3;
    static constexpr int HQ = (HID + 3) / 4;                        // groups of 4 hidden channels
    static constexpr int OQ = (COUT + 3) / 4;                       // groups of 4 output channels
    static constexpr int NCH = (HQ + GC - 1) / GC;                  // the hidden channels go through LDS in chunks of GC groups
    static constexpr int HS = 4 * (GC + 1 + (GC & 1));              // h1 position stride (floats): granule count odd
    // halo-row stride: a multiple of 16 granules, so that the lanes of the NEXT pixel row that a ds_read_b128 serves
    // together with this row's ({0-3, 12-15} of one row with {4-11} of the next) complete the bank permutation
    static constexpr int HR = HW * HS + 4 * ((16 - (HW * (HS / 4)) % 16) % 16);
    static constexpr int PWIN = REG / 2 + 2, PPL = PWIN * PWIN;
    static constexpr int NRING = 4 * REG + 4;
    static constexpr int NKB1 = (CIN + 15) / 16, NKBC = (4 * GC + 15) / 16, NKQ = (CIN + 3) / 4;
    static constexpr int F_H1 = 0;
    static constexpr int F_WIN = F_H1 + HW * HR;
    static constexpr int F_XR = F_WIN + ((CPREV * PPL + 3) & ~3);
    static constexpr int F_WA1 = F_XR + NRING * CINP;               // [HQ][NKB1][64]: lane 4t + i <- W1[4g + i][16 kb + t]
    static constexpr int F_WA3 = F_WA1 + HQ * NKB1 * 64;            // [NCH][OQ][NKBC][64]: lane 4t + i <- W3[4og + i][4 GC c + 16 kb + t]
    static constexpr int F_WR = F_WA3 + NCH * OQ * NKBC * 64;             // [HQ][NKQ][64]: lane 16s + 4t + i <- W1_side s[4g + i][4 kq + t]
    static constexpr int F_KD = F_WR + HQ * NKQ * 64;               // [9 taps][4 HQ]
    static constexpr int F_BN = F_KD + 9 * 4 * HQ;                  // folded BatchNorm rows [s1 | b1 | s2 | b2] x 4 HQ, [s3 | b3] x 4 OQ
    static constexpr int FLOATS = F_BN + 4 * 4 * HQ + 2 * 4 * OQ;
};

template <int CBSZ, int ABID>
__device__ __forceinline__ f32x4 bcast_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, 0);
}
// acc[g] += W[4g.., k0 + i] * x[k0 + i] for i < N and g < NG: one MFMA per (k, group), the A block of wv[g] selected by
// ABID = i; the NG accumulator chains are interleaved so that consecutive MFMAs are independent
template <int CBSZ, int N, int NG, int I = 0>
__device__ __forceinline__ void bcast_chains(const float (&wv)[NG], const float* x, f32x4 (&acc)[NG]) {
    if constexpr (I < N) {
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = bcast_mfma<CBSZ, I>(wv[g], x[I], acc[g]);
        bcast_chains<CBSZ, N, NG, I + 1>(wv, x, acc);
    }
}

template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int CIN, int CSKIP, int COUT, int HID, int GC>
__global__ __launch_bounds__(256)
void patch_ir_px_kernel(IrFusedArgs a) {
    using G = IrPxGeom<CIN, CSKIP, COUT, HID, GC>;
    constexpr int REG = G::REG, HW = G::HW, CPREV = G::CPREV, CINP = G::CINP, HQ = G::HQ, OQ = G::OQ, HS = G::HS, HR = G::HR;
    constexpr int PWIN = G::PWIN, PPL = G::PPL, NRING = G::NRING, NKB1 = G::NKB1, NKBC = G::NKBC, NKQ = G::NKQ, NCH = G::NCH;
    static_assert(CPREV > 0 && CSKIP > 0, "fused form: coords + skip + previous level");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds + G::F_H1;            // [halo row][HR] of [halo column][HS]: the current chunk's hidden channels
    float* win = lds + G::F_WIN;          // [CPREV][PWIN * PWIN]
    float* xr = lds + G::F_XR;            // [ring position][CINP]
    float* wa1 = lds + G::F_WA1;
    float* wa3 = lds + G::F_WA3;
    float* wr = lds + G::F_WR;
    float* kdt = lds + G::F_KD;
    float* bnl = lds + G::F_BN;           // BN rows, zero beyond the real channels (uniform reads: LDS broadcasts)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk = blockIdx.x;                 // XCD-contiguous region ranges (as the tiled kernels)
    if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
    const int rx = blk % a.regs_x; blk /= a.regs_x;
    const int ry = blk % a.regs_y;
    const int b = blk / a.regs_y;
    const int y0 = ry * REG, x0 = rx * REG;
    const int H = a.in.H, W = a.in.W;
    const unsigned plane = (unsigned)H * (unsigned)W;
    // lane <-> pixel: a wave takes 4 rows of the region
    const int ty = tid >> 4, tx = tid & 15;
    const float* __restrict__ skb = a.in.skip + (size_t)b * CSKIP * plane;

    // ring position r: 0..63 = the four sides (top, bottom, left, right; 16 each), 64..67 = corners; halo coordinates (u, v)
    auto ring_uv = [&](int r, int& u, int& v) {
        if (r < 64) {
            const int side = r >> 4, i = r & 15;
            u = side == 0 ? 0 : side == 1 ? HW - 1 : 1 + i;
            v = side == 2 ? 0 : side == 3 ? HW - 1 : 1 + i;
        } else {
            u = (r & 2) ? HW - 1 : 0; v = (r & 1) ? HW - 1 : 0;
        }
    };
    // the region's patch (i0, j0) and the eight ring owners (wave-uniform): patch rows of the halo's top / bottom rows,
    // patch columns of its left / right columns; a reflected row / column stays inside the region's own patch
    const int i0 = y0 / a.ph, j0 = x0 / a.pw;
    const int ry0 = y0 - i0 * a.ph, rx0 = x0 - j0 * a.pw;
    const int it = (ry0 == 0 && y0 > 0) ? i0 - 1 : i0, ib = (ry0 + REG == a.ph && y0 + REG < H) ? i0 + 1 : i0;
    const int jl = (rx0 == 0 && x0 > 0) ? j0 - 1 : j0, jr = (rx0 + REG == a.pw && x0 + REG < W) ? j0 + 1 : j0;
    auto patch_bank = [&](int i, int j) { return a.bank + (size_t)((b * a.fh + i) * a.fw + j) * (size_t)a.ld; };
    const float* __restrict__ own = patch_bank(i0, j0);

    // ---- loads ---------------------------------------------------------------------------------------------------
    // (1) low-res window of the previous level
    const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;
    constexpr int PQ = (CPREV * PPL + 255) / 256;
    float preg[PQ];
    {
        const float* __restrict__ pvb = a.in.prev + (size_t)b * CPREV * a.in.Hp * a.in.Wp;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int e = min(tid + q * 256, CPREV * PPL - 1);
            const int c = e / PPL, rq = e - c * PPL;
            const int r = rq / PWIN, qq = rq - r * PWIN;
            const int yy = min(max(ly0 + r, 0), a.in.Hp - 1), xx = min(max(lx0 + qq, 0), a.in.Wp - 1);
            preg[q] = pvb[(size_t)(c * a.in.Hp + yy) * a.in.Wp + xx];
        }
    }
    // (2) the filter banks, straight into the lane order of the broadcast MFMA's A operand (zero beyond the real channels)
    constexpr int N1 = HQ * NKB1 * 64, N3 = NCH * OQ * NKBC * 64, NR = HQ * NKQ * 64, NK = 9 * 4 * HQ;
    constexpr int Q1 = (N1 + 255) / 256, Q3 = (N3 + 255) / 256, QR = (NR + 255) / 256;
    constexpr int QK = (NK + 255) / 256;
    float g1[Q1], g3[Q3], gr[QR], gk[QK];
#pragma unroll
    for (int q = 0; q < Q1; ++q) {
        const int e = min(tid + q * 256, N1 - 1);
        const int ln = e & 63, gk_ = e >> 6, g = gk_ / NKB1, kb = gk_ - g * NKB1;
        const int h = 4 * g + (ln & 3), k = 16 * kb + (ln >> 2);
        g1[q] = (h < HID && k < CIN) ? own[h * CIN + k] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < Q3; ++q) {
        const int e = min(tid + q * 256, N3 - 1);
        const int ln = e & 63, gk_ = e >> 6, kb = gk_ % NKBC, og = (gk_ / NKBC) % OQ, c = gk_ / (NKBC * OQ);
        const int o = 4 * og + (ln & 3), hl = 16 * kb + (ln >> 2), h = 4 * GC * c + hl;
        g3[q] = (o < COUT && hl < 4 * GC && h < HID) ? own[CIN * HID + 9 * HID + o * HID + h] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < QR; ++q) {
        const int e = min(tid + q * 256, NR - 1);
        const int ln = e & 63, gk_ = e >> 6, g = gk_ / NKQ, kq = gk_ - g * NKQ;
        const int side = ln >> 4, h = 4 * g + (ln & 3), k = 4 * kq + ((ln >> 2) & 3);
        const float* __restrict__ src = side == 0 ? patch_bank(it, j0) : side == 1 ? patch_bank(ib, j0)
                                      : side == 2 ? patch_bank(i0, jl) : patch_bank(i0, jr);
        gr[q] = (h < HID && k < CIN) ? src[h * CIN + k] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < QK; ++q) {
        const int e = min(tid + q * 256, NK - 1);
        const int tap = e / (4 * HQ), h = e - tap * (4 * HQ);
        gk[q] = h < HID ? own[CIN * HID + h * 9 + tap] : 0.0f;
    }
    // folded BatchNorm rows (one element per thread: 4 * 4 HQ + 2 * 4 OQ <= 256)
    constexpr int NBN = 4 * 4 * HQ + 2 * 4 * OQ;
    static_assert(NBN <= 256, "BN rows: one element per thread");
    float gbn = 0.0f;
    if (tid < NBN) {
        const int seg = tid < 16 * HQ ? tid / (4 * HQ) : 4 + (tid - 16 * HQ) / (4 * OQ);
        const int idx = tid < 16 * HQ ? tid - seg * (4 * HQ) : (tid - 16 * HQ) - (seg - 4) * (4 * OQ);
        const float* src = seg == 0 ? a.s1 : seg == 1 ? a.b1 : seg == 2 ? a.s2 : seg == 3 ? a.b2 : seg == 4 ? a.s3 : a.b3;
        if (idx < (seg < 4 ? HID : COUT)) gbn = src[idx];
    }
    // (3) skip features of this thread's pixel (and of its ring position)
    const int yy_i = y0 + ty, xx_i = x0 + tx;
    int yy_r = y0, xx_r = x0;
    if (tid < NRING) {
        int u, v;
        ring_uv(tid, u, v);
        yy_r = pad_index(y0 + u - 1, H, HS_PAD_REFLECT); xx_r = pad_index(x0 + v - 1, W, HS_PAD_REFLECT);
    }
    float sk_i[CSKIP], sk_r[CSKIP];
#pragma unroll
    for (int c = 0; c < CSKIP; ++c) sk_i[c] = skb[(size_t)c * plane + (size_t)yy_i * W + xx_i];
#pragma unroll
    for (int c = 0; c < CSKIP; ++c) sk_r[c] = tid < NRING ? skb[(size_t)c * plane + (size_t)yy_r * W + xx_r] : 0.0f;
    // every load is in flight: LDS stores
#pragma unroll
    for (int q = 0; q < PQ; ++q) { const int e = tid + q * 256; if (e < CPREV * PPL) win[e] = preg[q]; }
#pragma unroll
    for (int q = 0; q < Q1; ++q) { const int e = tid + q * 256; if (e < N1) wa1[e] = g1[q]; }
#pragma unroll
    for (int q = 0; q < Q3; ++q) { const int e = tid + q * 256; if (e < N3) wa3[e] = g3[q]; }
#pragma unroll
    for (int q = 0; q < QR; ++q) { const int e = tid + q * 256; if (e < NR) wr[e] = gr[q]; }
#pragma unroll
    for (int q = 0; q < QK; ++q) { const int e = tid + q * 256; if (e < NK) kdt[e] = gk[q]; }
    if (tid < NBN) bnl[tid] = gbn;
    __syncthreads();                                       // window + filter banks + BN rows in LDS

    // input vector of pixel (yy, xx): [x coordinate, y coordinate, skip.., bilinear previous level..]
    auto build_x = [&](int yy, int xx, const float (&sk)[CSKIP], float (&x)[CINP]) {
        x[0] = linspace_pm1(xx, W, a.in.step_x);
        x[1] = linspace_pm1(yy, H, a.in.step_y);
#pragma unroll
        for (int c = 0; c < CSKIP; ++c) x[2 + c] = sk[c];
        const Tap tyv = bilinear_tap(yy, a.in.scale_y, a.in.Hp), txv = bilinear_tap(xx, a.in.scale_x, a.in.Wp);
        const int r0 = tyv.i0 - ly0, r1 = tyv.i1 - ly0, q0 = txv.i0 - lx0, q1 = txv.i1 - lx0;
        const int o00 = r0 * PWIN + q0, o01 = r0 * PWIN + q1, o10 = r1 * PWIN + q0, o11 = r1 * PWIN + q1;
#pragma unroll
        for (int c = 0; c < CPREV; ++c) {
            const float* p = win + c * PPL;
            x[2 + CSKIP + c] = tyv.l0 * (txv.l0 * p[o00] + txv.l1 * p[o01]) + tyv.l1 * (txv.l0 * p[o10] + txv.l1 * p[o11]);
        }
#pragma unroll
        for (int c = CIN; c < CINP; ++c) x[c] = 0.0f;
    };
    float xi[CINP];
    build_x(yy_i, xx_i, sk_i, xi);
    if (tid < NRING) {
        float xq[CINP];
        build_x(yy_r, xx_r, sk_r, xq);
#pragma unroll
        for (int q = 0; q < CINP / 4; ++q)
            *reinterpret_cast<float4*>(xr + tid * CINP + 4 * q) = make_float4(xq[4 * q], xq[4 * q + 1], xq[4 * q + 2], xq[4 * q + 3]);
    }

    __syncthreads();                                       // ring inputs in LDS
    // this lane's edge position (the 64 lanes of a wave <-> the 64 edge positions) and its input vector
    int ue, ve;
    ring_uv(lane, ue, ve);
    float xe[CINP];
#pragma unroll
    for (int q = 0; q < CINP / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(xr + lane * CINP + 4 * q);
        xe[4 * q] = t.x; xe[4 * q + 1] = t.y; xe[4 * q + 2] = t.z; xe[4 * q + 3] = t.w;
    }
    // BN1 + ReLU6 of hidden group g (wave-uniform) -> slot j of the chunk at this lane's position
    auto store_h1 = [&](float* dst, int j, int g, const f32x4& acc) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = min(4 * g + r, HID - 1);
            o[r] = 4 * g + r < HID ? relu6_(fmaf(acc[r], bnl[h], bnl[4 * HQ + h])) : 0.0f;
        }
        *reinterpret_cast<float4*>(dst + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
    };

    f32x4 acc3[OQ];
#pragma unroll
    for (int og = 0; og < OQ; ++og) acc3[og] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- chunks of GC hidden groups:  pw1(chunk) -> h1 | barrier | dw(chunk) in registers, pw3 += W3[:, chunk] . h2 | barrier
    static_for<NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int NGC = c == NCH - 1 ? HQ - GC * (NCH - 1) : GC;                         // groups of this chunk
        constexpr int CHC = 4 * GC * c + 4 * NGC <= HID ? 4 * NGC : HID - 4 * GC * c;        // its real channels
        if constexpr (c > 0) __syncthreads();              // every wave finished the previous chunk's depthwise reads
        // pw1, interior: D[4 channels of group g][this lane's pixel] += W1[4g.., k] * x[k]
        {
            f32x4 acc[NGC];
#pragma unroll
            for (int j = 0; j < NGC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB1; ++kb) {
                float wv[NGC];
#pragma unroll
                for (int j = 0; j < NGC; ++j) wv[j] = wa1[((GC * c + j) * NKB1 + kb) * 64 + lane];
                if (kb == NKB1 - 1) bcast_chains<4, CIN - 16 * (NKB1 - 1), NGC>(wv, xi + 16 * kb, acc);
                else bcast_chains<4, 16, NGC>(wv, xi + 16 * kb, acc);
            }
            float* dst = h1 + (ty + 1) * HR + (tx + 1) * HS;
#pragma unroll
            for (int j = 0; j < NGC; ++j) store_h1(dst, j, GC * c + j, acc[j]);
        }
        // pw1, ring.  Edges: lane = edge position (16 per side); wave w takes the chunk's groups w, w + 4, ..; CBSZ = 2: the
        // 4 blocks of a side share the A block ABID of their own 16 lanes = that side's owner
        {
            float* dst = h1 + ue * HR + ve * HS;
#pragma unroll
            for (int ji = 0; ji < (NGC + 3) / 4; ++ji) {
                const int j = wave + 4 * ji;               // wave-uniform
                if (j < NGC) {
                    const int g = GC * c + j;
                    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int kq = 0; kq < NKQ; ++kq) {
                        const float wv[1] = {wr[(g * NKQ + kq) * 64 + lane]};
                        if (kq == NKQ - 1) bcast_chains<2, CIN - 4 * (NKQ - 1), 1>(wv, xe + 4 * kq, acc);
                        else bcast_chains<2, 4, 1>(wv, xe + 4 * kq, acc);
                    }
                    store_h1(dst, j, g, acc[0]);
                }
            }
            // corners: thread = (corner, channel of the chunk), plain FMAs, weights straight from the owner's bank
            if (tid < 4 * 4 * NGC) {
                const int cr = tid / (4 * NGC), hl = tid - cr * (4 * NGC), h = 4 * GC * c + hl;
                int u, v;
                ring_uv(64 + cr, u, v);
                float val = 0.0f;
                if (h < HID) {
                    const float* __restrict__ wrow = (cr == 0 ? patch_bank(it, jl) : cr == 1 ? patch_bank(it, jr)
                                                      : cr == 2 ? patch_bank(ib, jl) : patch_bank(ib, jr)) + h * CIN;
                    const float* xc = xr + (64 + cr) * CINP;
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < CIN; ++k) acc = fmaf(wrow[k], xc[k], acc);
                    val = relu6_(fmaf(acc, bnl[h], bnl[4 * HQ + h]));
                }
                h1[u * HR + v * HS + hl] = val;
            }
        }
        __syncthreads();                                   // the chunk's h1 is complete

        // depthwise 3x3 + BN2 + ReLU6 of the chunk in this thread's registers
        float h2[16 * NKBC];
#pragma unroll
        for (int h = 0; h < 16 * NKBC; ++h) h2[h] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* src = h1 + (ty + ky) * HR + (tx + kx) * HS;
                const float* kt = kdt + (ky * 3 + kx) * (4 * HQ) + 4 * GC * c;
#pragma unroll
                for (int q = 0; q < NGC; ++q) {
                    const float4 t = *reinterpret_cast<const float4*>(src + 4 * q);
                    const float4 k4 = *reinterpret_cast<const float4*>(kt + 4 * q);   // uniform address: one broadcast read
                    h2[4 * q] = fmaf(k4.x, t.x, h2[4 * q]); h2[4 * q + 1] = fmaf(k4.y, t.y, h2[4 * q + 1]);
                    h2[4 * q + 2] = fmaf(k4.z, t.z, h2[4 * q + 2]); h2[4 * q + 3] = fmaf(k4.w, t.w, h2[4 * q + 3]);
                }
            }
#pragma unroll
        for (int hl = 0; hl < CHC; ++hl) h2[hl] = relu6_(fmaf(h2[hl], bnl[8 * HQ + 4 * GC * c + hl], bnl[12 * HQ + 4 * GC * c + hl]));
#pragma unroll
        for (int hl = CHC; hl < 4 * NGC; ++hl) h2[hl] = 0.0f;

        // pw3: D[4 output channels of group og][this lane's pixel] += W3[4og.., chunk] * h2
        constexpr int NKBR = (CHC + 15) / 16;              // 16-channel blocks with real channels
#pragma unroll
        for (int kb = 0; kb < NKBR; ++kb) {
            float wv[OQ];
#pragma unroll
            for (int og = 0; og < OQ; ++og) wv[og] = wa3[((c * OQ + og) * NKBC + kb) * 64 + lane];
            if (kb == NKBR - 1) bcast_chains<4, CHC - 16 * (NKBR - 1), OQ>(wv, h2 + 16 * kb, acc3);
            else bcast_chains<4, 16, OQ>(wv, h2 + 16 * kb, acc3);
        }
    });

    float* __restrict__ yo = a.y + (size_t)b * COUT * plane + (size_t)yy_i * W + xx_i;
#pragma unroll
    for (int og = 0; og < OQ; ++og)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * og + r;
            if (o < COUT) yo[(size_t)o * plane] = fmaf(acc3[og][r], bnl[16 * HQ + o], bnl[16 * HQ + 4 * OQ + o]);
        }
}

template <int CIN, int CSKIP, int COUT, int HID, int GC>
static int launch_irp(IrFusedArgs& a, hipStream_t stream) {
    using G = IrPxGeom<CIN, CSKIP, COUT, HID, GC>;
    if (a.hid != HID || a.in.H % 16 != 0 || a.in.W % 16 != 0) return 1;
    a.regs_y = a.in.H / 16; a.regs_x = a.in.W / 16;
    constexpr size_t lds = (size_t)G::FLOATS * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};
        const int e = allow_full_lds((const void*)patch_ir_px_kernel<CIN, CSKIP, COUT, HID, GC>, done);
        if (e != HS_OK) return e;
    }
    const long blocks = (long)a.in.B * a.regs_y * a.regs_x;
    hipLaunchKernelGGL((patch_ir_px_kernel<CIN, CSKIP, COUT, HID, GC>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    return launch_status();
}

// Op D levels with narrow channels and patch edges that are multiples of 16 pixels; 1 = no instantiation
int try_launch_ir_px(IrFusedArgs& a, int cin, int c_skip, int c_out, hipStream_t stream) {
    if (a.ph != a.pw || a.ph % 16 != 0) return 1;
// GC = hidden groups per LDS chunk: sets the h1 footprint and with it the workgroups per CU (level 5: 3 chunks of 8 channels,
// 35 KB -> 4 per CU; level 4: 4 chunks of 8, 40 KB -> 4 per CU; one chunk would be 53 / 71 KB -> 3 / 2 per CU)
#define HS_IRP_CASE(CI, CS, CO, HID, GC) \
    if (cin == CI && c_skip == CS && c_out == CO && a.hid == HID) return launch_irp<CI, CS, CO, HID, GC>(a, stream);
    HS_IRP_CASE(11, 3, 21, 22, 2)    // HyperSeg-L level 5 (PASCAL VOC: 21 classes)
    HS_IRP_CASE(16, 6, 6, 32, 2)     // HyperSeg-L level 4
#undef HS_IRP_CASE
    return 1;
}

}  // namespace hs

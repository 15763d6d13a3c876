// The inverted-residual decoder levels on the f32 matrix cores, one launch per level:
//   Op C  (MODE 0)  hyperseg_v1_0.py:328-376 / hyperseg_v1_0_unify.py:330-389 -- per patch, on its own reflect halo tile
//   Op D  (MODE 1)  hyperseg_v0_1.py:205-237 -- three image-level patch convolutions (pw1 -> depthwise 3x3 with reflect
//                   padding of the HIDDEN activation -> pw3): the ring around a region is pw1 of the neighbouring
//                   patches' inputs with the NEIGHBOURS' weights, recomputed here instead of being exchanged through HBM
// and the dominant kernel of the decoder (85 % of the FLOPs at HyperSeg-M).
//
// Per region the block is two small dense GEMMs around a depthwise 3x3,
//     pw1  [hid x cin] . [cin x halo positions]          pw3  [cout x hid] . [hid x pixels]
// on v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fma chain at the fp32 peak rate).  One workgroup (4 waves) = one
// REG x REG region; the tile maps (which 16 positions share a filter bank) are in hs_ir_tiles.h.
//
//   prologue   the stage input cat(coords, skip, bilinear2x(prev)) is built DIRECTLY in the layout of the matrix
//              cores' B operand (lane = (position, channel mod 4)): skip features are gathered from HBM straight into
//              the fragment registers, the previous level's low-resolution window goes through LDS once and is
//              sampled with the 4-tap stencil, coordinates are analytic.  The fragments stay in registers for the
//              whole kernel; all HBM loads of the workgroup are in flight together.
//   pw1        per owned position tile ceil(cin/4) MFMAs; D -> BN1 -> ReLU6 -> LDS h1[2][16][halo] (double-buffered)
//   dw         thread = (hidden channel, output row): rows of the halo as ds_read_b128, 9 per-lane weights (of the
//              patch that owns the OUTPUT pixel), -> BN2 -> ReLU6 -> LDS h2[16][pixels]
//   pw3        per owned pixel tile 4 MFMAs per 16 output channels, accumulators persistent across hidden chunks
//   epilogue   BN3, row runs to HBM.
// Hidden channels are processed in chunks of 16, software-pipelined: pw1 of chunk c+1 (matrix pipe) is issued in the
// same basic block as the depthwise stage of chunk c (VALU + LDS), so the two pipes of a SIMD overlap inside every
// wave instead of alternating behind barriers:
//     step c:   { pw1(c+1) -> h1[(c+1)&1]  ||  dw(c): h1[c&1] -> h2 }   barrier   { pw3(c): h2 -> acc }   barrier
// Filter-bank operands (the A fragments, the depthwise taps, BN rows) are plain vector loads from the bank in HBM/L2
// issued one phase ahead of their use ("load next after use"); nothing of the bank is staged in LDS, which is what
// lets h1 be double-buffered inside 80 KB (two workgroups per CU at HyperSeg-M level 4: 69.3 KB).
// Hidden activations never leave the CU.
#include "hs_common.h"
#include "hs_ir_tiles.h"
#include <type_traits>
#include <utility>

namespace hs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct IrFusedArgs {
    StageIn in;
    int fh, fw, ph, pw;
    const float* __restrict__ bank;
    long ld;
    int hid;
    const float* __restrict__ s1; const float* __restrict__ b1;
    const float* __restrict__ s2; const float* __restrict__ b2;
    const float* __restrict__ s3; const float* __restrict__ b3;
    float* __restrict__ y;
    int regs_y, regs_x;          // regions per image
};

constexpr int IRF_THREADS = 256;

// 1 = the mixed {pw1 || depthwise} block is emitted as explicit slices "one MFMA + its share of the depthwise stage's
// VALU / LDS instructions", pinned with sched_barrier (an in-order wave overlaps its matrix and vector work only if
// they alternate in program order; left alone, the compiler clusters the MFMAs and runs the depthwise stage after them).
// 0 = the two stages one after the other (dev A/B).
#ifndef HS_IRF_INTERLEAVE
#define HS_IRF_INTERLEAVE 1
#endif

template <int REG> struct IrfGeom {
    static constexpr int HW = REG + 2;
    static constexpr int RS = (HW + 3) & ~3;                    // h1 row stride (floats): 16-byte aligned rows
    // h1 plane per hidden channel, == 4 (mod 8) floats: the D-row groups of a half-wave then store to banks 16 apart.
    // The plane's tail [HW*RS, H1P) is padding; its first float is the DUMMY slot dead columns store to.
    static constexpr int H1P = ((HW * RS + 7) & ~7) + 4;
    static constexpr int DUMMY = HW * RS;
    static constexpr int PWIN = REG / 2 + 2;                    // low-res window edge of the previous level (exact 2x)
    static constexpr int PPL = PWIN * PWIN;
    static constexpr int RS2 = REG + 4;                         // h2 pixel-row stride
    static constexpr int H2S = ((REG * RS2 + 31) & ~31) + 16;   // h2 plane: == 16 (mod 32) -> conflict-free B reads
    static constexpr int H1_FLOATS = 2 * 16 * H1P;
    static constexpr int H2_FLOATS = 16 * H2S;
};

__device__ __forceinline__ float relu6_(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// Compile-time loop: f(std::integral_constant<int, LO>), ..., f(std::integral_constant<int, HI-1>).  The mixed stage's
// schedule needs every index as a constant expression (register arrays indexed by anything else end up in scratch).
template <int LO, int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, LO + I>{}), ...);
}
template <int LO, int HI, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (HI > LO) static_for_impl<LO>(std::make_integer_sequence<int, HI - LO>{}, f);
}

template <int CIN, int CSKIP, int COUT, int REG, int MODE, int PWR>
__global__ __launch_bounds__(IRF_THREADS, 2)
void patch_ir_fused_kernel(IrFusedArgs a) {
    constexpr bool INTERLEAVE = HS_IRF_INTERLEAVE != 0;
    using G = IrfGeom<REG>;
    using TM = IrTiles<REG, MODE, PWR>;
    constexpr int CPREV = CIN - 2 - CSKIP;
    static_assert(CPREV > 0 && CSKIP > 0, "fused form: coords + skip + previous level");
    constexpr int KS1 = (CIN + 3) / 4;
    constexpr int MT3 = (COUT + 15) / 16;
    constexpr int NT1 = TM::NT1, NT3 = TM::NT3;
    constexpr int J1 = (NT1 + 3) / 4, J3 = NT3 / 4;
    static_assert(NT3 % 4 == 0, "pixel tiles split evenly over the 4 waves");
    constexpr bool P1_UNI = (MODE == 0);        // every pw1 tile uses the region's own patch
    constexpr bool IN_UNI = (PWR == REG);       // every pixel of the region belongs to one patch
    constexpr int SEGS = REG / PWR;
    // depthwise stage: thread = (hidden channel of the chunk, output row, DWW-pixel run of that row): all 256 threads busy
    constexpr int DWW = REG * REG / 16;                    // 16 (a whole row) or 4 (half a row of an 8x8 region)
    constexpr int NRD = (DWW + 2 + 3) / 4;                 // 16-byte reads per halo row
    constexpr int NKD = (IN_UNI || DWW <= PWR) ? 1 : DWW / PWR;   // tap sets per thread (one per patch under its run)
    constexpr int NA1 = P1_UNI ? 1 : J1, NA3 = IN_UNI ? 1 : J3;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds;                                  // [2][16][H1P]
    float* h2 = lds + G::H1_FLOATS;                   // [16][H2S]
    float* bnl = h2 + G::H2_FLOATS;                   // [s1 | b1 | s2 | b2] x hid, [s3 | b3] x COUT
    float* pl = lds;                                  // prologue only: [CPREV][PWIN*PWIN], aliases h1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lk = lane >> 4;
    int blk = blockIdx.x;
    const int rx = blk % a.regs_x; blk /= a.regs_x;
    const int ry = blk % a.regs_y;
    const int b = blk / a.regs_y;
    const int y0 = ry * REG, x0 = rx * REG;
    const int hid = a.hid;
    const int H = a.in.H, W = a.in.W;
    auto owner_of = [&](int yy, int xx) { return (b * a.fh + yy / a.ph) * a.fw + xx / a.pw; };
    const float* __restrict__ bank = a.bank;
    const size_t off_kd = (size_t)CIN * hid, off_w3 = off_kd + 9 * (size_t)hid;

    // ---- prologue ------------------------------------------------------------------------------------------------
    // (1) low-res window of the previous level: rows [ly0, ly0+PWIN) x cols [lx0, lx0+PWIN), clamped at the image border;
    //     every bilinear tap of the halo grid, reflected positions included, falls inside it
    const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;
    constexpr int PQ = (CPREV * G::PPL + IRF_THREADS - 1) / IRF_THREADS;
    float preg[PQ];
    {
        const float* __restrict__ pvb = a.in.prev + (size_t)b * CPREV * a.in.Hp * a.in.Wp;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int e = min(tid + q * IRF_THREADS, CPREV * G::PPL - 1);
            const int c = e / G::PPL, rq = e - c * G::PPL;
            const int r = rq / G::PWIN, qq = rq - r * G::PWIN;
            const int yy = min(max(ly0 + r, 0), a.in.Hp - 1), xx = min(max(lx0 + qq, 0), a.in.Wp - 1);
            preg[q] = pvb[((size_t)c * a.in.Hp + yy) * a.in.Wp + xx];
        }
    }
    // (2) positions of this wave's pw1 tiles and the skip-feature gathers, straight into the B fragments
    float bf[J1][KS1];
    int hoff[J1];                 // LDS offset of this lane's position inside an h1 plane (DUMMY for a dead column)
    int pyx[J1];                  // image coordinates (yy << 16 | xx) of this lane's position
    int own1[NA1];                // patch that owns the tile (MODE 1)
    {
        const size_t plane = (size_t)H * W;
        const float* __restrict__ skb = a.in.skip + (size_t)b * CSKIP * plane;
#pragma unroll
        for (int jt = 0; jt < J1; ++jt) {
            const int t = wave + 4 * jt;
            const bool tile_ok = (NT1 % 4 == 0) || t < NT1;
            int u, v;
            const bool live = TM::halo(tile_ok ? t : 0, lrow, u, v) && tile_ok;
            const int yy = pad_index(y0 + u - 1, H, HS_PAD_REFLECT), xx = pad_index(x0 + v - 1, W, HS_PAD_REFLECT);
            hoff[jt] = live ? u * G::RS + v : G::DUMMY;
            pyx[jt] = live ? ((yy << 16) | xx) : -1;
            if constexpr (!P1_UNI) {
                int u0, v0;
                TM::halo(tile_ok ? t : 0, 0, u0, v0);
                own1[jt] = owner_of(pad_index(y0 + u0 - 1, H, HS_PAD_REFLECT), pad_index(x0 + v0 - 1, W, HS_PAD_REFLECT));
            }
            const float* __restrict__ sp = skb + (size_t)yy * W + xx;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int c = ks * 4 + lk;
                float val = 0.0f;
                if (live && c >= 2 && c < 2 + CSKIP) val = sp[(size_t)(c - 2) * plane];
                bf[jt][ks] = val;
            }
        }
        if constexpr (P1_UNI) own1[0] = owner_of(y0, x0);
    }
    // (3) folded BatchNorm rows -> LDS
    {
        const int nb = 4 * hid + 2 * COUT;
        for (int e = tid; e < nb; e += IRF_THREADS) {
            const float* __restrict__ srcp;
            int off;
            if (e < hid) { srcp = a.s1; off = e; }
            else if (e < 2 * hid) { srcp = a.b1; off = e - hid; }
            else if (e < 3 * hid) { srcp = a.s2; off = e - 2 * hid; }
            else if (e < 4 * hid) { srcp = a.b2; off = e - 3 * hid; }
            else if (e < 4 * hid + COUT) { srcp = a.s3; off = e - 4 * hid; }
            else { srcp = a.b3; off = e - 4 * hid - COUT; }
            bnl[e] = srcp[off];
        }
    }
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int e = tid + q * IRF_THREADS;
        if (e < CPREV * G::PPL) pl[e] = preg[q];
    }

    // ---- filter-bank operands ("load next after use") ------------------------------------------------------------
    float afa[NA1][KS1];               // pw1 A fragments  W1[h0 + lrow][4*ks + lk]  of the tile's owner
    float k9[NKD][9], sc2, sh2;        // depthwise taps (of the owner of the output pixels) + bn2 of this thread's channel
    float a3[NA3][MT3][4];             // pw3 A fragments  W3[16*m + lrow][h0 + 4*ks + lk]
    const int dw_hh = tid >> 4;
    const int dw_u = (tid & 15) % REG, dw_c0 = ((tid & 15) / REG) * DWW;
    int own3[NA3], ownd[NKD];
#pragma unroll
    for (int q = 0; q < NA3; ++q) {
        int row, col;
        TM::pixel(wave + 4 * q, 0, row, col);
        own3[q] = IN_UNI ? owner_of(y0, x0) : owner_of(y0 + row, x0 + col);
    }
#pragma unroll
    for (int q = 0; q < NKD; ++q)
        ownd[q] = IN_UNI ? owner_of(y0, x0) : owner_of(y0 + dw_u, x0 + dw_c0 + q * PWR);

    auto load_a1 = [&](int h0) {
        const int h = h0 + lrow;
        const bool hok = h < hid;
        const size_t row = (size_t)(hok ? h : 0) * CIN;
#pragma unroll
        for (int q = 0; q < NA1; ++q) {
            const float* __restrict__ wr = bank + (size_t)own1[q] * a.ld + row;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int k = ks * 4 + lk;
                const float v = wr[k < CIN ? k : 0];
                afa[q][ks] = (hok && k < CIN) ? v : 0.0f;
            }
        }
    };
    auto load_kd = [&](int h0) {
        const int h = h0 + dw_hh;
        const int hc = h < hid ? h : 0;
#pragma unroll
        for (int q = 0; q < NKD; ++q) {
            const float* __restrict__ kr = bank + (size_t)ownd[q] * a.ld + off_kd + (size_t)hc * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) k9[q][e] = kr[e];
        }
        sc2 = a.s2[hc]; sh2 = a.b2[hc];
    };
    auto load_a3 = [&](int h0) {
#pragma unroll
        for (int q = 0; q < NA3; ++q) {
            const float* __restrict__ w3 = bank + (size_t)own3[q] * a.ld + off_w3;
#pragma unroll
            for (int m = 0; m < MT3; ++m) {
                const int oc = m * 16 + lrow;
                const bool ook = oc < COUT;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int h = h0 + ks * 4 + lk;
                    const float v = w3[(size_t)(ook ? oc : 0) * hid + (h < hid ? h : 0)];
                    a3[q][m][ks] = (ook && h < hid) ? v : 0.0f;
                }
            }
        }
    };
    load_a1(0);
    load_kd(0);
    load_a3(0);
    __syncthreads();                                   // window + BN rows are in LDS

    // (4) coordinates and the bilinear-resized previous level complete the B fragments
#pragma unroll
    for (int jt = 0; jt < J1; ++jt) {
        const bool live = pyx[jt] >= 0;
        const int yy = live ? (pyx[jt] >> 16) : y0, xx = live ? (pyx[jt] & 0xffff) : x0;   // dead lanes: any valid pixel
        const Tap ty = bilinear_tap(yy, a.in.scale_y, a.in.Hp), tx = bilinear_tap(xx, a.in.scale_x, a.in.Wp);
        const int r0 = ty.i0 - ly0, r1 = ty.i1 - ly0, q0 = tx.i0 - lx0, q1 = tx.i1 - lx0;
        const int o00 = r0 * G::PWIN + q0, o01 = r0 * G::PWIN + q1, o10 = r1 * G::PWIN + q0, o11 = r1 * G::PWIN + q1;
        const float cx = linspace_pm1(xx, W, a.in.step_x), cy = linspace_pm1(yy, H, a.in.step_y);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int c = ks * 4 + lk;
            float val = bf[jt][ks];
            if (c < 2) val = (c == 0) ? cx : cy;
            if (c >= 2 + CSKIP && c < CIN) {
                const float* q = pl + (c - 2 - CSKIP) * G::PPL;
                val = ty.l0 * (tx.l0 * q[o00] + tx.l1 * q[o01]) + ty.l1 * (tx.l0 * q[o10] + tx.l1 * q[o11]);
            }
            bf[jt][ks] = live ? val : 0.0f;
        }
    }
    int h2off[J3];                                     // LDS offset of this lane's pixel inside an h2 plane
    int pix3[J3];                                      // (row << 8 | col) of this lane's pixel
#pragma unroll
    for (int jt = 0; jt < J3; ++jt) {
        int row, col;
        TM::pixel(wave + 4 * jt, lrow, row, col);
        h2off[jt] = row * G::RS2 + col;
        pix3[jt] = (row << 8) | col;
    }
    f32x4 acc3[MT3][J3];
#pragma unroll
    for (int m = 0; m < MT3; ++m)
#pragma unroll
        for (int jt = 0; jt < J3; ++jt) acc3[m][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                   // the window is dead: h1 may overwrite it

    // ---- stages -------------------------------------------------------------------------------------------------
    // pw1 of the chunk starting at h0 into h1 buffer hb, then the loads of the FOLLOWING chunk's A fragments
    auto stage_pw1 = [&](int h0, float* __restrict__ hb) {
        float sc1[4], sh1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hr = h0 + 4 * lk + r;
            const int hc = hr < hid ? hr : 0;
            sc1[r] = bnl[hc]; sh1[r] = bnl[hid + hc];
        }
        // branch-free on purpose (a dead tile multiplies zeros and stores to the DUMMY slot): the whole stage must stay
        // in ONE basic block with the depthwise stage for the two to be interleaved
#pragma unroll
        for (int jt = 0; jt < J1; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afa[P1_UNI ? 0 : jt][ks], bf[jt][ks], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                hb[(4 * lk + r) * G::H1P + hoff[jt]] = relu6_(fmaf(acc[r], sc1[r], sh1[r]));
        }
        load_a1(h0 + 16 < hid ? h0 + 16 : h0);          // next chunk's rows (the last chunk reloads its own)
    };
    // depthwise 3x3 + bn2 + relu6 of one chunk: h1 buffer hb -> h2; then the loads of the next chunk's taps
    auto stage_dw = [&](int h0, const float* __restrict__ hb) {
        const float* hp = hb + dw_hh * G::H1P + dw_u * G::RS + dw_c0;
        float o[DWW];
#pragma unroll
        for (int v = 0; v < DWW; ++v) o[v] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float rowv[NRD * 4];
#pragma unroll
            for (int q = 0; q < NRD; ++q) {
                const float4 v4 = *reinterpret_cast<const float4*>(hp + ky * G::RS + 4 * q);
                rowv[4 * q] = v4.x; rowv[4 * q + 1] = v4.y; rowv[4 * q + 2] = v4.z; rowv[4 * q + 3] = v4.w;
            }
#pragma unroll
            for (int v = 0; v < DWW; ++v)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    o[v] = fmaf(k9[NKD == 1 ? 0 : v / PWR][ky * 3 + kx], rowv[v + kx], o[v]);
        }
        float* dst = h2 + dw_hh * G::H2S + dw_u * G::RS2 + dw_c0;
#pragma unroll
        for (int q = 0; q < DWW / 4; ++q)
            *reinterpret_cast<float4*>(dst + 4 * q) =
                make_float4(relu6_(fmaf(o[4 * q], sc2, sh2)), relu6_(fmaf(o[4 * q + 1], sc2, sh2)),
                            relu6_(fmaf(o[4 * q + 2], sc2, sh2)), relu6_(fmaf(o[4 * q + 3], sc2, sh2)));
        load_kd(h0 + 16 < hid ? h0 + 16 : h0);
    };
    // pw3: acc3 += W3[:, chunk] . h2; then the loads of the next chunk's A fragments
    auto stage_pw3 = [&](int h0, auto full) {
        // FULL: all 16 channels of the chunk exist (every chunk but possibly the last): no k-step test, one block
        constexpr bool FULL = decltype(full)::value;
        float bv[4][J3];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int jt = 0; jt < J3; ++jt) bv[ks][jt] = h2[(ks * 4 + lk) * G::H2S + h2off[jt]];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (FULL || h0 + ks * 4 < hid) {           // uniform: skip k-steps past the last hidden channel
#pragma unroll
                for (int jt = 0; jt < J3; ++jt)
#pragma unroll
                    for (int m = 0; m < MT3; ++m)
                        acc3[m][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[IN_UNI ? 0 : jt][m][ks], bv[ks][jt], acc3[m][jt], 0, 0, 0);
            }
        }
        load_a3(h0 + 16 < hid ? h0 + 16 : h0);
    };
    // ---- software-pipelined chunk loop ---------------------------------------------------------------------------
    // The mixed stage: pw1 of chunk h0n -> hbn interleaved with the depthwise stage of chunk h0d: hbc -> h2.
    // MFMA order: tiles in pairs (two independent accumulator chains: a dependent MFMA needs 40 cycles, the pipe 32);
    // a tile's BN1/ReLU6/LDS-store epilogue is emitted three MFMAs after its last one.  The depthwise stage is a flat
    // list of micro-ops (halo-row reads through two row buffers, FMAs, BN2/ReLU6 + stores) spread evenly over the slices.
    auto stage_mixed = [&](int h0n, float* __restrict__ hbn, int h0d, const float* __restrict__ hbc) {
        constexpr int NSL = J1 * KS1;                  // slices = MFMAs of the pw1 stage
        constexpr int NF = DWW * 3;                    // FMAs per halo row
        constexpr int OP_R1 = NRD, OP_F0 = 2 * NRD, OP_R2 = OP_F0 + NF, OP_F1 = OP_R2 + NRD, OP_F2 = OP_F1 + NF,
                      OP_FIN = OP_F2 + NF, NOPS = OP_FIN + DWW / 4;
        float sc1[4], sh1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hr = h0n + 4 * lk + r;
            const int hc = hr < hid ? hr : 0;
            sc1[r] = bnl[hc]; sh1[r] = bnl[hid + hc];
        }
        const float* hp = hbc + dw_hh * G::H1P + dw_u * G::RS + dw_c0;
        float* dst = h2 + dw_hh * G::H2S + dw_u * G::RS2 + dw_c0;
        float o[DWW], rowv[2][NRD * 4];
#pragma unroll
        for (int v = 0; v < DWW; ++v) o[v] = 0.0f;
        f32x4 acc[J1];
#pragma unroll
        for (int jt = 0; jt < J1; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto dw_op = [&](auto I_) {
            constexpr int i = decltype(I_)::value;
            if constexpr (i < OP_F0 || (i >= OP_R2 && i < OP_F1)) {                 // one 16-byte read of a halo row
                constexpr int ky = i < OP_R1 ? 0 : (i < OP_F0 ? 1 : 2);
                constexpr int q = i < OP_R1 ? i : (i < OP_F0 ? i - OP_R1 : i - OP_R2);
                constexpr int buf = ky == 1 ? 1 : 0;
                const float4 v4 = *reinterpret_cast<const float4*>(hp + ky * G::RS + 4 * q);
                rowv[buf][4 * q] = v4.x; rowv[buf][4 * q + 1] = v4.y; rowv[buf][4 * q + 2] = v4.z; rowv[buf][4 * q + 3] = v4.w;
            } else if constexpr (i < OP_FIN) {                                       // one tap
                constexpr int ky = i < OP_R2 ? 0 : (i < OP_F2 ? 1 : 2);
                constexpr int f = i < OP_R2 ? i - OP_F0 : (i < OP_F2 ? i - OP_F1 : i - OP_F2);
                constexpr int buf = ky == 1 ? 1 : 0;
                constexpr int v = f / 3, kx = f - 3 * v;
                o[v] = fmaf(k9[NKD == 1 ? 0 : v / PWR][ky * 3 + kx], rowv[buf][v + kx], o[v]);
            } else {                                                                 // BN2 + ReLU6 + store of 4 pixels
                constexpr int q = i - OP_FIN;
                *reinterpret_cast<float4*>(dst + 4 * q) =
                    make_float4(relu6_(fmaf(o[4 * q], sc2, sh2)), relu6_(fmaf(o[4 * q + 1], sc2, sh2)),
                                relu6_(fmaf(o[4 * q + 2], sc2, sh2)), relu6_(fmaf(o[4 * q + 3], sc2, sh2)));
            }
        };
        auto tile_epilogue = [&](auto T_) {
            constexpr int jt = decltype(T_)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                hbn[(4 * lk + r) * G::H1P + hoff[jt]] = relu6_(fmaf(acc[jt][r], sc1[r], sh1[r]));
        };
        constexpr int NPAIR = J1 / 2;
        static_for<0, NSL>([&](auto M_) {
            constexpr int m = decltype(M_)::value;
            // (tile, k-step) of MFMA m
            constexpr bool paired = m < NPAIR * 2 * KS1;
            constexpr int pr = m / (2 * KS1), r2 = m - pr * 2 * KS1;
            constexpr int jt = paired ? 2 * pr + (r2 & 1) : J1 - 1;
            constexpr int ks = paired ? (r2 >> 1) : m - NPAIR * 2 * KS1;
            acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afa[P1_UNI ? 0 : jt][ks], bf[jt][ks], acc[jt], 0, 0, 0);
            // epilogues that fall due in this slice (three MFMAs after the tile's last one)
            static_for<0, J1>([&](auto T_) {
                constexpr int t = decltype(T_)::value;
                constexpr int last = (t < NPAIR * 2) ? (t / 2) * 2 * KS1 + 2 * (KS1 - 1) + (t & 1) : NSL - 1;
                if constexpr (last + 3 < NSL && m == last + 3) tile_epilogue(T_);
            });
            if constexpr (INTERLEAVE) {
                static_for<m * NOPS / NSL, (m + 1) * NOPS / NSL>(dw_op);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // tiles whose epilogue could not be placed three MFMAs later
        static_for<0, J1>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            constexpr int last = (t < NPAIR * 2) ? (t / 2) * 2 * KS1 + 2 * (KS1 - 1) + (t & 1) : NSL - 1;
            if constexpr (last + 3 >= NSL) tile_epilogue(T_);
        });
        if constexpr (!INTERLEAVE) static_for<0, NOPS>(dw_op);
        load_a1(h0n + 16 < hid ? h0n + 16 : h0n);
        load_kd(h0d + 16 < hid ? h0d + 16 : h0d);
    };

    const int nchunks = (hid + 15) >> 4;
    stage_pw1(0, h1);
    __syncthreads();
    for (int ch = 0; ch + 1 < nchunks; ++ch) {
        const int h0 = ch * 16;
        float* cur = h1 + (ch & 1) * (16 * G::H1P);
        float* nxt = h1 + ((ch + 1) & 1) * (16 * G::H1P);
        stage_mixed(h0 + 16, nxt, h0, cur);            // matrix pipe under the VALU / LDS work of the previous chunk
        __syncthreads();
        stage_pw3(h0, std::true_type{});
        __syncthreads();                               // h2 is rewritten by the next depthwise stage
    }
    {
        const int ch = nchunks - 1;
        stage_dw(ch * 16, h1 + (ch & 1) * (16 * G::H1P));
        __syncthreads();
        stage_pw3(ch * 16, std::false_type{});
    }

    // ---- epilogue: bn3 + store -----------------------------------------------------------------------------------
#pragma unroll
    for (int m = 0; m < MT3; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = m * 16 + 4 * lk + r;
            if (o < COUT) {
                const float sc = bnl[4 * hid + o], sh = bnl[4 * hid + COUT + o];
#pragma unroll
                for (int jt = 0; jt < J3; ++jt) {
                    const int row = pix3[jt] >> 8, col = pix3[jt] & 0xff;
                    a.y[(((size_t)b * COUT + o) * H + (y0 + row)) * W + (x0 + col)] = fmaf(acc3[m][jt][r], sc, sh);
                }
            }
        }
    }
}

template <int CIN, int CSKIP, int COUT, int REG, int MODE, int PWR>
static int launch_irf(IrFusedArgs& a, hipStream_t stream) {
    using G = IrfGeom<REG>;
    if (a.in.H % REG != 0 || a.in.W % REG != 0) return 1;      // regions must tile the level
    a.regs_y = a.in.H / REG; a.regs_x = a.in.W / REG;
    const size_t bn_floats = ((size_t)4 * a.hid + 2 * COUT + 3) & ~(size_t)3;
    const size_t lds = ((size_t)G::H1_FLOATS + G::H2_FLOATS + bn_floats) * sizeof(float);
    if (lds > 160 * 1024) return HS_ERR_LDS;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};       // one per instantiation
        const int e = allow_full_lds((const void*)patch_ir_fused_kernel<CIN, CSKIP, COUT, REG, MODE, PWR>, done);
        if (e != HS_OK) return e;
    }
    const long blocks = (long)a.in.B * a.regs_y * a.regs_x;
    hipLaunchKernelGGL((patch_ir_fused_kernel<CIN, CSKIP, COUT, REG, MODE, PWR>), dim3((unsigned)blocks),
                       dim3(IRF_THREADS), lds, stream, a);
    return launch_status();
}

// Called by hs_patch_ir_fwd (mode 0) / hs_patch_ir_v0_fwd (mode 1) for the decoder's fused form.  Returns 1 if no
// instantiation matches (the caller then uses the generic kernels), otherwise the launch status.
int try_launch_ir_fused(int mode, const StageIn& in, int fh, int fw, const float* bank, long ld, int cin, int c_skip,
                        int hid, int c_out, const float* s1, const float* b1, const float* s2, const float* b2,
                        const float* s3, const float* b3, float* y, hipStream_t stream) {
    IrFusedArgs a;
    a.in = in; a.fh = fh; a.fw = fw; a.ph = in.H / fh; a.pw = in.W / fw;
    a.bank = bank; a.ld = ld; a.hid = hid;
    a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3; a.y = y;
    if (in.Hp * 2 != in.H || in.Wp * 2 != in.W) return 1;      // the LDS window assumes the exact 2x pyramid
    if (in.H >= 32768 || in.W >= 32768) return 1;              // packed (yy << 16 | xx) positions
    if (a.ph != a.pw) return 1;
    const int p = a.ph;
#define HS_IRF_CASE(CI, CS, CO, REG, MODE, PWR) \
    if (cin == CI && c_skip == CS && c_out == CO) return launch_irf<CI, CS, CO, REG, MODE, PWR>(a, stream);
    if (mode == 0) {
        if (p % 16 == 0) {
            HS_IRF_CASE(34, 16, 19, 16, 0, 16)   // HyperSeg-M level 4 (Cityscapes, 19 classes)
            HS_IRF_CASE(26, 16, 19, 16, 0, 16)   // HyperSeg-S level 4
            HS_IRF_CASE(22, 4, 12, 16, 0, 16)    // CamVid-S level 4 (12 classes)
            HS_IRF_CASE(24, 6, 16, 16, 0, 16)    // level-3 shapes on larger patches
        }
        if (p % 8 == 0) {
            HS_IRF_CASE(24, 6, 16, 8, 0, 8)      // HyperSeg-M / CamVid-S level 3
            HS_IRF_CASE(14, 4, 8, 8, 0, 8)       // HyperSeg-S level 3
            HS_IRF_CASE(34, 16, 19, 8, 0, 8)
            HS_IRF_CASE(22, 4, 12, 8, 0, 8)
        }
        return 1;
    }
    // Op D (HyperSeg-L / v0_1): patch edge 4 -> 8x8 regions of 2x2 patches; 8 -> 16x16 regions of 2x2 patches;
    // >= 16 -> 16x16 regions inside one patch
    if (p == 4) { HS_IRF_CASE(48, 12, 12, 8, 1, 4) }
    if (p == 8) { HS_IRF_CASE(22, 8, 8, 16, 1, 8) }
    if (p % 16 == 0) {
        HS_IRF_CASE(16, 6, 6, 16, 1, 16)
        HS_IRF_CASE(11, 3, 21, 16, 1, 16)
    }
#undef HS_IRF_CASE
    return 1;
}

}  // namespace hs

using namespace hs;

// Introspection for the tests: the tile map of one (region edge, mode, patch edge) combination.
// out[(t*16 + n)*3 + {0,1,2}] = (u, v, live) for t < *n_tiles; returns 0, or HS_ERR_UNSUPPORTED.
extern "C" int hs_ir_tile_map(int32_t reg, int32_t mode, int32_t pwr, int32_t* n_tiles, int32_t* n_pixel_tiles,
                              int32_t* out, int32_t capacity) {
    if (!n_tiles || !n_pixel_tiles) return HS_ERR_BAD_ARG;
#define HS_TM_CASE(R, M, P) \
    if (reg == R && mode == M && pwr == P) { \
        using TM = IrTiles<R, M, P>; \
        *n_tiles = TM::NT1; *n_pixel_tiles = TM::NT3; \
        if (out) { \
            if (capacity < TM::NT1 * 16 * 3) return HS_ERR_BAD_ARG; \
            for (int t = 0; t < TM::NT1; ++t) \
                for (int n = 0; n < 16; ++n) { \
                    int u, v; \
                    const bool live = TM::halo(t, n, u, v); \
                    out[(t * 16 + n) * 3] = u; out[(t * 16 + n) * 3 + 1] = v; out[(t * 16 + n) * 3 + 2] = live ? 1 : 0; \
                } \
        } \
        return HS_OK; \
    }
    HS_TM_CASE(16, 0, 16) HS_TM_CASE(8, 0, 8) HS_TM_CASE(8, 1, 4) HS_TM_CASE(16, 1, 8) HS_TM_CASE(16, 1, 16)
    HS_TM_CASE(16, 1, 4) HS_TM_CASE(8, 1, 8)
#undef HS_TM_CASE
    return HS_ERR_UNSUPPORTED;
}

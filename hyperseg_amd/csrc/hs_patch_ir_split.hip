// The inverted-residual decoder levels on the f16 matrix cores with SPLIT operands: f32-class results at ~4x the
// matrix rate of the exact-f32 form (hs_patch_ir_fused.hip, whose structure -- regions, tile maps, chunk loop -- this
// kernel shares; read that file's header first).
//
// Why: v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate and excludes the SIMD's other VALU work while it does
// (profiles/round2_ubench_mfma_valu_overlap.txt), so the exact form spends 27 of its 66 kcycles per CU inside pw1/pw3.
// v_mfma_f32_16x16x32_f16 retires 8x the MACs per cycle.  Every f32 operand x is scaled by a power of two and split
//     x * 2^e = hi + lo + r,   hi = f16(x * 2^e),  lo = f16(x * 2^e - hi),  |r| <= 2^-22 |x * 2^e|
// and a product a*b is taken as  ah*bh + al*bh + ah*bl  (three f16 MFMAs accumulating in f32; al*bl ~ 2^-22 is dropped).
// On the GPU probe tools/ubench/f16_probe.hip the worst |error| / sum|a||b| of a 32-term dot product is 1.3e-7, BELOW
// the 2.2e-7 of an f32 fmaf chain; tests/test_hip_parity.py holds this kernel to the same tolerance as the exact one.
// The power-of-two scales keep hi/lo inside f16's range whatever the data's magnitude:
//     pw1 weights   per (owner patch, hidden row, chunk): row maximum -> 2^15; undone through the row's BN1 scale
//     pw1 input     per position tile (16 positions x cin): tile maximum -> 2^15; undone with the same multiply
//     pw3 weights   per output row, a RUNNING exponent over the chunks (the accumulators live across chunks): it starts
//                   with 2 bits of headroom and only grows; when a row's grows the waves rescale their accumulators (exact)
//     pw3 input     h2 = relu6(.) is in [0, 6]: a static 2^12
// K layout.  The three products are CONCATENATED along K: any permutation of K is free as long as A and B agree.  With
// q the channel in kernel order [skip | previous level | coordinates],
//     group g (32 channels, 3 MFMAs):  lane (n, kg) element j  <->  q = 32 g + 4 j + kg       (hi*hi, lo*hi, hi*lo)
//     tail  t (<= 8 channels, 1 MFMA): lane (n, kg) element j  <->  q = 32 NG + 8 t + j, and kg picks the PRODUCT:
//                                      kg 0: ah*bh   kg 1: al*bh   kg 2: ah*bl   kg 3: zero (B side)
// so cin = 34 (HyperSeg-M level 4) costs 4 MFMAs of 16 cycles per tile and chunk instead of 9 of 32.  pw3's K is the
// chunk's 16 hidden channels x 3 products = 48 -> two MFMAs:  [ah*bh(0..7) | ah*bh(8..15) | al*bh(0..7) | al*bh(8..15)]
// and [ah*bl(0..7) | ah*bl(8..15) | 0 | 0] (the zero again on the B side, read from a zeroed LDS block).
//
// LDS.  h1 stays f32 (consumed by the depthwise stage on the VALU; channels interleaved in pairs, see SplitH1).  h2 is written by the depthwise stage as
// two f16 planes per hidden channel ([piece][channel][pixel], 16-byte stores) and read by pw3 with ds_read_b64_tr_b16,
// which hands each lane 4 consecutive CHANNELS of its pixel -- the transpose the B fragment needs, in the load.  The
// chunk's weights are split ONCE per workgroup by the thread that stages them (thread = (row, 16-lane segment): the row
// maximum is a DPP reduction inside the segment, no barrier) and stored in fragment order, so a wave's A fragment is
// one ds_read_b128 per quad.
#include "hs_ir_common.h"
#include <cstdlib>
#include <cstring>

namespace hs {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
typedef __fp16 half4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define HS_LDS_H4(p) ((__attribute__((address_space(3))) half4v*)(p))

// K layout of pw1 for `CIN` input channels
template <int CIN> struct SplitK {
    static constexpr int NG = CIN / 32 + ((CIN % 32) > 16 ? 1 : 0);       // 32-channel groups (the last may be zero-padded)
    static constexpr int REST = CIN > 32 * NG ? CIN - 32 * NG : 0;        // channels left for the tails (<= 16)
    static constexpr int NT = (REST + 7) / 8;
    static constexpr int NQ = 2 * NG + NT;                                // quads per tile (B) / per owner (A)
    static constexpr int NSRC = NG + NT;                                  // 8-value source sets per lane and tile
};

// Operand buffer of one chunk (double-buffered).  Halfs first, then float-indexed sections.
template <int CIN, int COUT, int MODE> struct SplitOps {
    using K = SplitK<CIN>;
    static constexpr int NS1 = MODE == 0 ? 1 : 9;            // owners: the region's patch | the 3 x 3 patches around it
    static constexpr int MT3 = (COUT + 15) / 16, CP = 16 * MT3;
    static constexpr int GQ_H = 16 * 4 * 8;                  // a group quad: [16 rows][4 kg][8]
    static constexpr int TQ_H = 2 * 16 * 8;                  // a tail: [hi | lo][16 rows][8]
    static constexpr int SLOT_H = 2 * K::NG * GQ_H + K::NT * TQ_H;       // [hi groups | lo groups | tails]
    static constexpr int W1_H = NS1 * SLOT_H;
    static constexpr int W3_H = 2 * MT3 * 256;               // [piece][m][16 rows][16 hidden channels]
    static constexpr int F_WE1 = (W1_H + W3_H) / 2;          // 1 / scale of every staged W1 row
    static constexpr int F_KD = F_WE1 + NS1 * 16;            // depthwise taps [8 channel pairs][9][2]
    static constexpr int F_WE3 = F_KD + 144;                 // 1 / scale of the W3 rows
    static constexpr int F_ZERO = F_WE3 + 2 * CP;            // (inv | ratio to the previous chunk) x CP; then 128 zero bytes
    static constexpr int OPF = (F_ZERO + 32 + 3) & ~3;       // floats per buffer
};

// h2 as f16: plane p (hidden channel of the chunk) of a piece at p * PS + (p >> 3) * 64 halfs.  PS = pixels + 16 puts
// the 4 planes one transpose read touches 8 banks apart, the + 64 halfs shifts channels 8..15 onto the other 32 banks:
// the two 16-lane groups that are served together (channels 0..3 / 8..11, then 4..7 / 12..15) never collide.
template <int REG> struct SplitH2 {
    static constexpr int PS = REG * REG + 16;
    static constexpr int PIECE_H = 16 * PS + 64;
    static constexpr int HALFS = 2 * PIECE_H;
    static __host__ __device__ constexpr int plane(int p) { return p * PS + (p >> 3) * 64; }
};

// h1 as f32 with hidden channels interleaved in PAIRS: plane cp (channels 2 cp, 2 cp + 1 of the chunk) holds
// [halo row][halo column][2].  The depthwise thread works on a channel pair with v_pk_fma_f32 (both halves of a 64-bit
// register pair carry the same pixel of the two channels, so every tap offset is an aligned pair), and pw1 stores its
// four D rows as two 8-byte writes.  Row stride 2 * (REG + 2) floats = an odd number of 16-byte granules: the 16 rows a
// ds_read_b128 serves together fall on distinct bank groups.
template <int REG> struct SplitH1 {
    static constexpr int HW = REG + 2;
    static constexpr int RSP = 2 * HW;                       // floats per halo row
    static constexpr int PSP = HW * RSP + 8;                 // floats per channel-pair plane; DUMMY = first pad float
    static constexpr int DUMMY = HW * RSP;
    static constexpr int FLOATS = 8 * PSP;
};

constexpr float H2_SCALE = 4096.0f;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// max over the 16 lanes of a DPP row (every lane gets it)
__device__ __forceinline__ float rowmax16(float v) {
    int x = __float_as_int(v);
    auto step = [&](int ctrl_moved) { x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(ctrl_moved))); };
    step(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    step(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    step(__builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));     // row_half_mirror
    step(__builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false));     // row_mirror
    return __int_as_float(x);
}
// biased exponent eb of a non-negative m (m < 2^(eb - 126)), clamped so that both 2^(141 - eb) and 2^(eb - 141) are
// normal floats; m * 2^(141 - eb) < 2^15
__device__ __forceinline__ int exp_of(float m) { return min(max(__float_as_int(m) >> 23, 27), 254); }
__device__ __forceinline__ float scale_of(int eb) { return __int_as_float((268 - eb) << 23); }
__device__ __forceinline__ float inv_scale_of(int eb) { return __int_as_float((eb - 14) << 23); }

__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

template <int CIN, int CSKIP, int COUT, int REG, int MODE, int NW, int WPS>
__global__ __launch_bounds__(64 * NW, WPS)
void patch_ir_split_kernel(IrFusedArgs a) {
    // NW waves per workgroup.  4 everywhere: 8 (half the tiles per wave, 4 waves per SIMD) measured SLOWER, 31.4 vs 28.6 us
    // at HyperSeg-M level 4 -- the kernel is bound by VALU issue slots, and per-thread overheads double with the threads.
    constexpr int NTHR = 64 * NW;
    using G = IrfGeom<REG>;
    using TM = IrTiles<REG, MODE, REG>;
    using K = SplitK<CIN>;
    using OP = SplitOps<CIN, COUT, MODE>;
    using H2 = SplitH2<REG>;
    using H1 = SplitH1<REG>;
    constexpr int CPREV = CIN - 2 - CSKIP;
    static_assert(CPREV > 0 && CSKIP > 0, "fused form: coords + skip + previous level");
    constexpr int NG = K::NG, NT = K::NT, NQ = K::NQ, NSRC = K::NSRC;
    constexpr int MT3 = OP::MT3, CP = OP::CP, NS1 = OP::NS1;
    constexpr int NT1 = TM::NT1, NT3 = TM::NT3;
    constexpr int J1 = (NT1 + NW - 1) / NW, J3 = NT3 / NW;
    static_assert(NT3 % NW == 0, "pixel tiles split evenly over the waves");
    constexpr bool P1_UNI = (MODE == 0);
    constexpr int SIN = P1_UNI ? 0 : 4;                    // slot of the patch the region lies in
    constexpr int PXT = 8 * REG * REG / NTHR;              // depthwise: thread = (channel pair, row, run of PXT pixels)
    constexpr int NI = (CIN + 15) / 16;                    // staged W1 elements per thread and owner

    const int hid = a.hid;
    const int HP = (hid + 15) & ~15;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds;                                       // [8 channel pairs][halo][2] f32
    _Float16* h2 = reinterpret_cast<_Float16*>(lds + H1::FLOATS);       // [2][16 planes] f16
    float* bnl = lds + H1::FLOATS + H2::HALFS / 2;       // [s1 | b1 | s2 | b2] x HP, [s3 | b3] x CP, zero-padded
    float* opsb = bnl + 4 * HP + 2 * CP;                   // [2][OPF]
    float* pl = lds;                                       // prologue only: [CPREV][PWIN^2], aliases h1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lk = lane >> 4;
    int blk = blockIdx.x;                                  // XCD-contiguous region ranges (see the exact kernel)
    if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
    const int rx = blk % a.regs_x; blk /= a.regs_x;
    const int ry = blk % a.regs_y;
    const int b = blk / a.regs_y;
    const int y0 = ry * REG, x0 = rx * REG;
    const int H = a.in.H, W = a.in.W;
    const unsigned plane = (unsigned)H * (unsigned)W;
    const float* __restrict__ bank = a.bank;
    const unsigned off_kd = (unsigned)CIN * hid, off_w3 = off_kd + 9u * hid;
    const int i0 = y0 / a.ph, j0 = x0 / a.pw;              // the patch the region lies in
    auto slot_ob = [&](int slot) {                         // bank offset of owner `slot`
        int pi = i0, pj = j0;
        if constexpr (!P1_UNI) {
            pi = min(max(i0 + slot / 3 - 1, 0), a.fh - 1); pj = min(max(j0 + slot % 3 - 1, 0), a.fw - 1);
        }
        return (unsigned)((b * a.fh + pi) * a.fw + pj) * (unsigned)a.ld;
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------
    // (0) both operand buffers start as zeros: K slots past cin, the unused product slots and the zero block stay so
    {
        float4* z = reinterpret_cast<float4*>(opsb);
        for (int e = tid; e < 2 * OP::OPF / 4; e += NTHR) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // (1) low-res window of the previous level
    const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;
    constexpr int PQ = (CPREV * G::PPL + NTHR - 1) / NTHR;
    float preg[PQ];
    {
        const float* __restrict__ pvb = a.in.prev + (size_t)b * CPREV * a.in.Hp * a.in.Wp;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int e = min(tid + q * NTHR, CPREV * G::PPL - 1);
            const int c = e / G::PPL, rq = e - c * G::PPL;
            const int r = rq / G::PWIN, qq = rq - r * G::PWIN;
            const int yy = min(max(ly0 + r, 0), a.in.Hp - 1), xx = min(max(lx0 + qq, 0), a.in.Wp - 1);
            preg[q] = pvb[(unsigned)((c * a.in.Hp + yy) * a.in.Wp + xx)];
        }
    }
    // (2) positions of this wave's pw1 tiles; gathers of the skip features in K order.  Source set s < NG is group s
    //     (element j <-> q = 32 s + 4 j + lk), set NG + t is tail t (q = 32 NG + 8 t + j, the same for every lane group).
    auto q_of = [&](int s, int j) { return s < NG ? 32 * s + 4 * j + lk : 32 * NG + 8 * (s - NG) + j; };
    auto q_lo = [](int s, int j) { return s < NG ? 32 * s + 4 * j : 32 * NG + 8 * (s - NG) + j; };
    auto q_hi = [](int s, int j) { return s < NG ? 32 * s + 4 * j + 3 : 32 * NG + 8 * (s - NG) + j; };
    float sv[J1][NSRC][8];
    int hoff[J1], pyx[J1], s1[P1_UNI ? 1 : J1];
    {
        const float* __restrict__ skb = a.in.skip + (size_t)b * CSKIP * plane;
#pragma unroll
        for (int jt = 0; jt < J1; ++jt) {
            const int t = wave + NW * jt;
            const bool tile_ok = (NT1 % NW == 0) || t < NT1;
            int u, v;
            const bool live = TM::halo(tile_ok ? t : 0, lrow, u, v) && tile_ok;
            const int yy = pad_index(y0 + u - 1, H, HS_PAD_REFLECT), xx = pad_index(x0 + v - 1, W, HS_PAD_REFLECT);
            hoff[jt] = live ? (u * H1::HW + v) * 2 : H1::DUMMY;
            pyx[jt] = live ? ((yy << 16) | xx) : -1;
            if constexpr (!P1_UNI) {
                int u0, v0;
                TM::halo(tile_ok ? t : 0, 0, u0, v0);
                const int oy = pad_index(y0 + u0 - 1, H, HS_PAD_REFLECT), ox = pad_index(x0 + v0 - 1, W, HS_PAD_REFLECT);
                s1[jt] = (oy / a.ph - i0 + 1) * 3 + (ox / a.pw - j0 + 1);
            }
            const int pix = yy * W + xx;
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float val = 0.0f;
                    if (q_lo(s, j) < CSKIP) {              // compile time: this element holds a skip feature for some lane
                        const int q = q_of(s, j);
                        if (live && q < CSKIP) val = skb[(unsigned)q * plane + (unsigned)pix];
                    }
                    sv[jt][s][j] = val;
                }
        }
        if constexpr (P1_UNI) s1[0] = 0;
    }
    // (3) folded BatchNorm rows; BN2 carries h2's static scale
    const int nbp = 4 * HP + 2 * CP;
    auto bn_val = [&](int e) -> float {
        const int seg = e < 4 * HP ? e / HP : 4 + (e - 4 * HP) / CP;
        const int idx = e < 4 * HP ? e - seg * HP : (e - 4 * HP) - (seg - 4) * CP;
        const float* src = seg == 0 ? a.s1 : seg == 1 ? a.b1 : seg == 2 ? a.s2 : seg == 3 ? a.b2 : seg == 4 ? a.s3 : a.b3;
        const int n = seg < 4 ? hid : COUT;
        const float v = src[min(idx, n - 1)] * (idx < n ? 1.0f : 0.0f);     // no branch around the load (it was waited for)
        return (seg == 2 || seg == 3) ? v * H2_SCALE : v;
    };
    float bnreg[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) bnreg[q] = bn_val(min(tid + q * NTHR, nbp - 1));

    // ---- filter-bank operands: staged per chunk.  A 16-lane segment handles one ROW of a matrix at a time (unit):
    //      W1 units = (owner, hidden row of the chunk), W3 units = output rows; group sg = tid / 16 takes units sg, sg + NGRP, ..
    constexpr int NGRP = NTHR / 16;
    constexpr int NU1 = NS1 * 16, NK1 = (NU1 + NGRP - 1) / NGRP;
    constexpr int NU3 = MT3 * 16, NK3 = (NU3 + NGRP - 1) / NGRP;
    const int sg = tid >> 4, seg = tid & 15;
    const unsigned nw_1 = off_w3 + (unsigned)hid * COUT - 1;             // last float of a patch's bank
    float sw[NK1 * NI], sk = 0.0f, s3[NK3];
    int e3cur[NK3];                                                      // running exponent of this segment's W3 rows
#pragma unroll
    for (int k = 0; k < NK3; ++k) e3cur[k] = 0;
    int wofs_hi[NI], wofs_lo[NI];                                        // where row 0's elements of this lane go (halfs)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = min(seg + 16 * i, CIN - 1);
        const int q = c >= 2 ? c - 2 : CIN - 2 + c;
        if (q < 32 * NG) {
            wofs_hi[i] = (q >> 5) * OP::GQ_H + (q & 3) * 8 + ((q >> 2) & 7);
            wofs_lo[i] = wofs_hi[i] + NG * OP::GQ_H;
        } else {
            const int r = q - 32 * NG;
            wofs_hi[i] = 2 * NG * OP::GQ_H + (r >> 3) * OP::TQ_H + (r & 7);
            wofs_lo[i] = wofs_hi[i] + 128;
        }
    }
    // row `row` of a quad: + row * 32 halfs in a group quad, + row * 8 in a tail
    const int wrow_g = 32, wrow_t = 8;
    auto ops_load = [&](int h0) {
        const unsigned ob_in = slot_ob(SIN);
#pragma unroll
        for (int k = 0; k < NK1; ++k) {
            const int unit = min(sg + k * NGRP, NU1 - 1);
            const unsigned ob = slot_ob(unit >> 4) + (unsigned)((h0 + (unit & 15)) * CIN);
#pragma unroll
            for (int i = 0; i < NI; ++i) sw[k * NI + i] = bank[ob + (unsigned)min(seg + 16 * i, CIN - 1)];
        }
        sk = bank[ob_in + off_kd + (unsigned)(h0 * 9 + min(tid, 143))];
#pragma unroll
        for (int k = 0; k < NK3; ++k) {
            const int unit = min(sg + k * NGRP, NU3 - 1);
            s3[k] = bank[ob_in + min(off_w3 + (unsigned)(min(unit, COUT - 1) * hid + h0 + seg), nw_1)];
        }
    };
    auto ops_store = [&](float* __restrict__ buf) {
        _Float16* hb = reinterpret_cast<_Float16*>(buf);
#pragma unroll
        for (int k = 0; k < NK1; ++k) {
            const int unit = sg + k * NGRP;
            if (NU1 % NGRP == 0 || k < NK1 - 1 || unit < NU1) {          // uniform per 16-lane segment
                const int slot = unit >> 4, row = unit & 15;
                float m = 0.0f;
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (16 * i + 15 < CIN || seg + 16 * i < CIN) m = fmaxf(m, fabsf(sw[k * NI + i]));
                const int eb = exp_of(rowmax16(m));
                const float sc = scale_of(eb);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if (16 * i + 15 < CIN || seg + 16 * i < CIN) {
                        _Float16 hi, lo;
                        split_f16(sw[k * NI + i] * sc, hi, lo);
                        const int c = min(seg + 16 * i, CIN - 1);
                        const bool grp = (c >= 2 ? c - 2 : CIN - 2 + c) < 32 * NG;
                        const int ro = slot * OP::SLOT_H + row * (grp ? wrow_g : wrow_t);
                        hb[ro + wofs_hi[i]] = hi;
                        hb[ro + wofs_lo[i]] = lo;
                    }
                }
                if (seg == 0) buf[OP::F_WE1 + unit] = inv_scale_of(eb);
            }
        }
        if (tid < 144) buf[OP::F_KD + ((tid / 9) >> 1) * 18 + (tid % 9) * 2 + ((tid / 9) & 1)] = sk;   // [pair][tap][2]
#pragma unroll
        for (int k = 0; k < NK3; ++k) {
            const int unit = sg + k * NGRP;
            if (NU3 % NGRP == 0 || k < NK3 - 1 || unit < NU3) {
                const float rm = rowmax16(fabsf(s3[k]));
                const int eb = exp_of(rm);
                const int eprev = max(e3cur[k], 27);
                if (rm > 0.0f && eb > e3cur[k]) e3cur[k] = min(eb + 2, 254);   // grows only; 2 bits of headroom when it does
                const int ee = max(e3cur[k], 27);
                _Float16 hi, lo;
                split_f16(s3[k] * scale_of(ee), hi, lo);
                // [piece][m][16 rows][16]: unit = 16 m + row
                hb[OP::W1_H + unit * 16 + seg] = hi;
                hb[OP::W1_H + MT3 * 256 + unit * 16 + seg] = lo;
                if (seg == 0) {
                    buf[OP::F_WE3 + unit] = inv_scale_of(ee);
                    buf[OP::F_WE3 + CP + unit] = __int_as_float(max(127 + eprev - ee, 0) << 23);      // 2^(eprev - ee) <= 1: what an
                }                                                                             // accumulator in the old unit is worth
            }
        }
    };
    ops_load(0);
    // every load of the prologue is in flight: now the LDS stores
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int e = tid + q * NTHR;
        if (e < CPREV * G::PPL) pl[e] = preg[q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + q * NTHR;
        if (e < nbp) bnl[e] = bnreg[q];
    }
    for (int e = tid + 2 * NTHR; e < nbp; e += NTHR) bnl[e] = bn_val(e);
    __syncthreads();                                       // window + BN rows in LDS; operand buffers zeroed
    ops_store(opsb);                                       // chunk 0 (visible after the next barrier)

    // (4) the B fragments: assemble each tile's values (skip features, bilinear previous level, coordinates), scale the
    //     tile to 2^15 and split
    half8 bq[J1][NQ];                                      // [hi groups | lo groups | tails]
    float invb[J1];
#pragma unroll
    for (int jt = 0; jt < J1; ++jt) {
        const bool live = pyx[jt] >= 0;
        const int yy = live ? (pyx[jt] >> 16) : y0, xx = live ? (pyx[jt] & 0xffff) : x0;
        const Tap ty = bilinear_tap(yy, a.in.scale_y, a.in.Hp), tx = bilinear_tap(xx, a.in.scale_x, a.in.Wp);
        const int r0 = ty.i0 - ly0, r1 = ty.i1 - ly0, q0 = tx.i0 - lx0, q1 = tx.i1 - lx0;
        const int o00 = r0 * G::PWIN + q0, o01 = r0 * G::PWIN + q1, o10 = r1 * G::PWIN + q0, o11 = r1 * G::PWIN + q1;
        const float cx = linspace_pm1(xx, W, a.in.step_x), cy = linspace_pm1(yy, H, a.in.step_y);
        float m = 0.0f;
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float val = sv[jt][s][j];
                const int q = q_of(s, j);
                if (q_hi(s, j) >= CSKIP && q_lo(s, j) < CSKIP + CPREV) {           // may be a previous-level channel
                    const int c = min(max(q - CSKIP, 0), CPREV - 1);
                    const float* p = pl + c * G::PPL;
                    const float bl = ty.l0 * (tx.l0 * p[o00] + tx.l1 * p[o01]) + ty.l1 * (tx.l0 * p[o10] + tx.l1 * p[o11]);
                    if (q >= CSKIP && q < CSKIP + CPREV) val = bl;
                }
                if (q_hi(s, j) >= CIN - 2 && q_lo(s, j) < CIN) {                   // may be a coordinate
                    if (q == CIN - 2) val = cx;
                    if (q == CIN - 1) val = cy;
                }
                val = live ? val : 0.0f;
                sv[jt][s][j] = val;
                m = fmaxf(m, fabsf(val));
            }
        m = rowmax16(m);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const int eb = exp_of(m);
        const float sc = scale_of(eb);
        invb[jt] = inv_scale_of(eb);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 hi, lo;
                split_f16(sv[jt][g][j] * sc, hi, lo);
                bq[jt][g][j] = hi; bq[jt][NG + g][j] = lo;
            }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 hi, lo;
                split_f16(sv[jt][NG + t][j] * sc, hi, lo);
                // the lane group picks the product: 0, 1 -> b hi;  2 -> b lo;  3 -> nothing
                bq[jt][2 * NG + t][j] = lk < 2 ? hi : (lk == 2 ? lo : (_Float16)0.0f);
            }
    }
    // pw3: transpose-read addresses (halfs).  Lane i of a 16-lane group supplies the 4-pixel chunk (i & 3) of plane
    // (i >> 2) of its block and receives pixel i of the block's 4 planes.
    int trh[2], trl[2], trl_step;                          // B hi / B lo blocks of this lane group (rd = 0, 1)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int p = 8 * (lk & 1) + 4 * rd + (lrow >> 2);
        trh[rd] = H2::plane(p) + 4 * (lrow & 3);
        trl[rd] = lk < 2 ? H2::PIECE_H + trh[rd] : -1;
    }
    trl_step = lk < 2 ? 16 : 0;
    const int zero_h = (int)((opsb + OP::F_ZERO) - (lds + H1::FLOATS)) * 2;      // the zero block, relative to h2
    if (lk >= 2) { trl[0] = zero_h; trl[1] = zero_h; }
    unsigned yoff[J3];
#pragma unroll
    for (int jt = 0; jt < J3; ++jt) {
        int row, col;
        TM::pixel(wave + NW * jt, lrow, row, col);
        yoff[jt] = (unsigned)(4 * lk) * plane + (unsigned)((y0 + row) * W + x0 + col);
    }
    f32x4 acc3[MT3][J3];
#pragma unroll
    for (int m = 0; m < MT3; ++m)
#pragma unroll
        for (int jt = 0; jt < J3; ++jt) acc3[m][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                       // the window is dead; chunk 0's operands are staged
    const int dw_cp = tid / (NTHR / 8);                    // depthwise: channel pair, output row, first pixel of the run
    const int dw_u = (tid % (NTHR / 8)) % REG, dw_c0 = ((tid % (NTHR / 8)) / REG) * PXT;

    // ---- stages ------------------------------------------------------------------------------------------------------
    auto stage_pw1 = [&](int h0) {
        const float4 sc1 = *reinterpret_cast<const float4*>(bnl + h0 + 4 * lk);
        const float4 sh1 = *reinterpret_cast<const float4*>(bnl + HP + h0 + 4 * lk);
        const float* fb = opsb + ((h0 >> 4) & 1) * OP::OPF;
        const _Float16* hb = reinterpret_cast<const _Float16*>(fb);
        half8 aq[NQ];
        float4 ia = make_float4(0.f, 0.f, 0.f, 0.f);
        auto load_a = [&](int slot) {
            const _Float16* ab = hb + slot * OP::SLOT_H;
#pragma unroll
            for (int g = 0; g < 2 * NG; ++g) aq[g] = *reinterpret_cast<const half8*>(ab + g * OP::GQ_H + lrow * 32 + lk * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t)        // lane group 1 multiplies a lo; 3 meets zeros on the B side
                aq[2 * NG + t] = *reinterpret_cast<const half8*>(ab + 2 * NG * OP::GQ_H + t * OP::TQ_H + (lk == 1 ? 128 : 0) + lrow * 8);
            const float4 w = *reinterpret_cast<const float4*>(fb + OP::F_WE1 + slot * 16 + 4 * lk);
            ia = make_float4(w.x * sc1.x, w.y * sc1.y, w.z * sc1.z, w.w * sc1.w);
        };
        if constexpr (P1_UNI) load_a(0);
#pragma unroll
        for (int jt = 0; jt < J1; ++jt) {
            if ((NT1 % NW == 0) || jt < J1 - 1 || wave < NT1 - NW * (J1 - 1)) {      // uniform: the last round may be short
                if constexpr (!P1_UNI) load_a(s1[jt]);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[2 * NG + t], bq[jt][2 * NG + t], acc, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[NG + g], bq[jt][g], acc, 0, 0, 0);        // lo * hi
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[g], bq[jt][NG + g], acc, 0, 0, 0);        // hi * lo
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[g], bq[jt][g], acc, 0, 0, 0);             // hi * hi
                }
                const float ib = invb[jt];
                float* dst = h1 + (2 * lk) * H1::PSP + hoff[jt];
                const f32x2 v01 = {__builtin_amdgcn_fmed3f(fmaf(acc[0], ia.x * ib, sh1.x), 0.0f, 6.0f),
                                   __builtin_amdgcn_fmed3f(fmaf(acc[1], ia.y * ib, sh1.y), 0.0f, 6.0f)};
                const f32x2 v23 = {__builtin_amdgcn_fmed3f(fmaf(acc[2], ia.z * ib, sh1.z), 0.0f, 6.0f),
                                   __builtin_amdgcn_fmed3f(fmaf(acc[3], ia.w * ib, sh1.w), 0.0f, 6.0f)};
                *reinterpret_cast<f32x2*>(dst) = v01;
                *reinterpret_cast<f32x2*>(dst + H1::PSP) = v23;
            }
        }
    };
    // depthwise 3x3 + bn2 + relu6 (x 2^12) of one chunk: h1 -> the two f16 planes of h2
    auto stage_dw = [&](int h0) {
        if (h0 + 16 < HP) ops_load(h0 + 16);               // next chunk's operands: in flight during this stage
        const f32x2* kb = reinterpret_cast<const f32x2*>(opsb + ((h0 >> 4) & 1) * OP::OPF + OP::F_KD) + dw_cp * 9;
        const f32x2 sc2 = *reinterpret_cast<const f32x2*>(bnl + 2 * HP + h0 + 2 * dw_cp);
        const f32x2 sh2 = *reinterpret_cast<const f32x2*>(bnl + 3 * HP + h0 + 2 * dw_cp);
        const float* hp = h1 + dw_cp * H1::PSP + dw_u * H1::RSP + dw_c0 * 2;
        f32x2 o[PXT];
#pragma unroll
        for (int v = 0; v < PXT; ++v) o[v] = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            f32x2 rowv[PXT + 2], k3[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) k3[kx] = kb[ky * 3 + kx];
#pragma unroll
            for (int q = 0; q < (PXT + 2) / 2; ++q) {
                const float4 v4 = *reinterpret_cast<const float4*>(hp + ky * H1::RSP + 4 * q);
                rowv[2 * q] = f32x2{v4.x, v4.y}; rowv[2 * q + 1] = f32x2{v4.z, v4.w};
            }
#pragma unroll
            for (int v = 0; v < PXT; ++v)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) o[v] = __builtin_elementwise_fma(k3[kx], rowv[v + kx], o[v]);
        }
        using hv = __attribute__((ext_vector_type(PXT))) _Float16;
        hv hi[2], lo[2];
#pragma unroll
        for (int v = 0; v < PXT; ++v) {
            const f32x2 t = __builtin_elementwise_fma(o[v], sc2, sh2);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                _Float16 vh, vl;
                split_f16(__builtin_amdgcn_fmed3f(t[c], 0.0f, 6.0f * H2_SCALE), vh, vl);
                hi[c][v] = vh; lo[c][v] = vl;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            _Float16* dst = h2 + H2::plane(2 * dw_cp + c) + dw_u * REG + dw_c0;
            *reinterpret_cast<hv*>(dst) = hi[c];
            *reinterpret_cast<hv*>(dst + H2::PIECE_H) = lo[c];
        }
        if (h0 + 16 < HP) ops_store(opsb + (((h0 >> 4) + 1) & 1) * OP::OPF);       // visible after the barrier that follows
    };
    // pw3: acc3 += W3[:, chunk] . h2
    auto stage_pw3 = [&](int h0) {
        const float* fb = opsb + ((h0 >> 4) & 1) * OP::OPF;
        const _Float16* hb = reinterpret_cast<const _Float16*>(fb) + OP::W1_H;
        half8 a3[MT3];
#pragma unroll
        for (int m = 0; m < MT3; ++m) {
            // lane groups 0, 1: a hi (channels 0..7 / 8..15);  2, 3: a lo
            a3[m] = *reinterpret_cast<const half8*>(hb + ((lk >> 1) * MT3 + m) * 256 + lrow * 16 + (lk & 1) * 8);
            const float4 rt = *reinterpret_cast<const float4*>(fb + OP::F_WE3 + CP + 16 * m + 4 * lk);
            if (__any(rt.x != 1.0f || rt.y != 1.0f || rt.z != 1.0f || rt.w != 1.0f)) {     // a row's exponent grew (rare)
#pragma unroll
                for (int jt = 0; jt < J3; ++jt) {
                    acc3[m][jt][0] *= rt.x; acc3[m][jt][1] *= rt.y; acc3[m][jt][2] *= rt.z; acc3[m][jt][3] *= rt.w;
                }
            }
        }
#pragma unroll
        for (int jt = 0; jt < J3; ++jt) {
            const int t16 = (wave + NW * jt) * 16;
            half8 bh, bl;
            {
                const half4v x0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(HS_LDS_H4(h2 + trh[0] + t16));
                const half4v x1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(HS_LDS_H4(h2 + trh[1] + t16));
                const half4v y0v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(HS_LDS_H4(h2 + trl[0] + (wave + NW * jt) * trl_step));
                const half4v y1v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(HS_LDS_H4(h2 + trl[1] + (wave + NW * jt) * trl_step));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bh[e] = (_Float16)x0[e]; bh[4 + e] = (_Float16)x1[e];
                    bl[e] = (_Float16)y0v[e]; bl[4 + e] = (_Float16)y1v[e];
                }
            }
#pragma unroll
            for (int m = 0; m < MT3; ++m) {
                acc3[m][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3[m], bl, acc3[m][jt], 0, 0, 0);
                acc3[m][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3[m], bh, acc3[m][jt], 0, 0, 0);
            }
        }
    };

    // ---- chunk loop:  dw(c) | barrier | pw3(c), pw1(c+1) | barrier ---------------------------------------------------
    stage_pw1(0);
    __syncthreads();
    for (int h0 = 0; h0 < HP; h0 += 16) {
        stage_dw(h0);
        __syncthreads();
        stage_pw3(h0);
        if (h0 + 16 < HP) {
            stage_pw1(h0 + 16);
            __syncthreads();
        }
    }

    // ---- epilogue: bn3 + store ---------------------------------------------------------------------------------------
    float* __restrict__ yb = a.y + (size_t)b * COUT * plane;
    const float* we3_last = opsb + (((HP >> 4) - 1) & 1) * OP::OPF + OP::F_WE3;
#pragma unroll
    for (int m = 0; m < MT3; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = m * 16 + 4 * lk + r;
            if (o < COUT) {
                const float sc = bnl[4 * HP + o] * (we3_last[o] * (1.0f / H2_SCALE)), sh = bnl[4 * HP + CP + o];
                float* __restrict__ yo = yb + (size_t)(m * 16 + r) * plane;
#pragma unroll
                for (int jt = 0; jt < J3; ++jt) yo[yoff[jt]] = fmaf(acc3[m][jt][r], sc, sh);
            }
        }
    }
}

template <int CIN, int CSKIP, int COUT, int REG, int MODE, int NW, int WPS>
static int launch_irs(IrFusedArgs& a, hipStream_t stream) {
    using G = IrfGeom<REG>;
    using OP = SplitOps<CIN, COUT, MODE>;
    using H1 = SplitH1<REG>;
    if (a.in.H % REG != 0 || a.in.W % REG != 0) return 1;
    a.regs_y = a.in.H / REG; a.regs_x = a.in.W / REG;
    const size_t hp = ((size_t)a.hid + 15) & ~(size_t)15;
    const size_t bn_floats = 4 * hp + 2 * OP::CP;
    const size_t lds = ((size_t)H1::FLOATS + SplitH2<REG>::HALFS / 2 + bn_floats + 2 * OP::OPF) * sizeof(float);
    // the prologue window aliases h1 and must end before h2
    static_assert((CIN - 2 - CSKIP) * G::PPL <= H1::FLOATS + SplitH2<REG>::HALFS / 2, "previous-level window fits");
    // operand rows past the last hidden channel are read unmasked: they must stay inside the patch's bank
    if (16 * CIN + 4 > a.hid * (9 + COUT) || 144 > a.hid * COUT) return 1;
    if (lds > 160 * 1024) return HS_ERR_LDS;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};
        const int e = allow_full_lds((const void*)patch_ir_split_kernel<CIN, CSKIP, COUT, REG, MODE, NW, WPS>, done);
        if (e != HS_OK) return e;
    }
    const long blocks = (long)a.in.B * a.regs_y * a.regs_x;
    hipLaunchKernelGGL((patch_ir_split_kernel<CIN, CSKIP, COUT, REG, MODE, NW, WPS>), dim3((unsigned)blocks), dim3(64 * NW),
                       lds, stream, a);
    return launch_status();
}

// Math mode: 0 = auto (the split form where it is the FASTER one: 16x16 regions of Op C), 1 = exact f32 everywhere,
// 2 = split wherever it is instantiated.  HS_IR_MATH=auto|f32|split overrides the default at load time.
// Measured on MI355X (profiles/round2_ir_split_vs_f32.txt): HyperSeg-M level 4 28.6 us split vs 32.5 us exact; the 8x8
// region shapes (level 3: 11.4 vs 9.9 us) and Op D (HyperSeg-L level 5: 1088 vs 800 us -- 9 owner patches to split per
// chunk, only 2 chunks to amortise the operand set-up over) are faster exact, so auto leaves them there.
static std::atomic<int> g_ir_math{-1};
static int ir_math() {
    int m = g_ir_math.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = std::getenv("HS_IR_MATH");
        m = (e && std::strcmp(e, "f32") == 0) ? 1 : (e && std::strcmp(e, "split") == 0) ? 2 : 0;
        g_ir_math.store(m, std::memory_order_relaxed);
    }
    return m;
}

int try_launch_ir_split(int mode, IrFusedArgs& a, int cin, int c_skip, int c_out, hipStream_t stream) {
    const int math = ir_math();
    if (math == 1) return 1;
    const int p = a.ph;
#define HS_IRS_CASE(CI, CS, CO, REG, MODE) HS_IRS_CASE_W(CI, CS, CO, REG, MODE, 2)
#define HS_IRS_CASE_W(CI, CS, CO, REG, MODE, WPS) \
    if (cin == CI && c_skip == CS && c_out == CO) return launch_irs<CI, CS, CO, REG, MODE, 4, WPS>(a, stream);
    if (mode == 0) {
        if (p % 16 == 0) {
            HS_IRS_CASE(34, 16, 19, 16, 0)   // HyperSeg-M level 4
            HS_IRS_CASE(26, 16, 19, 16, 0)   // HyperSeg-S level 4 (a third workgroup per CU -- 168 VGPRs, 30 spilled -- measured
                                             // slower at 1536x768: 56.8 vs ~50 us)
            HS_IRS_CASE(22, 4, 12, 16, 0)    // CamVid-S level 4
            HS_IRS_CASE(24, 6, 16, 16, 0)
        }
        if (math == 2 && p % 8 == 0) {
            HS_IRS_CASE(24, 6, 16, 8, 0)     // HyperSeg-M / CamVid-S level 3
            HS_IRS_CASE(14, 4, 8, 8, 0)      // HyperSeg-S level 3
            HS_IRS_CASE(34, 16, 19, 8, 0)
            HS_IRS_CASE(22, 4, 12, 8, 0)
        }
        return 1;
    }
    if (math == 2 && p % 16 == 0) {          // Op D regions inside one patch (HyperSeg-L levels 4, 5)
        HS_IRS_CASE(16, 6, 6, 16, 1)
        HS_IRS_CASE(11, 3, 21, 16, 1)
    }
#undef HS_IRS_CASE
#undef HS_IRS_CASE_W
    return 1;
}

}  // namespace hs

using namespace hs;

// Math mode of the fused inverted-residual levels (include/hyperseg_hip.h, hs_ir_math).
extern "C" int hs_set_ir_math(int32_t mode) {
    if (mode < 0 || mode > 2) return HS_ERR_BAD_ARG;
    g_ir_math.store(mode, std::memory_order_relaxed);
    return HS_OK;
}
extern "C" int hs_get_ir_math(void) { return ir_math(); }

// The inverted-residual decoder levels on the f32 matrix cores, one launch per level:
//   Op C  (MODE 0)  hyperseg_v1_0.py:328-376 / hyperseg_v1_0_unify.py:330-389 -- per patch, on its own reflect halo tile
//   Op D  (MODE 1)  hyperseg_v0_1.py:205-237 -- three image-level patch convolutions (pw1 -> depthwise 3x3 with reflect
//                   padding of the HIDDEN activation -> pw3): the ring around a region is pw1 of the neighbouring
//                   patches' inputs with the NEIGHBOURS' weights, recomputed here instead of being exchanged through HBM
// and the dominant kernel of the decoder (85 % of the FLOPs at HyperSeg-M).
//
// Per region the block is two small dense GEMMs around a depthwise 3x3,
//     pw1  [hid x cin] . [cin x halo positions]          pw3  [cout x hid] . [hid x pixels]
// on v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fma chain).  One workgroup (4 waves) = one REG x REG region; the
// tile maps (which 16 positions share a filter bank) are in hs_ir_tiles.h.
//
//   prologue   the stage input cat(coords, skip, bilinear2x(prev)) is built DIRECTLY in the layout of the matrix
//              cores' B operand (lane = (position, channel mod 4)): skip features are gathered from HBM straight into
//              the fragment registers, the previous level's low-resolution window goes through LDS once and is
//              sampled with the 4-tap stencil, coordinates are analytic.  The fragments stay in registers for the
//              whole kernel; all HBM loads of the workgroup are in flight together.
//   pw1        per owned position tile ceil(cin/4) MFMAs; D -> BN1 -> ReLU6 -> LDS h1[16][halo]
//   dw         thread = (hidden channel, output row [, half row]): halo rows as ds_read_b128, 9 per-lane taps (of the
//              patch that owns the OUTPUT pixel), -> BN2 -> ReLU6 -> LDS h2[16][pixels]
//   pw3        per owned pixel tile 4 MFMAs per 16 output channels, accumulators persistent across hidden chunks
//   epilogue   BN3, row runs to HBM.
// Hidden channels go through in chunks of 16:   dw(c) | barrier | pw3(c), pw1(c+1) | barrier.
//
// What bounds it (measured on MI355X, tools/ubench/mfma_valu.hip + profiles/round2_*): the f32-input MFMA executes on the
// SIMD's vector FMA lanes -- while one is in flight the SIMD issues no other VALU instruction, from this wave or from a
// co-resident one (38 cycles per MFMA alone, 86 with six v_fma behind it, 119 per wave with two waves per SIMD).  So a
// SIMD's time is the SUM of its MFMA cycles and its VALU/LDS issue cycles; software-pipelining pw1 under the depthwise
// stage (built and measured in round 2) bought nothing.  The levers that remain are instruction count and memory
// latency, hence: no masks or selects behind loads (operand rows past the last channel read finite neighbours and meet a
// zero BN scale/shift instead), 32-bit offsets from uniform bases, every prologue load in flight before the first wait,
// and the chunk's filter-bank operands copied into a double-buffered LDS block with COALESCED loads one stage ahead
// (STAGE below; regions that span several patches keep the direct per-lane operand loads).
// Hidden activations never leave the CU.  Op C on the f16 matrix cores (split products): hs_patch_irc.hip;
// one lane per pixel with a broadcast MFMA for HyperSeg-L's narrow levels: hs_patch_ir_px.hip.
#include "hs_ir_common.h"

namespace hs {

template <int CIN, int CSKIP, int COUT, int REG, int MODE, int PWR>
__global__ __launch_bounds__(IRF_THREADS, 2)
void patch_ir_fused_kernel(IrFusedArgs a) {
    using G = IrfGeom<REG>;
    using TM = IrTiles<REG, MODE, PWR>;
    constexpr int CPREV = CIN - 2 - CSKIP;
    static_assert(CPREV > 0 && CSKIP > 0, "fused form: coords + skip + previous level");
    constexpr int KS1 = (CIN + 3) / 4;
    constexpr int MT3 = (COUT + 15) / 16;
    constexpr int CP = 16 * MT3;                                // output channels, padded
    constexpr int NT1 = TM::NT1, NT3 = TM::NT3;
    constexpr int J1 = (NT1 + 3) / 4, J3 = NT3 / 4;
    static_assert(NT3 % 4 == 0, "pixel tiles split evenly over the 4 waves");
    constexpr bool P1_UNI = (MODE == 0);        // every pw1 tile uses the region's own patch
    constexpr bool IN_UNI = (PWR == REG);       // every pixel of the region belongs to one patch
    // depthwise stage: thread = (hidden channel of the chunk, output row, DWW-pixel run of that row): all 256 threads busy
    constexpr int DWW = REG * REG / 16;                    // 16 (a whole row) or 4 (half a row of an 8x8 region)
    constexpr int NRD = (DWW + 2 + 3) / 4;                 // 16-byte reads per halo row
    constexpr int NKD = (IN_UNI || DWW <= PWR) ? 1 : DWW / PWR;   // tap sets per thread (one per patch under its run)
    constexpr int SEGS = REG / PWR;                        // patches per region edge
    constexpr int NA3 = IN_UNI ? 1 : J3;
    // pw1 A fragments: ONE set when the region lies in one patch; with per-tile owners (Op D) a set per tile, all fetched
    // a stage ahead when that fits the register budget, otherwise fetched one tile ahead inside the stage
    constexpr bool A1_ALL = !P1_UNI && J1 * KS1 <= 32;
    constexpr int NA1 = P1_UNI ? 1 : (A1_ALL ? J1 : 1);
    // STAGE: the chunk's filter-bank operands go through LDS -- per chunk the workgroup copies W1[16 rows] (of its own patch
    // and, for Op D, of the 8 patches around it), the 16 x 9 depthwise taps and W3[:, 16 columns] with COALESCED loads
    // (4-8 per thread, double-buffered one chunk ahead) and every wave reads its fragments from there.  The direct form
    // (each lane fetching its own fragment elements from the bank: ~26 scattered 4-byte loads per lane and chunk, ~400
    // 16..64-byte requests per wave) was bound by the CU's address path: profiles/round2_ir_fused_phase_cycles_*.
    // Kept for the Op D shapes whose regions span several patches (the 16 neighbour slots would not fit the LDS budget).
    constexpr bool STAGE = P1_UNI || (MODE == 1 && SEGS == 1);
    constexpr int NS1 = P1_UNI ? 1 : 9;                     // W1 slots: own patch | 3 x 3 patches around the region's patch
    constexpr int SIN = P1_UNI ? 0 : 4;                     // slot of the patch the region lies in
    constexpr int W1F = 16 * CIN, KDF = 144, W3S = 17, W3F = COUT * W3S;
    constexpr int OPF = STAGE ? ((NS1 * W1F + KDF + W3F + 3) & ~3) : 0;     // floats per staged operand buffer
    constexpr int NLW = (NS1 * W1F + IRF_THREADS - 1) / IRF_THREADS, NL3 = (COUT * 16 + IRF_THREADS - 1) / IRF_THREADS;

    const int hid = a.hid;
    const int HP = (hid + 15) & ~15;                  // hidden channels, padded to whole chunks
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds;                                  // [16][H1P]
    float* h2 = lds + G::H1_FLOATS;                   // [16][H2S]
    // folded BatchNorm rows [s1 | b1 | s2 | b2] x HP, [s3 | b3] x CP, ZERO beyond the real channels: a hidden channel
    // past hid (the tail of the last chunk) gets scale = shift = 0, i.e. h1 = h2 = relu6(0) = 0 whatever its (finite)
    // operand rows held -- which is why no operand load below needs a mask
    float* bnl = h2 + G::H2_FLOATS;
    float* opsb = bnl + 4 * HP + 2 * CP;              // STAGE: [2][OPF] staged operands of the current / next chunk
    float* pl = lds;                                  // prologue only: [CPREV][PWIN*PWIN], aliases h1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lk = lane >> 4;
    // blocks are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, each with its own L2): give every XCD a
    // CONTIGUOUS range of regions, so that the halo rows / neighbour banks two adjacent regions share are fetched into
    // one L2 instead of two.  A pure relabelling (speed only; any placement is correct).
    int blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
    const int rx = blk % a.regs_x; blk /= a.regs_x;
    const int ry = blk % a.regs_y;
    const int b = blk / a.regs_y;
    const int y0 = ry * REG, x0 = rx * REG;
    const int H = a.in.H, W = a.in.W;
    const unsigned plane = (unsigned)H * (unsigned)W;
    // element offset of a patch's bank (host guarantees patches * ld < 2^30: 32-bit offsets from the uniform base)
    auto bank_of = [&](int yy, int xx) { return (unsigned)((b * a.fh + yy / a.ph) * a.fw + xx / a.pw) * (unsigned)a.ld; };
    const float* __restrict__ bank = a.bank;
    const unsigned off_kd = (unsigned)CIN * hid, off_w3 = off_kd + 9u * hid;

    // ---- prologue: issue every load, then wait ---------------------------------------------------------------------
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
            preg[q] = pvb[(unsigned)((c * a.in.Hp + yy) * a.in.Wp + xx)];
        }
    }
    // (2) positions of this wave's pw1 tiles and the skip-feature gathers, straight into the B fragments.  k-steps whose
    //     four channels are not all skip features (they also hold coordinates / previous-level channels, merged in (4))
    //     are gathered FIRST: loads return in order, so (4) then waits for those few only, not for the whole batch.
    float bf[J1][KS1];
    int hoff[J1];                 // LDS offset of this lane's position inside an h1 plane (DUMMY for a dead column)
    int pyx[J1];                  // image coordinates (yy << 16 | xx) of this lane's position, -1 for a dead column
    unsigned ob1[P1_UNI ? 1 : J1];   // bank offset of the patch that owns the tile
    int s1[P1_UNI ? 1 : J1];         // STAGE: its W1 slot (3 x 3 patches around the region's patch, row-major)
    const int i0 = y0 / a.ph, j0 = x0 / a.pw;       // the patch the region's first pixel lies in
    {
        const float* __restrict__ skb = a.in.skip + (size_t)b * CSKIP * plane;
        int spo[J1];              // element offset of channel (lk - 2) at this lane's position (lanes lk < 2 never use it)
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
                const int oy = pad_index(y0 + u0 - 1, H, HS_PAD_REFLECT), ox = pad_index(x0 + v0 - 1, W, HS_PAD_REFLECT);
                ob1[jt] = bank_of(oy, ox);
                s1[jt] = (oy / a.ph - i0 + 1) * 3 + (ox / a.pw - j0 + 1);
            }
            spo[jt] = yy * W + xx + (lk - 2) * (int)plane;
        }
        if constexpr (P1_UNI) { ob1[0] = bank_of(y0, x0); s1[0] = 0; }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int jt = 0; jt < J1; ++jt) {
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    const bool pure = ks * 4 >= 2 && ks * 4 + 3 < 2 + CSKIP;       // all four channels are skip features
                    if (pure == (pass == 1)) {
                        const int c = ks * 4 + lk;
                        float val = 0.0f;
                        if (pyx[jt] >= 0 && c >= 2 && c < 2 + CSKIP) val = (skb + (size_t)(ks * 4) * plane)[spo[jt]];
                        bf[jt][ks] = val;
                    }
                }
            }
        }
    }
    // (3) folded BatchNorm rows (two per thread cover 4*HP + 2*CP <= 512)
    const int nbp = 4 * HP + 2 * CP;
    auto bn_val = [&](int e) -> float {
        const int seg = e < 4 * HP ? e / HP : 4 + (e - 4 * HP) / CP;
        const int idx = e < 4 * HP ? e - seg * HP : (e - 4 * HP) - (seg - 4) * CP;
        const float* src = seg == 0 ? a.s1 : seg == 1 ? a.b1 : seg == 2 ? a.s2 : seg == 3 ? a.b2 : seg == 4 ? a.s3 : a.b3;
        const int n = seg < 4 ? hid : COUT;
        float v = 0.0f;
        if (idx < n) v = src[idx];
        return v;
    };
    float bnreg[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) bnreg[q] = bn_val(min(tid + q * IRF_THREADS, nbp - 1));

    // ---- filter-bank operands: plain loads from the bank, issued one stage ahead of their use ----------------------
    // pw1 A fragments  W1[h0 + lrow][4*ks + lk] of the tile's owner (see NA1 above; in the rolling form afa[0] is tile 0's
    // set of the next chunk)
    float afa[NA1][KS1];
    float k9[NKD][9];                  // depthwise taps of the owner of this thread's output pixels
    float a3[NA3][MT3][4];             // pw3 A fragments  W3[16*m + lrow][h0 + 4*ks + lk]
    const int dw_hh = tid >> 4;
    const int dw_u = (tid & 15) % REG, dw_c0 = ((tid & 15) / REG) * DWW;
    unsigned ob3[NA3], obd[NKD];
#pragma unroll
    for (int q = 0; q < NA3; ++q) {
        int row, col;
        TM::pixel(wave + 4 * q, 0, row, col);
        ob3[q] = IN_UNI ? bank_of(y0, x0) : bank_of(y0 + row, x0 + col);
    }
#pragma unroll
    for (int q = 0; q < NKD; ++q) obd[q] = IN_UNI ? bank_of(y0, x0) : bank_of(y0 + dw_u, x0 + dw_c0 + q * PWR);
    // lane parts of the operand offsets; rows past hid / k past cin read the finite floats that follow inside the bank
    // (host-checked), rows past cout are clamped
    const unsigned o1 = (unsigned)(lrow * CIN + lk);
    const unsigned okd = off_kd + (unsigned)(dw_hh * 9);
    unsigned o3[MT3];
#pragma unroll
    for (int m = 0; m < MT3; ++m) o3[m] = off_w3 + (unsigned)(min(m * 16 + lrow, COUT - 1) * hid + lk);

    auto load_a1 = [&](int h0, unsigned ob, float (&dst)[KS1]) {
        const float* __restrict__ wr = bank + (size_t)(h0 * CIN);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) dst[ks] = wr[ob + o1 + 4 * ks];
    };
    auto load_kd = [&](int h0) {
#pragma unroll
        for (int q = 0; q < NKD; ++q) {
            const float* __restrict__ kr = bank + (size_t)(h0 * 9);
#pragma unroll
            for (int e = 0; e < 9; ++e) k9[q][e] = kr[obd[q] + okd + e];
        }
    };
    auto load_a3 = [&](int h0, bool last) {
#pragma unroll
        for (int q = 0; q < NA3; ++q) {
#pragma unroll
            for (int m = 0; m < MT3; ++m) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    // the last chunk may end inside a k-step: stay inside the row (h2 is zero there)
                    const int hcol = last ? min(h0 + 4 * ks, hid - 1 - lk) : h0 + 4 * ks;
                    a3[q][m][ks] = bank[ob3[q] + o3[m] + (unsigned)hcol];
                }
            }
        }
    };
    const int nchunks = HP >> 4;
    // STAGE: coalesced copy of one chunk's operands, bank -> registers (ops_load) -> LDS buffer (ops_store)
    const unsigned nw_1 = off_w3 + (unsigned)hid * COUT - 1;               // last float of a patch's bank
    auto slot_ob = [&](int slot) {                                       // bank offset of W1 slot `slot`
        if constexpr (P1_UNI) return ob1[0];
        const int pi = min(max(i0 + slot / 3 - 1, 0), a.fh - 1), pj = min(max(j0 + slot % 3 - 1, 0), a.fw - 1);
        return (unsigned)((b * a.fh + pi) * a.fw + pj) * (unsigned)a.ld;
    };
    float sw[STAGE ? NLW : 1], sk = 0.0f, s3[STAGE ? NL3 : 1];
    auto ops_load = [&](int h0) {
        const unsigned ob_in = slot_ob(SIN);
#pragma unroll
        for (int q = 0; q < NLW; ++q) {
            const int e = min(tid + q * IRF_THREADS, NS1 * W1F - 1);
            const int slot = e / W1F, r = e - slot * W1F;
            sw[q] = bank[slot_ob(slot) + (unsigned)(h0 * CIN + r)];
        }
        sk = bank[ob_in + off_kd + (unsigned)(h0 * 9 + min(tid, KDF - 1))];
#pragma unroll
        for (int q = 0; q < NL3; ++q) {
            const int e = min(tid + q * IRF_THREADS, COUT * 16 - 1);
            s3[q] = bank[ob_in + min(off_w3 + (unsigned)((e >> 4) * hid + h0 + (e & 15)), nw_1)];
        }
    };
    auto ops_store = [&](float* __restrict__ buf) {
#pragma unroll
        for (int q = 0; q < NLW; ++q) {
            const int e = tid + q * IRF_THREADS;
            if (e < NS1 * W1F) buf[e] = sw[q];
        }
        if (tid < KDF) buf[NS1 * W1F + tid] = sk;
#pragma unroll
        for (int q = 0; q < NL3; ++q) {
            const int e = tid + q * IRF_THREADS;
            if (e < COUT * 16) buf[NS1 * W1F + KDF + (e >> 4) * W3S + (e & 15)] = s3[q];
        }
    };
    auto load_a1_stage = [&](int h0) {                // the sets a pw1 stage starts from
        if constexpr (P1_UNI) load_a1(h0, ob1[0], afa[0]);
        else if constexpr (A1_ALL) {
#pragma unroll
            for (int q = 0; q < J1; ++q) load_a1(h0, ob1[q], afa[q]);
        } else {
            load_a1(h0, ob1[0], afa[0]);
        }
    };
    if constexpr (STAGE) ops_load(0);
    else {
        load_a1_stage(0);
        load_kd(0);
        load_a3(0, nchunks == 1);
    }
    // every load of the prologue is in flight: now the LDS stores (they wait for the OLDEST loads only)
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int e = tid + q * IRF_THREADS;
        if (e < CPREV * G::PPL) pl[e] = preg[q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + q * IRF_THREADS;
        if (e < nbp) bnl[e] = bnreg[q];
    }
    for (int e = tid + 2 * IRF_THREADS; e < nbp; e += IRF_THREADS) bnl[e] = bn_val(e);     // hid > 112: not a decoder shape
    if constexpr (STAGE) ops_store(opsb);
    __syncthreads();                                   // window + BN rows (+ chunk 0's operands) are in LDS

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
            const bool pure = ks * 4 >= 2 && ks * 4 + 3 < 2 + CSKIP;   // gathered skip features only: nothing to merge
            if (!pure) {
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
    }
    int h2off[J3];                                     // LDS offset of this lane's pixel inside an h2 plane
    unsigned yoff[J3];                                 // element offset of output channel 4*lk at this lane's pixel
#pragma unroll
    for (int jt = 0; jt < J3; ++jt) {
        int row, col;
        TM::pixel(wave + 4 * jt, lrow, row, col);
        h2off[jt] = row * G::RS2 + col;
        yoff[jt] = (unsigned)(4 * lk) * plane + (unsigned)((y0 + row) * W + x0 + col);
    }
    f32x4 acc3[MT3][J3];
#pragma unroll
    for (int m = 0; m < MT3; ++m)
#pragma unroll
        for (int jt = 0; jt < J3; ++jt) acc3[m][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                   // the window is dead: h1 may overwrite it

    // ---- stages ------------------------------------------------------------------------------------------------------
    // pw1 of the chunk starting at h0 -> h1, then the loads of the FOLLOWING chunk's A fragments
    auto stage_pw1 = [&](int h0) {
        const float4 sc1 = *reinterpret_cast<const float4*>(bnl + h0 + 4 * lk);
        const float4 sh1 = *reinterpret_cast<const float4*>(bnl + HP + h0 + 4 * lk);
        const float* ob = opsb + ((h0 >> 4) & 1) * OPF;          // STAGE: this chunk's operand buffer
        float at[2][KS1];                              // rolling form: fragments of tiles jt and jt+1
        if constexpr (!STAGE && !P1_UNI && !A1_ALL) {
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) at[0][ks] = afa[0][ks];
        }
        if constexpr (STAGE && P1_UNI) {
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) afa[0][ks] = ob[o1 + 4 * ks];
        }
#pragma unroll
        for (int jt = 0; jt < J1; ++jt) {
            if constexpr (!STAGE && !P1_UNI && !A1_ALL) {
                if (jt + 1 < J1) load_a1(h0, ob1[jt + 1], at[(jt + 1) & 1]);
            }
            if ((NT1 % 4 == 0) || jt < J1 - 1 || wave < NT1 - 4 * (J1 - 1)) {      // uniform: the last round may be short
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    float av;
                    if constexpr (P1_UNI) av = afa[0][ks];
                    else if constexpr (STAGE) av = ob[s1[jt] * W1F + o1 + 4 * ks];
                    else av = A1_ALL ? afa[jt][ks] : at[jt & 1][ks];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf[jt][ks], acc, 0, 0, 0);
                }
                float* dst = h1 + (4 * lk) * G::H1P + hoff[jt];
                dst[0] = relu6_(fmaf(acc[0], sc1.x, sh1.x));
                dst[G::H1P] = relu6_(fmaf(acc[1], sc1.y, sh1.y));
                dst[2 * G::H1P] = relu6_(fmaf(acc[2], sc1.z, sh1.z));
                dst[3 * G::H1P] = relu6_(fmaf(acc[3], sc1.w, sh1.w));
            }
        }
        if constexpr (!STAGE) load_a1_stage(h0 + 16 < HP ? h0 + 16 : h0);
    };
    // depthwise 3x3 + bn2 + relu6 of one chunk: h1 -> h2; then the loads of the next chunk's taps
    auto stage_dw = [&](int h0) {
        if constexpr (STAGE) {
            if (h0 + 16 < HP) ops_load(h0 + 16);       // next chunk's operands: in flight during this stage
            const float* kb = opsb + ((h0 >> 4) & 1) * OPF + NS1 * W1F + dw_hh * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) k9[0][e] = kb[e];
        }
        const float sc2 = bnl[2 * HP + h0 + dw_hh], sh2 = bnl[3 * HP + h0 + dw_hh];
        const float* hp = h1 + dw_hh * G::H1P + dw_u * G::RS + dw_c0;
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
        if constexpr (STAGE) {
            if (h0 + 16 < HP) ops_store(opsb + (((h0 >> 4) + 1) & 1) * OPF);     // visible after the barrier that follows
        } else {
            load_kd(h0 + 16 < HP ? h0 + 16 : h0);
        }
    };
    // pw3: acc3 += W3[:, chunk] . h2; then the loads of the next chunk's A fragments
    auto stage_pw3 = [&](int h0) {
        if constexpr (STAGE) {
            const float* wb = opsb + ((h0 >> 4) & 1) * OPF + NS1 * W1F + KDF + lk;
#pragma unroll
            for (int m = 0; m < MT3; ++m)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a3[0][m][ks] = wb[min(m * 16 + lrow, COUT - 1) * W3S + 4 * ks];
        }
        float bv[4][J3];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int jt = 0; jt < J3; ++jt) bv[ks][jt] = h2[(ks * 4 + lk) * G::H2S + h2off[jt]];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (h0 + ks * 4 < hid) {                   // uniform: skip k-steps past the last hidden channel
#pragma unroll
                for (int jt = 0; jt < J3; ++jt)
#pragma unroll
                    for (int m = 0; m < MT3; ++m)
                        acc3[m][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[IN_UNI ? 0 : jt][m][ks], bv[ks][jt], acc3[m][jt], 0, 0, 0);
            }
        }
        if constexpr (!STAGE) {
            const int hn = h0 + 16 < HP ? h0 + 16 : h0;
            load_a3(hn, hn + 16 >= HP);
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
            stage_pw1(h0 + 16);                        // h1 is free: every wave finished dw(c) before the barrier above
            __syncthreads();                           // h1 ready for dw(c+1); h2 no longer read by pw3(c)
        }
    }

    // ---- epilogue: bn3 + store ---------------------------------------------------------------------------------------
    float* __restrict__ yb = a.y + (size_t)b * COUT * plane;
#pragma unroll
    for (int m = 0; m < MT3; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = m * 16 + 4 * lk + r;
            if (o < COUT) {
                const float sc = bnl[4 * HP + o], sh = bnl[4 * HP + CP + o];
                float* __restrict__ yo = yb + (size_t)(m * 16 + r) * plane;
#pragma unroll
                for (int jt = 0; jt < J3; ++jt) yo[yoff[jt]] = fmaf(acc3[m][jt][r], sc, sh);
            }
        }
    }
}

template <int CIN, int CSKIP, int COUT, int REG, int MODE, int PWR>
static int launch_irf(IrFusedArgs& a, hipStream_t stream) {
    using G = IrfGeom<REG>;
    if (a.in.H % REG != 0 || a.in.W % REG != 0) return 1;      // regions must tile the level
    a.regs_y = a.in.H / REG; a.regs_x = a.in.W / REG;
    const size_t hp = ((size_t)a.hid + 15) & ~(size_t)15;
    const size_t bn_floats = 4 * hp + 2 * 16 * ((COUT + 15) / 16);
    constexpr bool stage = MODE == 0 || REG == PWR;                            // as the kernel's STAGE
    constexpr size_t opf = stage ? (((MODE == 0 ? 1 : 9) * 16 * CIN + 144 + COUT * 17 + 3) & ~3) : 0;
    const size_t lds = ((size_t)G::H1_FLOATS + G::H2_FLOATS + bn_floats + 2 * opf) * sizeof(float);
    // operand rows past the last hidden channel / k past cin are read unmasked: they must stay inside the patch's bank
    if (16 * CIN + 4 > a.hid * (9 + COUT) || 144 > a.hid * COUT) return 1;
    if (lds > 160 * 1024) return HS_ERR_LDS;
    if (a.y == nullptr) return HS_OK;                         // route query (hs_patch_ir_route): the shape is covered, nothing is launched
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
    // 32-bit element offsets from uniform bases
    if ((size_t)in.B * fh * fw * (size_t)ld >= (1u << 30) || (size_t)in.H * in.W * (size_t)(c_out > c_skip ? c_out : c_skip) >= (1u << 30) ||
        (size_t)(cin - 2 - c_skip) * in.Hp * in.Wp >= (1u << 30)) return 1;
    if (a.ph != a.pw) return 1;
    const int p = a.ph;
    if (mode == 1) {   // the narrow Op D levels: one lane per pixel (hs_patch_ir_px.hip)
        const int e = try_launch_ir_px(a, cin, c_skip, c_out, stream);
        if (e != 1) return e;
    }
#define HS_IRF_CASE(CI, CS, CO, REG, MODE, PWR) \
    if (cin == CI && c_skip == CS && c_out == CO) return launch_irf<CI, CS, CO, REG, MODE, PWR>(a, stream);
    if (mode == 0) {
        if (p % 16 == 0) {
            HS_IRF_CASE(34, 16, 19, 16, 0, 16)   // HyperSeg-M level 4 (Cityscapes, 19 classes)
            HS_IRF_CASE(26, 16, 19, 16, 0, 16)   // HyperSeg-S level 4
            HS_IRF_CASE(22, 4, 12, 16, 0, 16)    // CamVid-S level 4 (12 classes)
            HS_IRF_CASE(24, 6, 16, 16, 0, 16)    // level-3 shapes on larger patches
            HS_IRF_CASE(22, 4, 16, 16, 0, 16)    // CamVid HyperSeg-L level 4 (configs/train/camvid_efficientnet_b1_hyperseg-l.py: 22 -> 44 -> 16)
            HS_IRF_CASE(21, 3, 12, 16, 0, 16)    // CamVid HyperSeg-L level 5 (21 -> 42 -> 12 on 32 x 32-pixel patches of the raw image)
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

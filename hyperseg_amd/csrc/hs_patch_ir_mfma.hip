// Op C on the f32 matrix cores: the decoder's dominant kernel (85 % of the decoder FLOPs at HyperSeg-M).
//
// Per patch the inverted residual is two small dense GEMMs around a depthwise 3x3:
//     pw1  [hid x cin] . [cin x (T+2)^2]      pw3  [cout x hid] . [hid x T^2]          (T = 16 or 8 pixel tile)
// i.e. exactly "a patch's (out_ch x in_ch) collapses to a dense tile".  v_mfma_f32_16x16x4_f32 is exact fp32
// (a k-ordered fma chain) at the fp32 peak rate, and -- unlike the VALU form, where the patch's weights are
// wave-uniform and have to trickle in through the scalar cache one row at a time -- its A operand lives
// distributed over the lanes, so the weights are ordinary prefetchable vector loads.
//
// One workgroup (4 waves) = one T x T tile of one patch; hidden channels are processed in chunks of 16:
//   prologue  thread = halo position: coords (analytic) + skip gather + 4-tap bilinear of the previous level,
//             all HBM loads of a column in flight together; column -> LDS T[c][pos]; each wave then lifts the
//             B fragments of ITS position tiles into registers for the whole kernel and the LDS is recycled.
//   pw1       per owned position tile: ceil(cin/4) MFMAs; D -> BN1 -> ReLU6 -> LDS h1[16][(T+2) x RS] (RS = row
//             stride padded to a multiple of 4 floats so that dw can read rows as 16-byte vectors)
//   dw        thread = (hidden channel, output row): 3 rows of T+2 activations as ds_read_b128, 9 per-lane
//             weights, T outputs -> BN2 -> ReLU6 -> LDS h2[16][T*T (+16 pad)]
//   pw3       per owned pixel tile: 4 MFMAs per 16 output channels, accumulators persistent across chunks
//   epilogue  BN3, 64-byte row runs to HBM.
// LDS: max(T tile, h1 + h2) = 48 KB at HyperSeg-M level 4 -> 3 workgroups / CU; hidden activations never leave the CU.
#include "hs_common.h"
#include <cstdlib>

#ifndef HS_IRM_ABLATE
#define HS_IRM_ABLATE 0     // dev-only timing ablations (tools/ablate_ir.py): 1 no prologue loads, 2 no pw1 MFMA,
#endif                      // 4 no dw math, 8 no pw3 MFMA, 16 no per-chunk operand loads.  0 = the product kernel.

#ifndef HS_IRM_TIMING
#define HS_IRM_TIMING 0     // dev-only: per-phase s_memtime stamps of wave 0 of every workgroup (tools/ablate_ir.py)
#endif
#if HS_IRM_TIMING
__device__ long long hs_irm_stamps[4096 * 32];
#define HS_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) hs_irm_stamps[blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int hs_debug_read_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hs_irm_stamps), sizeof(long long) * n);
}
#else
#define HS_STAMP(k) do {} while (0)
#endif

namespace hs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct IrMfmaArgs {
    StageIn in;
    int fh, fw, ph, pw;
    const float* __restrict__ bank;
    long ld;
    int hid;
    const float* __restrict__ s1; const float* __restrict__ b1;
    const float* __restrict__ s2; const float* __restrict__ b2;
    const float* __restrict__ s3; const float* __restrict__ b3;
    float* __restrict__ y;
    int tiles_y, tiles_x;
};

constexpr int IRM_THREADS = 256;

template <int TILE> struct IrmGeom {
    static constexpr int HW = TILE + 2;                         // halo tile edge
    static constexpr int NPOS = HW * HW;
    static constexpr int NT1 = (NPOS + 15) / 16;                // position tiles (pw1 N)
    static constexpr int NP1 = NT1 * 16;
    static constexpr int J1 = (NT1 + 3) / 4;                    // position tiles per wave
    static constexpr int RS = (HW + 3) & ~3;                    // h1 row stride (floats), 16-byte aligned rows
    // h1 plane per hidden channel, padded to == 4 (mod 8) floats: the two D-row groups (lk = 0, 1) of a half-wave
    // then write banks 16 apart (an unpadded 360 makes all four row groups collide: 4-way conflicts on every store)
    static constexpr int H1P = ((HW * RS + 7) & ~7) + 4;
    static constexpr int PW = TILE / 2 + 2;                     // low-res window edge of the previous level (exact 2x)
    static constexpr int PPL = PW * PW;
    static constexpr int NPIX = TILE * TILE;
    static constexpr int NT3 = NPIX / 16;                       // pixel tiles (pw3 N)
    static constexpr int J3 = (NT3 + 3) / 4;
    static constexpr int RS2 = TILE + 4;                        // h2 pixel-row stride: 16-byte rows, b128 stores of
                                                                // 8 consecutive rows fall on 8 distinct bank quads
    static constexpr int H2S = ((TILE * RS2 + 31) & ~31) + 16;  // h2 plane: == 16 (mod 32) -> conflict-free B reads
};

__device__ __forceinline__ float relu6m(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

template <int CIN, int CSKIP, int COUT, int TILE>
__global__ __launch_bounds__(IRM_THREADS)
void patch_ir_mfma_kernel(IrMfmaArgs a) {
    using G = IrmGeom<TILE>;
    constexpr int CPREV = CIN - 2 - CSKIP;
    constexpr int KS1 = (CIN + 3) / 4;
    constexpr int MT3 = (COUT + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* T = lds;                                   // [KS1*4][NP1]            (prologue only)
    float* h1 = lds;                                  // [16][H1P]               (aliases T)
    float* h2 = lds + 16 * G::H1P;                    // [16][H2S]
    constexpr int TILE_FLOATS = (KS1 * 4 * G::NP1 > 16 * G::H1P + 16 * G::H2S) ? KS1 * 4 * G::NP1
                                                                               : 16 * G::H1P + 16 * G::H2S;
    float* wl = lds + TILE_FLOATS;                    // the patch's whole filter bank (rows in reference order)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lk = lane >> 4;
    int blk = blockIdx.x;
    const int tx_i = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty_i = blk % a.tiles_y; blk /= a.tiles_y;
    const int patch = blk;
    const int j = patch % a.fw, i = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const int y0 = i * a.ph + ty_i * TILE, x0 = j * a.pw + tx_i * TILE;
    const int hid = a.hid;
    const float* __restrict__ wp = a.bank + (size_t)patch * a.ld;
    const int nw = CIN * hid + 9 * hid + hid * COUT;           // bank rows of this block
    const int nw4 = (nw + 3) >> 2;
    float* bnl = wl + nw4 * 4;                                  // [s1 | b1 | s2 | b2] x hid, [s3 | b3] x COUT
    const float* w1 = wl;                                       // LDS copies (filled below)
    const float* kd = wl + CIN * hid;
    const float* w3 = kd + 9 * hid;

    HS_STAMP(0);
    // low-res window of the previous level: rows [ly0, ly0+PW) x cols [lx0, lx0+PW) (indices clamped at the image
    // border) -- every bilinear tap of the halo tile, reflected positions included, falls inside it
    float* pl = bnl + ((4 * hid + 2 * COUT + 3) & ~3);         // [CPREV][PW*PW]
    const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;
    constexpr int PQ = (CPREV * G::PPL + IRM_THREADS - 1) / IRM_THREADS;
    float preg[PQ];
    {
        const float* __restrict__ pvb = a.in.prev + (size_t)b * CPREV * a.in.Hp * a.in.Wp;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int e = min(tid + q * IRM_THREADS, CPREV * G::PPL - 1);
            const int c = e / G::PPL, rq = e - c * G::PPL;
            const int r = rq / G::PW, qq = rq - r * G::PW;
            const int yy = min(max(ly0 + r, 0), a.in.Hp - 1), xx = min(max(lx0 + qq, 0), a.in.Wp - 1);
            preg[q] = pvb[((size_t)c * a.in.Hp + yy) * a.in.Wp + xx];
        }
    }
    // The bank is read from HBM exactly once, as 16-byte loads issued BEFORE the prologue's gathers, so the (cold)
    // HBM latency of weights and inputs is paid once and together; every later operand fetch is an LDS read.
    constexpr int WQ = 6;                                       // float4 per thread: covers banks up to 6144 floats
    float4 wreg[WQ];
    const bool w_vec = (((size_t)wp) & 15) == 0;
    {
        // unconditional loads from a clamped index keep wreg in registers (ld is a multiple of 4 >= rows, so the
        // last float4 of the bank stays inside this patch's row)
        const float4* __restrict__ src = reinterpret_cast<const float4*>(w_vec ? wp : a.bank);
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int e = tid + q * IRM_THREADS;
            wreg[q] = src[e < nw4 ? e : nw4 - 1];
        }
    }
    // the skip-feature gathers of BOTH position passes are issued now as well: together with the window and the bank
    // this is the workgroup's ONLY exposed HBM round trip
    constexpr int NPASS = (G::NP1 + IRM_THREADS - 1) / IRM_THREADS;
    float skv[NPASS][CSKIP];
    {
        const size_t plane = (size_t)a.in.H * a.in.W;
        const float* __restrict__ skb = a.in.skip + (size_t)b * CSKIP * plane;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int pos = tid + ps * IRM_THREADS;
            const bool live = pos < G::NPOS;
            const int pu = live ? pos / G::HW : 0, pv = live ? pos - pu * G::HW : 0;
            const int yy = pad_index(y0 + pu - 1, a.in.H, HS_PAD_REFLECT);
            const int xx = pad_index(x0 + pv - 1, a.in.W, HS_PAD_REFLECT);
            const float* __restrict__ sp = skb + (size_t)yy * a.in.W + xx;
#pragma unroll
            for (int c = 0; c < CSKIP; ++c) skv[ps][c] = sp[c * plane];
        }
    }

    // folded BatchNorm rows (4*hid + 2*COUT floats), also part of the single up-front load batch
    float bnreg[2];
    {
        const int nb = 4 * hid + 2 * COUT;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = min(tid + q * IRM_THREADS, nb - 1);
            const float* __restrict__ srcp;
            int off;
            if (e < hid) { srcp = a.s1; off = e; }
            else if (e < 2 * hid) { srcp = a.b1; off = e - hid; }
            else if (e < 3 * hid) { srcp = a.s2; off = e - 2 * hid; }
            else if (e < 4 * hid) { srcp = a.b2; off = e - 3 * hid; }
            else if (e < 4 * hid + COUT) { srcp = a.s3; off = e - 4 * hid; }
            else { srcp = a.b3; off = e - 4 * hid - COUT; }
            bnreg[q] = srcp[off];
        }
    }
    HS_STAMP(24);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + q * IRM_THREADS;
        if (e < 4 * hid + 2 * COUT) bnl[e] = bnreg[q];
    }
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int e = tid + q * IRM_THREADS;
        if (e < CPREV * G::PPL) pl[e] = preg[q];
    }
    if (w_vec) {
        float4* dst = reinterpret_cast<float4*>(wl);
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int e = tid + q * IRM_THREADS;
            if (e < nw4) dst[e] = wreg[q];
        }
        for (int e = tid + WQ * IRM_THREADS; e < nw4; e += IRM_THREADS) dst[e] = reinterpret_cast<const float4*>(wp)[e];
    } else {
        for (int e = tid; e < nw; e += IRM_THREADS) wl[e] = wp[e];
    }
    HS_STAMP(25);
    __syncthreads();
    HS_STAMP(26);
    // ---- prologue: stage-input columns -> LDS --------------------------------------------------
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int pos = tid + ps * IRM_THREADS;
        if (pos < G::NP1) {
            const bool live = pos < G::NPOS;
            const int pu = live ? pos / G::HW : 0, pv = live ? pos - pu * G::HW : 0;
            const int yy = pad_index(y0 + pu - 1, a.in.H, HS_PAD_REFLECT);
            const int xx = pad_index(x0 + pv - 1, a.in.W, HS_PAD_REFLECT);
            float col[KS1 * 4];
            col[0] = linspace_pm1(xx, a.in.W, a.in.step_x);
            col[1] = linspace_pm1(yy, a.in.H, a.in.step_y);
#pragma unroll
            for (int c = 0; c < CSKIP; ++c) col[2 + c] = skv[ps][c];
            const Tap ty = bilinear_tap(yy, a.in.scale_y, a.in.Hp), tx = bilinear_tap(xx, a.in.scale_x, a.in.Wp);
            // bilinear taps from the LDS window (tap indices are already clamped to the image by bilinear_tap)
            const int r0 = ty.i0 - ly0, r1 = ty.i1 - ly0, q0 = tx.i0 - lx0, q1 = tx.i1 - lx0;
            const int o00 = r0 * G::PW + q0, o01 = r0 * G::PW + q1, o10 = r1 * G::PW + q0, o11 = r1 * G::PW + q1;
#pragma unroll
            for (int c = 0; c < CPREV; ++c) {
                const float* q = pl + c * G::PPL;
                col[2 + CSKIP + c] = ty.l0 * (tx.l0 * q[o00] + tx.l1 * q[o01]) + ty.l1 * (tx.l0 * q[o10] + tx.l1 * q[o11]);
            }
#pragma unroll
            for (int c = CIN; c < KS1 * 4; ++c) col[c] = 0.0f;
#pragma unroll
            for (int c = 0; c < KS1 * 4; ++c) T[c * G::NP1 + pos] = live ? col[c] : 0.0f;
        }
    }
    HS_STAMP(27);
    __syncthreads();

    HS_STAMP(2);
    // ---- B fragments of this wave's position tiles, kept in registers for the whole kernel ------
    float bf[G::J1][KS1];
    int h1off[G::J1];                                  // LDS offset (pu*RS + pv) of this lane's position, -1 if none
#pragma unroll
    for (int jt = 0; jt < G::J1; ++jt) {
        const int nt = wave + 4 * jt;
        const int pos = nt * 16 + lrow;
        const bool ok = nt < G::NT1;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) bf[jt][ks] = ok ? T[(ks * 4 + lk) * G::NP1 + pos] : 0.0f;
        const int pu = pos / G::HW, pv = pos - pu * G::HW;
        h1off[jt] = (ok && pos < G::NPOS) ? pu * G::RS + pv : -1;
    }
    __syncthreads();                                   // T is dead: h1 / h2 may now overwrite it
    HS_STAMP(3);

    int h2off[G::J3];                                  // LDS offset of this lane's pixel inside an h2 plane
#pragma unroll
    for (int jt = 0; jt < G::J3; ++jt) {
        const int pix = (wave + 4 * jt) * 16 + lrow;
        h2off[jt] = (pix / TILE) * G::RS2 + (pix % TILE);
    }
    f32x4 acc3[MT3][G::J3];
#pragma unroll
    for (int m = 0; m < MT3; ++m)
#pragma unroll
        for (int jt = 0; jt < G::J3; ++jt) acc3[m][jt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // dw role of this thread: hidden channel hh of the chunk, output row u
    const int dw_hh = tid >> 4, dw_u = tid & 15;
    const bool dw_on = dw_u < TILE;

    // per-chunk operands: LDS reads of the staged bank (A fragments are distributed over the lanes)
    struct ChunkOps {
        float af[KS1];             // pw1 A fragments   W1[h0 + lrow][4*ks + lk]
        float sc1[4], sh1[4];      // bn1 rows of this lane's 4 D rows
        float k9[9], sc2, sh2;     // depthwise weights + bn2 of this thread's dw channel
        float a3[MT3][4];          // pw3 A fragments   W3[16*m + lrow][h0 + 4*ks + lk]
    };
    auto load_ops = [&](int h0, ChunkOps& o) {
        {
            const int h = h0 + lrow;
            const bool hok = h < hid;
            const float* wr = w1 + (hok ? h : 0) * CIN;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int k = ks * 4 + lk;
                const float v = wr[k < CIN ? k : 0];
                o.af[ks] = (hok && k < CIN) ? v : 0.0f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hr = h0 + 4 * lk + r;
            const int hc = hr < hid ? hr : 0;
            o.sc1[r] = bnl[hc]; o.sh1[r] = bnl[hid + hc];
        }
        {
            const int h = h0 + dw_hh;
            const int hc = h < hid ? h : 0;
#pragma unroll
            for (int q = 0; q < 9; ++q) o.k9[q] = kd[hc * 9 + q];
            o.sc2 = bnl[2 * hid + hc]; o.sh2 = bnl[3 * hid + hc];
        }
#pragma unroll
        for (int m = 0; m < MT3; ++m) {
            const int oc = m * 16 + lrow;
            const bool ook = oc < COUT;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int h = h0 + ks * 4 + lk;
                const float v = w3[(ook ? oc : 0) * hid + (h < hid ? h : 0)];
                o.a3[m][ks] = (ook && h < hid) ? v : 0.0f;
            }
        }
    };

    const int nchunks = (hid + 15) >> 4;
    ChunkOps ops;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int h0 = ch * 16;
        load_ops(h0, ops);
        const float (&af)[KS1] = ops.af;
        const float (&sc1)[4] = ops.sc1;
        const float (&sh1)[4] = ops.sh1;
        const float (&k9)[9] = ops.k9;
        const float sc2 = ops.sc2, sh2 = ops.sh2;
        const float (&a3)[MT3][4] = ops.a3;

        HS_STAMP(4 + 4 * (ch < 4 ? ch : 4));
        // ---- pw1: h1[16][pos] = relu6(bn1(W1 chunk . T)) ---------------------------------------
#pragma unroll
        for (int jt = 0; jt < G::J1; ++jt) {
            if (wave + 4 * jt < G::NT1) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    if (HS_IRM_ABLATE & 2) acc[ks & 3] += af[ks] * bf[jt][ks];
                    else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[jt][ks], acc, 0, 0, 0);
                }
                if (h1off[jt] >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h1[(4 * lk + r) * G::H1P + h1off[jt]] = relu6m(fmaf(acc[r], sc1[r], sh1[r]));
                }
            }
        }
        HS_STAMP(5 + 4 * (ch < 4 ? ch : 4));
        __syncthreads();
        HS_STAMP(6 + 4 * (ch < 4 ? ch : 4));

        // ---- dw 3x3 + bn2 + relu6: thread = (hidden channel, output row) ------------------------
        if (dw_on) {
            const float* hp = h1 + dw_hh * G::H1P + dw_u * G::RS;
            float rowv[3][G::RS];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int q = 0; q < G::RS / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(hp + ky * G::RS + 4 * q);
                    rowv[ky][4 * q] = v.x; rowv[ky][4 * q + 1] = v.y; rowv[ky][4 * q + 2] = v.z; rowv[ky][4 * q + 3] = v.w;
                }
            float o[TILE];
#pragma unroll
            for (int v = 0; v < TILE; ++v) {
                float d = 0.0f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        if (!(HS_IRM_ABLATE & 4) || (ky == 1 && kx == 1)) d = fmaf(k9[ky * 3 + kx], rowv[ky][v + kx], d);
                o[v] = relu6m(fmaf(d, sc2, sh2));
            }
            float* dst = h2 + dw_hh * G::H2S + dw_u * G::RS2;
#pragma unroll
            for (int q = 0; q < TILE / 4; ++q)
                *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
        __syncthreads();
        HS_STAMP(7 + 4 * (ch < 4 ? ch : 4));

        // ---- pw3: acc3 += W3[:, chunk] . h2 ------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (h0 + ks * 4 < hid) {                   // uniform: skip k-steps past the last hidden channel
#pragma unroll
                for (int jt = 0; jt < G::J3; ++jt) {
                    const int nt = wave + 4 * jt;
                    if (nt < G::NT3) {
                        const float bv = h2[(ks * 4 + lk) * G::H2S + h2off[jt]];
#pragma unroll
                        for (int m = 0; m < MT3; ++m) {
                            if (HS_IRM_ABLATE & 8) acc3[m][jt][ks] += a3[m][ks] * bv;
                            else acc3[m][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[m][ks], bv, acc3[m][jt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // no barrier here: the next chunk's pw1 writes h1 (all dw reads of h1 finished before the barrier above)
        // and the next dw writes h2 only after the next chunk's first barrier, which every wave reaches after pw3.
    }

    HS_STAMP(22);
    // ---- epilogue: bn3 + store -----------------------------------------------------------------
#pragma unroll
    for (int m = 0; m < MT3; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = m * 16 + 4 * lk + r;
            if (o < COUT) {
                const float sc = bnl[4 * hid + o], sh = bnl[4 * hid + COUT + o];
#pragma unroll
                for (int jt = 0; jt < G::J3; ++jt) {
                    const int nt = wave + 4 * jt;
                    if (nt < G::NT3) {
                        const int pix = nt * 16 + lrow;
                        const int u = pix / TILE, v = pix - u * TILE;
                        a.y[(((size_t)b * COUT + o) * a.in.H + (y0 + u)) * a.in.W + (x0 + v)] = fmaf(acc3[m][jt][r], sc, sh);
                    }
                }
            }
        }
    }
    HS_STAMP(23);
}

template <int CIN, int CSKIP, int COUT, int TILE>
static int launch_irm(const IrMfmaArgs& a, long blocks, hipStream_t stream) {
    using G = IrmGeom<TILE>;
    constexpr int KS1 = (CIN + 3) / 4;
    constexpr size_t t_floats = (size_t)KS1 * 4 * G::NP1;
    constexpr size_t h_floats = (size_t)16 * G::H1P + (size_t)16 * G::H2S;
    const size_t nw = (size_t)CIN * a.hid + 9 * (size_t)a.hid + (size_t)a.hid * COUT;
    constexpr size_t cprev = CIN - 2 - CSKIP;
    const size_t lds = ((t_floats > h_floats ? t_floats : h_floats) + ((nw + 3) & ~(size_t)3) +
                        ((4 * (size_t)a.hid + 2 * COUT + 3) & ~(size_t)3) + cprev * G::PPL) * sizeof(float);
    if (lds > 160 * 1024) return HS_ERR_LDS;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)patch_ir_mfma_kernel<CIN, CSKIP, COUT, TILE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((patch_ir_mfma_kernel<CIN, CSKIP, COUT, TILE>), dim3((unsigned)blocks), dim3(IRM_THREADS), lds, stream, a);
    return launch_status();
}

// Called by hs_patch_ir_fwd for the decoder's fused form.  Returns 1 if no instantiation matches (caller falls
// back to the generic VALU kernel), otherwise the launch status.
int try_launch_ir_mfma(const StageIn& in, int fh, int fw, const float* bank, long ld, int cin, int c_skip, int hid,
                       int c_out, const float* s1, const float* b1, const float* s2, const float* b2,
                       const float* s3, const float* b3, float* y, hipStream_t stream) {
    IrMfmaArgs a;
    a.in = in; a.fh = fh; a.fw = fw; a.ph = in.H / fh; a.pw = in.W / fw;
    a.bank = bank; a.ld = ld; a.hid = hid;
    a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3; a.y = y;
    int tile = 0;
    if (a.ph % 16 == 0 && a.pw % 16 == 0) tile = 16;
    else if (a.ph % 8 == 0 && a.pw % 8 == 0) tile = 8;
    {   // dev knob (read once): HS_IRM_TILE=8 forces 8x8 tiles on 16x16 patches
        static const int forced = [] { const char* e = getenv("HS_IRM_TILE"); return e ? atoi(e) : 0; }();
        if (forced == 8 && tile == 16) tile = 8;
    }
    if (!tile) return 1;
    if (in.Hp * 2 != in.H || in.Wp * 2 != in.W) return 1;      // the LDS window assumes the exact 2x pyramid
    a.tiles_y = a.ph / tile; a.tiles_x = a.pw / tile;
    const long blocks = (long)in.B * fh * fw * a.tiles_y * a.tiles_x;
#define HS_IRM_CASE(CI, CS, CO) \
    if (cin == CI && c_skip == CS && c_out == CO) \
        return tile == 16 ? launch_irm<CI, CS, CO, 16>(a, blocks, stream) : launch_irm<CI, CS, CO, 8>(a, blocks, stream);
    HS_IRM_CASE(24, 6, 16)    // HyperSeg-M / CamVid-S level 3
    HS_IRM_CASE(34, 16, 19)   // HyperSeg-M level 4 (Cityscapes, 19 classes)
    HS_IRM_CASE(14, 4, 8)     // HyperSeg-S level 3
    HS_IRM_CASE(26, 16, 19)   // HyperSeg-S level 4
    HS_IRM_CASE(22, 4, 12)    // CamVid-S level 4 (12 classes)
#undef HS_IRM_CASE
    return 1;
}

}  // namespace hs

// The lean form of hs_mbconv.hip's launch (round 6): same tiles, same chunk groups, same arithmetic order -- bit-identical y and
// pool partials -- for the shapes the benched encoders actually run (Cin a multiple of 4, Cmid a multiple of 16, whole tile columns, no
// squeeze-excite tail).  The general kernel spends more vector instructions on being general than on the block: 862 per wave for
// 36 matrix products, 80 transcendentals and 144 depthwise FMAs at HyperSeg-M's 24 -> 144 blocks (64-bit address arithmetic per
// load, clamps and mask multiplies for ragged channel counts, SGPR spills around the tail's arguments), at 160 registers = 3
// waves per SIMD: 20.5 us for a launch whose vector-issue floor is ~6 us (profiles/round6_mbconv_*).  Here:
//   * every global load is `uniform base + 32-bit lane offset` (the channel step of the input tile and of the weights is an
//     immediate or a scalar add), all of them issued before the first use;
//   * no channel clamps / masks (the host routes other shapes to the general kernel); positions outside the image still become
//     exact zeros in h1 (the depthwise conv pads the ACTIVATION);
//   * the block decode is scalar, with host-made magic numbers for the two divisions.
#include "hs_common.h"

#ifndef HS_MBL_STAGE
#define HS_MBL_STAGE 1      // output stores through a per-wave LDS staging area (dev A/B knob: tools/build_variants.py mbl_nostage)
#endif

namespace hs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct MblArgs {
    const float* __restrict__ x; const float* __restrict__ w_e; const float* __restrict__ s0; const float* __restrict__ b0;
    const float* __restrict__ w_dw; const float* __restrict__ s1; const float* __restrict__ b1;
    float* __restrict__ y; float* __restrict__ pool;
    int Cmid, H, W, Ho, Wo, pad_t, pad_l, tiles_y, tiles_x, chunks_per_wg, ngroups;
    unsigned m_ngroups, m_tiles_x, m_tiles_y;      // 2^32 / d + 1 (0: d == 1)
    int sH, sW, spad_t, spad_l;                    // STEM form: the raw image x (B, 3, sH, sW) and the stem conv's (top, left) padding
};

template <int K, int S, int OTH, int OTW> struct MblGeom {
    static constexpr int IH = (OTH - 1) * S + K, IW = (OTW - 1) * S + K;      // input halo tile
    static constexpr int NPOS = IH * IW;
    static constexpr int NT = (NPOS + 15) / 16;                               // position tiles (pw N)
    static constexpr int J = (NT + 3) / 4;                                    // position tiles per wave
    static constexpr int RS = (IW + 3) & ~3;                                  // h1 row stride: 16-byte aligned rows
    static constexpr int H1P = ((IH * RS + 7) & ~7) + 4;                      // plane == 4 (mod 8): as MbxGeom
    static constexpr int NSEG = 16 / OTH;                                     // row segments per output row
    static constexpr int NOUT = OTW / NSEG;                                   // outputs per dw thread
    static constexpr int NIN = (NOUT - 1) * S + K;                            // h1 values feeding them, per tap row
    static constexpr int NIN4 = (NIN + 3) & ~3;
    // output staging (per wave: its 4 hidden channels x OTH rows, rows padded to 20 floats: conflict-free 16-byte writes at a lane
    // stride of one row) -- see the store of the depthwise stage
    static constexpr int SROW = OTW + 4;
    // channels of a wave staged at a time.  (2 -- two passes, 10 KB less LDS, 4 workgroups per CU -- together with a single-chunk
    // specialisation of the kernel at 89 instead of 146 registers was measured in visit r6w17: 167.0 against 163.9 us over the 7 launches,
    // frame 0.7459 against 0.7408 ms.  More co-resident workgroups are not better here, as fewer were not in r6w15: three per CU it is.)
    static constexpr int STG_CH = 4;
    static constexpr int STG_WAVE = STG_CH * OTH * SROW;                      // floats per wave
    static constexpr int H1_FLOATS = 16 * H1P;
    static constexpr int LDS_FLOATS = H1_FLOATS + (HS_MBL_STAGE ? 4 * STG_WAVE : 0);
    static_assert(16 % OTH == 0 && OTW % NSEG == 0 && (NOUT * S) % 4 == 0 && NOUT % 4 == 0 && OTW == 16, "tile shape");
};

__device__ __forceinline__ unsigned mbl_div(unsigned x, unsigned m) { return m ? __umulhi(x, m) : x; }
inline unsigned mbl_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / d + 1); }
// base[byte offset] from a uniform base: saddr + voffset addressing
__device__ __forceinline__ float mbl_ld(const float* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
}
__device__ __forceinline__ f32x4 mbl_ld4(const float* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
}

// STEM (round 6): the "expand" is EfficientNet's stem -- Conv2d(3, Cmid, 3, stride 2, TF-"SAME" zero padding of the IMAGE) + BN + swish
// (efficientnet.py:321-322) -- feeding the first block's depthwise conv (that block has no expand conv): the same GEMM with
// K = 27 (+ 1 zero column: w_e is the stem weight flattened to (Cmid, 27) and padded to 28), its B operand gathered from the image
// through the 3 x 3 / stride-2 window, k = 9 c + 3 ky + kx.  The stem's output map (16.8 MB at 1024 x 512: written by one launch, read
// back by the next) never exists; a.H / a.W are ITS size (what the depthwise conv pads and tiles), a.sH / a.sW the image's.
template <int K, int S, int OTH, int OTW, int KS, bool STEM = false>
__global__ __launch_bounds__(256, 2)
void mbconv_lean_kernel(MblArgs a) {
    using G = MblGeom<K, S, OTH, OTW>;
    constexpr int Cin = 4 * KS;
    static_assert(!STEM || KS == 7, "the stem's K = 27 -> 28");
    extern __shared__ __attribute__((aligned(16))) float h1[];                 // [16][H1P]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lk = lane >> 4;
    // scalar block decode: blk = ((b * tiles_y + ty) * tiles_x + tx) * ngroups + grp
    unsigned blk = blockIdx.x;
    unsigned q = mbl_div(blk, a.m_ngroups); const int grp = (int)(blk - q * a.ngroups); blk = q;
    q = mbl_div(blk, a.m_tiles_x); const int tx = (int)(blk - q * a.tiles_x); blk = q;
    q = mbl_div(blk, a.m_tiles_y); const int ty = (int)(blk - q * a.tiles_y);
    const int b = (int)q;
    const int oy0 = ty * OTH, ox0 = tx * OTW;
    const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
    const int Cmid = a.Cmid, H = a.H, W = a.W;
    const unsigned plane = STEM ? (unsigned)a.sH * (unsigned)a.sW : (unsigned)H * (unsigned)W;
    const float* __restrict__ xb = a.x + (size_t)b * (STEM ? 3 : Cin) * plane;

    // dw role of this thread: hidden channel hh of the chunk, output row / row segment
    const int hh = tid >> 4, u = tid & 15;
    const int drow = u % OTH, dseg = u / OTH;
    const int ntiles = a.tiles_y * a.tiles_x;
    const int nchunks = Cmid >> 4;
    const int c_begin = grp * a.chunks_per_wg;
    const int c_end = min(c_begin + a.chunks_per_wg, nchunks);
    const unsigned a_off = (unsigned)(lrow * Cin + lk) << 2;
    const unsigned bn_off = (unsigned)(4 * lk) << 2;
    const unsigned dw_off = (unsigned)(hh * K * K) << 2;

    float af[KS];
    f32x4 sc0, sh0;
    float kd[K * K], sc1, sh1;
    float kdn[K * K], sc1n, sh1n;       // the next chunk's depthwise operands (kd is live while they are in flight)
    auto fetch_pw = [&](int ch) {       // what the expand GEMM of chunk ch needs
        const int h0 = ch * 16;
        const float* __restrict__ wr = a.w_e + (size_t)h0 * Cin;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = mbl_ld(wr + 4 * ks, a_off);
        sc0 = mbl_ld4(a.s0 + h0, bn_off); sh0 = mbl_ld4(a.b0 + h0, bn_off);
    };
    auto fetch_dw = [&](int ch) {       // what its depthwise stage needs
        const int h0 = ch * 16;
        const float* __restrict__ kr = a.w_dw + (size_t)h0 * K * K;
#pragma unroll
        for (int t = 0; t < K * K; ++t) kdn[t] = mbl_ld(kr + t, dw_off);
        sc1n = a.s1[h0 + hh]; sh1n = a.b1[h0 + hh];
    };
    auto take_dw = [&]() {
#pragma unroll
        for (int t = 0; t < K * K; ++t) kd[t] = kdn[t];
        sc1 = sc1n; sh1 = sh1n;
    };
    // the first chunk's operands BEFORE the tile: they are then the oldest requests, and the waits in front of their uses (which the
    // loop body shares between its first pass and the steady state) count the tile's loads as allowed-outstanding -- written after the
    // tile they would be the youngest, i.e. `vmcnt(0)` in the loop body, which in the steady state drains the previous chunk's stores
    fetch_pw(c_begin);
    fetch_dw(c_begin);
    __builtin_amdgcn_sched_barrier(0);

    // ---- the input halo tile -> B fragments (registers): bf[jt][ks] = x[4 ks + lk][position (wave + 4 jt) 16 + lrow] ----------
    float bf[G::J][KS];
    int h1off[G::J];                   // LDS offset of this lane's position (-1: none); bit 30: inside the image
    // STEM: this lane's K index of k-step ks is k = 4 ks + lk = 9 c + 3 ky + kx (k = 27: the zero column, any valid tap)
    int koff[STEM ? KS : 1], kyx[STEM ? KS : 1];
    unsigned smask[STEM ? G::J : 1];
    bool stem_inside = true;
    if constexpr (STEM) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = min(4 * ks + lk, 26), c = k / 9, r = k - 9 * c, ky = r / 3, kx = r - 3 * ky;
            koff[ks] = c * (int)plane + ky * a.sW + kx; kyx[ks] = 4 * ky + kx;
        }
#pragma unroll
        for (int jt = 0; jt < G::J; ++jt) smask[jt] = 0u;
        // rows / columns of the stem's map this tile's halo touches (clamped as the loader clamps them) and their windows
        const int y_lo = min(max(iy0, 0), H - 1), y_hi = min(max(iy0 + G::IH - 1, 0), H - 1);
        const int x_lo = min(max(ix0, 0), W - 1), x_hi = min(max(ix0 + G::IW - 1, 0), W - 1);
        stem_inside = 2 * y_lo - a.spad_t >= 0 && 2 * y_hi - a.spad_t + 2 < a.sH && 2 * x_lo - a.spad_l >= 0 && 2 * x_hi - a.spad_l + 2 < a.sW;
    }
#pragma unroll
    for (int jt = 0; jt < G::J; ++jt) {
        const int pos = (wave + 4 * jt) * 16 + lrow;
        const bool ok = pos < G::NPOS;
        const int pu = (STEM && !ok ? 0 : pos) / G::IW, pv = (STEM && !ok ? 0 : pos) - pu * G::IW;      // STEM: a dead lane gathers position 0's window
        const int yy = iy0 + pu, xx = ix0 + pv;
        const bool in = ok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        if constexpr (!STEM) {
            const unsigned off = ((in ? (unsigned)(yy * W + xx) : 0u) + (unsigned)lk * plane) << 2;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bf[jt][ks] = mbl_ld(xb + (size_t)(4 * ks) * plane, off);
        } else {
            // window origin of this stem-output position in the image (positions outside the stem's map: any valid one)
            const int by = 2 * min(max(yy, 0), H - 1) - a.spad_t, bx = 2 * min(max(xx, 0), W - 1) - a.spad_l;
            if (stem_inside) {                                                 // uniform: no tap of this tile leaves the image
                const int base = by * a.sW + bx;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bf[jt][ks] = mbl_ld(xb, (unsigned)(base + koff[ks]) << 2);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int iy = by + kyx[ks] / 4, ix = bx + (kyx[ks] & 3);
                    const int e = koff[ks] - (kyx[ks] / 4) * a.sW - (kyx[ks] & 3)            // c * plane
                                  + min(max(iy, 0), a.sH - 1) * a.sW + min(max(ix, 0), a.sW - 1);
                    bf[jt][ks] = mbl_ld(xb, (unsigned)e << 2);
                    smask[jt] |= ((unsigned)iy < (unsigned)a.sH && (unsigned)ix < (unsigned)a.sW ? 1u : 0u) << ks;
                }
            }
        }
        h1off[jt] = ok ? ((pu * G::RS + pv) | (in ? (1 << 30) : 0)) : -1;
    }
    if constexpr (STEM) {
        if (!stem_inside) {                                                    // zero padding of the image: after every load is out
#pragma unroll
            for (int jt = 0; jt < G::J; ++jt)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bf[jt][ks] = (smask[jt] >> ks) & 1u ? bf[jt][ks] : 0.0f;
        }
    }

    // every prologue load has landed before the loop is entered: the loop body then holds no wait that its first pass needs and the
    // steady state would pay for (with the tile's loads pending at the loop head the body's last matrix products wait `vmcnt(0)`)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    take_dw();
    // The chunk loop's barriers wait for the LDS only, and its ONE wait for vector memory sits right in front of a chunk's output
    // stores: by then the next chunk's operands (requested at the top of the depthwise stage) have arrived and the previous chunk's
    // stores have had a whole chunk to land, so nothing stalls -- and the stores issued after it stay in flight under the next chunk's
    // matrix products and depthwise FMAs.  (A __syncthreads() drains the vector-memory counter at every barrier: the launch's 19 MB of
    // stores were 5.6 of its 19.4 us with nothing overlapping them, visit r6w4.)
#define HS_MBL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    for (int ch = c_begin; ch < c_end; ++ch) {
        const int h0 = ch * 16;
        const bool more = ch + 1 < c_end;
        // ---- pw: h1[16][pos] = swish(BN0(W_e chunk . x tile)), exact zeros outside the image -------------------------
#pragma unroll
        for (int jt = 0; jt < G::J; ++jt) {
            if (wave + 4 * jt < G::NT) {                                       // uniform
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[jt][ks], acc, 0, 0, 0);
                if (h1off[jt] >= 0) {
                    const bool in = (h1off[jt] >> 30) & 1;
                    float* d = h1 + 4 * lk * G::H1P + (h1off[jt] & ((1 << 30) - 1));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sv = swishf(fmaf(acc[r], sc0[r], sh0[r]));
                        d[r * G::H1P] = in ? sv : 0.0f;
                    }
                }
            }
        }
        HS_MBL_BARRIER();
        if (more) { fetch_pw(ch + 1); fetch_dw(ch + 1); }

        // ---- dw K x K stride S + BN1 + swish: thread = (hidden channel, output row segment) --------------------------
        {
            const float* hp = h1 + hh * G::H1P + (drow * S) * G::RS + dseg * G::NOUT * S;
            float o[G::NOUT];
#pragma unroll
            for (int v = 0; v < G::NOUT; ++v) o[v] = 0.0f;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float rowv[G::NIN4];
#pragma unroll
                for (int qd = 0; qd < G::NIN4 / 4; ++qd) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(hp + ky * G::RS + 4 * qd);
                    rowv[4 * qd] = t[0]; rowv[4 * qd + 1] = t[1]; rowv[4 * qd + 2] = t[2]; rowv[4 * qd + 3] = t[3];
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int v = 0; v < G::NOUT; ++v) o[v] = fmaf(kd[ky * K + kx], rowv[v * S + kx], o[v]);
            }
            const int h = h0 + hh;
            float psum = 0.0f;
#pragma unroll
            for (int v = 0; v < G::NOUT; ++v) o[v] = swishf(fmaf(o[v], sc1, sh1));
#pragma unroll
            for (int qd = 0; qd < G::NOUT / 4; ++qd) psum += (o[4 * qd] + o[4 * qd + 1]) + (o[4 * qd + 2] + o[4 * qd + 3]);
            const bool rowok = oy0 + drow < a.Ho;            // the last tile row of a map whose height is not a multiple of OTH (CamVid: 72, 36)
            if (!rowok) psum = 0.0f;
            if (a.pool) psum = rowsum16(psum);      // SE pooling: one partial sum per (channel, tile), reduced over the channel's 16 lanes
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) as an instruction the compiler's own wait insertion sees (an asm string it does not)
            if (more) take_dw();
            __builtin_amdgcn_sched_barrier(0);
            // A lane owns NOUT consecutive pixels of ONE row: stored from here, a wave's store instruction touches 64 different 64-byte
            // row segments with 16 bytes each.  HS_MBL_STAGE = 1 sends the wave's 4 channels x OTH rows through its own LDS staging area
            // (no barrier: nobody else touches it) so that they leave as (channel, row, quarter) = 4 adjacent lanes per 64-byte segment:
            // 168.7 -> 162.0 us over HyperSeg-M's 7 launches, frame 0.757 -> 0.749 ms (profiles/round6_stem_dw_and_output_staging_ab_w9.txt).
            if constexpr (HS_MBL_STAGE) {
                float* stg = h1 + G::H1_FLOATS + wave * G::STG_WAVE;
                constexpr int NPASS = 4 / G::STG_CH;                              // passes over the wave's 4 channels
                constexpr int NLD = G::STG_CH * OTH * 4 / 64;                       // 16-byte pieces per lane and pass
                static_assert(NLD >= 1, "a pass fills every lane");
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    if (NPASS == 1 || ((hh & 3) >> 1) == ps) {
                        float* sw = stg + (((hh & 3) % G::STG_CH) * OTH + drow) * G::SROW + dseg * G::NOUT;
#pragma unroll
                        for (int qd = 0; qd < G::NOUT / 4; ++qd)
                            *reinterpret_cast<f32x4*>(sw + 4 * qd) = f32x4{o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]};
                    }
                    // a lane reads what OTHER lanes of its wave wrote: nothing in the language orders its own write (another address)
                    // before its read (the two-pass form of visit r6w17 came out reordered: rel err 1.2), so the order is pinned for the
                    // compiler -- the hardware executes a wave's LDS instructions in order, which is all it takes
                    asm volatile("" ::: "memory");
                    f32x4 ov[NLD];
#pragma unroll
                    for (int i = 0; i < NLD; ++i) {
                        const int idx = i * 64 + lane;
                        const int sq = idx & 3, sr = (idx >> 2) % OTH, sc = idx / (4 * OTH);
                        ov[i] = *reinterpret_cast<const f32x4*>(stg + (sc * OTH + sr) * G::SROW + 4 * sq);
                    }
                    asm volatile("" ::: "memory");                                  // ... and the next pass's writes stay behind these reads
#pragma unroll
                    for (int i = 0; i < NLD; ++i) {
                        const int idx = i * 64 + lane;
                        const int sq = idx & 3, sr = (idx >> 2) % OTH, sc = idx / (4 * OTH) + G::STG_CH * ps;
                        float* __restrict__ d4 = a.y + (((size_t)b * Cmid + (h0 + 4 * wave + sc)) * a.Ho + (oy0 + sr)) * a.Wo + ox0 + 4 * sq;
                        if (oy0 + sr < a.Ho) *reinterpret_cast<f32x4*>(d4) = ov[i];
                    }
                }
            } else {
                float* __restrict__ dst = a.y + (((size_t)b * Cmid + h) * a.Ho + oy0 + drow) * a.Wo + ox0 + dseg * G::NOUT;
                if (rowok) {
#pragma unroll
                    for (int qd = 0; qd < G::NOUT / 4; ++qd)
                        *reinterpret_cast<f32x4*>(dst + 4 * qd) = f32x4{o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]};
                }
            }
            if (a.pool && u == 0) a.pool[((size_t)b * Cmid + h) * ntiles + ty * a.tiles_x + tx] = psum;
        }
        if (more) HS_MBL_BARRIER();                 // h1 is rewritten by the next chunk's pw
    }
#undef HS_MBL_BARRIER
}

template <int K, int S, int OTH, int OTW, int KS, bool STEM = false>
static int launch_mbl(MblArgs& a, int batch, hipStream_t stream) {
    using G = MblGeom<K, S, OTH, OTW>;
    static const int lds_pad = [] { const char* e = getenv("HS_MBX_LDS_PAD"); return e ? atoi(e) : 0; }();      // dev knob: fewer co-resident workgroups (KB)
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float) + (size_t)lds_pad * 1024;
    const size_t blocks = (size_t)batch * a.tiles_y * a.tiles_x * a.ngroups;
    if (blocks > 0x7fffffffu) return 1;
    a.m_ngroups = mbl_magic((unsigned)a.ngroups); a.m_tiles_x = mbl_magic((unsigned)a.tiles_x); a.m_tiles_y = mbl_magic((unsigned)a.tiles_y);
    hipLaunchKernelGGL((mbconv_lean_kernel<K, S, OTH, OTW, KS, STEM>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    return launch_status();
}

// Returns 1 when the shape is outside what the lean kernel covers (the caller then takes the general kernel).
int try_launch_mbconv_lean(const float* x, int batch, int c_in, int H, int W, const float* w_expand, int c_mid, const float* scale0,
                           const float* shift0, const float* w_dw, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                           const float* scale1, const float* shift1, float* y, float* pool, int oth, int tiles_y, int tiles_x,
                           int chunks_per_wg, int ngroups, hipStream_t stream) {
    static const bool off = [] { const char* e = getenv("HS_MBX_LEAN"); return e && atoi(e) == 0; }();      // dev A/B knob
    if (off) return 1;
    if ((c_in & 3) != 0 || (c_mid & 15) != 0 || Wo % 16 != 0) return 1;             // (a ragged LAST TILE ROW is fine: rows past Ho are neither stored nor pooled)
    if ((size_t)c_in * H * W >= (1u << 30)) return 1;                       // 32-bit byte offsets from the batch element's base
    if ((((size_t)y | (size_t)scale0 | (size_t)shift0) & 15) != 0) return 1;
    MblArgs a;
    a.x = x; a.w_e = w_expand; a.s0 = scale0; a.b0 = shift0; a.w_dw = w_dw; a.s1 = scale1; a.b1 = shift1; a.y = y; a.pool = pool;
    a.Cmid = c_mid; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.pad_t = pad_t; a.pad_l = pad_l;
    a.tiles_y = tiles_y; a.tiles_x = tiles_x; a.chunks_per_wg = chunks_per_wg; a.ngroups = ngroups;
    a.sH = a.sW = a.spad_t = a.spad_l = 0;
    const int ks = c_in >> 2;
    if (k == 5 && stride == 2 && oth == 8 && ks > 6) return 1;      // that instantiation does not fit the register file (spills)
    // (Cin = 80 -- KS = 20, HyperSeg-M's 64 x 32 blocks -- was instantiated and measured in round 6: 11.5 us against the GEMM + depthwise
    // pair's 12.2 per block in isolation, but the whole frame LOSES 4.5 us with the fusion threshold raised to 80 channels:
    // profiles/round6_mbconv_lean_cin80_negative_w13.txt.  Not instantiated.)
#define HS_MBL_KS(K_, S_, OTH_) \
    if (k == K_ && stride == S_ && oth == OTH_) { \
        if (ks == 4) return launch_mbl<K_, S_, OTH_, 16, 4>(a, batch, stream); \
        if (ks == 6) return launch_mbl<K_, S_, OTH_, 16, 6>(a, batch, stream); \
        if (ks == 8) return launch_mbl<K_, S_, OTH_, 16, 8>(a, batch, stream);      /* EfficientNet-B3's 32-channel blocks (HyperSeg-L) */ \
        if (ks == 10) return launch_mbl<K_, S_, OTH_, 16, 10>(a, batch, stream); \
        return 1; \
    }
    HS_MBL_KS(3, 1, 16) HS_MBL_KS(3, 1, 8) HS_MBL_KS(5, 1, 16) HS_MBL_KS(5, 1, 8)
    HS_MBL_KS(3, 2, 8) HS_MBL_KS(3, 2, 4) HS_MBL_KS(5, 2, 8) HS_MBL_KS(5, 2, 4)
#undef HS_MBL_KS
    return 1;
}

// The stem form: x = the image (B, 3, sH, sW), w28 = the stem weight (c_mid, 27) padded to 28 columns, (Hs, Ws) = the stem's output map
// = the depthwise conv's input and (stride 1, "SAME") output size.  Returns 1 when the shape is not covered.
int try_launch_stem_dw_lean(const float* x, int batch, int sH, int sW, const float* w28, int c_mid, const float* scale0, const float* shift0,
                            int spad_t, int spad_l, int Hs, int Ws, const float* w_dw, int k, int pad_t, int pad_l, const float* scale1,
                            const float* shift1, float* y, float* pool, int oth, int tiles_y, int tiles_x, int chunks_per_wg, int ngroups,
                            hipStream_t stream) {
    if (k != 3 || (c_mid & 15) != 0 || Ws % 16 != 0 || spad_t < 0 || spad_l < 0) return 1;
    if ((size_t)3 * sH * sW >= (1u << 29)) return 1;                          // 32-bit byte offsets, with room for the clamped windows
    if (2 * (Hs - 1) - spad_t >= sH || 2 * (Ws - 1) - spad_l >= sW) return 1;  // every output's window starts inside the image
    if ((((size_t)y | (size_t)scale0 | (size_t)shift0) & 15) != 0) return 1;
    MblArgs a;
    a.x = x; a.w_e = w28; a.s0 = scale0; a.b0 = shift0; a.w_dw = w_dw; a.s1 = scale1; a.b1 = shift1; a.y = y; a.pool = pool;
    a.Cmid = c_mid; a.H = Hs; a.W = Ws; a.Ho = Hs; a.Wo = Ws; a.pad_t = pad_t; a.pad_l = pad_l;
    a.tiles_y = tiles_y; a.tiles_x = tiles_x; a.chunks_per_wg = chunks_per_wg; a.ngroups = ngroups;
    a.sH = sH; a.sW = sW; a.spad_t = spad_t; a.spad_l = spad_l;
    if (oth == 16) return launch_mbl<3, 1, 16, 16, 7, true>(a, batch, stream);
    if (oth == 8) return launch_mbl<3, 1, 8, 16, 7, true>(a, batch, stream);
    return 1;
}

}  // namespace hs

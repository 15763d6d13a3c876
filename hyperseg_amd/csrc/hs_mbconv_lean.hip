// The lean form of hs_mbconv.hip's launch (round 6): same tiles, same chunk groups, same arithmetic order -- bit-identical y and
// pool partials -- for the shapes the benched encoders actually run (Cin a multiple of 4, Cmid a multiple of 16, whole tiles, no
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

namespace hs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct MblArgs {
    const float* __restrict__ x; const float* __restrict__ w_e; const float* __restrict__ s0; const float* __restrict__ b0;
    const float* __restrict__ w_dw; const float* __restrict__ s1; const float* __restrict__ b1;
    float* __restrict__ y; float* __restrict__ pool;
    int Cmid, H, W, Ho, Wo, pad_t, pad_l, tiles_y, tiles_x, chunks_per_wg, ngroups;
    unsigned m_ngroups, m_tiles_x, m_tiles_y;      // 2^32 / d + 1 (0: d == 1)
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
    static constexpr int STG_WAVE = 4 * OTH * SROW;                           // floats per wave
    static constexpr int H1_FLOATS = 16 * H1P;
    static constexpr int LDS_FLOATS = H1_FLOATS + 4 * STG_WAVE;
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

template <int K, int S, int OTH, int OTW, int KS>
__global__ __launch_bounds__(256, 2)
void mbconv_lean_kernel(MblArgs a) {
    using G = MblGeom<K, S, OTH, OTW>;
    constexpr int Cin = 4 * KS;
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
    const unsigned plane = (unsigned)H * (unsigned)W;
    const float* __restrict__ xb = a.x + (size_t)b * Cin * plane;

    // dw role of this thread: hidden channel hh of the chunk, output row / row segment
    const int hh = tid >> 4, u = tid & 15;
    const int drow = u % OTH, dseg = u / OTH;
    const int oy = oy0 + drow, ox = ox0 + dseg * G::NOUT;
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
#pragma unroll
    for (int jt = 0; jt < G::J; ++jt) {
        const int pos = (wave + 4 * jt) * 16 + lrow;
        const bool ok = pos < G::NPOS;
        const int pu = pos / G::IW, pv = pos - pu * G::IW;
        const int yy = iy0 + pu, xx = ix0 + pv;
        const bool in = ok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const unsigned off = ((in ? (unsigned)(yy * W + xx) : 0u) + (unsigned)lk * plane) << 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bf[jt][ks] = mbl_ld(xb + (size_t)(4 * ks) * plane, off);
        h1off[jt] = ok ? ((pu * G::RS + pv) | (in ? (1 << 30) : 0)) : -1;
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
            if (a.pool) psum = rowsum16(psum);      // SE pooling: one partial sum per (channel, tile), reduced over the channel's 16 lanes
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) as an instruction the compiler's own wait insertion sees (an asm string it does not)
            if (more) take_dw();
            __builtin_amdgcn_sched_barrier(0);
            // A lane owns NOUT consecutive pixels of ONE row: stored from here, a wave's store instruction would touch 64 different
            // 64-byte row segments with 16 bytes each -- four 16-byte requests per segment where one 64-byte request does, and the
            // launch's stores are bound by requests, not bytes (visit r6w4: 5.6 of 19.4 us; 18.9 MB / 16 B at the L2s' ~128 requests
            // per clock = 4.4 us).  So the wave's 4 channels x OTH rows go through its own LDS staging area (no barrier: nobody else
            // touches it) and leave as (channel, row, quarter) = 4 adjacent lanes per 64-byte segment.
            {
                float* stg = h1 + G::H1_FLOATS + wave * G::STG_WAVE;
                float* sw = stg + ((hh & 3) * OTH + drow) * G::SROW + dseg * G::NOUT;
#pragma unroll
                for (int qd = 0; qd < G::NOUT / 4; ++qd)
                    *reinterpret_cast<f32x4*>(sw + 4 * qd) = f32x4{o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]};
                f32x4 ov[G::NOUT / 4];
#pragma unroll
                for (int i = 0; i < G::NOUT / 4; ++i) {
                    const int idx = i * 64 + lane;
                    const int sq = idx & 3, sr = (idx >> 2) % OTH, sc = idx / (4 * OTH);
                    ov[i] = *reinterpret_cast<const f32x4*>(stg + (sc * OTH + sr) * G::SROW + 4 * sq);
                }
#pragma unroll
                for (int i = 0; i < G::NOUT / 4; ++i) {
                    const int idx = i * 64 + lane;
                    const int sq = idx & 3, sr = (idx >> 2) % OTH, sc = idx / (4 * OTH);
                    float* __restrict__ d4 = a.y + (((size_t)b * Cmid + (h0 + 4 * wave + sc)) * a.Ho + (oy0 + sr)) * a.Wo + ox0 + 4 * sq;
                    *reinterpret_cast<f32x4*>(d4) = ov[i];
                }
            }
            if (a.pool && u == 0) a.pool[((size_t)b * Cmid + h) * ntiles + ty * a.tiles_x + tx] = psum;
            // (Tried: the wave's 4 channels x OTH rows through an LDS staging area so that 4 adjacent lanes write one 64-byte row segment
            // per store instruction instead of 16 bytes of 64 different segments -- 163.3 vs 162.0 us over the 7 launches, frame 0.7568 /
            // 0.7571 vs 0.7559 / 0.7573 ms: neutral, profiles/round6_mbconv_lean_output_staging_neutral_w7.txt.  The stores' cost is bytes.)
        }
        if (more) HS_MBL_BARRIER();                 // h1 is rewritten by the next chunk's pw
    }
#undef HS_MBL_BARRIER
}

template <int K, int S, int OTH, int OTW, int KS>
static int launch_mbl(MblArgs& a, int batch, hipStream_t stream) {
    using G = MblGeom<K, S, OTH, OTW>;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    const size_t blocks = (size_t)batch * a.tiles_y * a.tiles_x * a.ngroups;
    if (blocks > 0x7fffffffu) return 1;
    a.m_ngroups = mbl_magic((unsigned)a.ngroups); a.m_tiles_x = mbl_magic((unsigned)a.tiles_x); a.m_tiles_y = mbl_magic((unsigned)a.tiles_y);
    hipLaunchKernelGGL((mbconv_lean_kernel<K, S, OTH, OTW, KS>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    return launch_status();
}

// Returns 1 when the shape is outside what the lean kernel covers (the caller then takes the general kernel).
int try_launch_mbconv_lean(const float* x, int batch, int c_in, int H, int W, const float* w_expand, int c_mid, const float* scale0,
                           const float* shift0, const float* w_dw, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                           const float* scale1, const float* shift1, float* y, float* pool, int oth, int tiles_y, int tiles_x,
                           int chunks_per_wg, int ngroups, hipStream_t stream) {
    static const bool off = [] { const char* e = getenv("HS_MBX_LEAN"); return e && atoi(e) == 0; }();      // dev A/B knob
    if (off) return 1;
    if ((c_in & 3) != 0 || (c_mid & 15) != 0 || Ho % oth != 0 || Wo % 16 != 0) return 1;
    if ((size_t)c_in * H * W >= (1u << 30)) return 1;                       // 32-bit byte offsets from the batch element's base
    if ((((size_t)y | (size_t)scale0 | (size_t)shift0) & 15) != 0) return 1;
    MblArgs a;
    a.x = x; a.w_e = w_expand; a.s0 = scale0; a.b0 = shift0; a.w_dw = w_dw; a.s1 = scale1; a.b1 = shift1; a.y = y; a.pool = pool;
    a.Cmid = c_mid; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.pad_t = pad_t; a.pad_l = pad_l;
    a.tiles_y = tiles_y; a.tiles_x = tiles_x; a.chunks_per_wg = chunks_per_wg; a.ngroups = ngroups;
    const int ks = c_in >> 2;
    if (k == 5 && stride == 2 && oth == 8 && ks > 6) return 1;      // that instantiation does not fit the register file (spills)
#define HS_MBL_KS(K_, S_, OTH_) \
    if (k == K_ && stride == S_ && oth == OTH_) { \
        if (ks == 4) return launch_mbl<K_, S_, OTH_, 16, 4>(a, batch, stream); \
        if (ks == 6) return launch_mbl<K_, S_, OTH_, 16, 6>(a, batch, stream); \
        if (ks == 10) return launch_mbl<K_, S_, OTH_, 16, 10>(a, batch, stream); \
        return 1; \
    }
    HS_MBL_KS(3, 1, 16) HS_MBL_KS(3, 1, 8) HS_MBL_KS(5, 1, 16) HS_MBL_KS(5, 1, 8)
    HS_MBL_KS(3, 2, 8) HS_MBL_KS(3, 2, 4) HS_MBL_KS(5, 2, 8) HS_MBL_KS(5, 2, 4)
#undef HS_MBL_KS
    return 1;
}

}  // namespace hs

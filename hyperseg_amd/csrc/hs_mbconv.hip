// Encoder-side helper (opt-in through utils.inference): the first HALF of an MBConv block
// in one launch,
//     y = swish(BN1(depthwise_KxK_strideS(zero-pad(swish(BN0(W_e . x))))))      (+ per-tile sums of y for the SE pool)
// replacing {1x1 expand conv, BatchNorm, swish, F.pad, depthwise conv, BatchNorm, swish, adaptive_avg_pool2d} of
// hyperseg/models/backbones/efficientnet.py:101-106.  The 6x-expanded activation (50 MB at the first stride-2 block of
// HyperSeg-M 1024x512: written once and read once = 100 MB of HBM traffic around 0.4 GFLOP) never exists: it lives in LDS
// one 16-channel chunk of one spatial tile at a time.  Same structure as the decoder's patch_ir_mfma_kernel
// (hs_patch_ir_mfma.hip) minus the per-patch weights:
//
//   workgroup (4 waves) = one OTH x OTW output tile x a range of 16-channel chunks of the hidden dimension
//   prologue   the input halo tile ((OTH-1)S+K) x ((OTW-1)S+K) x Cin is loaded ONCE, straight into MFMA B fragments in
//              registers (lane = (position, k)): 64-byte runs, every load of the tile in flight together
//   per chunk  pw   v_mfma_f32_16x16x4_f32, A = W_e rows of the chunk (per-lane vector loads, L2 resident);
//                   D -> BN0 -> swish -> LDS h1[16][rows x RS]; positions outside the image are written as exact zeros
//                   (the depthwise conv pads the ACTIVATION, not the input)
//              dw   thread = (channel, output row segment): K rows of h1 as ds_read_b128, K*K per-lane taps, BN1, swish,
//                   16/32-byte row runs to HBM; 16-lane shuffle reduction -> one pool partial per (channel, tile)
// The depthwise taps accumulate in the same (ky, kx) order as depthwise_conv_kernel, so the two routes differ only by the
// k-order of the expand GEMM.
#include "hs_common.h"
#include "hs_se_tail.h"

namespace hs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct MbxArgs {
    const float* __restrict__ x; const float* __restrict__ w_e; const float* __restrict__ s0; const float* __restrict__ b0;
    const float* __restrict__ w_dw; const float* __restrict__ s1; const float* __restrict__ b1;
    float* __restrict__ y; float* __restrict__ pool;
    int Cin, Cmid, H, W, Ho, Wo, pad_t, pad_l, tiles_y, tiles_x, chunks_per_wg, ngroups;
    SeTail se;                          // hs_se_tail.h: ws != null -> the partials are published as granules and the last workgroups finish the gate
};

template <int K, int S, int OTH, int OTW> struct MbxGeom {
    static constexpr int IH = (OTH - 1) * S + K, IW = (OTW - 1) * S + K;      // input halo tile
    static constexpr int NPOS = IH * IW;
    static constexpr int NT = (NPOS + 15) / 16;                               // position tiles (pw N)
    static constexpr int J = (NT + 3) / 4;                                    // position tiles per wave
    static constexpr int RS = (IW + 3) & ~3;                                  // h1 row stride: 16-byte aligned rows
    static constexpr int H1P = ((IH * RS + 7) & ~7) + 4;                      // plane == 4 (mod 8): see IrmGeom::H1P
    static constexpr int NSEG = 16 / OTH;                                     // row segments per output row
    static constexpr int NOUT = OTW / NSEG;                                   // outputs per dw thread
    static constexpr int NIN = (NOUT - 1) * S + K;                            // h1 values feeding them, per tap row
    static constexpr int NIN4 = (NIN + 3) & ~3;
    static_assert(16 % OTH == 0 && OTW % NSEG == 0 && (NOUT * S) % 4 == 0, "tile shape");
};

// two workgroups per CU wherever the B fragments leave room for it (<= 256 registers incl. AGPRs): the per-chunk weight
// fetch and the two barriers of one workgroup then overlap the other's MFMA / depthwise phases
template <int K, int S, int OTH, int OTW, int KS>
__global__ __launch_bounds__(256, (MbxGeom<K, S, OTH, OTW>::J * KS <= 80 ? 2 : 1))
void mbconv_expand_dw_kernel(MbxArgs a) {
    using G = MbxGeom<K, S, OTH, OTW>;
    extern __shared__ __attribute__((aligned(16))) float h1[];                 // [16][H1P]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lk = lane >> 4;
    int blk = blockIdx.x;
    const int grp = blk % a.ngroups; blk /= a.ngroups;
    const int tx = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty = blk % a.tiles_y;
    const int b = blk / a.tiles_y;
    const unsigned se_gen = a.se.ws ? se_tag(a.se, b) : 0u;
    const int oy0 = ty * OTH, ox0 = tx * OTW;
    const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
    const int Cin = a.Cin, Cmid = a.Cmid;
    const size_t plane = (size_t)a.H * a.W;

    // ---- prologue: the input halo tile -> B fragments (registers), all loads in flight together ----------------------
    float bf[G::J][KS];
    int h1off[G::J];                   // LDS offset of this lane's position (-1: no such position); bit 30: inside the image
    unsigned inmask = 0;
#pragma unroll
    for (int jt = 0; jt < G::J; ++jt) {
        const int nt = wave + 4 * jt;
        const int pos = nt * 16 + lrow;
        const bool ok = nt < G::NT && pos < G::NPOS;
        const int pu = ok ? pos / G::IW : 0, pv = ok ? pos - pu * G::IW : 0;
        const int yy = iy0 + pu, xx = ix0 + pv;
        const bool in = ok && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        const float* __restrict__ src = a.x + (size_t)b * Cin * plane + (size_t)(in ? yy : 0) * a.W + (in ? xx : 0);
        // unconditional loads from clamped (always valid) addresses; the masks are applied in a second pass below so that
        // no load of the tile waits for an earlier one (a use between the loads made the compiler drain the first 8 of
        // them before issuing the rest: two memory round trips instead of one; tools/isa_phases.py)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 4 + lk;
            bf[jt][ks] = src[(size_t)(c < Cin ? c : Cin - 1) * plane];
        }
        inmask |= (in ? 1u : 0u) << jt;
        h1off[jt] = ok ? ((pu * G::RS + pv) | (in ? (1 << 30) : 0)) : -1;
    }

    // dw role of this thread: hidden channel hh of the chunk, output row / row segment
    const int hh = tid >> 4, u = tid & 15;
    const int drow = u % OTH, dseg = u / OTH;
    const int oy = oy0 + drow, ox = ox0 + dseg * G::NOUT;
    const int ntiles = a.tiles_y * a.tiles_x;
    const int nchunks = (Cmid + 15) >> 4;
    const int c_begin = grp * a.chunks_per_wg;
    const int c_end = min(c_begin + a.chunks_per_wg, nchunks);

    // Per-chunk operands (weights: L2-resident vector loads).  Every load is unconditional from a clamped address and all
    // of them are issued before the first use (masks applied afterwards by a multiply): with selects the compiler put each
    // A-fragment load behind its own exec-mask branch and waited for it -- four serialised L2 round trips per chunk
    // (tools/isa_phases.py).  The first chunk's GEMM operands are requested together with the input tile, the next chunk's
    // at the end of the current one (ahead of its closing barrier).
    float af[KS];                       // A fragments  W_e[h0 + lrow][4*ks + lk]
    float sc0[4], sh0[4];               // BN0 rows of this lane's 4 D rows
    float kd[K * K], sc1, sh1;          // depthwise taps and BN1 of this thread's hidden channel
    auto fetch_pw = [&](int ch) {       // what the expand GEMM of chunk ch needs
        const int h0 = ch * 16;
        const float* __restrict__ wr = a.w_e + (size_t)min(h0 + lrow, Cmid - 1) * Cin;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = wr[min(ks * 4 + lk, Cin - 1)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hr = min(h0 + 4 * lk + r, Cmid - 1);
            sc0[r] = a.s0[hr]; sh0[r] = a.b0[hr];
        }
    };
    auto fetch_dw = [&](int ch) {       // what its depthwise stage needs: in flight during the GEMM
        const int hd = min(ch * 16 + hh, Cmid - 1);
#pragma unroll
        for (int q = 0; q < K * K; ++q) kd[q] = a.w_dw[(size_t)hd * K * K + q];
        sc1 = a.s1[hd]; sh1 = a.b1[hd];
    };
    // (the 5x5 instances with a large input tile are within a few registers of the 256 the two-workgroups-per-CU bound
    // allows: they fetch at the top of the chunk instead -- still one round trip, not overlapped with the tile's)
    constexpr bool HOIST = !(K == 5 && G::J * KS > 40);
    if constexpr (HOIST) fetch_pw(min(c_begin, nchunks - 1));
#pragma unroll
    for (int jt = 0; jt < G::J; ++jt) {
        const bool in = (inmask >> jt) & 1u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bf[jt][ks] *= (in && ks * 4 + lk < Cin) ? 1.0f : 0.0f;   // a multiply, not a select:
    }                                                                                             // no exec-mask branches

    for (int ch = c_begin; ch < c_end; ++ch) {
        const int h0 = ch * 16;
        if constexpr (!HOIST) fetch_pw(ch);
        fetch_dw(ch);
        {
            const bool hok = h0 + lrow < Cmid;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[ks] *= (hok && ks * 4 + lk < Cin) ? 1.0f : 0.0f;
        }

        // ---- pw: h1[16][pos] = swish(BN0(W_e chunk . x tile)), exact zeros outside the image -------------------------
#pragma unroll
        for (int jt = 0; jt < G::J; ++jt) {
            if (wave + 4 * jt < G::NT) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[jt][ks], acc, 0, 0, 0);
                if (h1off[jt] >= 0) {
                    const bool in = (h1off[jt] >> 30) & 1;
                    const int off = h1off[jt] & ((1 << 30) - 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sv = swishf(fmaf(acc[r], sc0[r], sh0[r]));
                        h1[(4 * lk + r) * G::H1P + off] = in ? sv : 0.0f;
                    }
                }
            }
        }
        __syncthreads();

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
                for (int q = 0; q < G::NIN4 / 4; ++q) {
                    const float4 t = *reinterpret_cast<const float4*>(hp + ky * G::RS + 4 * q);
                    rowv[4 * q] = t.x; rowv[4 * q + 1] = t.y; rowv[4 * q + 2] = t.z; rowv[4 * q + 3] = t.w;
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int v = 0; v < G::NOUT; ++v) o[v] = fmaf(kd[ky * K + kx], rowv[v * S + kx], o[v]);
            }
            const int h = h0 + hh;
            float psum = 0.0f;
            if (h < Cmid && oy < a.Ho) {
                float* __restrict__ dst = a.y + (((size_t)b * Cmid + h) * a.Ho + oy) * a.Wo + ox;
#pragma unroll
                for (int v = 0; v < G::NOUT; ++v) o[v] = swishf(fmaf(o[v], sc1, sh1));
                if ((a.Wo & 3) == 0 && ox + G::NOUT <= a.Wo) {
#pragma unroll
                    for (int q = 0; q < G::NOUT / 4; ++q) {
                        *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
                        psum += (o[4 * q] + o[4 * q + 1]) + (o[4 * q + 2] + o[4 * q + 3]);
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < G::NOUT; ++v)
                        if (ox + v < a.Wo) { dst[v] = o[v]; psum += o[v]; }
                }
            }
            // SE pooling: one partial sum per (channel, tile), reduced over the channel's 16 lanes
            if (a.pool || a.se.ws) {
                psum = rowsum16(psum);         // the channel's 16 lanes are one DPP row
                const size_t pidx = ((size_t)b * Cmid + h) * ntiles + ty * a.tiles_x + tx;
                if (u == 0 && h < Cmid) {
                    if (a.se.ws) se_publish(a.se.ws, se_ws_pg(a.se) + pidx, psum, se_gen);
                    else a.pool[pidx] = psum;
                }
            }
        }
        if constexpr (HOIST) {
            if (ch + 1 < c_end) fetch_pw(ch + 1);
        }
        __syncthreads();               // h1 is rewritten by the next chunk's pw
    }
    if (a.se.ws) {                     // h1 is free now (the host sized the segment for the tail as well)
        const long per_b = (long)ntiles * a.ngroups;
        se_tail_run(a.se, b, (long)blockIdx.x - (long)b * per_b, per_b, se_gen, h1);
    }
}

template <int K, int S, int OTH, int OTW, int KS>
static int launch_mbx(const MbxArgs& a, int batch, hipStream_t stream) {
    using G = MbxGeom<K, S, OTH, OTW>;
    size_t lds = (size_t)16 * G::H1P * sizeof(float);
    if (a.se.ws) {
        const size_t tail = (size_t)se_lds_floats(a.se.Csq) * sizeof(float);
        if (tail > 64 * 1024) return HS_ERR_UNSUPPORTED;
        if (tail > lds) lds = tail;
    }
    const size_t blocks = (size_t)batch * a.tiles_y * a.tiles_x * a.ngroups;
    if (blocks > 0x7fffffffu) return HS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((mbconv_expand_dw_kernel<K, S, OTH, OTW, KS>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    return launch_status();
}

template <int K, int S, int OTH, int OTW>
static int dispatch_ks(const MbxArgs& a, int batch, hipStream_t stream) {
    const int ks = (a.Cin + 3) / 4;
    if (ks <= 4) return launch_mbx<K, S, OTH, OTW, 4>(a, batch, stream);
    if (ks <= 6) return launch_mbx<K, S, OTH, OTW, 6>(a, batch, stream);
    if (ks <= 10) return launch_mbx<K, S, OTH, OTW, 10>(a, batch, stream);
    if (ks <= 12) return launch_mbx<K, S, OTH, OTW, 12>(a, batch, stream);
    if (ks <= 20) return launch_mbx<K, S, OTH, OTW, 20>(a, batch, stream);
    return HS_ERR_UNSUPPORTED;         // wider inputs: the B fragments no longer fit the register file -> unfused route
}

int try_launch_mbconv_lean(const float* x, int batch, int c_in, int H, int W, const float* w_expand, int c_mid, const float* scale0,
                           const float* shift0, const float* w_dw, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                           const float* scale1, const float* shift1, float* y, float* pool, int oth, int tiles_y, int tiles_x,
                           int chunks_per_wg, int ngroups, hipStream_t stream);                 // hs_mbconv_lean.hip

int try_launch_stem_dw_lean(const float* x, int batch, int sH, int sW, const float* w28, int c_mid, const float* scale0, const float* shift0,
                            int spad_t, int spad_l, int Hs, int Ws, const float* w_dw, int k, int pad_t, int pad_l, const float* scale1,
                            const float* shift1, float* y, float* pool, int oth, int tiles_y, int tiles_x, int chunks_per_wg, int ngroups,
                            hipStream_t stream);                                               // hs_mbconv_lean.hip

}  // namespace hs

using namespace hs;

// dev A/B knobs of the tile shape (round 6): output-tile height per stride and the workgroup count below which a launch is cut
// into more channel-chunk groups
static int mbx_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static int mbx_oth(int stride) {
    static const int o1 = mbx_knob("HS_MBX_OTH1", 16), o2 = mbx_knob("HS_MBX_OTH2", 8);
    return stride == 1 ? o1 : o2;
}

extern "C" int hs_mbconv_tiles(int32_t k, int32_t stride, int32_t Ho, int32_t Wo) {
    (void)k;
    const int oth = mbx_oth(stride), otw = 16;
    return ((Ho + oth - 1) / oth) * ((Wo + otw - 1) / otw);
}

// tiles and channel-chunk groups of a launch (enough workgroups to fill 256 CUs a few times over, but as few re-loads of the
// input tile as that allows)
static void mbx_grid(int batch, int c_mid, int stride, int Ho, int Wo, int& tiles_y, int& tiles_x, int& cpw, int& ngroups) {
    const int oth = mbx_oth(stride), otw = 16;
    tiles_y = (Ho + oth - 1) / oth; tiles_x = (Wo + otw - 1) / otw;
    const int nchunks = (c_mid + 15) / 16;
    const long tiles = (long)batch * tiles_y * tiles_x;
    static const int min_wg = mbx_knob("HS_MBX_MIN_WG", 768);
    cpw = nchunks;
    while (cpw > 1 && tiles * ((nchunks + cpw - 1) / cpw) < min_wg) --cpw;
    ngroups = (nchunks + cpw - 1) / cpw;
}

extern "C" int64_t hs_mbconv_se_workgroups(int32_t batch, int32_t c_mid, int32_t k, int32_t stride, int32_t Ho, int32_t Wo) {
    (void)k;
    if (batch <= 0 || c_mid <= 0 || Ho <= 0 || Wo <= 0 || (stride != 1 && stride != 2)) return 0;
    int ty, tx, cpw, ng;
    mbx_grid(batch, c_mid, stride, Ho, Wo, ty, tx, cpw, ng);
    return (int64_t)ty * tx * ng;
}

static int mbconv_launch(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W,
                         const float* w_expand, int32_t c_mid, const float* scale0, const float* shift0,
                         const float* w_dw, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                         int32_t Ho, int32_t Wo, const float* scale1, const float* shift1, float* y,
                         float* pool_partial, const hs_se_tail* se_in, void* stream) {
    if (!x || !w_expand || !scale0 || !shift0 || !w_dw || !scale1 || !shift1 || !y) return HS_ERR_BAD_ARG;
    if (batch <= 0 || c_in <= 0 || c_mid <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || pad_t < 0 || pad_l < 0)
        return HS_ERR_BAD_ARG;
    MbxArgs a;
    a.x = x; a.w_e = w_expand; a.s0 = scale0; a.b0 = shift0; a.w_dw = w_dw; a.s1 = scale1; a.b1 = shift1;
    a.y = y; a.pool = pool_partial;
    a.Cin = c_in; a.Cmid = c_mid; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.pad_t = pad_t; a.pad_l = pad_l;
    if (stride != 1 && stride != 2) return HS_ERR_UNSUPPORTED;
    mbx_grid(batch, c_mid, stride, Ho, Wo, a.tiles_y, a.tiles_x, a.chunks_per_wg, a.ngroups);
    hipStream_t s = (hipStream_t)stream;
    if (!se_in && (stride == 1 || stride == 2)) {      // the lean form (hs_mbconv_lean.hip) wherever the shape allows: same tiles, same sums
        const int st = try_launch_mbconv_lean(x, batch, c_in, H, W, w_expand, c_mid, scale0, shift0, w_dw, k, stride, pad_t, pad_l, Ho, Wo,
                                              scale1, shift1, y, pool_partial, mbx_oth(stride), a.tiles_y, a.tiles_x, a.chunks_per_wg,
                                              a.ngroups, s);
        if (st != 1) return st;
    }
    a.se = SeTail{};
    if (se_in) {
        const int st = make_se_tail(se_in, batch, c_mid, a.tiles_y * a.tiles_x, (long)a.tiles_y * a.tiles_x * a.ngroups, Ho * Wo, a.se);
        if (st != HS_OK) return st;
    }
    const int oth = mbx_oth(stride);
    if (k == 3 && stride == 1) return oth == 8 ? dispatch_ks<3, 1, 8, 16>(a, batch, s) : dispatch_ks<3, 1, 16, 16>(a, batch, s);
    if (k == 3 && stride == 2) return oth == 4 ? dispatch_ks<3, 2, 4, 16>(a, batch, s) : dispatch_ks<3, 2, 8, 16>(a, batch, s);
    if (k == 5 && stride == 1) return oth == 8 ? dispatch_ks<5, 1, 8, 16>(a, batch, s) : dispatch_ks<5, 1, 16, 16>(a, batch, s);
    if (k == 5 && stride == 2) return oth == 4 ? dispatch_ks<5, 2, 4, 16>(a, batch, s) : dispatch_ks<5, 2, 8, 16>(a, batch, s);
    return HS_ERR_UNSUPPORTED;
}

extern "C" int hs_mbconv_expand_dw_fwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W,
                                       const float* w_expand, int32_t c_mid, const float* scale0, const float* shift0,
                                       const float* w_dw, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                                       int32_t Ho, int32_t Wo, const float* scale1, const float* shift1, float* y,
                                       float* pool_partial, void* stream) {
    return mbconv_launch(x, batch, c_in, H, W, w_expand, c_mid, scale0, shift0, w_dw, k, stride, pad_t, pad_l, Ho, Wo, scale1, shift1, y,
                         pool_partial, nullptr, stream);
}

// The same launch finishing the block's squeeze-excite gate in its last workgroups (hs_se_tail.h)
extern "C" int hs_mbconv_expand_dw_se_fwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W,
                                          const float* w_expand, int32_t c_mid, const float* scale0, const float* shift0,
                                          const float* w_dw, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                                          int32_t Ho, int32_t Wo, const float* scale1, const float* shift1, float* y,
                                          const hs_se_tail* se, void* stream) {
    if (!se) return HS_ERR_BAD_ARG;
    return mbconv_launch(x, batch, c_in, H, W, w_expand, c_mid, scale0, shift0, w_dw, k, stride, pad_t, pad_l, Ho, Wo, scale1, shift1, y,
                         nullptr, se, stream);
}

// The encoder's stem and the first block's depthwise half in ONE launch (hs_mbconv_lean.hip, STEM form):
//   y = swish(BN1(depthwise_3x3(zero-pad(swish(BN0(conv3x3/s2(zero-pad(x))))))))   (+ per-tile sums of y for the SE pool)
// x (B,3,H,W); w_stem28 (c_mid, 28) = the stem weight flattened to (c_mid, 27) with one zero column; (Hs, Ws) the stem's output size
// = y's; pool_partial (optional) (B*c_mid, hs_mbconv_tiles(3, 1, Hs, Ws)).  HS_ERR_UNSUPPORTED for shapes the launch does not cover
// (the caller then runs hs_stem_conv_fwd + hs_depthwise_conv_fwd).  Replaces efficientnet.py:321-322 + 59-66 / 101-103 of block 0.
extern "C" int hs_stem_dw_fwd(const float* x, int32_t batch, int32_t H, int32_t W, const float* w_stem28, int32_t c_mid,
                              const float* scale0, const float* shift0, int32_t stem_pad_t, int32_t stem_pad_l, int32_t Hs, int32_t Ws,
                              const float* w_dw, int32_t k, int32_t pad_t, int32_t pad_l, const float* scale1, const float* shift1,
                              float* y, float* pool_partial, void* stream) {
    if (!x || !w_stem28 || !scale0 || !shift0 || !w_dw || !scale1 || !shift1 || !y) return HS_ERR_BAD_ARG;
    if (batch <= 0 || c_mid <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0 || pad_t < 0 || pad_l < 0) return HS_ERR_BAD_ARG;
    int tiles_y, tiles_x, cpw, ngroups;
    mbx_grid(batch, c_mid, 1, Hs, Ws, tiles_y, tiles_x, cpw, ngroups);
    const int st = try_launch_stem_dw_lean(x, batch, H, W, w_stem28, c_mid, scale0, shift0, stem_pad_t, stem_pad_l, Hs, Ws, w_dw, k, pad_t,
                                           pad_l, scale1, shift1, y, pool_partial, mbx_oth(1), tiles_y, tiles_x, cpw, ngroups,
                                           (hipStream_t)stream);
    return st == 1 ? HS_ERR_UNSUPPORTED : st;
}

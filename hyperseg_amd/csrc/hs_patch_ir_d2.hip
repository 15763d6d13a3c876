// Op D (hyperseg_v0_1.py:205-237) for the SMALL-PATCH levels of HyperSeg-L -- level 2 (48 -> 96 -> 12 channels on the 64 x 64 map,
// 4 x 4-pixel patches) and level 3 (22 -> 44 -> 8 on 128 x 128, 8 x 8 patches) -- as the reference states it: three image-level
// patch convolutions, in TWO launches with the hidden map in HBM between them.
//
// Why not one launch: a depthwise tap that crosses a patch edge reads a hidden activation computed with the NEIGHBOUR's W1.  The
// tiled kernel (hs_patch_ir_fused.hip, MODE 1) recomputes that ring per 16 x 16 region with per-position owner lookups -- at these
// levels every region spans 4 / 16 patches and its ring 12 / 20 more, W1 is 56-70 % of a bank row, and the launch sits at 6-10 % of
// its HBM floor (197 / 307 us per bs-32 batch, profiles/round2_decoder_L_kernel_stats.csv).  Writing h1 once costs 50 / 100 MB of
// the ~350 / 300 MB a batch moves here anyway (the level-2 bank alone is 220 MB: 27 KB of weights for 16 pixels), and both halves
// become what these levels really are: a stream of small per-patch GEMMs whose N dimension IS the patch --
//   16 pixels = one 16-column MFMA tile (4 x 4 patches: one tile per patch, 8 x 8: four tiles sharing the A fragments).
//
// pass 1  h1 = relu6(bn1(W1[patch] . x)),  x = cat(coords, skip, bilinear2x(prev)) assembled per lane, branch-free (every candidate
//         load issued, the channel's kind selected afterwards); wave = patch, v_mfma_f32_16x16x4_f32 with A = the bank row's W1
//         straight from HBM into registers -- each weight is used exactly once per patch, LDS would only add a hop.  A lane fetches
//         4 consecutive k of its row in one load; MFMA j then multiplies k-set {16 q + 4 kgroup + j}, B is assembled to match.
//         h1 is stored CHANNELS-LAST (B, H, W, hid rounded up to 16): the lane's four accumulator rows are four consecutive channels
//         = one 16-byte store, and exactly the unit pass 2 reads.
// pass 2  h2 = relu6(bn2(dw3x3 reflect (own patch's taps))) for (pixel, 4 channels) per lane -- 9 x 16-byte loads of h1 -- which IS
//         the B fragment of y = bn3(W3[patch] . h2): 4 MFMAs per 16 hidden channels, A = W3 rows as 16-byte loads.
// Exact f32.  The hidden map comes from the caller (hs_patch_ir_v0_ws_fwd's workspace): the library owns no global state.
#include "hs_ir_common.h"

namespace hs {

using d2_f32x4 = __attribute__((ext_vector_type(4))) float;
using d2_f32x2 = __attribute__((ext_vector_type(2))) float;

struct IrdArgs {
    StageIn in;
    const float* __restrict__ bank;
    long ld;
    int fh, fw, cin, hid, cout, hidp;
    const float* __restrict__ s1; const float* __restrict__ b1;
    const float* __restrict__ s2; const float* __restrict__ b2;
    const float* __restrict__ s3; const float* __restrict__ b3;
    float* __restrict__ h1;                // (B, H, W, hidp)
    float* __restrict__ y;                 // (B, cout, H, W)
};

// lane n of a 16-pixel tile t of a PW x PW patch -> position inside the patch
template <int PW> __device__ __forceinline__ void d2_pixel(int t, int n, int& ly, int& lx) {
    if constexpr (PW == 4) { ly = n >> 2; lx = n & 3; }
    else { ly = 2 * t + (n >> 3); lx = n & 7; }
}

// 4 consecutive floats at p (AL = the alignment the host guarantees, in floats)
template <int AL> __device__ __forceinline__ d2_f32x4 d2_load4(const float* __restrict__ p) {
    if constexpr (AL == 4) return *reinterpret_cast<const d2_f32x4*>(p);
    else if constexpr (AL == 2) {
        const d2_f32x2 lo = *reinterpret_cast<const d2_f32x2*>(p), hi = *reinterpret_cast<const d2_f32x2*>(p + 2);
        return d2_f32x4{lo[0], lo[1], hi[0], hi[1]};
    } else return d2_f32x4{p[0], p[1], p[2], p[3]};
}

// RT 16-row tiles of hidden channels, KQ 16-deep k blocks of input channels; 4 waves = 4 patches next to each other along x.
// The stage input x = cat(coords, skip, bilinear2x(prev)) of the workgroup's 4 PW x PW pixel region is assembled ONCE, cooperatively,
// into LDS as [pixel][16 KQ channels] -- thread = pixel (x channel group at 4 x 4 patches), the channel index uniform per wave, so
// each kind of channel is a loop without divergence: skip channels are coalesced global loads (all in flight together), previous-
// level channels are four ds_read_b32 of the low-resolution window (PW/2 + 2 rows x 2 PW + 2 columns per channel, clamped at the
// image border exactly as the bilinear taps clamp; the level below is at exactly half the resolution: the host checks) -- and a
// lane's B fragment is then ONE ds_read_b128 per k block and tile.  (First form: every lane assembled its own 4 KQ values per tile
// branch-free, i.e. all five candidate loads and the bilinear arithmetic for every k whatever its kind: 2650 instructions per wave at
// level 3, issue-bound at 52 us; before the window went through LDS the corner loads came from global memory, 68 us.)
template <int PW, int KQ> struct IrdP1 {
    static constexpr int WROWS = PW / 2 + 2, WCOLS = 2 * PW + 2, WP = WROWS * WCOLS;
    static constexpr int RW = 4 * PW, NPX = RW * PW, CG = 256 / NPX;     // pixels per region; channel groups per pixel (4 x 4: 4)
    static constexpr int KP = 16 * KQ, XS = KP + 4;                      // floats per pixel: an odd number of 16-byte granules
    static constexpr int ROWP = PW == 4 ? 20 : 40;                       // row pitch in pixels, == PW (mod 16): see pass 2
    static constexpr int XL = PW * ROWP * XS;                            // floats of the assembled region; the window follows
    static constexpr int MAXS = 16 / CG;                                 // skip channels per thread (c_skip <= 16)
};

template <int PW, int RT, int KQ, int AL>
__global__ __launch_bounds__(256)
void ird_pw1_kernel(IrdArgs a) {
    using P = IrdP1<PW, KQ>;
    constexpr int NT = PW * PW / 16;
    extern __shared__ __attribute__((aligned(16))) float d2_p1[];
    float* xl = d2_p1;                                                  // [PW][ROWP][XS]
    float* win = d2_p1 + P::XL;                                         // [c_prev][WROWS][WCOLS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int pj = blockIdx.x * 4 + wave, pi = blockIdx.y, b = blockIdx.z;
    const StageIn& s = a.in;
    const float* __restrict__ w1 = a.bank + (size_t)((b * a.fh + pi) * a.fw + min(pj, a.fw - 1)) * (size_t)a.ld;
    const int kq_max = a.cin - AL;                                      // last k a vector piece may start at

    // ---- A fragments of the whole patch: row 16 rt + n, k = 16 q + 4 kg .. + 3 (clamped; the matching B values are zero)
    d2_f32x4 aw[RT][KQ];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const float* __restrict__ row = w1 + (size_t)min(16 * rt + n, a.hid - 1) * a.cin;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int k0 = 16 * q + 4 * kg;
            if constexpr (AL == 4) aw[rt][q] = d2_load4<4>(row + min(k0, kq_max));
            else {
                const d2_f32x2 lo = *reinterpret_cast<const d2_f32x2*>(row + min(k0, kq_max));
                const d2_f32x2 hi = *reinterpret_cast<const d2_f32x2*>(row + min(k0 + 2, kq_max));
                aw[rt][q] = d2_f32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
    }
    // ---- this thread's pixel of the region and channel group; its skip values (all requested now)
    const int ncoord = 2 * s.coords;
    const int p = tid % P::NPX, cg = __builtin_amdgcn_readfirstlane(tid / P::NPX);
    const int ry = p / P::RW, rx = p - ry * P::RW;
    const int Y = pi * PW + ry, X = min((int)blockIdx.x * P::RW + rx, s.W - 1);     // columns of patches beyond the grid: any pixel
    float* __restrict__ xp = xl + (ry * P::ROWP + rx) * P::XS;
    float vs[P::MAXS];
    {
        const float* __restrict__ sk = s.skip + (size_t)b * s.c_skip * s.H * s.W + (size_t)Y * s.W + X;
#pragma unroll
        for (int i = 0; i < P::MAXS; ++i) vs[i] = sk[(size_t)min(cg + P::CG * i, s.c_skip - 1) * s.H * s.W];
    }
    // ---- the previous level's window
    const int wy0 = pi * (PW / 2) - 1, wx0 = blockIdx.x * (2 * PW) - 1;
    {
        const float* __restrict__ pv = s.prev + (size_t)b * s.c_prev * s.Hp * s.Wp;
        const int total = s.c_prev * P::WP;
        for (int base = 0; base < total; base += 1024) {                // 4 requests in flight per thread and round trip
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = min(base + tid + 256 * i, total - 1);
                const int ch = e / P::WP, r = e - ch * P::WP, wy = r / P::WCOLS, wx = r - wy * P::WCOLS;
                const int gy = min(max(wy0 + wy, 0), s.Hp - 1), gx = min(max(wx0 + wx, 0), s.Wp - 1);
                v[i] = pv[((size_t)ch * s.Hp + gy) * s.Wp + gx];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (base + tid + 256 * i < total) win[base + tid + 256 * i] = v[i];
        }
    }
    // coords, skip and the zero tail do not need the window
    if (s.coords) {
        if (cg == 0) xp[0] = linspace_pm1(X, s.W, s.step_x);
        if (cg == 1 % P::CG) xp[1] = linspace_pm1(Y, s.H, s.step_y);
    }
#pragma unroll
    for (int i = 0; i < P::MAXS; ++i)
        if (cg + P::CG * i < s.c_skip) xp[ncoord + cg + P::CG * i] = vs[i];
    for (int c = a.cin + cg; c < P::KP; c += P::CG) xp[c] = 0.0f;
    __syncthreads();
    {
        const Tap ty = bilinear_tap(Y, s.scale_y, s.Hp), tx = bilinear_tap(X, s.scale_x, s.Wp);
        const int r0 = (ty.i0 - wy0) * P::WCOLS - wx0, r1 = (ty.i1 - wy0) * P::WCOLS - wx0;
        const int o00 = r0 + tx.i0, o01 = r0 + tx.i1, o10 = r1 + tx.i0, o11 = r1 + tx.i1;
        float* __restrict__ xq = xp + ncoord + s.c_skip;
#pragma unroll 4
        for (int cp = cg; cp < s.c_prev; cp += P::CG) {
            const float* pl = win + cp * P::WP;
            const float top = tx.l0 * pl[o00] + tx.l1 * pl[o01], bot = tx.l0 * pl[o10] + tx.l1 * pl[o11];
            xq[cp] = ty.l0 * top + ty.l1 * bot;                          // stage_value's expression (hs_common.h)
        }
    }
    __syncthreads();
    if (pj >= a.fw) return;                                             // no barrier below: a wave beyond the grid leaves here
    constexpr int RTB = NT > 1 ? RT : 1;                                // several tiles: the BatchNorm rows of the lane's channels
    float s1v[RTB][4], b1v[RTB][4];                                     // 16 rt + 4 kg + r are fetched once (one tile: where they are used)
    if constexpr (NT > 1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = min(16 * rt + 4 * kg + r, a.hid - 1);
                s1v[rt][r] = a.s1[c]; b1v[rt][r] = a.b1[c];
            }
    }

#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int ly, lx;
        d2_pixel<PW>(t, n, ly, lx);
        const float* __restrict__ xb = xl + (ly * P::ROWP + wave * PW + lx) * P::XS + 4 * kg;
        d2_f32x4 xv[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) xv[q] = *reinterpret_cast<const d2_f32x4*>(xb + 16 * q);
        d2_f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = d2_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[rt][q][j], xv[q][j], acc[rt], 0, 0, 0);
        // ---- BN1 + ReLU6, channels-last: D row 4 kg + r of tile rt = channel 16 rt + 4 kg + r of pixel n
        float* __restrict__ dst = a.h1 + ((size_t)(b * s.H + pi * PW + ly) * s.W + pj * PW + lx) * a.hidp + 4 * kg;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            d2_f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sc, sh;
                if constexpr (NT > 1) { sc = s1v[rt][r]; sh = b1v[rt][r]; }
                else { const int c = min(16 * rt + 4 * kg + r, a.hid - 1); sc = a.s1[c]; sh = a.b1[c]; }
                const float v = fminf(fmaxf(fmaf(acc[rt][r], sc, sh), 0.0f), 6.0f);
                o[r] = 16 * rt + 4 * kg + r < a.hid ? v : 0.0f;
            }
            if (16 * rt < a.hidp) *reinterpret_cast<d2_f32x4*>(dst + 16 * rt) = o;
        }
    }
}

// KQ 16-deep blocks of hidden channels (hidp = 16 KQ); cout <= 16.  Workgroup = the 4 patches of pass 1 = a 4 PW x PW pixel region.
// Per k block the region's halo (reflect-mapped at the image border) of 16 channels -- 64 contiguous bytes per pixel in the
// channels-last map -- is staged into LDS by the whole workgroup with fully used loads (the first form had every lane fetch its own
// nine neighbours: 16 different 64-byte segments per load instruction, 120 16-byte loads per lane, and the L1 return path alone
// was 26 us of its 76), together with each patch's 144 taps of the block; both double-buffered one block ahead in registers.
// LDS position stride 20 floats (5 granules: odd) and row pitch == PW (mod 16) positions: the 16 pixels of a tile read 16 distinct
// granules (tools/lds_conflicts.py model).
template <int PW> struct IrdP2 {
    static constexpr int RW = 4 * PW, HWW = RW + 2, HH = PW + 2;
    static constexpr int ROWP = PW == 4 ? 20 : 40;                       // >= HWW, == PW (mod 16)
    static constexpr int PS = 20;                                        // floats per position (16 channels + 4 pad)
    static constexpr int HBUF = HH * ROWP * PS;                          // floats per halo buffer
    static constexpr int NE = HH * HWW * 4;                              // 16-byte pieces per k block
    static constexpr int NLD = (NE + 255) / 256;                         // per thread
    static constexpr int TBUF = 4 * 4 * 36;                              // taps: [wave][kg][36]
    static constexpr int FLOATS = 2 * HBUF + 2 * TBUF;
};

template <int PW, int KQ>
__global__ __launch_bounds__(256)
void ird_dw_pw3_kernel(IrdArgs a) {
    using P = IrdP2<PW>;
    constexpr int NT = PW * PW / 16;
    extern __shared__ __attribute__((aligned(16))) float d2_lds[];
    float* hbuf = d2_lds;                       // [2][HH][ROWP][PS]
    float* tbuf = d2_lds + 2 * P::HBUF;         // [2][wave][kg][36]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int pj = blockIdx.x * 4 + wave, pi = blockIdx.y, b = blockIdx.z;
    const bool live = pj < a.fw;                                        // block barriers below: a dead wave stays, its stores do not
    const int H = a.in.H, W = a.in.W;
    const float* __restrict__ wt = a.bank + (size_t)((b * a.fh + pi) * a.fw + min(pj, a.fw - 1)) * (size_t)a.ld;
    const float* __restrict__ taps = wt + (size_t)a.cin * a.hid;
    const float* __restrict__ w3 = taps + 9 * (size_t)a.hid + (size_t)min(n, a.cout - 1) * a.hid;     // this lane's A row
    // ---- staging map: piece e = (halo pixel, 16-byte part); the pixel's image position through reflect padding
    const float* __restrict__ hb = a.h1 + (size_t)b * H * W * a.hidp;
    unsigned gofs[P::NLD], lofs[P::NLD];
#pragma unroll
    for (int i = 0; i < P::NLD; ++i) {
        const int e = min(tid + 256 * i, P::NE - 1);
        const int px = e >> 2, part = e & 3, hy = px / P::HWW, hx = px - hy * P::HWW;
        int Y = pi * PW + hy - 1, X = blockIdx.x * P::RW + hx - 1;
        Y = Y < 0 ? -Y : (Y > H - 1 ? 2 * (H - 1) - Y : Y);
        X = X < 0 ? -X : (X > W - 1 ? 2 * (W - 1) - X : X);
        X = min(max(X, 0), W - 1);                                      // columns of patches beyond the grid (fw % 4 != 0): any pixel
        gofs[i] = (unsigned)(Y * W + X) * (unsigned)a.hidp + 4 * part;
        lofs[i] = (hy * P::ROWP + hx) * P::PS + 4 * part;
    }
    auto fetch_h = [&](int q, d2_f32x4 (&r)[P::NLD]) {
#pragma unroll
        for (int i = 0; i < P::NLD; ++i) r[i] = *reinterpret_cast<const d2_f32x4*>(hb + gofs[i] + 16 * q);
    };
    auto store_h = [&](int buf, const d2_f32x4 (&r)[P::NLD]) {
#pragma unroll
        for (int i = 0; i < P::NLD; ++i)
            if (tid + 256 * i < P::NE) *reinterpret_cast<d2_f32x4*>(hbuf + buf * P::HBUF + lofs[i]) = r[i];
    };
    // taps of block q for this lane's channel quad: 36 consecutive floats; lane n fetches piece min(n, 8)
    auto tap_src = [&](int q) { return taps + 9 * min(16 * q + 4 * kg, a.hid - 4) + 4 * min(n, 8); };
    float* tdst = tbuf + (wave * 4 + kg) * 36 + 4 * min(n, 8);

    // per-block operands that stay in registers: the lane's W3 quad (A fragment) and BatchNorm-2 rows, fetched one block ahead too
    auto fetch_w = [&](int q, d2_f32x4& w, d2_f32x4& sc, d2_f32x4& sh) {
        const int c4 = min(16 * q + 4 * kg, a.hid - 4);                // hid % 4 == 0 (host); beyond it h2 is masked to zero
        w = *reinterpret_cast<const d2_f32x4*>(w3 + c4);
        sc = d2_f32x4{a.s2[c4], a.s2[c4 + 1], a.s2[c4 + 2], a.s2[c4 + 3]};
        sh = d2_f32x4{a.b2[c4], a.b2[c4 + 1], a.b2[c4 + 2], a.b2[c4 + 3]};
    };
    d2_f32x4 hr[P::NLD], tr, w3n, s2n, b2n;
    fetch_h(0, hr);
    tr = *reinterpret_cast<const d2_f32x4*>(tap_src(0));
    fetch_w(0, w3n, s2n, b2n);
    // the 3 x 3 neighbourhood of the lane's pixel in every tile, as LDS offsets (halo position = region position + 1)
    unsigned pos[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int ly, lx;
        d2_pixel<PW>(t, n, ly, lx);
        pos[t] = (ly * P::ROWP + wave * PW + lx) * P::PS + 4 * kg;      // tap (dy, dx) adds (dy ROWP + dx) PS
    }
    store_h(0, hr);
    if (n < 9) *reinterpret_cast<d2_f32x4*>(tdst) = tr;
    __syncthreads();

    d2_f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = d2_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const d2_f32x4 w3v = w3n, s2v = s2n, b2v = b2n;
        if (q + 1 < KQ) {                                              // next block's global requests first ...
            fetch_h(q + 1, hr);
            tr = *reinterpret_cast<const d2_f32x4*>(tap_src(q + 1));
            fetch_w(q + 1, w3n, s2n, b2n);
            __builtin_amdgcn_sched_barrier(0);                         // ... and they stay first (the scheduler sinks them to their use)
        }
        const float* hq = hbuf + (q & 1) * P::HBUF;
        const float* tq = tbuf + (q & 1) * P::TBUF + (wave * 4 + kg) * 36;
        d2_f32x4 tp[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) tp[i] = *reinterpret_cast<const d2_f32x4*>(tq + 4 * i);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            d2_f32x4 hv[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) hv[i] = *reinterpret_cast<const d2_f32x4*>(hq + pos[t] + ((i / 3) * P::ROWP + (i % 3)) * P::PS);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float h = 0.0f;
#pragma unroll
                for (int i = 0; i < 9; ++i) h = fmaf(tp[(9 * j + i) >> 2][(9 * j + i) & 3], hv[i][j], h);   // taps[c][ky][kx]
                h = fminf(fmaxf(fmaf(h, s2v[j], b2v[j]), 0.0f), 6.0f);
                h = 16 * q + 4 * kg + j < a.hid ? h : 0.0f;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3v[j], h, acc[t], 0, 0, 0);
            }
        }
        if (q + 1 < KQ) {
            store_h((q + 1) & 1, hr);
            if (n < 9) *reinterpret_cast<d2_f32x4*>(tdst + ((q + 1) & 1) * P::TBUF) = tr;
            __syncthreads();                                           // one barrier per block: the other buffer was read a block ago
        }
    }
    float s3v[4], b3v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int o = min(4 * kg + r, a.cout - 1); s3v[r] = a.s3[o]; b3v[r] = a.b3[o]; }
    if (!live) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int ly, lx;
        d2_pixel<PW>(t, n, ly, lx);
        float* __restrict__ yb = a.y + (size_t)b * a.cout * H * W + (size_t)(pi * PW + ly) * W + pj * PW + lx;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * kg + r < a.cout) yb[(size_t)(4 * kg + r) * H * W] = fmaf(acc[t][r], s3v[r], b3v[r]);
    }
}

}  // namespace hs

using namespace hs;

// 0 = launched, 1 = not covered (the caller falls through to the single-launch kernels), else an error code.  Coverage: square
// patches of 4 or 8 pixels, coords + skip + bilinear prev at exactly half the resolution, cin <= 48 and even, hid <= 96 and % 4 == 0, cout <= 16, no residual
// (cin != cout), 16-byte aligned bank rows.  workspace: B H W hidp floats (hs_patch_ir_v0_workspace).
size_t hs::ird_workspace_bytes(const StageIn& si, int fh, int fw, int cin, int hid, int c_out) {
    if (fh <= 0 || fw <= 0 || si.H % fh || si.W % fw) return 0;
    const int ph = si.H / fh, pw = si.W / fw;
    if (ph != pw || (pw != 4 && pw != 8)) return 0;
    if (!si.coords || si.prev_mode != HS_PREV_BILINEAR || si.c_skip < 1 || si.c_prev < 1) return 0;
    if (cin > 48 || (cin & 1) || hid > 96 || (hid & 3) || hid < 4 || c_out > 16 || cin == c_out || cin < 2) return 0;
    if (si.H < 2 || si.W < 2 || fh > 65535 || si.B > 65535 || si.Hp * 2 != si.H || si.Wp * 2 != si.W || si.c_prev > 64 || si.c_skip > 16) return 0;
    const int hidp = (hid + 15) & ~15;
    return (size_t)si.B * si.H * si.W * hidp * sizeof(float);
}

int hs::try_launch_ird(const StageIn& si, int fh, int fw, const float* bank, long ld, int cin, int hid, int c_out,
                       const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                       float* workspace, size_t workspace_bytes, float* y, hipStream_t stream) {
    const size_t need = ird_workspace_bytes(si, fh, fw, cin, hid, c_out);
    if (need == 0 || !workspace || workspace_bytes < need) return 1;
    if ((ld & 3) || (((size_t)bank) & 15) || (((size_t)workspace) & 15)) return 1;
    const int pw = si.W / fw, hidp = (hid + 15) & ~15;
    IrdArgs a{si, bank, ld, fh, fw, cin, hid, c_out, hidp, s1, b1, s2, b2, s3, b3, workspace, y};
    const dim3 grid((fw + 3) / 4, fh, si.B), block(256);
    const int al = (cin & 3) == 0 ? 4 : 2;                              // cin even (hid = expand * cin is a multiple of 4)
    const bool big = hid > 48 || cin > 32;                              // (RT, KQ) = (6, 3), else (3, 2)
    // beyond 64 KiB of dynamic LDS (8 x 8 patches with wide inputs; pass 2 at 8 x 8: 69 KB) the kernel's ceiling is raised once per device
#define HS_D2A(PWV, RTV, KQV, ALV) do { \
        const size_t lds1 = ((size_t)IrdP1<PWV, KQV>::XL + (size_t)si.c_prev * IrdP1<PWV, KQV>::WP) * sizeof(float); \
        if (lds1 > 64 * 1024) { \
            static std::atomic<unsigned long long> done{0}; \
            const int e = allow_full_lds((const void*)ird_pw1_kernel<PWV, RTV, KQV, ALV>, done); \
            if (e != HS_OK) return e; \
        } \
        hipLaunchKernelGGL((ird_pw1_kernel<PWV, RTV, KQV, ALV>), grid, block, lds1, stream, a); } while (0)
#define HS_D2B(PWV, RTV, KQV) do { if (al == 4) HS_D2A(PWV, RTV, KQV, 4); else HS_D2A(PWV, RTV, KQV, 2); } while (0)
    if (pw == 4) { if (big) HS_D2B(4, 6, 3); else HS_D2B(4, 3, 2); }
    else { if (big) HS_D2B(8, 6, 3); else HS_D2B(8, 3, 2); }
#undef HS_D2B
#undef HS_D2A
    int st = launch_status();
    if (st != HS_OK) return st;
#define HS_D2C(PWV, KQV) do { \
        constexpr size_t lds2 = IrdP2<PWV>::FLOATS * sizeof(float); \
        if (lds2 > 64 * 1024) { \
            static std::atomic<unsigned long long> done{0}; \
            const int e = allow_full_lds((const void*)ird_dw_pw3_kernel<PWV, KQV>, done); \
            if (e != HS_OK) return e; \
        } \
        hipLaunchKernelGGL((ird_dw_pw3_kernel<PWV, KQV>), grid, block, lds2, stream, a); } while (0)
    const int kq = hidp / 16;
    if (pw == 4) { switch (kq) { case 1: HS_D2C(4, 1); break; case 2: HS_D2C(4, 2); break; case 3: HS_D2C(4, 3); break; case 4: HS_D2C(4, 4); break;
                                 case 5: HS_D2C(4, 5); break; default: HS_D2C(4, 6); break; } }
    else { switch (kq) { case 1: HS_D2C(8, 1); break; case 2: HS_D2C(8, 2); break; case 3: HS_D2C(8, 3); break; case 4: HS_D2C(8, 4); break;
                         case 5: HS_D2C(8, 5); break; default: HS_D2C(8, 6); break; } }
#undef HS_D2C
    return launch_status();
}

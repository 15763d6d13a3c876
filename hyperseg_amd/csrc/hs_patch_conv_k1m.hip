// Op A, k = 1, for BATCHED 2 x 2-pixel patches: HyperSeg-L's level 1 at batch 32 (132 -> 34 channels on 32 x 32 maps) -- 8192 patches per
// launch, each with its own 18 KB of weights for 4 pixels: a pure weight stream (147 MB per batch), which the LDS-staged kernel of
// hs_patch_conv.hip (built for batch 1: one workgroup per patch, bank -> LDS -> dot products) moved at 1.8 TB/s (85 us).
//
// Here, as in hs_patch_ir_d2.hip pass 1: wave = patch, the bank row goes straight from HBM into A fragments of
// v_mfma_f32_16x16x4_f32 (a lane fetches 4 consecutive k of its output row per 16-byte load; MFMA j multiplies k-set
// {16 q + 4 kgroup + j}), 16 k per step, next step's fragments requested before this step's MFMAs; the 16-column N dimension holds the
// patch's 4 pixels (the rest of the tile is idle: the f32 matrix pipe has 10x the headroom this stream needs, the point is that
// no weight takes a detour through LDS and no lane does index arithmetic per weight).  The stage input cat(coords, skip,
// bilinear2x(prev)) of the workgroup's 4 patches is assembled once into LDS, branch-free (every candidate load of an element issued, the
// kind selected afterwards: a handful of elements per thread).  BatchNorm + activation in the epilogue, NCHW stores.  Exact f32.
// 85 -> 54 us (2.7 TB/s).  The same kernel on level 0 (ONE pixel per patch, 98 -> 96 channels, 8-byte aligned 392-byte rows) measured
// 126-132 us against the staged kernel's 82, with 16 or 32 k per step alike (profiles/round3_L_k1m.txt): kept OFF there (PW = 1 is still
// instantiable for a future look; the dispatch below takes 2 x 2 only).
#include "hs_common.h"

namespace hs {

using k1m_f32x4 = __attribute__((ext_vector_type(4))) float;
using k1m_f32x2 = __attribute__((ext_vector_type(2))) float;

struct K1mArgs {
    StageIn in;
    const float* __restrict__ bank;
    long ld;
    int fh, fw, cin, cout, kp;             // kp = cin rounded up to 16
    const float* __restrict__ scale; const float* __restrict__ shift;
    int act;
    float* __restrict__ y;
};

#ifndef HS_K1M_MIN_PATCHES
#define HS_K1M_MIN_PATCHES 1024        // below: the batch-1 kernel of hs_patch_conv.hip (one workgroup per patch, 512 patches at HyperSeg-M);
                                       // a bs-4 shard of HyperSeg-L (1024) takes the same kernel as the bs-32 batch: outputs stay shard-invariant
#endif
constexpr int K1M_MAXC = 144;              // channels (rounded up to 16) the staging loop is unrolled for

template <int PW, int RT, int AL, bool PREV>
__global__ __launch_bounds__(256)
void patch_conv_k1m_kernel(K1mArgs a) {
    constexpr int NPX = PW * PW, NP4 = 4 * NPX, K1M_MAXE = (NP4 * K1M_MAXC + 255) / 256;    // staged elements per thread
    extern __shared__ __attribute__((aligned(16))) float k1m_x[];         // [4 patches x NPX pixels][kp + 4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int pj = blockIdx.x * 4 + wave, pi = blockIdx.y, b = blockIdx.z;
    const StageIn& s = a.in;
    const int xs = a.kp + 4;                                            // floats per pixel: an odd number of 16-byte granules
    const float* __restrict__ wrow[RT];
    {
        const float* __restrict__ w = a.bank + (size_t)((b * a.fh + pi) * a.fw + min(pj, a.fw - 1)) * (size_t)a.ld;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) wrow[rt] = w + (size_t)min(16 * rt + n, a.cout - 1) * a.cin;
    }
    // 16 k per step.  (32 k per step -- the four lanes of a row fetching 128 contiguous bytes -- measured SLOWER: 60 vs 54 us at
    // level 1, profiles/round3_L_k1m.txt: the half lines left for the next step do survive in L1.)
    const int kmax = a.cin - AL;                                        // last k a vector piece may start at
    auto load_a = [&](int q, k1m_f32x4 (&f)[RT]) {
        const int k0 = 16 * q + 4 * kg;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if constexpr (AL == 4) f[rt] = *reinterpret_cast<const k1m_f32x4*>(wrow[rt] + min(k0, kmax));
            else {
                const k1m_f32x2 lo = *reinterpret_cast<const k1m_f32x2*>(wrow[rt] + min(k0, kmax));
                const k1m_f32x2 hi = *reinterpret_cast<const k1m_f32x2*>(wrow[rt] + min(k0 + 2, kmax));
                f[rt] = k1m_f32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
    };
    k1m_f32x4 a0[RT], a1[RT];
    load_a(0, a0);                                                      // in flight during the assembly below

    // ---- stage input of the 4 patches: element e = (pixel p, channel c), pixels fastest
    const int ncoord = 2 * s.coords, total = NP4 * a.kp;
    constexpr bool has_prev = PREV;
    const float* __restrict__ skb = s.skip + (size_t)b * s.c_skip * s.H * s.W;
    const float* __restrict__ pvb = has_prev ? s.prev + (size_t)b * s.c_prev * s.Hp * s.Wp : skb;   // never selected without a previous level
    float v_skip[K1M_MAXE], v00[K1M_MAXE], v01[K1M_MAXE], v10[K1M_MAXE], v11[K1M_MAXE];
#pragma unroll
    for (int i = 0; i < K1M_MAXE; ++i) {
        const int e = min(tid + 256 * i, total - 1);
        const int c = e / NP4, p = e - c * NP4, w = p / NPX, l = p - w * NPX;
        const int Y = pi * PW + l / PW, X = min((blockIdx.x * 4 + w) * PW + l % PW, s.W - 1);
        const int cs = min(max(c - ncoord, 0), s.c_skip - 1);
        v_skip[i] = skb[((size_t)cs * s.H + Y) * s.W + X];
        if constexpr (has_prev) {
            const Tap ty = bilinear_tap(Y, s.scale_y, s.Hp), tx = bilinear_tap(X, s.scale_x, s.Wp);
            const float* __restrict__ pl = pvb + (size_t)min(max(c - ncoord - s.c_skip, 0), s.c_prev - 1) * s.Hp * s.Wp;
            v00[i] = pl[ty.i0 * s.Wp + tx.i0]; v01[i] = pl[ty.i0 * s.Wp + tx.i1];
            v10[i] = pl[ty.i1 * s.Wp + tx.i0]; v11[i] = pl[ty.i1 * s.Wp + tx.i1];
        } else {
            v00[i] = v01[i] = v10[i] = v11[i] = 0.0f;
        }
    }
#pragma unroll
    for (int i = 0; i < K1M_MAXE; ++i) {
        const int e = tid + 256 * i;
        if (e < total) {
            const int c = e / NP4, p = e - c * NP4, w = p / NPX, l = p - w * NPX;
            const int Y = pi * PW + l / PW, X = min((blockIdx.x * 4 + w) * PW + l % PW, s.W - 1);
            float v;
            if (c < ncoord) v = c == 0 ? linspace_pm1(X, s.W, s.step_x) : linspace_pm1(Y, s.H, s.step_y);
            else if (c - ncoord < s.c_skip) v = v_skip[i];
            else if (has_prev && c < a.cin) {
                const Tap ty = bilinear_tap(Y, s.scale_y, s.Hp), tx = bilinear_tap(X, s.scale_x, s.Wp);
                const float top = tx.l0 * v00[i] + tx.l1 * v01[i], bot = tx.l0 * v10[i] + tx.l1 * v11[i];
                v = ty.l0 * top + ty.l1 * bot;                           // stage_value's expression (hs_common.h)
            } else v = 0.0f;
            k1m_x[p * xs + c] = v;
        }
    }
    __syncthreads();
    if (pj >= a.fw) return;                                             // no barrier below: a wave beyond the grid leaves here

    // ---- the weight stream: 16 k per step, the next step's fragments in flight during this step's products
    const float* __restrict__ xb = k1m_x + (wave * NPX + min(n, NPX - 1)) * xs + 4 * kg;
    k1m_f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = k1m_f32x4{0.f, 0.f, 0.f, 0.f};
    const int kq = a.kp >> 4;
    auto step = [&](int q, const k1m_f32x4 (&f)[RT]) {
        const k1m_f32x4 xv = *reinterpret_cast<const k1m_f32x4*>(xb + 16 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[rt][j], xv[j], acc[rt], 0, 0, 0);
    };
    int q = 0;
    for (; q + 1 < kq; q += 2) {
        load_a(q + 1, a1);
        step(q, a0);
        load_a(min(q + 2, kq - 1), a0);
        step(q + 1, a1);
    }
    if (q < kq) step(q, a0);

    // ---- BatchNorm + activation; D row 4 kg + r of tile rt = output channel 16 rt + 4 kg + r, column n = pixel
    float sc[RT][4], sh[RT][4];                                         // all rows requested together (clamped), used below
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = min(16 * rt + 4 * kg + r, a.cout - 1);
            sc[rt][r] = a.scale ? a.scale[o] : 1.0f;
            sh[rt][r] = a.scale ? a.shift[o] : 0.0f;
        }
    if (n < NPX) {
        const int Y = pi * PW + n / PW, X = pj * PW + n % PW;
        float* __restrict__ yb = a.y + (size_t)b * a.cout * s.H * s.W + (size_t)Y * s.W + X;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * rt + 4 * kg + r;
                if (o < a.cout) yb[(size_t)o * s.H * s.W] = apply_act(fmaf(acc[rt][r], sc[rt][r], sh[rt][r]), a.act);
            }
    }
}

int try_launch_k1m(const StageIn& si, int fh, int fw, const float* bank, long ld, int cin, int c_out,
                   const float* scale, const float* shift, int act, float* y, hipStream_t stream);

}  // namespace hs

using namespace hs;

// 0 = launched, 1 = not covered (hs_patch_conv_fwd goes on to its other forms), else an error code.  Coverage: k = 1, groups = 1,
// square patches of 1 or 2 pixels, at least 1024 patches, even cin <= 144, cout <= 96, a previous level (if any) bilinear at exactly
// half the resolution, 16-byte aligned bank rows.
int hs::try_launch_k1m(const StageIn& si, int fh, int fw, const float* bank, long ld, int cin, int c_out,
                       const float* scale, const float* shift, int act, float* y, hipStream_t stream) {
    if (si.H % fh || si.W % fw) return 1;
    const int ph = si.H / fh, pw = si.W / fw;
    if (ph != pw || pw != 2) return 1;     // one-pixel patches (level 0, 98 -> 96) measured SLOWER here: 126 vs 82 us -- the staged kernel keeps them
    if ((long)si.B * fh * fw < HS_K1M_MIN_PATCHES || si.B > 65535 || fh > 65535) return 1;
    if (cin > K1M_MAXC || (cin & 1) || cin < 4 || c_out > 96 || si.c_skip < 1) return 1;
    if (si.c_prev > 0 && (si.prev_mode != HS_PREV_BILINEAR || si.Hp * 2 != si.H || si.Wp * 2 != si.W)) return 1;
    if ((ld & 3) || (((size_t)bank) & 15)) return 1;
    const int kp = (cin + 15) & ~15;
    if (kp > K1M_MAXC) return 1;
    K1mArgs a{si, bank, ld, fh, fw, cin, c_out, kp, scale, shift, act, y};
    const dim3 grid((fw + 3) / 4, fh, si.B), block(256);
    const size_t lds = (size_t)4 * pw * pw * (kp + 4) * sizeof(float);
    const int al = (cin & 3) == 0 ? 4 : 2;
    const int rt = (c_out + 15) / 16;
#define HS_K1M_(PWV, RTV, ALV) do { if (si.c_prev > 0) hipLaunchKernelGGL((patch_conv_k1m_kernel<PWV, RTV, ALV, true>), grid, block, lds, stream, a); \
                                    else hipLaunchKernelGGL((patch_conv_k1m_kernel<PWV, RTV, ALV, false>), grid, block, lds, stream, a); } while (0)
#define HS_K1M(PWV, RTV) do { if (al == 4) HS_K1M_(PWV, RTV, 4); else HS_K1M_(PWV, RTV, 2); } while (0)
#define HS_K1M_RT(PWV) do { if (rt <= 3) HS_K1M(PWV, 3); else HS_K1M(PWV, 6); } while (0)
    HS_K1M_RT(2);
#undef HS_K1M_RT
#undef HS_K1M
#undef HS_K1M_
    return launch_status();
}

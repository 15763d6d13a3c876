// Op C: the fused per-patch inverted residual of HyperSeg's k=3 decoder levels
// (hyperseg_v1_0.py:328-376) -- the dominant kernel of the decoder (85 % of its FLOPs at
// HyperSeg-M).  One workgroup = one (<=16x16)-pixel tile of one patch, one launch per level:
//
//   prologue   every thread owns ONE position of the (TH+2)x(TW+2) reflect-halo tile and builds
//              its CIN-channel input column in REGISTERS straight from HBM (coords analytic, skip
//              gather, previous level bilinear 2x) -- the concatenated / padded / unfolded input
//              of the reference never exists.
//   pw1        h1[h][pos] = relu6(bn1(sum_c W1[h][c] * T[c])): the patch's weights are uniform
//              over the workgroup, so they arrive through the SCALAR cache (s_load) as SGPR
//              operands of v_fmac -- no LDS traffic and no VGPRs for weights.  h1 goes to LDS
//              in chunks of HC hidden channels (double-buffered, one barrier per chunk).
//   dw + pw3   every thread owns ONE output pixel: 9 LDS reads of h1 per hidden channel,
//              depthwise 3x3 + bn2 + relu6 in registers, then COUT x chunk FMAs into register
//              accumulators with W3[o][chunk] again as scalar operands (natural bank order).
//   epilogue   bn3 (+ residual) and one store per (o, pixel); rows are contiguous.
//
// Hidden activations (hid x 324 floats per patch at HyperSeg-M level 4) never leave the CU.
// Algorithmic HBM bytes per patch: bank (ld*4) + input tile + output tile.
#include "hs_common.h"

namespace hs {

constexpr int IR_MAX_TILE = 16;                       // output tile edge
constexpr int IR_HC = 16;                             // hidden channels per LDS chunk
constexpr int IR_MAX_THREADS = 384;                   // >= 18*18 positions, multiple of 64

struct IrArgs {
    StageIn in;
    int fh, fw, ph, pw;
    const float* __restrict__ bank;
    long ld;
    int cin, hid, cout;
    const float* __restrict__ s1; const float* __restrict__ b1;
    const float* __restrict__ s2; const float* __restrict__ b2;
    const float* __restrict__ s3; const float* __restrict__ b3;
    float* __restrict__ y;
    int TH, TW, tiles_y, tiles_x, residual;
};

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// EXACT instantiations are the decoder's fused form: coords + CSKIP skip channels + (CIN-2-CSKIP) channels of
// the previous level resized bilinearly.  Their prologue issues every HBM load of the column (CSKIP skip loads +
// 4 taps x CPREV) before the first use, so the ~1 us HBM latency is paid once per tile instead of once per channel.
template <int CIN, int CSKIP, int COUT, bool EXACT>
__global__ __launch_bounds__(IR_MAX_THREADS)
void patch_ir_kernel(IrArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x [IR_HC][npos]
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    const int tx_i = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty_i = blk % a.tiles_y; blk /= a.tiles_y;
    const int patch = blk;
    const int j = patch % a.fw, i = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const int cin = EXACT ? CIN : a.cin;
    const int cout = EXACT ? COUT : a.cout;
    const int hid = a.hid;

    const int y0 = i * a.ph + ty_i * a.TH, x0 = j * a.pw + tx_i * a.TW;
    const int th = min(a.TH, (i + 1) * a.ph - y0), tw = min(a.TW, (j + 1) * a.pw - x0);
    const int HW = a.TW + 2, npos = (a.TH + 2) * HW, npix = a.TH * a.TW;

    // ---- prologue: this thread's input column ------------------------------------------------
    const bool has_pos = tid < npos;
    float T[CIN];
    {
        const int pu = tid / HW, pv = tid - pu * HW;
        const bool live = has_pos && pu < th + 2 && pv < tw + 2;
        int yy = pad_index(y0 + pu - 1, a.in.H, HS_PAD_REFLECT);
        int xx = pad_index(x0 + pv - 1, a.in.W, HS_PAD_REFLECT);
        if constexpr (EXACT) {
            constexpr int CPREV = CIN - 2 - CSKIP;
            yy = live ? yy : 0; xx = live ? xx : 0;          // dead lanes read a valid address, result unused
            T[0] = linspace_pm1(xx, a.in.W, a.in.step_x);
            T[1] = linspace_pm1(yy, a.in.H, a.in.step_y);
            const size_t plane = (size_t)a.in.H * a.in.W;
            const float* __restrict__ sp = a.in.skip + (size_t)b * CSKIP * plane + (size_t)yy * a.in.W + xx;
#pragma unroll
            for (int c = 0; c < CSKIP; ++c) T[2 + c] = sp[c * plane];
            const Tap ty = bilinear_tap(yy, a.in.scale_y, a.in.Hp), tx = bilinear_tap(xx, a.in.scale_x, a.in.Wp);
            const size_t pplane = (size_t)a.in.Hp * a.in.Wp;
            const float* __restrict__ pp = a.in.prev + (size_t)b * CPREV * pplane;
            const int o00 = ty.i0 * a.in.Wp + tx.i0, o01 = ty.i0 * a.in.Wp + tx.i1;
            const int o10 = ty.i1 * a.in.Wp + tx.i0, o11 = ty.i1 * a.in.Wp + tx.i1;
            float v00[CPREV], v01[CPREV], v10[CPREV], v11[CPREV];
#pragma unroll
            for (int c = 0; c < CPREV; ++c) {
                const float* __restrict__ q = pp + c * pplane;
                v00[c] = q[o00]; v01[c] = q[o01]; v10[c] = q[o10]; v11[c] = q[o11];
            }
#pragma unroll
            for (int c = 0; c < CPREV; ++c)
                T[2 + CSKIP + c] = ty.l0 * (tx.l0 * v00[c] + tx.l1 * v01[c]) + ty.l1 * (tx.l0 * v10[c] + tx.l1 * v11[c]);
        } else {
            const StagePos sp = stage_pos(a.in, live ? yy : -1, live ? xx : -1);
#pragma unroll
            for (int c = 0; c < CIN; ++c) T[c] = (c < cin) ? stage_value(a.in, b, c, sp) : 0.0f;
        }
    }
    const bool has_pix = tid < npix;
    const int u = tid / a.TW, v = tid - u * a.TW;
    const int dw_off = u * HW + v;

    const float* __restrict__ wp = a.bank + (size_t)patch * a.ld;   // uniform -> scalar loads
    const float* __restrict__ w1 = wp;
    const float* __restrict__ kd = wp + (size_t)cin * hid;
    const float* __restrict__ w3 = kd + (size_t)9 * hid;

    float out[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) out[o] = 0.0f;

    auto pw1_chunk = [&](int h0, float* buf) {
        if (!has_pos) return;
        const int hc = min(IR_HC, hid - h0);
        for (int hh = 0; hh < hc; ++hh) {
            const int h = h0 + hh;
            const float* __restrict__ wr = w1 + (size_t)h * cin;
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < CIN; ++c)
                if (EXACT || c < cin) acc = fmaf(wr[c], T[c], acc);
            buf[hh * npos + tid] = relu6f(fmaf(acc, a.s1[h], a.b1[h]));
        }
    };
    auto dw_pw3_chunk = [&](int h0, const float* buf) {
        if (!has_pix) return;
        const int hc = min(IR_HC, hid - h0);
        float h2c[IR_HC];                       // depthwise outputs of the chunk for this pixel
#pragma unroll
        for (int hh = 0; hh < IR_HC; ++hh) {
            h2c[hh] = 0.0f;
            if (hh < hc) {
                const int h = h0 + hh;
                const float* __restrict__ kr = kd + (size_t)h * 9;
                const float* t = buf + hh * npos + dw_off;
                float d = 0.0f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) d = fmaf(kr[ky * 3 + kx], t[ky * HW + kx], d);
                h2c[hh] = relu6f(fmaf(d, a.s2[h], a.b2[h]));
            }
        }
        // pw3 in the natural bank order W3[o][h]: a contiguous run of the chunk's weights per output channel
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            if (EXACT || o < cout) {
                const float* __restrict__ w3r = w3 + (size_t)o * hid + h0;
#pragma unroll
                for (int hh = 0; hh < IR_HC; ++hh)
                    if (hh < hc) out[o] = fmaf(w3r[hh], h2c[hh], out[o]);
            }
        }
    };

    float* buf0 = lds;
    float* buf1 = lds + IR_HC * npos;
    pw1_chunk(0, buf0);
    __syncthreads();
    int cur = 0;
    for (int h0 = 0; h0 < hid; h0 += IR_HC) {
        float* bc = cur ? buf1 : buf0;
        float* bn = cur ? buf0 : buf1;
        if (h0 + IR_HC < hid) pw1_chunk(h0 + IR_HC, bn);
        dw_pw3_chunk(h0, bc);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue --------------------------------------------------------------------------
    if (has_pix && u < th && v < tw) {
        const int yy = y0 + u, xx = x0 + v;
        StagePos sp;
        if (a.residual) sp = stage_pos(a.in, yy, xx);
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            if (EXACT || o < cout) {
                float r = fmaf(out[o], a.s3[o], a.b3[o]);
                if (a.residual) r += stage_value(a.in, b, o, sp);
                a.y[(((size_t)b * cout + o) * a.in.H + yy) * a.in.W + xx] = r;
            }
        }
    }
}

template <int CIN, int CSKIP, int COUT, bool EXACT>
static int launch_ir(const IrArgs& a, int threads, size_t lds, long blocks, hipStream_t stream) {
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};       // one per instantiation
        const int e = allow_full_lds((const void*)patch_ir_kernel<CIN, CSKIP, COUT, EXACT>, done);
        if (e != HS_OK) return e;
    }
    hipLaunchKernelGGL((patch_ir_kernel<CIN, CSKIP, COUT, EXACT>), dim3((unsigned)blocks), dim3(threads), lds, stream, a);
    return launch_status();
}

int try_launch_ir_fused(int mode, const StageIn& in, int fh, int fw, const float* bank, long ld, int cin, int c_skip,
                        int hid, int c_out, const float* s1, const float* b1, const float* s2, const float* b2,
                        const float* s3, const float* b3, float* y, hipStream_t stream);   // hs_patch_ir_fused.hip
int try_launch_irc(const StageIn& in, int fh, int fw, const float* bank, long ld, int hid, int c_out,
                   const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                   float* y, hipStream_t stream);                                           // hs_patch_irc.hip
size_t ird_workspace_bytes(const StageIn& si, int fh, int fw, int cin, int hid, int c_out);  // hs_patch_ir_d2.hip
int try_launch_ird(const StageIn& si, int fh, int fw, const float* bank, long ld, int cin, int hid, int c_out,
                   const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                   float* workspace, size_t workspace_bytes, float* y, hipStream_t stream);

}  // namespace hs

using namespace hs;

// HS_IR_MATH_AUTO = "the faster form".  Measured (round 6, profiles/round6_auto_math_narrow_shapes_v17.txt): for the narrow blocks of
// CamVid HyperSeg-L -- <= 4 skip channels, so most of the split kernel's per-region cost is its fixed prologue -- in launches of more than
// one generation of 16 x 16 regions (> 2 per CU) the exact-f32 matrix-core kernel wins (level 5, 21 -> 42 -> 12 on 3072 regions: 81.4 vs
// 88-90 us; level 4, 22 -> 44 -> 16 on 768: 25.4 vs 26.9), while the same width in ONE generation (CamVid-S level 4, 432 regions: 18.3-18.7 vs
// 19.8) and every wider shape (HyperSeg-M 25.2 vs 31.6, HyperSeg-S 43.3 vs 49.8) stay on the split form.
static bool ir_auto_prefers_f32(int math, int c_skip, long patches, int ph, int pw) {
    if (math != HS_IR_MATH_AUTO || c_skip > 4 || ph % 16 != 0 || pw % 16 != 0) return false;
    return patches * (ph / 16) * (pw / 16) > 2 * 256;
}

extern "C" int hs_patch_ir_fwd(const hs_stage_input* in, int32_t fh, int32_t fw, const float* bank, int64_t ld,
                               int32_t hidden, int32_t c_out, const hs_epilogue* bn1, const hs_epilogue* bn2,
                               const hs_epilogue* bn3, int32_t residual, int32_t math, float* y, void* stream) {
    if (math < HS_IR_MATH_AUTO || math > HS_IR_MATH_SPLIT) return HS_ERR_BAD_ARG;
    IrArgs a;
    int st = make_stage(in, &a.in);
    if (st != HS_OK) return st;
    if (!bank || !y || !bn1 || !bn2 || !bn3 || fh <= 0 || fw <= 0 || hidden <= 0 || c_out <= 0) return HS_ERR_BAD_ARG;
    if (!bn1->scale || !bn1->shift || !bn2->scale || !bn2->shift || !bn3->scale || !bn3->shift) return HS_ERR_BAD_ARG;
    if (in->H % fh != 0 || in->W % fw != 0) return HS_ERR_NOT_DIVISIBLE;
    if (in->H < 2 || in->W < 2) return HS_ERR_BAD_ARG;           // reflect padding by 1 needs >= 2 pixels
    a.fh = fh; a.fw = fw; a.ph = in->H / fh; a.pw = in->W / fw;
    a.bank = bank; a.ld = ld;
    a.cin = a.in.cin(); a.hid = hidden; a.cout = c_out;
    if (ld < (int64_t)a.cin * hidden + 9 * hidden + (int64_t)hidden * c_out) return HS_ERR_BAD_ARG;
    a.s1 = bn1->scale; a.b1 = bn1->shift; a.s2 = bn2->scale; a.b2 = bn2->shift; a.s3 = bn3->scale; a.b3 = bn3->shift;
    a.y = y;
    if (residual && a.cin != c_out) return HS_ERR_BAD_ARG;        // use_res_connect (hyperseg_v1_0.py:295)
    a.residual = residual ? 1 : 0;
    a.TH = a.ph > IR_MAX_TILE ? IR_MAX_TILE : a.ph;
    a.TW = a.pw > IR_MAX_TILE ? IR_MAX_TILE : a.pw;
    a.tiles_y = (a.ph + a.TH - 1) / a.TH;
    a.tiles_x = (a.pw + a.TW - 1) / a.TW;
    const int npos = (a.TH + 2) * (a.TW + 2);
    const int threads = ((npos + kWave - 1) / kWave) * kWave;
    const size_t lds = (size_t)2 * IR_HC * npos * sizeof(float);
    const long blocks = (long)in->batch * fh * fw * a.tiles_y * a.tiles_x;
    hipStream_t s = (hipStream_t)stream;
    const bool fused_form = in->coords && a.in.prev_mode == HS_PREV_BILINEAR && !a.residual;
    if (fused_form && ir_auto_prefers_f32(math, in->c_skip, (long)in->batch * fh * fw, a.ph, a.pw)) {
        const int st_m = try_launch_ir_fused(0, a.in, fh, fw, bank, (long)ld, a.cin, in->c_skip, hidden, c_out,
                                             a.s1, a.b1, a.s2, a.b2, a.s3, a.b3, y, s);
        if (st_m != 1) return st_m;
    }
    if (fused_form && math != HS_IR_MATH_F32) {
        // f16 matrix cores on split operands: any channel counts up to 16 + 16 -> 32 on patches >= 8 x 16 pixels
        const int st_c = try_launch_irc(a.in, fh, fw, bank, (long)ld, hidden, c_out, a.s1, a.b1, a.s2, a.b2, a.s3, a.b3, y, s);
        if (st_c != 1) return st_c;
    }
    if (fused_form) {
        // exact f32 matrix cores at the decoder's own shapes; anything else falls through to the generic kernel
        const int st_m = try_launch_ir_fused(0, a.in, fh, fw, bank, (long)ld, a.cin, in->c_skip, hidden, c_out,
                                             a.s1, a.b1, a.s2, a.b2, a.s3, a.b3, y, s);
        if (st_m != 1) return st_m;
    }
#define HS_IR_CASE(CI, CS, CO) \
    if (fused_form && a.cin == CI && in->c_skip == CS && c_out == CO) return launch_ir<CI, CS, CO, true>(a, threads, lds, blocks, s);
    HS_IR_CASE(24, 6, 16)    // HyperSeg-M / CamVid-S level 3
    HS_IR_CASE(34, 16, 19)   // HyperSeg-M level 4 (Cityscapes, 19 classes)
    HS_IR_CASE(14, 4, 8)     // HyperSeg-S level 3
    HS_IR_CASE(26, 16, 19)   // HyperSeg-S level 4
    HS_IR_CASE(22, 4, 12)    // CamVid-S level 4 (12 classes)
#undef HS_IR_CASE
    if (a.cin <= 16 && c_out <= 16) return launch_ir<16, 0, 16, false>(a, threads, lds, blocks, s);
    if (a.cin <= 32 && c_out <= 32) return launch_ir<32, 0, 32, false>(a, threads, lds, blocks, s);
    if (a.cin <= 64 && c_out <= 32) return launch_ir<64, 0, 32, false>(a, threads, lds, blocks, s);
    if (a.cin <= 128 && c_out <= 64) return launch_ir<128, 0, 64, false>(a, threads, lds, blocks, s);
    return HS_ERR_UNSUPPORTED;
}

// Which kernel hs_patch_ir_fwd would run (host only): the same dispatch with nothing launched.
extern "C" int hs_patch_ir_route(const hs_stage_input* in, int32_t fh, int32_t fw, int32_t hidden, int32_t c_out,
                                 int32_t residual, int32_t math) {
    if (math < HS_IR_MATH_AUTO || math > HS_IR_MATH_SPLIT) return HS_ERR_BAD_ARG;
    hs_stage_input probe = *in;
    alignas(16) static const float dummy = 0.0f;                 // make_stage only checks that the pointers exist; 16-byte aligned like a real bank
    if (probe.c_skip > 0 && !probe.skip) probe.skip = &dummy;
    if (probe.c_prev > 0 && !probe.prev) probe.prev = &dummy;
    StageIn si;
    const int st = make_stage(&probe, &si);
    if (st != HS_OK) return st;
    if (fh <= 0 || fw <= 0 || hidden <= 0 || c_out <= 0) return HS_ERR_BAD_ARG;
    if (in->H % fh != 0 || in->W % fw != 0) return HS_ERR_NOT_DIVISIBLE;
    const int cin = si.cin();
    const long ld = (((long)cin * hidden + 9L * hidden + (long)hidden * c_out) + 3) & ~3L;
    const bool fused_form = in->coords && si.prev_mode == HS_PREV_BILINEAR && !residual;
    if (fused_form && ir_auto_prefers_f32(math, in->c_skip, (long)in->batch * fh * fw, in->H / fh, in->W / fw) &&
        try_launch_ir_fused(0, si, fh, fw, &dummy, ld, cin, in->c_skip, hidden, c_out, &dummy, &dummy, &dummy, &dummy,
                            &dummy, &dummy, nullptr, nullptr) == HS_OK)
        return HS_IR_ROUTE_F32_MFMA;
    if (fused_form && math != HS_IR_MATH_F32 &&
        try_launch_irc(si, fh, fw, &dummy, ld, hidden, c_out, &dummy, &dummy, &dummy, &dummy, &dummy, &dummy, nullptr, nullptr) == HS_OK)
        return HS_IR_ROUTE_SPLIT_MFMA;
    if (fused_form && try_launch_ir_fused(0, si, fh, fw, &dummy, ld, cin, in->c_skip, hidden, c_out, &dummy, &dummy, &dummy, &dummy,
                                          &dummy, &dummy, nullptr, nullptr) == HS_OK)
        return HS_IR_ROUTE_F32_MFMA;
    return HS_IR_ROUTE_GENERIC;
}

// Op D, fused form only (the decoder's own shapes); anything else returns HS_ERR_UNSUPPORTED and the caller runs the
// block as three hs_patch_conv_fwd launches (pw1, depthwise, pw3), which is what the reference's module structure is.
extern "C" int64_t hs_patch_ir_v0_workspace(const hs_stage_input* in, int32_t fh, int32_t fw, int32_t hidden, int32_t c_out) {
    StageIn si;
    if (make_stage(in, &si) != HS_OK || fh <= 0 || fw <= 0 || hidden <= 0 || c_out <= 0) return 0;
    return (int64_t)ird_workspace_bytes(si, fh, fw, si.cin(), hidden, c_out);
}

extern "C" int hs_patch_ir_v0_fwd(const hs_stage_input* in, int32_t fh, int32_t fw, const float* bank, int64_t ld,
                                  int32_t hidden, int32_t c_out, const hs_epilogue* bn1, const hs_epilogue* bn2,
                                  const hs_epilogue* bn3, int32_t math, float* y, void* stream) {
    return hs_patch_ir_v0_ws_fwd(in, fh, fw, bank, ld, hidden, c_out, bn1, bn2, bn3, math, nullptr, 0, y, stream);
}

extern "C" int hs_patch_ir_v0_ws_fwd(const hs_stage_input* in, int32_t fh, int32_t fw, const float* bank, int64_t ld,
                                     int32_t hidden, int32_t c_out, const hs_epilogue* bn1, const hs_epilogue* bn2,
                                     const hs_epilogue* bn3, int32_t math, float* workspace, int64_t workspace_bytes, float* y,
                                     void* stream) {
    if (math < HS_IR_MATH_AUTO || math > HS_IR_MATH_SPLIT) return HS_ERR_BAD_ARG;   // Op D: every mode runs the exact-f32 kernels
    if (workspace_bytes < 0 || (workspace_bytes > 0 && !workspace)) return HS_ERR_BAD_ARG;
    StageIn si;
    int st = make_stage(in, &si);
    if (st != HS_OK) return st;
    if (!bank || !y || !bn1 || !bn2 || !bn3 || fh <= 0 || fw <= 0 || hidden <= 0 || c_out <= 0) return HS_ERR_BAD_ARG;
    if (!bn1->scale || !bn1->shift || !bn2->scale || !bn2->shift || !bn3->scale || !bn3->shift) return HS_ERR_BAD_ARG;
    if (in->H % fh != 0 || in->W % fw != 0) return HS_ERR_NOT_DIVISIBLE;
    if (in->H < 2 || in->W < 2) return HS_ERR_BAD_ARG;
    const int cin = si.cin();
    if (ld < (int64_t)cin * hidden + 9 * hidden + (int64_t)hidden * c_out) return HS_ERR_BAD_ARG;
    if (!(in->coords && si.prev_mode == HS_PREV_BILINEAR)) return HS_ERR_UNSUPPORTED;
    if (workspace) {                       // small patches: two launches through the caller's hidden map (hs_patch_ir_d2.hip)
        const int rd = try_launch_ird(si, fh, fw, bank, (long)ld, cin, hidden, c_out, bn1->scale, bn1->shift, bn2->scale, bn2->shift,
                                      bn3->scale, bn3->shift, workspace, (size_t)workspace_bytes, y, (hipStream_t)stream);
        if (rd != 1) return rd;
    }
    const int r = try_launch_ir_fused(1, si, fh, fw, bank, (long)ld, cin, in->c_skip, hidden, c_out, bn1->scale, bn1->shift,
                                      bn2->scale, bn2->shift, bn3->scale, bn3->shift, y, (hipStream_t)stream);
    return r == 1 ? HS_ERR_UNSUPPORTED : r;
}

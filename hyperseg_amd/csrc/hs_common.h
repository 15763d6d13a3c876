// Shared device helpers of the HyperSeg decoder kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "hyperseg_hip.h"

namespace hs {

constexpr int kWave = 64;   // CDNA wavefront

// Device-side copy of hs_stage_input with derived constants.
struct StageIn {
    const float* __restrict__ skip;
    const float* __restrict__ prev;
    int B, H, W, c_skip, c_prev, Hp, Wp, coords, prev_mode;
    float step_x, step_y;     // linspace steps 2/(W-1), 2/(H-1)
    float scale_y, scale_x;   // Hp/H, Wp/W (bilinear source scale, align_corners=False)
    __host__ __device__ int cin() const { return 2 * coords + c_skip + c_prev; }
};

// torch.linspace(-1, 1, n)[i]: symmetric evaluation (start + step*i below the midpoint,
// end - step*(n-1-i) above it), as ATen's linspace kernel does.
__device__ __forceinline__ float linspace_pm1(int i, int n, float step) {
    if (n == 1) return -1.0f;
    return (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

// Index map of F.pad for one axis.  Returns -1 for "zero" (HS_PAD_ZEROS outside the image).
__device__ __forceinline__ int pad_index(int i, int n, int mode) {
    if (i >= 0 && i < n) return i;
    switch (mode) {
        case HS_PAD_REFLECT:   i = (i < 0) ? -i : 2 * (n - 1) - i; return i;
        case HS_PAD_REPLICATE: return (i < 0) ? 0 : n - 1;
        case HS_PAD_CIRCULAR:  i %= n; return (i < 0) ? i + n : i;
        default:               return -1;
    }
}

// Bilinear taps of F.interpolate(mode='bilinear', align_corners=False) for one axis.
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap bilinear_tap(int dst, float scale, int in_size) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    int i0 = (int)src;
    i0 = i0 > in_size - 1 ? in_size - 1 : i0;
    Tap t;
    t.i0 = i0;
    t.i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    t.l1 = src - (float)i0;
    t.l0 = 1.0f - t.l1;
    return t;
}

// Precomputed per-position sampling state of a stage input: everything that does not depend
// on the channel.  (y, x) are in-image coordinates (already mapped through pad_index).
// n / d for 0 <= n < 2^21 and a run-time divisor d whose reciprocal the host passes as a float: floor((n + 0.5) * (1 / d)) -- (n + 0.5) / d
// sits >= 0.5 / d away from the next integer on either side while the float error is <= 2^-22 n / d, so the truncation is exact.  Three
// instructions; the compiler's 32-bit division by a run-time value is ~25 (the image-level training kernels did up to 18 per thread).
__device__ __forceinline__ int div_by_inv(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }

// Storage types of the training-path kernels: fp32, or bf16 storage with fp32 arithmetic (rounded to nearest-even once on store).
struct bf16_t { uint16_t v; };

template <typename T> struct Store;
template <> struct Store<float> {
    static __device__ __forceinline__ float ld(const float* p, size_t i) { return p[i]; }
    // raw / cvt: a load whose widening is kept out of the (predicated) load itself, so that a batch of loads stays in flight together
    typedef float raw_t;
    static __device__ __forceinline__ raw_t raw(const float* p, size_t i) { return p[i]; }
    static __device__ __forceinline__ float cvt(raw_t r) { return r; }
    static __device__ __forceinline__ void pin(raw_t&) {}
    static __device__ __forceinline__ void st(float* p, size_t i, float x) { p[i] = x; }
};
template <> struct Store<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p, size_t i) { return __uint_as_float((uint32_t)p[i].v << 16); }
    typedef uint32_t raw_t;           // a 16-bit raw type gets packed two to a register, which waits for each load
    static __device__ __forceinline__ raw_t raw(const bf16_t* p, size_t i) { return p[i].v; }
    static __device__ __forceinline__ float cvt(raw_t r) { return __uint_as_float(r << 16); }
    // pin: keeps the widening on this side of a predicated load's branch (the compiler otherwise moves the shift -- and a wait -- into it)
    static __device__ __forceinline__ void pin(raw_t& r) { asm volatile("" : "+v"(r)); }
    static __device__ __forceinline__ void st(bf16_t* p, size_t i, float x) {
        uint32_t u = __float_as_uint(x);
        if ((u & 0x7fffffffu) > 0x7f800000u) { p[i].v = (uint16_t)((u >> 16) | 0x40); return; }     // NaN stays NaN
        u += 0x7fffu + ((u >> 16) & 1u);                                                           // round to nearest even
        p[i].v = (uint16_t)(u >> 16);
    }
};

// two adjacent elements as one aligned load / store (the address must be a multiple of two elements)
using bw_f32x2 = __attribute__((ext_vector_type(2))) float;
template <typename T> struct Pair;
template <> struct Pair<float> {
    static __device__ __forceinline__ void ld(const float* p, size_t i, float& a, float& b) {
        const bw_f32x2 v = *reinterpret_cast<const bw_f32x2*>(p + i); a = v[0]; b = v[1];
    }
    static __device__ __forceinline__ void st(float* p, size_t i, float a, float b) { *reinterpret_cast<bw_f32x2*>(p + i) = bw_f32x2{a, b}; }
};
template <> struct Pair<bf16_t> {
    static __device__ __forceinline__ void ld(const bf16_t* p, size_t i, float& a, float& b) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(p + i); a = __uint_as_float(v << 16); b = __uint_as_float(v & 0xffff0000u);
    }
    // one 4-byte store of the two rounded values (round 6: two 2-byte stores before; Store<bf16_t>::st's rounding, NaN stays NaN)
    static __device__ __forceinline__ uint32_t bits(float x) {
        uint32_t u = __float_as_uint(x);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    }
    static __device__ __forceinline__ void st(bf16_t* p, size_t i, float a, float b) { *reinterpret_cast<uint32_t*>(p + i) = bits(a) | (bits(b) << 16); }
};

// Slices a training-mode BatchNorm channel is cut into: partial[(c * BN_CHUNKS + chunk) * 2 + {0, 1}] = {sum, sum of squares} of
// (x - shift) over the slice, shift = the channel's first element (hs_train_aux.hip: bn_stats_kernel; consumers that normalise on load
// re-derive mean / invstd from the 32 pairs: hs_patch_conv_bwd.hip dw_tiles_*).
constexpr int BN_CHUNKS = 32;

struct StagePos {
    int y, x;             // -1 in either => zero padding
    Tap ty, tx;           // only valid for prev_mode == HS_PREV_BILINEAR
};

__device__ __forceinline__ StagePos stage_pos(const StageIn& s, int y, int x) {
    StagePos p;
    p.y = y; p.x = x;
    if (s.prev_mode == HS_PREV_BILINEAR && y >= 0 && x >= 0) {
        p.ty = bilinear_tap(y, s.scale_y, s.Hp);
        p.tx = bilinear_tap(x, s.scale_x, s.Wp);
    }
    return p;
}

// Value of stage-input channel c of batch b at a sampled position.  TP: storage type of the previous level (fp32 everywhere but the
// training path under bf16 autocast, hs_stage_input_typed_fwd).
template <typename TP = float>
__device__ __forceinline__ float stage_value(const StageIn& s, int b, int c, const StagePos& p) {
    if (p.y < 0 || p.x < 0) return 0.0f;
    if (s.coords) {
        if (c == 0) return linspace_pm1(p.x, s.W, s.step_x);
        if (c == 1) return linspace_pm1(p.y, s.H, s.step_y);
        c -= 2;
    }
    if (c < s.c_skip)
        return s.skip[(((size_t)b * s.c_skip + c) * s.H + p.y) * s.W + p.x];
    c -= s.c_skip;
    const TP* base = (const TP*)s.prev + ((size_t)b * s.c_prev + c) * s.Hp * s.Wp;
    if (s.prev_mode == HS_PREV_SAME) return Store<TP>::ld(base, (size_t)p.y * s.Wp + p.x);
    const TP* r0 = base + (size_t)p.ty.i0 * s.Wp;
    const TP* r1 = base + (size_t)p.ty.i1 * s.Wp;
    float top = p.tx.l0 * Store<TP>::ld(r0, p.tx.i0) + p.tx.l1 * Store<TP>::ld(r0, p.tx.i1);
    float bot = p.tx.l0 * Store<TP>::ld(r1, p.tx.i0) + p.tx.l1 * Store<TP>::ld(r1, p.tx.i1);
    return p.ty.l0 * top + p.ty.l1 * bot;
}

// x * sigmoid(x) on the hardware transcendental units: v_exp_f32 (2^x) and v_rcp_f32 are accurate to 1 ulp, so the
// result is within ~3 ulp of the IEEE expression, for 5 VALU operations instead of the ~25 that expf() + an IEEE division
// expand to.  The encoder evaluates ~1e8 swishes per frame: at 4 cycles per wave64 VALU instruction that difference was
// a third of the depthwise / fused-MBConv kernels' time.  t -> -inf: 2^(+big) = inf, rcp(inf) = 0, t * 0 = -0.
__device__ __forceinline__ float swishf(float t) {
    return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.44269504088896341f));
}
__device__ __forceinline__ float sigmoidf_fast(float t) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.44269504088896341f));
}

// Sums across lanes on the DPP data path (no LDS crossbar: a __shfl_xor is a ds_bpermute_b32, ~150 cycles of latency each,
// and a 6-step butterfly of them is ~0.4 us in a kernel whose whole budget is 2-4 us).  rowsum16: every lane of a row of 16
// gets the row's total (xor 1, xor 2, half-mirror, mirror: the same pairs from both sides, so all 16 results are
// bit-identical); wave_sum64: the four row totals combined through v_readlane, identical in all 64 lanes.
__device__ __forceinline__ float rowsum16(float v) {
    int x = __float_as_int(v);
    auto step = [&](int moved) { x = __float_as_int(__int_as_float(x) + __int_as_float(moved)); };
    step(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    step(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    step(__builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));     // row_half_mirror
    step(__builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false));     // row_mirror
    return __int_as_float(x);
}
__device__ __forceinline__ float wave_sum64(float v) {
    const int x = __float_as_int(rowsum16(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == HS_ACT_RELU)  return fmaxf(v, 0.0f);
    if (act == HS_ACT_RELU6) return fminf(fmaxf(v, 0.0f), 6.0f);
    if (act == HS_ACT_SWISH) return swishf(v);        // not used by any reference decoder (ReLU / ReLU6 only: SURVEY appendix D-5)
    return v;
}

// ---- host side ---------------------------------------------------------------------------
inline int make_stage(const hs_stage_input* in, StageIn* out) {
    if (!in || (!in->skip && in->c_skip > 0)) return HS_ERR_BAD_ARG;
    if (in->batch <= 0 || in->H <= 0 || in->W <= 0 || in->c_skip < 0 || in->c_prev < 0) return HS_ERR_BAD_ARG;
    if (in->c_prev > 0 && (in->prev == nullptr || in->prev_mode == HS_PREV_NONE)) return HS_ERR_BAD_ARG;
    if (in->c_prev > 0 && (in->Hp <= 0 || in->Wp <= 0)) return HS_ERR_BAD_ARG;
    if (in->prev_mode == HS_PREV_SAME && in->c_prev > 0 && (in->Hp != in->H || in->Wp != in->W)) return HS_ERR_BAD_ARG;
    out->skip = in->skip; out->prev = in->prev;
    out->B = in->batch; out->H = in->H; out->W = in->W;
    out->c_skip = in->c_skip; out->c_prev = in->c_prev;
    out->Hp = in->c_prev > 0 ? in->Hp : 1; out->Wp = in->c_prev > 0 ? in->Wp : 1;
    out->coords = in->coords ? 1 : 0;
    out->prev_mode = in->c_prev > 0 ? in->prev_mode : HS_PREV_NONE;
    out->step_x = in->W > 1 ? 2.0f / (float)(in->W - 1) : 0.0f;
    out->step_y = in->H > 1 ? 2.0f / (float)(in->H - 1) : 0.0f;
    out->scale_y = (float)out->Hp / (float)in->H;
    out->scale_x = (float)out->Wp / (float)in->W;
    return HS_OK;
}

// A kernel that may need more than 64 KiB of dynamic LDS has its ceiling raised to the full 160 KiB ONCE per device
// (hipFuncSetAttribute is a driver call: not something to repeat on every launch).  `done` is the call site's static
// per-device bit mask -- write-once, idempotent, so the library stays re-entrant.
inline int allow_full_lds(const void* kernel, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return HS_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return HS_OK;
}

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HS_OK : (int)e;
}

}  // namespace hs

// Shared by the fused inverted-residual kernels (hs_patch_ir_fused.hip, hs_patch_ir_px.hip: exact f32 matrix cores;
// hs_patch_irc.hip: Op C as f16 split products on the f16 matrix cores): launch arguments and the LDS geometry of a region.
#pragma once
#include "hs_common.h"
#include "hs_ir_tiles.h"

namespace hs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct IrFusedArgs {
    StageIn in;
    int fh, fw, ph, pw;
    const float* __restrict__ bank;
    long ld;
    int hid;
    const float* __restrict__ s1; const float* __restrict__ b1;
    const float* __restrict__ s2; const float* __restrict__ b2;
    const float* __restrict__ s3; const float* __restrict__ b3;
    float* __restrict__ y;
    int regs_y, regs_x;          // regions per image
};

constexpr int IRF_THREADS = 256;

template <int REG> struct IrfGeom {
    static constexpr int HW = REG + 2;
    static constexpr int RS = (HW + 3) & ~3;                    // h1 row stride (floats): 16-byte aligned rows
    // h1 plane per hidden channel, == 4 (mod 8) floats: the D-row groups of a half-wave then store to banks 16 apart.
    // The plane's tail [HW*RS, H1P) is padding; its first float is the DUMMY slot dead columns store to.
    static constexpr int H1P = ((HW * RS + 7) & ~7) + 4;
    static constexpr int DUMMY = HW * RS;
    static constexpr int PWIN = REG / 2 + 2;                    // low-res window edge of the previous level (exact 2x)
    static constexpr int PPL = PWIN * PWIN;
    static constexpr int RS2 = REG + 4;                         // h2 pixel-row stride
    static constexpr int H2S = ((REG * RS2 + 31) & ~31) + 16;   // h2 plane: == 16 (mod 32) -> conflict-free B reads
    static constexpr int H1_FLOATS = 16 * H1P;
    static constexpr int H2_FLOATS = 16 * H2S;
};

__device__ __forceinline__ float relu6_(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// hs_patch_ir_px.hip: Op D, one lane per pixel, for the narrow HyperSeg-L levels; 1 = no instantiation
int try_launch_ir_px(IrFusedArgs& a, int cin, int c_skip, int c_out, hipStream_t stream);

// hs_patch_ir_d2.hip: Op D for 4 x 4 / 8 x 8-pixel patches in two launches with the hidden map (channels-last) in the caller's
// workspace; ird_workspace_bytes = 0 and try_launch_ird = 1 when the shape is not covered
size_t ird_workspace_bytes(const StageIn& si, int fh, int fw, int cin, int hid, int c_out);
int try_launch_ird(const StageIn& si, int fh, int fw, const float* bank, long ld, int cin, int hid, int c_out,
                   const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                   float* workspace, size_t workspace_bytes, float* y, hipStream_t stream);

}  // namespace hs

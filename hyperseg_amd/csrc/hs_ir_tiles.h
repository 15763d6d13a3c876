// Tile maps of the fused inverted-residual kernel (hs_patch_ir_fused.hip), shared by device code and by the host
// introspection entry point hs_ir_tile_map (the CPU tests check them against the reference's semantics).
//
// One workgroup owns a REG x REG pixel REGION of one level and works on its (REG+2)^2 HALO grid, halo coordinate
// (u, v) <-> image pixel (y0 + u - 1, x0 + v - 1) mapped through reflect padding.  Every matrix-core tile is
// 16 columns (positions) wide and all of its columns must share ONE filter bank (the A operand):
//
//   MODE 0 -- Op C (hyperseg_v1_0.py:328-376): the region lies inside one patch and that patch's weights are
//             applied to the whole halo tile, so the (REG+2)^2 positions are simply enumerated 16 at a time.
//   MODE 1 -- Op D (hyperseg_v0_1.py:205-237): three IMAGE-level patch convolutions, i.e. a halo position is
//             filtered with the weights of the patch that OWNS it (the neighbouring patch for the ring around the
//             region, and -- when patches are smaller than the region, PWR < REG -- per patch inside it).
//             pw1 tiles = NT3 interior tiles, each inside one patch, then 4*SEG edge segments (PWR live columns
//             each), then 4 corners (one live column each).
// The interior tiles (16 pixels inside one patch) are also the pixel tiles of pw3.
#pragma once
#ifndef HS_HD
#define HS_HD __host__ __device__
#endif

namespace hs {

template <int REG, int MODE, int PWR>
struct IrTiles {
    static_assert(REG == 8 || REG == 16, "region edge");
    static_assert(PWR >= 4 && PWR <= REG && REG % PWR == 0 && (PWR * PWR) % 16 == 0, "patch edge inside the region");
    static_assert(MODE == 1 || PWR == REG, "Op C regions lie inside one patch");
    static constexpr int HW = REG + 2;                     // halo grid edge
    static constexpr int NPOS = HW * HW;
    static constexpr int SEG = REG / PWR;                  // patches per region edge
    static constexpr int TPP = PWR * PWR / 16;             // interior tiles per patch-in-region
    static constexpr int NT3 = REG * REG / 16;             // interior (pixel) tiles
    static constexpr int NT1 = MODE == 0 ? (NPOS + 15) / 16 : NT3 + 4 * SEG + 4;

    // region-relative pixel (row, col) of column n of interior tile t
    static HS_HD inline void pixel(int t, int n, int& row, int& col) {
        const int q = t / TPP, sub = t - q * TPP;
        const int py = q / SEG, px = q - py * SEG;
        const int e = sub * 16 + n;
        row = py * PWR + e / PWR;
        col = px * PWR + e % PWR;
    }

    // halo position (u, v) of column n of pw1 tile t; false = dead column (u, v then name a valid position anyway)
    static HS_HD inline bool halo(int t, int n, int& u, int& v) {
        if (MODE == 0) {
            const int pos = t * 16 + n;
            const bool live = pos < NPOS;
            const int p = live ? pos : 0;
            u = p / HW; v = p - u * HW;
            return live;
        }
        if (t < NT3) {
            int row, col;
            pixel(t, n, row, col);
            u = row + 1; v = col + 1;
            return true;
        }
        const int r = t - NT3;
        if (r < 4 * SEG) {
            const int side = r / SEG, s = r - side * SEG;
            const bool live = n < PWR;
            const int e = 1 + s * PWR + (live ? n : 0);
            if (side == 0) { u = 0; v = e; }
            else if (side == 1) { u = HW - 1; v = e; }
            else if (side == 2) { u = e; v = 0; }
            else { u = e; v = HW - 1; }
            return live;
        }
        const int c = r - 4 * SEG;
        u = (c & 2) ? HW - 1 : 0;
        v = (c & 1) ? HW - 1 : 0;
        return n == 0;
    }
};

}  // namespace hs

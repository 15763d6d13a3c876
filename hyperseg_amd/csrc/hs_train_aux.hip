// Training-path helpers around the patch convolutions (BASELINE config 5; round 3).  Plain tensors, storage type T = float | bf16_t.
//
// 1. Halo tiles.  A train-mode v1_0 inverted residual (hyperseg_v1_0.py:328-376) applies each patch's weights to the patch's own
//    reflect-padded (ph + 2) x (pw + 2) tile; hyperseg_amd lays those tiles side by side as one "tiled image" so that the three layers
//    are ordinary patch convolutions (models/hyperseg_v1_0.py _run_train).  With stock ops that is F.pad(reflect) -> unfold -> unfold ->
//    permute -> reshape (and a slice -> reshape to drop the halo again): 3 launches forward and ~6 backward per level, all of them
//    copies.  Here each direction of each of the two re-layouts is ONE gather kernel (no atomics: the backward of the tiling sums, per
//    image pixel, the tile positions that map onto it -- own tile, the neighbours' halos, the reflections at the image border).
//       hs_halo_tiles_fwd      tiled[b,c, i(ph+2)+u, j(pw+2)+v] = x[b,c, reflect(i ph + u - 1), reflect(j pw + v - 1)]
//       hs_halo_tiles_bwd      dx = adjoint of the above
//       hs_tile_interior_fwd   y[b,c, i ph + u, j pw + v] = tiled[b,c, i(ph+2)+u+1, j(pw+2)+v+1]
//       hs_tile_interior_bwd   dtiled = dy on the interiors, 0 on the halos
// 2. hs_bootstrap_mean_{fwd,bwd}: the reduction of hyperseg/losses/bootstrapped_ce_loss.py:19-25 for one image without a sort and
//    without a host read (what a captured training step needs): the k-th largest loss by a three-level radix histogram of the float
//    bit patterns (losses are >= 0, so their bits order like integers), then both branches of the rule as sums.
#include "hs_common.h"

#ifndef HS_HALO_BWD_CH_SMALL
#define HS_HALO_BWD_CH_SMALL 2
#endif
#ifndef HS_HALO_BWD_CH_BIG
#define HS_HALO_BWD_CH_BIG 4          // channels per thread of halo_tiles_bwd_patch_kernel: 256-pixel patches (BIG) 2 / 4 / 8 -> 17.8 / 16.4 / 20.2 us at config 5's
                                       // level 4, smaller ones (SMALL) 1 / 2 / 4 -> 9.3 / 8.2 / 9.6 us at level 3 (visits x11 - x14)
#endif

namespace hs {

struct TileArgs { int B, C, H, W, fh, fw, ph, pw; float inv_ph, inv_pw, inv_ph2, inv_pw2; int pm; float inv_c; };      // reciprocals of ph, pw, ph + 2, pw + 2 (div_by_inv)
// Where tile (i, j) of plane pl = b C + c keeps its position (U, V).  pm = 0: the IMAGE of tiles (B, C, fh (ph+2), fw (pw+2)) -- tiles side by
// side, what the generic patch convolutions take.  pm = 1: PATCH-MAJOR (B fh fw, C, ph+2, pw+2) -- every operand of a patch one contiguous
// run, each patch a 1 x 1-grid "image" of its own for the patch convolutions (round 4: in the image of tiles a tile row is 72 bytes of a
// 128-byte line that the neighbouring tile's workgroup fetches again).
struct TilePlane { size_t pm_base, im_base; };                         // per thread, once: where plane pl = b C + c starts in either layout
__device__ __forceinline__ TilePlane tile_plane(const TileArgs& a, size_t pl) {
    const int b = div_by_inv((int)pl, a.inv_c), c = (int)pl - b * a.C;
    TilePlane t;
    t.pm_base = ((size_t)b * a.fh * a.fw * a.C + c) * (size_t)((a.ph + 2) * (a.pw + 2));        // + (i fw + j) C tile_size
    t.im_base = pl * (size_t)(a.fh * (a.ph + 2)) * (size_t)(a.fw * (a.pw + 2));
    return t;
}
__device__ __forceinline__ size_t tile_addr(const TileArgs& a, const TilePlane& t, int i, int U, int j, int V) {
    if (a.pm) return t.pm_base + (size_t)(i * a.fw + j) * a.C * (size_t)((a.ph + 2) * (a.pw + 2)) + (size_t)U * (a.pw + 2) + V;
    return t.im_base + ((size_t)i * (a.ph + 2) + U) * (size_t)(a.fw * (a.pw + 2)) + (size_t)j * (a.pw + 2) + V;
}

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

template <typename T>
__global__ __launch_bounds__(256)
void halo_tiles_fwd_kernel(TileArgs a, const T* __restrict__ x, T* __restrict__ t) {
    // four tile rows (Y, Y + 4, Y + 8, Y + 12: a wave still walks 64 consecutive columns of one row) of TWO planes per thread, the eight loads
    // requested together; the row / column arithmetic and the offset inside a plane's tiles are shared by the planes (round 6: one element per
    // thread -- 21 k workgroups of a load and a store each -- ran at 2.2 TB/s)
    const int TW = a.fw * (a.pw + 2), TH = a.fh * (a.ph + 2);
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y0 = blockIdx.y * 16 + (threadIdx.x >> 6);
    if (X >= TW || Y0 >= TH) return;
    const int j = div_by_inv(X, a.inv_pw2), v = X - j * (a.pw + 2), xx = reflect1(j * a.pw + v - 1, a.W);
    const size_t pl0 = (size_t)blockIdx.z * 2, npl = (size_t)a.B * a.C;
    const bool two = pl0 + 1 < npl;
    const TilePlane tp0 = tile_plane(a, pl0), tp1 = tile_plane(a, two ? pl0 + 1 : pl0), zero{0, 0};
    const size_t base0 = a.pm ? tp0.pm_base : tp0.im_base, base1 = a.pm ? tp1.pm_base : tp1.im_base;
    typename Store<T>::raw_t val[2][4];
    size_t dst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int Y = min(Y0 + 4 * r, TH - 1);                        // a row past the end repeats the last one (and is not stored)
        const int i = div_by_inv(Y, a.inv_ph2), u = Y - i * (a.ph + 2), y = reflect1(i * a.ph + u - 1, a.H);
        const size_t src = (size_t)y * a.W + xx;
        val[0][r] = Store<T>::raw(x, pl0 * a.H * a.W + src);
        val[1][r] = Store<T>::raw(x, (two ? pl0 + 1 : pl0) * a.H * a.W + src);
        dst[r] = tile_addr(a, zero, i, u, j, v);                      // offset inside a plane's tiles: the same for every plane
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (Y0 + 4 * r < TH) {
            Store<T>::st(t, base0 + dst[r], Store<T>::cvt(val[0][r]));
            if (two) Store<T>::st(t, base1 + dst[r], Store<T>::cvt(val[1][r]));
        }
}

// candidates (tile index, position inside the tile) of one axis that map onto image index y: every padded coordinate that reflects
// onto y (itself; -1 for y = 1; n for y = n - 2), in every tile whose p + 2 padded rows contain it (two at a patch border, three when
// patches are one pixel wide)
__device__ __forceinline__ int tile_sources(int y, int n, int p, float inv_p, int f, int (&tpos)[9]) {
    int cnt = 0;
    const int yps[3] = {y, y == 1 ? -1 : -2, y == n - 2 ? n : -2};      // -2: none
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (yps[q] == -2) continue;
        const int Yp = yps[q] + 1;                                      // index in the padded image [0, n + 1]
        for (int i = div_by_inv(Yp, inv_p); i >= 0 && Yp - i * p <= p + 1; --i)
            if (i < f) tpos[cnt++] = i * (p + 2) + (Yp - i * p);
    }
    return cnt;
}

// the same list for patches of >= 2 rows, without the search: the own tile always; the tile above / below when y is the patch's first / last
// row (tiles overlap by their rings); the image's reflections (padded row -1 = row 1, padded row n = row n - 2) in the first / last tile
__device__ __forceinline__ int tile_sources_p2(int y, int n, int p, float inv_p, int f, int (&tpos)[3]) {
    const int i = div_by_inv(y, inv_p), u = y - i * p;
    int cnt = 0;
    tpos[cnt++] = i * (p + 2) + u + 1;
    if (u == 0 && i > 0) tpos[cnt++] = (i - 1) * (p + 2) + p + 1;
    if (u == p - 1 && i + 1 < f) tpos[cnt++] = (i + 1) * (p + 2);
    if (y == 1) tpos[cnt++] = 0;
    if (y == n - 2) tpos[cnt++] = (f - 1) * (p + 2) + p + 1;
    return cnt;                                                         // <= 3: y == 1 or n - 2 is a first / last row only when p == 2, f == 1
}

// (Round 6, visits x3 / x4: two forms with the candidate loads in flight together -- 2 x 2 unpredicated loads per pixel; four rows per thread with
//  shared column parts, 32-bit offsets and predicated loads -- both measured SLOWER on config 5 (23.0 -> 45.6 / 38.4 us at level 4, 10.0 -> 34.2 /
//  32.5 us at level 3) although they execute fewer instructions and wait once.  What moved the launch in the end is below: mapped by patch, several
//  channels per thread (halo_tiles_bwd_patch_kernel); this kernel stays for the patch shapes that one does not take.)
template <typename T>
__global__ __launch_bounds__(256)
void halo_tiles_bwd_kernel(TileArgs a, const T* __restrict__ dt, T* __restrict__ dx) {
    const int TW = a.fw * (a.pw + 2), TH = a.fh * (a.ph + 2);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const size_t pl = blockIdx.z;
    const TilePlane tp = tile_plane(a, pl);
    float acc = 0.0f;
    if (a.ph >= 3 && a.pw >= 3) {                                       // (uniform) at most 2 x 2 sources... 3 with a reflection: no search, no division
        int ys[3], xs[3];
        const int ny = tile_sources_p2(y, a.H, a.ph, a.inv_ph, a.fh, ys), nx = tile_sources_p2(x, a.W, a.pw, a.inv_pw, a.fw, xs);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (p < ny && q < nx) {
                    const int ti = div_by_inv(ys[p], a.inv_ph2), tj = div_by_inv(xs[q], a.inv_pw2);
                    acc += Store<T>::ld(dt, tile_addr(a, tp, ti, ys[p] - ti * (a.ph + 2), tj, xs[q] - tj * (a.pw + 2)));
                }
    } else {
        int ys[9], xs[9];
        const int ny = tile_sources(y, a.H, a.ph, a.inv_ph, a.fh, ys), nx = tile_sources(x, a.W, a.pw, a.inv_pw, a.fw, xs);
        for (int p = 0; p < ny; ++p)
            for (int q = 0; q < nx; ++q) {
                const int ti = div_by_inv(ys[p], a.inv_ph2), tj = div_by_inv(xs[q], a.inv_pw2);
                acc += Store<T>::ld(dt, tile_addr(a, tp, ti, ys[p] - ti * (a.ph + 2), tj, xs[q] - tj * (a.pw + 2)));
            }
    }
    Store<T>::st(dx, (pl * a.H + y) * a.W + x, acc);
}

// Round 6, third form -- mapped by PATCH: a workgroup owns the pixels of one patch for 256 / (ph pw) consecutive channels, thread = pixel (u, v).
// The sources of a pixel are then read off its position with no search and no division by the tile size: its own tile at (u + 1, v + 1); the
// tile above / below (left / right) when it sits in the patch's first / last row (column) -- their ring rows; the image's reflections for row /
// column 1 and n - 2 -- in tile_sources_p2's order, so the sum is the one-pixel kernel's bit for bit.  A wave reads four 16-pixel rows of ONE
// tile (consecutive lines in either layout) where the image-mapped kernel above reads a row across four tiles, and the candidate loads are
// requested before the first sum.  Patches of 3 x 3 ... 256 pixels whose pixel count divides 256; the rest keeps the kernel above.
// HB_CH channels per thread: the sources and their offsets depend on the pixel only, so one source list serves HB_CH planes and the candidate
// loads of all of them are in flight together (one channel per thread -- a wave of one dependent load and one store behind ~150 instructions of
// set-up -- took the same 23 us as the image-mapped kernel: 57 k such waves, 4.6 generations of them on the chip).
template <typename T, int HB_CH>
__global__ __launch_bounds__(256)
void halo_tiles_bwd_patch_kernel(TileArgs a, const T* __restrict__ dt, T* __restrict__ dx, float inv_npx) {
    const int npx = a.ph * a.pw, per = 256 / npx;
    const int lt = div_by_inv((int)threadIdx.x, inv_npx), k = (int)threadIdx.x - lt * npx;
    const int c0 = (blockIdx.y * per + lt) * HB_CH, b = blockIdx.z;
    if (c0 >= a.C) return;
    const int j = (int)blockIdx.x % a.fw, i = (int)blockIdx.x / a.fw;                       // (uniform)
    const int u = div_by_inv(k, a.inv_pw), v = k - u * a.pw;
    const int y = i * a.ph + u, x = j * a.pw + v;
    const size_t pl0 = (size_t)b * a.C + c0;
    const TilePlane tp = tile_plane(a, pl0);
    const size_t cstride = a.pm ? (size_t)((a.ph + 2) * (a.pw + 2)) : (size_t)(a.fh * (a.ph + 2)) * (size_t)(a.fw * (a.pw + 2));   // next channel's tile
    int ty[3], uy[3], tx[3], vx[3], ny = 0, nx = 0;
    ty[ny] = i; uy[ny++] = u + 1;
    if (u == 0 && i > 0) { ty[ny] = i - 1; uy[ny++] = a.ph + 1; }
    if (u == a.ph - 1 && i + 1 < a.fh) { ty[ny] = i + 1; uy[ny++] = 0; }
    if (y == 1) { ty[ny] = 0; uy[ny++] = 0; }
    if (y == a.H - 2) { ty[ny] = a.fh - 1; uy[ny++] = a.ph + 1; }
    tx[nx] = j; vx[nx++] = v + 1;
    if (v == 0 && j > 0) { tx[nx] = j - 1; vx[nx++] = a.pw + 1; }
    if (v == a.pw - 1 && j + 1 < a.fw) { tx[nx] = j + 1; vx[nx++] = 0; }
    if (x == 1) { tx[nx] = 0; vx[nx++] = 0; }
    if (x == a.W - 2) { tx[nx] = a.fw - 1; vx[nx++] = a.pw + 1; }
    const int nch = min(HB_CH, a.C - c0);
    typename Store<T>::raw_t val[HB_CH][3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const bool on = p < ny && q < nx;
            const size_t at = on ? tile_addr(a, tp, ty[p], uy[p], tx[q], vx[q]) : 0;
#pragma unroll
            for (int ch = 0; ch < HB_CH; ++ch)
                val[ch][p][q] = (on && ch < nch) ? Store<T>::raw(dt, at + (size_t)ch * cstride) : typename Store<T>::raw_t(0);
        }
#pragma unroll
    for (int ch = 0; ch < HB_CH; ++ch) {
        float acc = 0.0f;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                typename Store<T>::raw_t w = val[ch][p][q];
                Store<T>::pin(w);
                if (p < ny && q < nx) acc += Store<T>::cvt(w);
            }
        if (ch < nch) Store<T>::st(dx, ((pl0 + ch) * a.H + y) * a.W + x, acc);
    }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256)
void tile_interior_kernel(TileArgs a, const T* __restrict__ src, T* __restrict__ dst) {
    const int TW = a.fw * (a.pw + 2), TH = a.fh * (a.ph + 2);
    const size_t pl = blockIdx.z;
    if constexpr (!BWD) {
        const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
        if (x >= a.W || y >= a.H) return;
        const int i = div_by_inv(y, a.inv_ph), j = div_by_inv(x, a.inv_pw);
        Store<T>::st(dst, (pl * a.H + y) * a.W + x, Store<T>::ld(src, (pl * TH + y + 2 * i + 1) * TW + x + 2 * j + 1));
    } else {
        const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
        if (X >= TW || Y >= TH) return;
        const int i = div_by_inv(Y, a.inv_ph2), u = Y - i * (a.ph + 2), j = div_by_inv(X, a.inv_pw2), v = X - j * (a.pw + 2);
        const bool in = u >= 1 && u <= a.ph && v >= 1 && v <= a.pw;
        const float g = Store<T>::ld(src, (pl * a.H + min(max(i * a.ph + u - 1, 0), a.H - 1)) * a.W + min(max(j * a.pw + v - 1, 0), a.W - 1));
        Store<T>::st(dst, (pl * TH + Y) * TW + X, in ? g : 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// bootstrapped mean.  The k-th largest of n non-negative floats by THREE radix levels over their bit patterns (bits 30..20, 19..10,
// 9..0): per level a histogram of the losses that match the prefix selected so far -- privatised per workgroup in LDS (cross-entropy
// losses crowd into a few hundred high-bit bins, and every ignore_index pixel is an exact 0: global atomics on those few addresses
// serialised; the first version of this file measured slower than the sort for that reason), flushed with one global atomic per
// non-empty bin -- then one workgroup walks the bins from the top.
// ws (uint32): [0, 2048) level-0 histogram, [2048, 3072) level 1, [3072, 4096) level 2, then state, one pair PER LEVEL:
// [4096 + 2 l] the prefix bits selected after level l, [4097 + 2 l] the number of losses known to be above that prefix' bin
// (level 2's pair = t's bits and the count > t).  Round 5: the bin walk of level l runs at the TOP of the next launch (the histogram
// of level l + 1, the sums after level 2), redundantly in each of its workgroups, instead of as a one-workgroup launch of its own
// (3 launches of 3.6 us per step); a pair per level because workgroup 0 stores level l's pair while the others still read level l - 1's.
// partial (float): per workgroup {sum over v > thresh, count > thresh, sum over v > t, count == t}, combined in workgroup order.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int BM_BINS0 = 2048, BM_BINS12 = 1024, BM_STATE = BM_BINS0 + 2 * BM_BINS12, BM_WG = 128;
constexpr int BM_WS_U32 = BM_STATE + 8 + BM_WG * 4;          // 32-bit words of one image's workspace: histograms | state | partial sums
__device__ __forceinline__ int bm_shift(int level) { return level == 0 ? 20 : (level == 1 ? 10 : 0); }
__device__ __forceinline__ int bm_bins(int level) { return level == 0 ? BM_BINS0 : BM_BINS12; }
__device__ __forceinline__ int bm_base(int level) { return level == 0 ? 0 : (level == 1 ? BM_BINS0 : BM_BINS0 + BM_BINS12); }

// The bin (scanning from the top) in which the cumulative count of level `level`'s histogram reaches the wanted rank, given the pair
// after level - 1; every thread of the workgroup gets the pair after `level` (all 256 threads call it; part / sel: LDS).
__device__ __forceinline__ void bm_find_body(const unsigned* __restrict__ ws, int level, int k, unsigned* part, unsigned* sel,
                                             unsigned& prefix_out, unsigned& above_out) {
    const int tid = threadIdx.x;
    const int nbins = bm_bins(level), per = nbins / 256, sh = bm_shift(level);
    const unsigned* h = ws + bm_base(level);
    const unsigned prefix_in = level == 0 ? 0u : ws[BM_STATE + 2 * (level - 1)];
    const unsigned above = level == 0 ? 0u : ws[BM_STATE + 2 * (level - 1) + 1];
    const unsigned want = (unsigned)k - above;                          // rank inside the selected prefix
    // thread t owns bins [nbins - (t + 1) per, nbins - t per): descending order of value; exclusive prefix of the threads' sums by a
    // wave scan + the four wave totals (a serial walk over 256 partial sums by one thread was 8 us of this 9 us kernel)
    unsigned s = 0;
    for (int q = 0; q < per; ++q) s += h[nbins - 1 - (tid * per + q)];
    unsigned incl = s;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += part[w];
    const unsigned excl = base + incl - s;
    if (excl < want && want <= excl + s) { sel[0] = (unsigned)tid; sel[1] = excl; }
    if (tid == 255 && want > excl + s) { sel[0] = 255u; sel[1] = excl; }      // rank beyond the total (cannot happen for n > k): last bin
    __syncthreads();
    if (tid == (int)sel[0]) {
        unsigned cum = sel[1]; int q = 0;
        for (; q < per - 1 && cum + h[nbins - 1 - (tid * per + q)] < want; ++q) cum += h[nbins - 1 - (tid * per + q)];
        const unsigned bin = (unsigned)(nbins - 1 - (tid * per + q));
        sel[2] = prefix_in | (bin << sh); sel[3] = above + cum;
    }
    __syncthreads();
    prefix_out = sel[2]; above_out = sel[3];
}

__global__ __launch_bounds__(256)
void bm_hist_kernel(const float* __restrict__ v, int n, unsigned* __restrict__ ws, int level, int k) {
    __shared__ unsigned h[BM_BINS0];
    __shared__ unsigned part[4], sel[4];
    v += (size_t)blockIdx.y * n; ws += (size_t)blockIdx.y * BM_WS_U32;           // image blockIdx.y of a batch (its own values and workspace)
    const int nb = bm_bins(level), sh = bm_shift(level);
    for (int i = threadIdx.x; i < nb; i += 256) h[i] = 0;
    unsigned prefix = 0u;                                               // the bits above this level's, already shifted into place
    if (level > 0) {
        unsigned above;
        bm_find_body(ws, level - 1, k, part, sel, prefix, above);       // (its barriers also cover the zero fill above)
        if (blockIdx.x == 0 && threadIdx.x == 0) { ws[BM_STATE + 2 * (level - 1)] = prefix; ws[BM_STATE + 2 * (level - 1) + 1] = above; }
    } else {
        __syncthreads();
    }
    const unsigned himask = level == 0 ? 0u : (0x7fffffffu >> (sh + 10)) << (sh + 10);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const unsigned b = __float_as_uint(fmaxf(v[e], 0.0f)) & 0x7fffffffu;
        if ((b & himask) == prefix) atomicAdd(&h[(b >> sh) & (nb - 1)], 1u);
    }
    __syncthreads();
    unsigned* g = ws + bm_base(level);
    for (int i = threadIdx.x; i < nb; i += 256)
        if (h[i]) atomicAdd(&g[i], h[i]);
}

// out[0] = the image's loss; out[1] = branch (1: everything above thresh, 0: the k largest), out[2] = 1 / count or 1 / k,
// out[3] = t, out[4] = weight of a loss equal to t (the k-th largest may be tied).  One wave; lane 0 writes.  `partial` may have been written by
// other workgroups of the SAME launch (bm_sums_kernel's tail): read past this CU's L1.
__device__ __forceinline__ float bm_final_body(const float* __restrict__ partial, int nwg, const unsigned* __restrict__ ws, int k, float thresh,
                                               float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    float s_thr = 0.f, c_thr = 0.f, s_top = 0.f, c_eq = 0.f;
    for (int i = lane; i < nwg; i += 64) {
        s_thr += __hip_atomic_load(partial + 4 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c_thr += __hip_atomic_load(partial + 4 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_top += __hip_atomic_load(partial + 4 * i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c_eq += __hip_atomic_load(partial + 4 * i + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_thr = wave_sum64(s_thr); c_thr = wave_sum64(c_thr); s_top = wave_sum64(s_top); c_eq = wave_sum64(c_eq);
    const float t = __uint_as_float(__hip_atomic_load(ws + BM_STATE + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const float c_gt = (float)__hip_atomic_load(ws + BM_STATE + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float o0, o1, o2, o3, o4;
    if (c_thr > (float)k) {                    // the (k+1)-th largest exceeds thresh exactly when more than k losses do
        o0 = s_thr / c_thr; o1 = 1.0f; o2 = 1.0f / c_thr; o3 = thresh; o4 = 0.0f;
    } else {
        const float ties = (float)k - c_gt;    // how many of the losses equal to t belong to the k largest
        o0 = (s_top + ties * t) / (float)k; o1 = 0.0f; o2 = 1.0f / (float)k; o3 = t;
        o4 = c_eq > 0.0f ? ties / c_eq : 0.0f;
    }
    if (lane == 0) { out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3; out[4] = o4; }
    return o0;
}

// Round 6: the per-image final and the batch mean run in the TAIL of this launch -- the workgroup that finishes last (a counter in image 0's state
// word 7, which the tail returns to zero) combines every image's partial sums in workgroup order, exactly as the two one-workgroup launches it
// replaces did (bm_final_kernel, bm_batch_mean_kernel: ~5 us each at the step's launch floor).
__global__ __launch_bounds__(256)
void bm_sums_kernel(const float* __restrict__ v, int n, float thresh, unsigned* __restrict__ ws, float* __restrict__ partial, int k,
                    float* __restrict__ out8, float* __restrict__ mean_out) {
    __shared__ float red[4][4];
    __shared__ unsigned part[4], sel[4];
    unsigned* const ws0 = ws; float* const partial0 = partial;
    v += (size_t)blockIdx.y * n; ws += (size_t)blockIdx.y * BM_WS_U32; partial += (size_t)blockIdx.y * BM_WS_U32;
    unsigned tbits, above;
    bm_find_body(ws, 2, k, part, sel, tbits, above);                    // the last level's bin walk: t and the count above it
    if (blockIdx.x == 0 && threadIdx.x == 0) {                          // (agent-scope stores: the tail below reads them from another CU)
        __hip_atomic_store(ws + BM_STATE + 4, tbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ws + BM_STATE + 5, above, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float t = __uint_as_float(tbits);
    float s_thr = 0.f, c_thr = 0.f, s_top = 0.f, c_eq = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const float raw = v[e];
        const float x = fmaxf(raw, 0.0f);
        if (raw != raw) { s_thr += raw; s_top += raw; }     // fmaxf(NaN, 0) = 0 would hide a diverged step: a NaN loss makes BOTH branch sums NaN
        if (x > thresh) { s_thr += x; c_thr += 1.0f; }
        if (x > t) s_top += x;
        if (x == t) c_eq += 1.0f;
    }
    s_thr = wave_sum64(s_thr); c_thr = wave_sum64(c_thr); s_top = wave_sum64(s_top); c_eq = wave_sum64(c_eq);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = s_thr; red[wave][1] = c_thr; red[wave][2] = s_top; red[wave][3] = c_eq; }
    __syncthreads();
    if (threadIdx.x < 4)
        __hip_atomic_store(partial + blockIdx.x * 4 + threadIdx.x, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- tail: the last workgroup of the launch finishes every image.  The sums and the state pair leave as agent-scope (write-through) stores,
    // each thread waits for its own to be acknowledged, then the workgroup takes its ticket; the tail reads with agent-scope loads.  (A
    // __threadfence() here -- an L2 write-back per workgroup on this part -- cost the launch 15 us, visit x5.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sel[0] = __hip_atomic_fetch_add(ws0 + BM_STATE + 7, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sel[0] != gridDim.x * gridDim.y - 1 || threadIdx.x >= 64) return;
    const int images = gridDim.y;
    float mean = 0.0f;
    for (int i = 0; i < images; ++i)
        mean += bm_final_body(partial0 + (size_t)i * BM_WS_U32, (int)gridDim.x, ws0 + (size_t)i * BM_WS_U32, k, thresh, out8 + (size_t)i * 8);
    if (threadIdx.x == 0) {
        if (mean_out) mean_out[0] = mean / (float)images;
        ws0[BM_STATE + 7] = 0u;                                         // the next call finds its counter at zero without a memset of its own
    }
}

// gout_stride 1: one upstream gradient per image; 0: ONE gradient of the batch mean (scale = 1 / images)
__global__ __launch_bounds__(256)
void bm_bwd_kernel(const float* __restrict__ v, int n, const float* __restrict__ state, const float* __restrict__ gout, int gout_stride,
                   float scale, float* __restrict__ gv) {
    v += (size_t)blockIdx.y * n; gv += (size_t)blockIdx.y * n; state += (size_t)blockIdx.y * 8; gout += blockIdx.y * gout_stride;
    const float w = state[2] * (gout[0] * scale), t = state[3], tie = state[4];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const float x = fmaxf(v[e], 0.0f);
        gv[e] = x > t ? w : (x == t ? w * tie : 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// BatchNorm2d in TRAINING mode fused with the activation that follows it (none / ReLU / ReLU6), forward and backward, two launches
// each: per-channel partial sums over NCHUNK slices of the (batch, pixel) range, then every workgroup of the second launch combines its
// channel's partials (in slice order: deterministic) and streams its slice once.  torch.nn.functional.batch_norm semantics: biased
// variance for the normalisation, unbiased for the running estimate, running = (1 - momentum) running + momentum batch.  The sums are
// taken around the channel's first element (a shift: E[(x - s)^2] - E[x - s]^2 does not cancel when |mean| >> std).
// MIOpen's spatial kernels + a clamp kernel (+ hardtanh_backward) were 0.53 + 0.1 ms of a 2.5 ms config-5 step.
// ---------------------------------------------------------------------------------------------------------------------------------

struct BnArgs { int B, C, HW, act; float eps, momentum; };

__device__ __forceinline__ void bn_slice(const BnArgs& a, int chunk, int& lo, int& hi) {      // B HW < 2^31 (host)
    const int n = a.B * a.HW, per = (n + BN_CHUNKS - 1) / BN_CHUNKS;
    lo = min(chunk * per, n); hi = min(lo + per, n);
}
// walks the elements e = lo + tid, + 256, ... of channel c: one division at the start, then carries
struct BnWalk {
    int e, b, p;
    __device__ __forceinline__ BnWalk(const BnArgs& a, int lo) { e = lo + (int)threadIdx.x; b = e / a.HW; p = e - b * a.HW; }
    __device__ __forceinline__ size_t at(const BnArgs& a, int c) const { return ((size_t)b * a.C + c) * a.HW + p; }
    // BIG (HW >= 256: every real layer): at most one image boundary per step, taken as a select -- no branch between a batch's loads
    template <bool BIG> __device__ __forceinline__ void next(const BnArgs& a) {
        e += 256; p += 256;
        if (BIG) { const bool wrap = p >= a.HW; p -= wrap ? a.HW : 0; b += wrap ? 1 : 0; }
        else while (p >= a.HW) { p -= a.HW; ++b; }
    }
};
// A slice is walked BN_BATCH elements per thread at a time: the batch's loads are all requested before the first is used (round 6: the
// one-load-per-iteration loops waited out a full memory round trip 20 times per thread at config 5's level 4: 9-11 us per launch for
// 13 MB); a thread still takes its elements in the same order, so every sum is bit-identical to the serial walk's.
constexpr int BN_BATCH = 8;
__device__ __forceinline__ void bn_block_sum2(float& s0, float& s1, float (*red)[2]) {
    s0 = wave_sum64(s0); s1 = wave_sum64(s1);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = s0; red[wave][1] = s1; }
    __syncthreads();
    s0 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    s1 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
}

__device__ __forceinline__ float bn_act(float z, int act) { return act == HS_ACT_RELU ? fmaxf(z, 0.f) : (act == HS_ACT_RELU6 ? fminf(fmaxf(z, 0.f), 6.f) : z); }
__device__ __forceinline__ float bn_act_grad(float z, int act) {
    return act == HS_ACT_RELU ? (z > 0.f ? 1.f : 0.f) : (act == HS_ACT_RELU6 ? ((z > 0.f && z < 6.f) ? 1.f : 0.f) : 1.f);
}

// ---- bf16 storage, PAIR mode (round 6): a lane takes two adjacent elements per step (one 4-byte load / store) -- the one-element walk moves 2 bytes per
// lane and instruction and did not get faster with the bytes bf16 saves (bn_bwd_apply 16.2 us against fp32's 22.7 for half the bytes).  Needs an even
// HW >= 256 (pairs stay inside an image; at most two image boundaries per step -- the 18 x 18 tiles of the patch-major tile tensor are 324); slices are cut in pairs (a slice's share of the channel differs from the
// one-element walk's: only the association of the 32 slice sums changes).  fp32 keeps the one-element walk: its sums stay bit-identical to round 5's.
__device__ __forceinline__ bool bn_pairs_ok(const BnArgs& a) { return (a.HW & 1) == 0 && a.HW >= 256; }
__device__ __forceinline__ void bn_slice2(const BnArgs& a, int chunk, int& lo2, int& hi2) {        // in pairs
    const int n2 = (a.B * a.HW) >> 1, per2 = (n2 + BN_CHUNKS - 1) / BN_CHUNKS;
    lo2 = min(chunk * per2, n2); hi2 = min(lo2 + per2, n2);
}
struct BnWalk2 {
    int e2, b, p;                                                     // pair index in the channel; its first element's image and position
    __device__ __forceinline__ BnWalk2(const BnArgs& a, int lo2) { e2 = lo2 + (int)threadIdx.x; const int e = 2 * e2; b = e / a.HW; p = e - b * a.HW; }
    __device__ __forceinline__ size_t at(const BnArgs& a, int c) const { return ((size_t)b * a.C + c) * a.HW + p; }
    __device__ __forceinline__ void next(const BnArgs& a) {           // 512 elements on: at most two image boundaries (HW >= 256), taken as selects
        e2 += 256; p += 512;
        bool wrap = p >= a.HW; p -= wrap ? a.HW : 0; b += wrap ? 1 : 0;
        wrap = p >= a.HW; p -= wrap ? a.HW : 0; b += wrap ? 1 : 0;
    }
};
__device__ __forceinline__ void bn_unpack2(uint32_t r, float& v0, float& v1) { v0 = __uint_as_float(r << 16); v1 = __uint_as_float(r & 0xffff0000u); }
// KIND 0: forward statistics (s, q about `k0` = shift); 1: forward apply (y = act(x k0 + ms)); 2: backward statistics; 3: backward apply
template <int KIND>
__device__ __forceinline__ void bn_pair_walk(const BnArgs& a, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ out, int c,
                                             int chunk, float mean, float invstd, float g, float bb, float k0, float ms, float mq, float& s, float& q) {
    int lo2, hi2;
    bn_slice2(a, chunk, lo2, hi2);
    const size_t at0 = (size_t)c * a.HW;
    for (BnWalk2 w(a, lo2); w.e2 < hi2;) {
        uint32_t vx[BN_BATCH], vd[BN_BATCH];
        size_t at[BN_BATCH];
        const int e0 = w.e2;
#pragma unroll
        for (int u = 0; u < BN_BATCH; ++u) {
            at[u] = w.e2 < hi2 ? w.at(a, c) : at0;
            vx[u] = *reinterpret_cast<const uint32_t*>(x + at[u]);
            if (KIND >= 2) vd[u] = *reinterpret_cast<const uint32_t*>(dy + at[u]);
            w.next(a);
        }
#pragma unroll
        for (int u = 0; u < BN_BATCH; ++u)
            if (e0 + 256 * u < hi2) {
                float xv[2], dv[2] = {0.f, 0.f}, o[2];
                bn_unpack2(vx[u], xv[0], xv[1]);
                if (KIND >= 2) bn_unpack2(vd[u], dv[0], dv[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (KIND == 0) { const float d = xv[h] - k0; s += d; q = fmaf(d, d, q); }
                    else if (KIND == 1) o[h] = bn_act(fmaf(xv[h], k0, ms), a.act);
                    else {
                        const float xh = (xv[h] - mean) * invstd;
                        const float d = dv[h] * bn_act_grad(fmaf(xh, g, bb), a.act);
                        if (KIND == 3) o[h] = k0 * (d - ms - xh * mq);
                        else { s += d; q = fmaf(d, xh, q); }
                    }
                }
                if (KIND == 1 || KIND == 3) Pair<bf16_t>::st(out, at[u], o[0], o[1]);
            }
    }
}
template <int KIND, typename T>
__device__ __forceinline__ bool bn_try_pairs(const BnArgs& a, const T* x, const T* dy, T* out, int c, int chunk, float mean, float invstd, float g, float bb,
                                             float k0, float ms, float mq, float& s, float& q) {
    if constexpr (sizeof(T) == 2) {
        if (bn_pairs_ok(a) && ((((size_t)x) | ((size_t)dy) | ((size_t)out)) & 3) == 0) {        // (4-byte aligned tensors: every pair is)
            bn_pair_walk<KIND>(a, x, dy, out, c, chunk, mean, invstd, g, bb, k0, ms, mq, s, q);
            return true;
        }
    }
    return false;
}

template <typename T, bool BIG>
__device__ __forceinline__ void bn_stats_walk(const BnArgs& a, const T* __restrict__ x, int c, int lo, int hi, float shift, float& s, float& q) {
    for (BnWalk w(a, lo); w.e < hi;) {
        typename Store<T>::raw_t v[BN_BATCH];
        const int e0 = w.e;
        const size_t at0 = (size_t)c * a.HW;        // the channel's first element: what a lane past the slice's end loads instead (unpredicated: a
                                                    // predicated 16-bit load drags its widening, and with it a wait, into the branch)
#pragma unroll
        for (int u = 0; u < BN_BATCH; ++u) { v[u] = Store<T>::raw(x, w.e < hi ? w.at(a, c) : at0); w.template next<BIG>(a); }
#pragma unroll
        for (int u = 0; u < BN_BATCH; ++u)
            if (e0 + 256 * u < hi) { const float d = Store<T>::cvt(v[u]) - shift; s += d; q = fmaf(d, d, q); }
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void bn_stats_kernel(BnArgs a, const T* __restrict__ x, float* __restrict__ partial) {
    __shared__ float red[4][2];
    const int c = blockIdx.x, chunk = blockIdx.y;
    const float shift = Store<T>::ld(x, (size_t)c * a.HW);
    int lo, hi;
    bn_slice(a, chunk, lo, hi);
    float s = 0.f, q = 0.f;
    if (bn_try_pairs<0, T>(a, x, (const T*)nullptr, (T*)nullptr, c, chunk, 0.f, 0.f, 0.f, 0.f, shift, 0.f, 0.f, s, q)) {}
    else if (a.HW >= 256) bn_stats_walk<T, true>(a, x, c, lo, hi, shift, s, q);
    else bn_stats_walk<T, false>(a, x, c, lo, hi, shift, s, q);
    bn_block_sum2(s, q, red);
    if (threadIdx.x == 0) { partial[((size_t)c * BN_CHUNKS + chunk) * 2] = s; partial[((size_t)c * BN_CHUNKS + chunk) * 2 + 1] = q; }
}


template <typename T, bool BIG>
__device__ __forceinline__ void bn_apply_walk(const BnArgs& a, const T* __restrict__ x, T* __restrict__ y, int c, int lo, int hi, float g, float bb) {
    for (BnWalk w(a, lo); w.e < hi;) {
        typename Store<T>::raw_t v[BN_BATCH];
        size_t at[BN_BATCH];
        const int e0 = w.e;
        const size_t at0 = (size_t)c * a.HW;        // the channel's first element: what a lane past the slice's end loads instead (unpredicated: a
                                                    // predicated 16-bit load drags its widening, and with it a wait, into the branch)
#pragma unroll
        for (int u = 0; u < BN_BATCH; ++u) { at[u] = w.e < hi ? w.at(a, c) : at0; v[u] = Store<T>::raw(x, at[u]); w.template next<BIG>(a); }
#pragma unroll
        for (int u = 0; u < BN_BATCH; ++u)
            if (e0 + 256 * u < hi) Store<T>::st(y, at[u], bn_act(fmaf(Store<T>::cvt(v[u]), g, bb), a.act));
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void bn_apply_kernel(BnArgs a, const T* __restrict__ x, const float* __restrict__ partial, const float* __restrict__ gamma,
                     const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                     float* __restrict__ save_mean, float* __restrict__ save_invstd, T* __restrict__ y, long long* __restrict__ counter) {
    const int c = blockIdx.x, chunk = blockIdx.y;
    if (counter && c == 0 && chunk == 0 && threadIdx.x == 0) *counter += 1;      // nn.BatchNorm2d.num_batches_tracked: one launch less per layer and step
    const float shift = Store<T>::ld(x, (size_t)c * a.HW);
    float s = 0.f, q = 0.f;
    for (int i = 0; i < BN_CHUNKS; ++i) { s += partial[((size_t)c * BN_CHUNKS + i) * 2]; q += partial[((size_t)c * BN_CHUNKS + i) * 2 + 1]; }
    const float n = (float)((long)a.B * a.HW);
    const float md = s / n, var = fmaxf(q / n - md * md, 0.f), mean = md + shift, invstd = rsqrtf(var + a.eps);
    if (chunk == 0 && threadIdx.x == 0) {
        save_mean[c] = mean; save_invstd[c] = invstd;
        if (running_mean) {
            running_mean[c] = (1.f - a.momentum) * running_mean[c] + a.momentum * mean;
            running_var[c] = (1.f - a.momentum) * running_var[c] + a.momentum * (n > 1.f ? var * n / (n - 1.f) : var);
        }
    }
    const float g = gamma ? gamma[c] * invstd : invstd, bb = (beta ? beta[c] : 0.f) - mean * g;
    int lo, hi;
    bn_slice(a, chunk, lo, hi);
    float z0 = 0.f, z1 = 0.f;
    if (bn_try_pairs<1, T>(a, x, (const T*)nullptr, y, c, chunk, 0.f, 0.f, 0.f, 0.f, g, bb, 0.f, z0, z1)) {}
    else if (a.HW >= 256) bn_apply_walk<T, true>(a, x, y, c, lo, hi, g, bb);
    else bn_apply_walk<T, false>(a, x, y, c, lo, hi, g, bb);
}

// the backward pair's walk: APPLY false -> the two sums (s, q), APPLY true -> dx from the finished sums (k0, ms, mq)
template <typename T, bool BIG, bool APPLY>
__device__ __forceinline__ void bn_bwd_walk(const BnArgs& a, const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int c, int lo,
                                            int hi, float mean, float invstd, float g, float bb, float k0, float ms, float mq, float& s, float& q) {
    for (BnWalk w(a, lo); w.e < hi;) {
        typename Store<T>::raw_t vx[BN_BATCH], vd[BN_BATCH];
        size_t at[BN_BATCH];
        const int e0 = w.e;
        const size_t at0 = (size_t)c * a.HW;        // the channel's first element: what a lane past the slice's end loads instead (unpredicated: a
                                                    // predicated 16-bit load drags its widening, and with it a wait, into the branch)
#pragma unroll
        for (int u = 0; u < BN_BATCH; ++u) {
            at[u] = w.e < hi ? w.at(a, c) : at0;
            vx[u] = Store<T>::raw(x, at[u]); vd[u] = Store<T>::raw(dy, at[u]);
            w.template next<BIG>(a);
        }
#pragma unroll
        for (int u = 0; u < BN_BATCH; ++u)
            if (e0 + 256 * u < hi) {
                const float xh = (Store<T>::cvt(vx[u]) - mean) * invstd;
                const float d = Store<T>::cvt(vd[u]) * bn_act_grad(fmaf(xh, g, bb), a.act);
                if (APPLY) Store<T>::st(dx, at[u], k0 * (d - ms - xh * mq));
                else { s += d; q = fmaf(d, xh, q); }
            }
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void bn_bwd_stats_kernel(BnArgs a, const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                         const float* __restrict__ beta, const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                         float* __restrict__ partial) {
    __shared__ float red[4][2];
    const int c = blockIdx.x, chunk = blockIdx.y;
    const float mean = save_mean[c], invstd = save_invstd[c], g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    int lo, hi;
    bn_slice(a, chunk, lo, hi);
    float s = 0.f, q = 0.f;
    if (bn_try_pairs<2, T>(a, x, dy, (T*)nullptr, c, chunk, mean, invstd, g, bb, 0.f, 0.f, 0.f, s, q)) {}
    else if (a.HW >= 256) bn_bwd_walk<T, true, false>(a, x, dy, (T*)nullptr, c, lo, hi, mean, invstd, g, bb, 0.f, 0.f, 0.f, s, q);
    else bn_bwd_walk<T, false, false>(a, x, dy, (T*)nullptr, c, lo, hi, mean, invstd, g, bb, 0.f, 0.f, 0.f, s, q);
    bn_block_sum2(s, q, red);
    if (threadIdx.x == 0) { partial[((size_t)c * BN_CHUNKS + chunk) * 2] = s; partial[((size_t)c * BN_CHUNKS + chunk) * 2 + 1] = q; }
}

template <typename T>
__global__ __launch_bounds__(256)
void bn_bwd_apply_kernel(BnArgs a, const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ partial,
                         const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save_mean,
                         const float* __restrict__ save_invstd, T* __restrict__ dx, float* __restrict__ dgamma,
                         float* __restrict__ dbeta) {
    const int c = blockIdx.x, chunk = blockIdx.y;
    float s = 0.f, q = 0.f;
    for (int i = 0; i < BN_CHUNKS; ++i) { s += partial[((size_t)c * BN_CHUNKS + i) * 2]; q += partial[((size_t)c * BN_CHUNKS + i) * 2 + 1]; }
    if (chunk == 0 && threadIdx.x == 0) { if (dgamma) dgamma[c] = q; if (dbeta) dbeta[c] = s; }
    const float n = (float)((long)a.B * a.HW);
    const float mean = save_mean[c], invstd = save_invstd[c], g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    const float k0 = g * invstd, ms = s / n, mq = q / n;
    int lo, hi;
    bn_slice(a, chunk, lo, hi);
    float u0 = 0.f, u1 = 0.f;
    if (bn_try_pairs<3, T>(a, x, dy, dx, c, chunk, mean, invstd, g, bb, k0, ms, mq, u0, u1)) {}
    else if (a.HW >= 256) bn_bwd_walk<T, true, true>(a, x, dy, dx, c, lo, hi, mean, invstd, g, bb, k0, ms, mq, u0, u1);
    else bn_bwd_walk<T, false, true>(a, x, dy, dx, c, lo, hi, mean, invstd, g, bb, k0, ms, mq, u0, u1);
}

// bn_bwd_apply_kernel with the channel's sums given as `np` pairs (any count: the producer's workgroups -- hs_dw_tiles_bn_bwd_in -- instead of a
// statistics launch's 32 slices), combined here in pair order: thread t takes pairs t, t + 256, ..., then the workgroup's fixed tree.
template <typename T>
__global__ __launch_bounds__(256)
void bn_bwd_apply_np_kernel(BnArgs a, const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ partial, int np,
                            const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save_mean,
                            const float* __restrict__ save_invstd, T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[4][2];
    const int c = blockIdx.x, chunk = blockIdx.y;
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) { s += partial[((size_t)c * np + i) * 2]; q += partial[((size_t)c * np + i) * 2 + 1]; }
    bn_block_sum2(s, q, red);
    if (chunk == 0 && threadIdx.x == 0) { if (dgamma) dgamma[c] = q; if (dbeta) dbeta[c] = s; }
    const float n = (float)((long)a.B * a.HW);
    const float mean = save_mean[c], invstd = save_invstd[c], g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    const float k0 = g * invstd, ms = s / n, mq = q / n;
    int lo, hi;
    bn_slice(a, chunk, lo, hi);
    float u0 = 0.f, u1 = 0.f;
    if (bn_try_pairs<3, T>(a, x, dy, dx, c, chunk, mean, invstd, g, bb, k0, ms, mq, u0, u1)) {}
    else if (a.HW >= 256) bn_bwd_walk<T, true, true>(a, x, dy, dx, c, lo, hi, mean, invstd, g, bb, k0, ms, mq, u0, u1);
    else bn_bwd_walk<T, false, true>(a, x, dy, dx, c, lo, hi, mean, invstd, g, bb, k0, ms, mq, u0, u1);
}

// Small channels (B HW <= BN_SMALL_MAX = 16 x 1024 elements: the k = 1 levels at config 5) in ONE launch per direction: a workgroup of 1024
// threads owns a channel and holds it in registers -- every load of the channel in flight at once, statistics through LDS, the result from
// the registers.  The two-launch form costs such a layer 4 x ~5.5 us per step -- the launches' own floor -- for a few tens of KB.
// Same shift (the channel's first element), same formulas; the sums associate differently (one workgroup instead of 32 slices).
// (A first version that also took the 41 472-element channels of level 3, streaming them twice through one workgroup, measured 14.6 /
// 19.3 us per launch against 2 x 6.5 / 2 x 8.5: a channel that does not fit the registers wants the 32-slice grid.)
constexpr int BN_SMALL_PER = 16, BN_SMALL_THREADS = 1024, BN_SMALL_MAX = BN_SMALL_PER * BN_SMALL_THREADS;
__device__ __forceinline__ void bn_block_sum2_n(float& s0, float& s1, float (*red)[2]) {       // 16 waves, summed in wave order
    s0 = wave_sum64(s0); s1 = wave_sum64(s1);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = s0; red[wave][1] = s1; }
    __syncthreads();
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int w = 0; w < BN_SMALL_THREADS / 64; ++w) { t0 += red[w][0]; t1 += red[w][1]; }
    s0 = t0; s1 = t1;
}
// element k of this thread: e = tid + 1024 k of the channel's (batch, pixel) range; clamped index (the mask is applied by the caller)
__device__ __forceinline__ void bn_small_index(const BnArgs& a, int c, size_t (&idx)[BN_SMALL_PER]) {
    const int total = a.B * a.HW;
    const float inv_hw = 1.0f / (float)a.HW;
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; ++k) {
        const int e = min((int)threadIdx.x + BN_SMALL_THREADS * k, total - 1), b = div_by_inv(e, inv_hw), p = e - b * a.HW;
        idx[k] = ((size_t)b * a.C + c) * a.HW + p;
    }
}

template <typename T>
__global__ __launch_bounds__(BN_SMALL_THREADS)
void bn_fwd_small_kernel(BnArgs a, const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                         float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ save_mean,
                         float* __restrict__ save_invstd, T* __restrict__ y, long long* __restrict__ counter) {
    __shared__ float red[BN_SMALL_THREADS / 64][2];
    const int c = blockIdx.x, total = a.B * a.HW;
    if (counter && c == 0 && threadIdx.x == 0) *counter += 1;
    size_t idx[BN_SMALL_PER];
    bn_small_index(a, c, idx);
    float v[BN_SMALL_PER];
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; ++k) v[k] = Store<T>::ld(x, idx[k]);
    const float shift = Store<T>::ld(x, (size_t)c * a.HW);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; ++k) {
        const float d = ((int)threadIdx.x + BN_SMALL_THREADS * k < total) ? v[k] - shift : 0.0f;
        s += d; q = fmaf(d, d, q);
    }
    bn_block_sum2_n(s, q, red);
    const float n = (float)total;
    const float md = s / n, var = fmaxf(q / n - md * md, 0.f), mean = md + shift, invstd = rsqrtf(var + a.eps);
    if (threadIdx.x == 0) {
        save_mean[c] = mean; save_invstd[c] = invstd;
        if (running_mean) {
            running_mean[c] = (1.f - a.momentum) * running_mean[c] + a.momentum * mean;
            running_var[c] = (1.f - a.momentum) * running_var[c] + a.momentum * (n > 1.f ? var * n / (n - 1.f) : var);
        }
    }
    const float g = gamma ? gamma[c] * invstd : invstd, bb = (beta ? beta[c] : 0.f) - mean * g;
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; ++k)
        if ((int)threadIdx.x + BN_SMALL_THREADS * k < total) Store<T>::st(y, idx[k], bn_act(fmaf(v[k], g, bb), a.act));
}

template <typename T>
__global__ __launch_bounds__(BN_SMALL_THREADS)
void bn_bwd_small_kernel(BnArgs a, const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                         const float* __restrict__ beta, const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                         T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[BN_SMALL_THREADS / 64][2];
    const int c = blockIdx.x, total = a.B * a.HW;
    size_t idx[BN_SMALL_PER];
    bn_small_index(a, c, idx);
    float xh[BN_SMALL_PER], d[BN_SMALL_PER];
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; ++k) { xh[k] = Store<T>::ld(x, idx[k]); d[k] = Store<T>::ld(dy, idx[k]); }
    const float mean = save_mean[c], invstd = save_invstd[c], g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; ++k) {
        xh[k] = (xh[k] - mean) * invstd;
        d[k] = ((int)threadIdx.x + BN_SMALL_THREADS * k < total) ? d[k] * bn_act_grad(fmaf(xh[k], g, bb), a.act) : 0.0f;
        s += d[k]; q = fmaf(d[k], xh[k], q);
    }
    bn_block_sum2_n(s, q, red);
    if (threadIdx.x == 0) { if (dgamma) dgamma[c] = q; if (dbeta) dbeta[c] = s; }
    const float n = (float)total;
    const float k0 = g * invstd, ms = s / n, mq = q / n;
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; ++k)
        if ((int)threadIdx.x + BN_SMALL_THREADS * k < total) Store<T>::st(dx, idx[k], k0 * (d[k] - ms - xh[k] * mq));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Adjoint of F.interpolate(x, (Ho, Wo), 'bilinear', align_corners=False): dx[yi][xi] = sum over the output pixels whose taps touch
// (yi, xi) of their weights x dy -- a gather (no atomics): the candidate output rows / columns are those within the tap footprint,
// and each one's weight on this input index is read off the same bilinear_tap the forward uses (so the border clamping, where both
// taps of an output land on the edge sample, comes out by itself).
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void upsample_bilinear_bwd_kernel(const T* __restrict__ dy, int channels, long dy_batch_stride, int Hi, int Wi, int Ho, int Wo, float sy, float sx,
                                  T* __restrict__ dx) {
    const int xi = blockIdx.x * 64 + (threadIdx.x & 63), yi = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xi >= Wi || yi >= Hi) return;
    // outputs whose source coordinate lies within (yi - 1, yi + 1): o in ((yi - 0.5) / s - 0.5, (yi + 1.5) / s - 0.5)
    const int oy0 = max((int)floorf(((float)yi - 0.5f) / sy - 0.5f) - 1, 0), oy1 = min((int)ceilf(((float)yi + 1.5f) / sy - 0.5f) + 1, Ho - 1);
    const int ox0 = max((int)floorf(((float)xi - 0.5f) / sx - 0.5f) - 1, 0), ox1 = min((int)ceilf(((float)xi + 1.5f) / sx - 0.5f) + 1, Wo - 1);
    const int pb = (int)blockIdx.z / channels, pc = (int)blockIdx.z - pb * channels;             // dy may be a channel range of a wider tensor
    const T* __restrict__ g = dy + (size_t)pb * dy_batch_stride + (size_t)pc * Ho * Wo;
    float acc = 0.0f;
    for (int oy = oy0; oy <= oy1; ++oy) {
        const Tap ty = bilinear_tap(oy, sy, Hi);
        const float wy = (ty.i0 == yi ? ty.l0 : 0.0f) + (ty.i1 == yi ? ty.l1 : 0.0f);
        if (wy == 0.0f) continue;
        float row = 0.0f;
        for (int ox = ox0; ox <= ox1; ++ox) {
            const Tap tx = bilinear_tap(ox, sx, Wi);
            const float wx = (tx.i0 == xi ? tx.l0 : 0.0f) + (tx.i1 == xi ? tx.l1 : 0.0f);
            row = fmaf(wx, Store<T>::ld(g, (size_t)oy * Wo + ox), row);
        }
        acc = fmaf(wy, row, acc);
    }
    Store<T>::st(dx, ((size_t)blockIdx.z * Hi + yi) * Wi + xi, acc);
}

// The exact-2x case (every use in the decoder): out[2i] = 0.25 in[i-1] + 0.75 in[i], out[2i+1] = 0.75 in[i] + 0.25 in[i+1] with the source index
// clamped at both ends, so dIn[i] = 0.25 g[2i-1] + 0.75 g[2i] + 0.75 g[2i+1] + 0.25 g[2i+2], the weight of a tap that would leave the
// image folded onto the border output (1.0 g[0] at i = 0, 1.0 g[2L+1] at the last row / column).  16 loads and fused multiply-adds per
// input pixel, no tap arithmetic (the general kernel above walks the candidate outputs and recomputes their taps: 21 us per launch at
// config 5, the second-largest line of the training step in visit r4m).
// Round 6: a thread owns a 2 x 2 block of input pixels -- their footprints are 6 rows x 6 columns of g, fetched as 6 x 4 aligned pairs (24
// 8-byte loads for four values where the one-pixel form issued 64 4-byte loads: it ran at 2.3 TB/s on the step's largest map); per value the
// same two fma chains (columns, then rows), so dx is bit-identical.  A pair that would start outside the row is clamped inside it: every
// element read in its place carries weight 0, as the clamped single loads did.
template <typename T>
__global__ __launch_bounds__(256)
void upsample2x_bwd_kernel(const T* __restrict__ dy, int channels, long dy_batch_stride, int Hi, int Wi, T* __restrict__ dx) {
    const int xi0 = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)), yi0 = 2 * (blockIdx.y * 4 + (threadIdx.x >> 6));
    if (xi0 >= Wi || yi0 >= Hi) return;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const int pb = (int)blockIdx.z / channels, pc = (int)blockIdx.z - pb * channels;
    const T* __restrict__ g = dy + (size_t)pb * dy_batch_stride + (size_t)pc * Ho * Wo;
    float v[6][8];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const T* __restrict__ row = g + (size_t)min(max(2 * yi0 - 1 + r, 0), Ho - 1) * Wo;
#pragma unroll
        for (int q = 0; q < 4; ++q) Pair<T>::ld(row, (size_t)min(max(2 * xi0 - 2 + 2 * q, 0), Wo - 2), v[r][2 * q], v[r][2 * q + 1]);
    }
#pragma unroll
    for (int dyi = 0; dyi < 2; ++dyi)
#pragma unroll
        for (int dxi = 0; dxi < 2; ++dxi) {
            const int yi = yi0 + dyi, xi = xi0 + dxi;
            float wy[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wx[4] = {0.25f, 0.75f, 0.75f, 0.25f};
            if (yi == 0) { wy[0] = 0.0f; wy[1] = 1.0f; }
            if (yi == Hi - 1) { wy[3] = 0.0f; wy[2] = 1.0f; }
            if (xi == 0) { wx[0] = 0.0f; wx[1] = 1.0f; }
            if (xi == Wi - 1) { wx[3] = 0.0f; wx[2] = 1.0f; }
            float acc = 0.0f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float r = 0.0f;
#pragma unroll
                for (int b = 0; b < 4; ++b) r = fmaf(wx[b], v[2 * dyi + a][2 * dxi + b + 1], r);
                acc = fmaf(wy[a], r, acc);
            }
            if (yi < Hi && xi < Wi) Store<T>::st(dx, ((size_t)blockIdx.z * Hi + yi) * Wi + xi, acc);
        }
}

// Adjoint of hs_bank_pack_fwd: the patch-major gradient (B fh fw, ld) back to the reference's channel-major layout (B, hp_total, fh, fw),
// channels [ch_offset, ch_offset + rows) from the bank and exact zeros everywhere else -- 32 x 32 LDS transpose tiles, one launch
// (stock ops: a zeros fill + a permuted, uncoalesced copy).
__global__ __launch_bounds__(256)
void bank_unpack_kernel(const float* __restrict__ bank, long ld, int hp_total, int grid_sz, int ch_offset, int rows,
                        float* __restrict__ w) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int m_base = blockIdx.x * 32, q_base = blockIdx.y * 32;       // m: channel of w, q = i fw + j
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;             // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int q = q_base + r, m = m_base + tx - ch_offset;
        float v = 0.0f;
        if (m >= 0 && m < rows && q < grid_sz) v = bank[((size_t)b * grid_sz + q) * ld + m];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int m = m_base + r, q = q_base + tx;
        if (m < hp_total && q < grid_sz) w[((size_t)b * hp_total + m) * grid_sz + q] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Per-pixel cross entropy of (N, C, H, W) logits against (N, H, W) int64 labels (F.cross_entropy(..., reduction='none'), no class
// weights: what BootstrappedCrossEntropyLoss feeds its top-k rule, bootstrapped_ce_loss.py:20-23), forward and adjoint, one launch each
// (stock: log-softmax + gather, and their two adjoints -- four launches of ~25 us at config 5).  One thread per pixel; the C logits of a
// pixel are HW floats apart, so a wave reads C coalesced rows.  loss = log(sum exp(x - max)) + max - x[t]; ignored labels give 0 and
// no gradient; d x[c] = (softmax[c] - [c == t]) * g.
// ---------------------------------------------------------------------------------------------------------------------------------
// CF: the class count at compile time (12 CamVid, 19 Cityscapes, 21 VOC: the reference's datasets) -- a pixel's logits are then loaded ONCE
// into registers, all in flight together, instead of three dependent passes over them (max, sum, result); 0 = any count, the three passes.
// The upstream gradient of pixel e: g[e] as it is, or -- bootstrapped form (hs_bootstrapped_ce_bwd, round 6) -- formed HERE from the pixel's own loss
// g[e], its image's selection state (out8 of the forward: 1 / count, t, tie weight) and the ONE gradient of the batch mean, exactly as
// bm_bwd_kernel writes it (whose launch and whose (N, HW) gradient tensor this replaces).
__device__ __forceinline__ float ce_upstream(const float* __restrict__ g, long e, long n, const float* __restrict__ bstate,
                                             const float* __restrict__ gmean, float gscale) {
    const float v = g[e];
    if (!bstate) return v;
    const float* __restrict__ st = bstate + n * 8;
    const float w = st[2] * (gmean[0] * gscale), t = st[3], tie = st[4], xv = fmaxf(v, 0.0f);
    return xv > t ? w : (xv == t ? w * tie : 0.0f);
}

// (Round 6, visit x8: a forward variant that also took level 0 of the selection's histograms while it made the losses -- an LDS histogram per
//  workgroup pass, flushed with one global atomic per non-empty bin -- measured 38.5 us against 10.3 + 6.8 for the two launches: the loss launch
//  has 2592 workgroups where bm_hist_kernel has 256, and their flushes meet on the few hundred bins the losses crowd into.  Removed.)
template <bool BWD, typename T, int CF>
__global__ __launch_bounds__(256)
void cross_entropy_kernel(const T* __restrict__ x, const long long* __restrict__ target, int C, long hw, long total, long long ignore_index,
                          const float* __restrict__ g, void* __restrict__ out_, const float* __restrict__ bstate, const float* __restrict__ gmean,
                          float gscale) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long n = e / hw, p = e - n * hw;
        const T* __restrict__ xp = x + n * C * hw + p;
        const long long t = target[e];
        const bool live = t != ignore_index && t >= 0 && t < C;
        if constexpr (CF > 0) {
            float v[CF];
#pragma unroll
            for (int c = 0; c < CF; ++c) v[c] = Store<T>::ld(xp, (long)c * hw);
            const float gi = BWD ? ce_upstream(g, e, n, bstate, gmean, gscale) : 0.0f;
            float m = v[0];
#pragma unroll
            for (int c = 1; c < CF; ++c) m = fmaxf(m, v[c]);
            float sum = 0.0f, xt = 0.0f;
#pragma unroll
            for (int c = 0; c < CF; ++c) {
                sum += expf(v[c] - m);
                xt = (c == (int)t) ? v[c] : xt;
            }
            if constexpr (!BWD) {
                ((float*)out_)[e] = live ? (logf(sum) + m) - xt : 0.0f;
            } else {
                T* __restrict__ dp = (T*)out_ + n * C * hw + p;
                const float gl = live ? gi : 0.0f, inv = 1.0f / sum;
#pragma unroll
                for (int c = 0; c < CF; ++c) Store<T>::st(dp, (long)c * hw, (expf(v[c] - m) * inv - (c == (int)t ? 1.0f : 0.0f)) * gl);
            }
        } else {
            float m = Store<T>::ld(xp, 0);
            for (int c = 1; c < C; ++c) m = fmaxf(m, Store<T>::ld(xp, (long)c * hw));
            float sum = 0.0f;
            for (int c = 0; c < C; ++c) sum += expf(Store<T>::ld(xp, (long)c * hw) - m);
            if constexpr (!BWD) {
                ((float*)out_)[e] = live ? (logf(sum) + m) - Store<T>::ld(xp, (long)(live ? t : 0) * hw) : 0.0f;
            } else {
                T* __restrict__ dp = (T*)out_ + n * C * hw + p;
                const float gi = live ? ce_upstream(g, e, n, bstate, gmean, gscale) : 0.0f, inv = 1.0f / sum;
                for (int c = 0; c < C; ++c) Store<T>::st(dp, (long)c * hw, (expf(Store<T>::ld(xp, (long)c * hw) - m) * inv - (c == (int)t ? 1.0f : 0.0f)) * gi);
            }
        }
    }
}

template <bool BWD, typename T>
static void launch_cross_entropy(dim3 blocks, hipStream_t s, const T* x, const long long* target, int C, long hw, long total, long long ignore_index,
                                 const float* g, void* out, const float* bstate = nullptr, const float* gmean = nullptr, float gscale = 1.0f) {
    if (C == 12) hipLaunchKernelGGL((cross_entropy_kernel<BWD, T, 12>), blocks, dim3(256), 0, s, x, target, C, hw, total, ignore_index, g, out, bstate, gmean, gscale);
    else if (C == 19) hipLaunchKernelGGL((cross_entropy_kernel<BWD, T, 19>), blocks, dim3(256), 0, s, x, target, C, hw, total, ignore_index, g, out, bstate, gmean, gscale);
    else if (C == 21) hipLaunchKernelGGL((cross_entropy_kernel<BWD, T, 21>), blocks, dim3(256), 0, s, x, target, C, hw, total, ignore_index, g, out, bstate, gmean, gscale);
    else hipLaunchKernelGGL((cross_entropy_kernel<BWD, T, 0>), blocks, dim3(256), 0, s, x, target, C, hw, total, ignore_index, g, out, bstate, gmean, gscale);
}

}  // namespace hs

using namespace hs;

extern "C" int hs_bank_unpack_fwd(const float* bank, int64_t ld, int32_t batch, int32_t hp_total, int32_t fh, int32_t fw,
                                  int32_t ch_offset, int32_t rows, float* w, void* stream) {
    if (!bank || !w || batch <= 0 || hp_total <= 0 || fh <= 0 || fw <= 0 || rows <= 0 || ch_offset < 0) return HS_ERR_BAD_ARG;
    if (ch_offset + rows > hp_total || ld < rows || batch > 65535) return HS_ERR_BAD_ARG;
    const int grid_sz = fh * fw;
    hipLaunchKernelGGL(bank_unpack_kernel, dim3((hp_total + 31) / 32, (grid_sz + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream,
                       bank, (long)ld, hp_total, grid_sz, ch_offset, rows, w);
    return launch_status();
}

extern "C" int hs_upsample_bilinear_typed_bwd(int32_t dtype, const void* dy, int64_t dy_batch_stride, int32_t batch, int32_t channels, int32_t Hi,
                                              int32_t Wi, int32_t Ho, int32_t Wo, void* dx, void* stream) {
    if (!dy || !dx || batch <= 0 || channels <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return HS_ERR_BAD_ARG;
    if (dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16) return HS_ERR_BAD_ARG;
    if (dy_batch_stride <= 0) dy_batch_stride = (int64_t)channels * Ho * Wo;                         // 0 = packed (B, C, Ho, Wo)
    if (dy_batch_stride < (int64_t)channels * Ho * Wo) return HS_ERR_BAD_ARG;
    if ((long)batch * channels > 65535 || Ho < Hi || Wo < Wi) return HS_ERR_UNSUPPORTED;          // upsampling only (the decoder's use)
    const dim3 grid((Wi + 63) / 64, (Hi + 3) / 4, batch * channels);
    hipStream_t q = (hipStream_t)stream;
    const long bs = (long)dy_batch_stride;
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    // the 2 x 2 form reads aligned pairs: every row of g must start on a pair boundary (Wo = 2 Wi is even; the base and the batch stride decide)
    const size_t pair_bytes = dtype == HS_DTYPE_F32 ? 8 : 4;
    if (Ho == 2 * Hi && Wo == 2 * Wi && (reinterpret_cast<size_t>(dy) % pair_bytes) == 0 && (bs & 1) == 0) {
        const dim3 grid2(((Wi + 1) / 2 + 63) / 64, ((Hi + 1) / 2 + 3) / 4, batch * channels);
        if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, grid2, dim3(256), 0, q, (const float*)dy, channels, bs, Hi, Wi, (float*)dx);
        else hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16_t>, grid2, dim3(256), 0, q, (const bf16_t*)dy, channels, bs, Hi, Wi, (bf16_t*)dx);
        return launch_status();
    }
    if (dtype == HS_DTYPE_F32)
        hipLaunchKernelGGL(upsample_bilinear_bwd_kernel<float>, grid, dim3(256), 0, q, (const float*)dy, channels, bs, Hi, Wi, Ho, Wo, sy, sx, (float*)dx);
    else
        hipLaunchKernelGGL(upsample_bilinear_bwd_kernel<bf16_t>, grid, dim3(256), 0, q, (const bf16_t*)dy, channels, bs, Hi, Wi, Ho, Wo, sy, sx, (bf16_t*)dx);
    return launch_status();
}

extern "C" int hs_upsample_bilinear_bwd(const float* dy, int64_t dy_batch_stride, int32_t batch, int32_t channels, int32_t Hi, int32_t Wi,
                                        int32_t Ho, int32_t Wo, float* dx, void* stream) {
    return hs_upsample_bilinear_typed_bwd(HS_DTYPE_F32, dy, dy_batch_stride, batch, channels, Hi, Wi, Ho, Wo, dx, stream);
}

static int bn_args(BnArgs& a, int B, int C, long hw, int act, float eps, float momentum) {
    if (B <= 0 || C <= 0 || hw <= 0 || act < HS_ACT_NONE || act > HS_ACT_RELU6 || eps < 0.f) return HS_ERR_BAD_ARG;
    if (C > 65535 || (long)B * hw > 0x7fffffffL) return HS_ERR_UNSUPPORTED;
    a = BnArgs{B, C, (int)hw, act, eps, momentum};
    return HS_OK;
}

extern "C" int64_t hs_bn_train_workspace(int32_t channels) { return (int64_t)channels * BN_CHUNKS * 2 * 4; }

extern "C" int hs_bn_act_train_fwd(int32_t dtype, const void* x, int32_t batch, int32_t channels, int64_t pixels, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum, float eps, int32_t act,
                                   float* save_mean, float* save_invstd, void* workspace, void* y, int64_t* num_batches_tracked,
                                   void* stream) {
    BnArgs a;
    const int st = bn_args(a, batch, channels, pixels, act, eps, momentum);
    if (st != HS_OK) return st;
    if (!x || !y || !save_mean || !save_invstd || !workspace || ((running_mean != nullptr) != (running_var != nullptr))) return HS_ERR_BAD_ARG;
    const dim3 grid(channels, BN_CHUNKS);
    hipStream_t s = (hipStream_t)stream;
    if ((long)batch * pixels <= BN_SMALL_MAX && (dtype == HS_DTYPE_F32 || dtype == HS_DTYPE_BF16)) {            // one launch: a workgroup per channel
        if (dtype == HS_DTYPE_F32)
            hipLaunchKernelGGL(bn_fwd_small_kernel<float>, dim3(channels), dim3(BN_SMALL_THREADS), 0, s, a, (const float*)x, gamma, beta, running_mean,
                               running_var, save_mean, save_invstd, (float*)y, (long long*)num_batches_tracked);
        else
            hipLaunchKernelGGL(bn_fwd_small_kernel<bf16_t>, dim3(channels), dim3(BN_SMALL_THREADS), 0, s, a, (const bf16_t*)x, gamma, beta, running_mean,
                               running_var, save_mean, save_invstd, (bf16_t*)y, (long long*)num_batches_tracked);
        return launch_status();
    }
    if (dtype == HS_DTYPE_F32) {
        hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, s, a, (const float*)x, (float*)workspace);
        hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, s, a, (const float*)x, (const float*)workspace, gamma, beta,
                           running_mean, running_var, save_mean, save_invstd, (float*)y, (long long*)num_batches_tracked);
    } else if (dtype == HS_DTYPE_BF16) {
        hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, dim3(256), 0, s, a, (const bf16_t*)x, (float*)workspace);
        hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, grid, dim3(256), 0, s, a, (const bf16_t*)x, (const float*)workspace, gamma, beta,
                           running_mean, running_var, save_mean, save_invstd, (bf16_t*)y, (long long*)num_batches_tracked);
    } else return HS_ERR_BAD_ARG;
    return launch_status();
}

// The statistics pass alone (partial sums per channel and slice into `workspace`): for consumers that normalise on load
// (hs_dw_tiles_bn_fwd) instead of reading a normalised copy
extern "C" int hs_bn_train_stats_fwd(int32_t dtype, const void* x, int32_t batch, int32_t channels, int64_t pixels, void* workspace, void* stream) {
    BnArgs a;
    const int st = bn_args(a, batch, channels, pixels, HS_ACT_NONE, 0.f, 0.f);
    if (st != HS_OK) return st;
    if (!x || !workspace) return HS_ERR_BAD_ARG;
    const dim3 grid(channels, BN_CHUNKS);
    if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a, (const float*)x, (float*)workspace);
    else if (dtype == HS_DTYPE_BF16) hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a, (const bf16_t*)x, (float*)workspace);
    else return HS_ERR_BAD_ARG;
    return launch_status();
}

extern "C" int hs_bn_act_train_bwd(int32_t dtype, const void* x, const void* dy, int32_t batch, int32_t channels, int64_t pixels,
                                   const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, float eps,
                                   int32_t act, void* workspace, void* dx, float* dgamma, float* dbeta, void* stream) {
    BnArgs a;
    const int st = bn_args(a, batch, channels, pixels, act, eps, 0.f);
    if (st != HS_OK) return st;
    if (!x || !dy || !dx || !save_mean || !save_invstd || !workspace) return HS_ERR_BAD_ARG;
    const dim3 grid(channels, BN_CHUNKS);
    hipStream_t s = (hipStream_t)stream;
    if ((long)batch * pixels <= BN_SMALL_MAX && (dtype == HS_DTYPE_F32 || dtype == HS_DTYPE_BF16)) {
        if (dtype == HS_DTYPE_F32)
            hipLaunchKernelGGL(bn_bwd_small_kernel<float>, dim3(channels), dim3(BN_SMALL_THREADS), 0, s, a, (const float*)x, (const float*)dy, gamma, beta,
                               save_mean, save_invstd, (float*)dx, dgamma, dbeta);
        else
            hipLaunchKernelGGL(bn_bwd_small_kernel<bf16_t>, dim3(channels), dim3(BN_SMALL_THREADS), 0, s, a, (const bf16_t*)x, (const bf16_t*)dy, gamma, beta,
                               save_mean, save_invstd, (bf16_t*)dx, dgamma, dbeta);
        return launch_status();
    }
    if (dtype == HS_DTYPE_F32) {
        hipLaunchKernelGGL(bn_bwd_stats_kernel<float>, grid, dim3(256), 0, s, a, (const float*)x, (const float*)dy, gamma, beta, save_mean,
                           save_invstd, (float*)workspace);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, s, a, (const float*)x, (const float*)dy, (const float*)workspace,
                           gamma, beta, save_mean, save_invstd, (float*)dx, dgamma, dbeta);
    } else if (dtype == HS_DTYPE_BF16) {
        hipLaunchKernelGGL(bn_bwd_stats_kernel<bf16_t>, grid, dim3(256), 0, s, a, (const bf16_t*)x, (const bf16_t*)dy, gamma, beta, save_mean,
                           save_invstd, (float*)workspace);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, s, a, (const bf16_t*)x, (const bf16_t*)dy, (const float*)workspace,
                           gamma, beta, save_mean, save_invstd, (bf16_t*)dx, dgamma, dbeta);
    } else return HS_ERR_BAD_ARG;
    return launch_status();
}

// The second half of hs_bn_act_train_bwd alone, from sums some producer left as `n_partials` {sum d, sum d x_hat} pairs per channel
// (partial[(c n_partials + i) 2 + {0, 1}]; hs_dw_tiles_bn_bwd_in): dx, dgamma, dbeta.  One launch.
extern "C" int hs_bn_act_train_bwd_apply(int32_t dtype, const void* x, const void* dy, int32_t batch, int32_t channels, int64_t pixels,
                                         const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int32_t act,
                                         const float* partial, int64_t n_partials, void* dx, float* dgamma, float* dbeta, void* stream) {
    BnArgs a;
    const int st = bn_args(a, batch, channels, pixels, act, 0.f, 0.f);
    if (st != HS_OK) return st;
    if (!x || !dy || !dx || !save_mean || !save_invstd || !partial || n_partials <= 0 || n_partials > 0x7fffffffL) return HS_ERR_BAD_ARG;
    const dim3 grid(channels, BN_CHUNKS);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == HS_DTYPE_F32)
        hipLaunchKernelGGL(bn_bwd_apply_np_kernel<float>, grid, dim3(256), 0, s, a, (const float*)x, (const float*)dy, partial, (int)n_partials, gamma, beta,
                           save_mean, save_invstd, (float*)dx, dgamma, dbeta);
    else if (dtype == HS_DTYPE_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_np_kernel<bf16_t>, grid, dim3(256), 0, s, a, (const bf16_t*)x, (const bf16_t*)dy, partial, (int)n_partials, gamma, beta,
                           save_mean, save_invstd, (bf16_t*)dx, dgamma, dbeta);
    else return HS_ERR_BAD_ARG;
    return launch_status();
}

static int tile_args(TileArgs& a, int B, int C, int H, int W, int fh, int fw, int pm = 0) {
    if (B <= 0 || C <= 0 || H < 2 || W < 2 || fh <= 0 || fw <= 0) return HS_ERR_BAD_ARG;
    if (H % fh || W % fw) return HS_ERR_NOT_DIVISIBLE;
    if ((long)B * C > 65535) return HS_ERR_UNSUPPORTED;
    a = TileArgs{B, C, H, W, fh, fw, H / fh, W / fw, 1.0f / (float)(H / fh), 1.0f / (float)(W / fw), 1.0f / (float)(H / fh + 2), 1.0f / (float)(W / fw + 2), pm ? 1 : 0, 1.0f / (float)C};
    if (H + 2 * fh >= (1 << 21) || W + 2 * fw >= (1 << 21)) return HS_ERR_UNSUPPORTED;          // div_by_inv's range
    return HS_OK;
}

#define HS_TILE_LAUNCH(dtype, KERNEL_F32, KERNEL_BF16, grid, SRC, DST) \
    if ((dtype) == HS_DTYPE_F32) hipLaunchKernelGGL(KERNEL_F32, grid, dim3(256), 0, (hipStream_t)stream, a, (const float*)(SRC), (float*)(DST)); \
    else if ((dtype) == HS_DTYPE_BF16) hipLaunchKernelGGL(KERNEL_BF16, grid, dim3(256), 0, (hipStream_t)stream, a, (const bf16_t*)(SRC), (bf16_t*)(DST)); \
    else return HS_ERR_BAD_ARG;

extern "C" int hs_halo_tiles_fwd(int32_t dtype, const void* x, int32_t batch, int32_t channels, int32_t H, int32_t W, int32_t fh,
                                 int32_t fw, void* tiled, int32_t patch_major, void* stream) {
    TileArgs a;
    const int st = tile_args(a, batch, channels, H, W, fh, fw, patch_major);
    if (st != HS_OK) return st;
    if (!x || !tiled) return HS_ERR_BAD_ARG;
    const dim3 grid((fw * (a.pw + 2) + 63) / 64, (fh * (a.ph + 2) + 15) / 16, (batch * channels + 1) / 2);        // four rows of two planes per thread
    HS_TILE_LAUNCH(dtype, halo_tiles_fwd_kernel<float>, halo_tiles_fwd_kernel<bf16_t>, grid, x, tiled)
    return launch_status();
}

extern "C" int hs_halo_tiles_bwd(int32_t dtype, const void* dtiled, int32_t batch, int32_t channels, int32_t H, int32_t W,
                                 int32_t fh, int32_t fw, void* dx, int32_t patch_major, void* stream) {
    TileArgs a;
    const int st = tile_args(a, batch, channels, H, W, fh, fw, patch_major);
    if (st != HS_OK) return st;
    if (!dtiled || !dx) return HS_ERR_BAD_ARG;
    const int npx = a.ph * a.pw;
    if (a.ph >= 3 && a.pw >= 3 && npx <= 256 && 256 % npx == 0 && batch <= 65535) {       // mapped by patch
        const int per = 256 / npx;
        const float inv_npx = 1.0f / (float)npx;
        const int ch = per == 1 ? HS_HALO_BWD_CH_BIG : HS_HALO_BWD_CH_SMALL;      // channels per thread
        const dim3 gridp((unsigned)(fh * fw), (unsigned)((channels + per * ch - 1) / (per * ch)), (unsigned)batch);
        if (dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16) return HS_ERR_BAD_ARG;
#define HS_HB_LAUNCH(CH) \
        if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((halo_tiles_bwd_patch_kernel<float, CH>), gridp, dim3(256), 0, (hipStream_t)stream, a, (const float*)dtiled, (float*)dx, inv_npx); \
        else hipLaunchKernelGGL((halo_tiles_bwd_patch_kernel<bf16_t, CH>), gridp, dim3(256), 0, (hipStream_t)stream, a, (const bf16_t*)dtiled, (bf16_t*)dx, inv_npx);
        if (ch == 8) { HS_HB_LAUNCH(8) } else if (ch == 4) { HS_HB_LAUNCH(4) } else { HS_HB_LAUNCH(2) }
#undef HS_HB_LAUNCH
        return launch_status();
    }
    const dim3 grid((W + 63) / 64, (H + 3) / 4, batch * channels);
    HS_TILE_LAUNCH(dtype, halo_tiles_bwd_kernel<float>, halo_tiles_bwd_kernel<bf16_t>, grid, dtiled, dx)
    return launch_status();
}

extern "C" int hs_tile_interior_fwd(int32_t dtype, const void* tiled, int32_t batch, int32_t channels, int32_t H, int32_t W,
                                    int32_t fh, int32_t fw, void* y, void* stream) {
    TileArgs a;
    const int st = tile_args(a, batch, channels, H, W, fh, fw);
    if (st != HS_OK) return st;
    if (!tiled || !y) return HS_ERR_BAD_ARG;
    const dim3 grid((W + 63) / 64, (H + 3) / 4, batch * channels);
    HS_TILE_LAUNCH(dtype, (tile_interior_kernel<float, false>), (tile_interior_kernel<bf16_t, false>), grid, tiled, y)
    return launch_status();
}

extern "C" int hs_tile_interior_bwd(int32_t dtype, const void* dy, int32_t batch, int32_t channels, int32_t H, int32_t W,
                                    int32_t fh, int32_t fw, void* dtiled, void* stream) {
    TileArgs a;
    const int st = tile_args(a, batch, channels, H, W, fh, fw);
    if (st != HS_OK) return st;
    if (!dy || !dtiled) return HS_ERR_BAD_ARG;
    const dim3 grid((fw * (a.pw + 2) + 63) / 64, (fh * (a.ph + 2) + 3) / 4, batch * channels);
    HS_TILE_LAUNCH(dtype, (tile_interior_kernel<float, true>), (tile_interior_kernel<bf16_t, true>), grid, dy, dtiled)
    return launch_status();
}

extern "C" int64_t hs_bootstrap_mean_workspace(void) { return (int64_t)(BM_STATE + 8) * 4 + (int64_t)BM_WG * 4 * 4; }

// `images` independent reductions in one set of launches (grid.y = image): values (images, n), workspace images x
// hs_bootstrap_mean_workspace() bytes, out (images, 8) floats [loss, branch, 1 / count, t, tie weight, -, -, -]
static int bootstrap_mean_fwd(const float* values, int32_t images, int32_t n, int32_t k, float thresh, void* workspace, float* out8,
                              float* mean_out, void* stream) {
    if (!values || !workspace || !out8 || n <= 0 || k <= 0 || images <= 0 || images > 65535) return HS_ERR_BAD_ARG;
    if (n <= k) return HS_ERR_UNSUPPORTED;                               // the reference indexes ranked[k]
    hipStream_t s = (hipStream_t)stream;
    unsigned* ws = (unsigned*)workspace;
    float* partial = (float*)(ws + BM_STATE + 8);
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)images * BM_WS_U32 * 4, s);
    if (e != hipSuccess) return (int)e;
    for (int level = 0; level < 3; ++level)
        hipLaunchKernelGGL(bm_hist_kernel, dim3(BM_WG, images), dim3(256), 0, s, values, n, ws, level, k);
    hipLaunchKernelGGL(bm_sums_kernel, dim3(BM_WG, images), dim3(256), 0, s, values, n, thresh, ws, partial, k, out8, mean_out);
    return launch_status();
}
extern "C" int hs_bootstrap_mean_batched_fwd(const float* values, int32_t images, int32_t n, int32_t k, float thresh, void* workspace,
                                             float* out8, void* stream) {
    return bootstrap_mean_fwd(values, images, n, k, thresh, workspace, out8, nullptr, stream);
}

extern "C" int hs_bootstrap_mean_batched_bwd(const float* values, int32_t images, int32_t n, const float* state8, const float* grad_out,
                                             float* grad_values, void* stream) {
    if (!values || !state8 || !grad_out || !grad_values || n <= 0 || images <= 0 || images > 65535) return HS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bm_bwd_kernel, dim3(BM_WG, images), dim3(256), 0, (hipStream_t)stream, values, n, state8, grad_out, 1, 1.0f, grad_values);
    return launch_status();
}

// The batch form the loss module uses (round 5): hs_bootstrap_mean_batched_fwd + the mean of the per-image losses in `mean_out` (one
// float), and its adjoint from ONE upstream gradient (of that mean).
extern "C" int hs_bootstrap_mean_of_batch_fwd(const float* values, int32_t images, int32_t n, int32_t k, float thresh, void* workspace,
                                              float* out8, float* mean_out, void* stream) {
    if (!mean_out) return HS_ERR_BAD_ARG;
    return bootstrap_mean_fwd(values, images, n, k, thresh, workspace, out8, mean_out, stream);
}

extern "C" int hs_bootstrap_mean_of_batch_bwd(const float* values, int32_t images, int32_t n, const float* state8, const float* grad_mean,
                                              float* grad_values, void* stream) {
    if (!values || !state8 || !grad_mean || !grad_values || n <= 0 || images <= 0 || images > 65535) return HS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bm_bwd_kernel, dim3(BM_WG, images), dim3(256), 0, (hipStream_t)stream, values, n, state8, grad_mean, 0, 1.0f / (float)images,
                       grad_values);
    return launch_status();
}

// one image: out5 = the first five floats of the batched form's row (five floats are written)
extern "C" int hs_bootstrap_mean_fwd(const float* values, int32_t n, int32_t k, float thresh, void* workspace, float* out5,
                                     void* stream) {
    return hs_bootstrap_mean_batched_fwd(values, 1, n, k, thresh, workspace, out5, stream);
}

extern "C" int hs_bootstrap_mean_bwd(const float* values, int32_t n, const float* state5, const float* grad_out, float* grad_values,
                                     void* stream) {
    return hs_bootstrap_mean_batched_bwd(values, 1, n, state5, grad_out, grad_values, stream);
}

extern "C" int hs_cross_entropy_typed_fwd(int32_t dtype, const void* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                                          int64_t ignore_index, float* loss, void* stream) {
    if (!logits || !target || !loss || batch <= 0 || classes <= 0 || pixels <= 0) return HS_ERR_BAD_ARG;
    if (dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16) return HS_ERR_BAD_ARG;
    const long total = (long)batch * pixels;
    const dim3 blocks((unsigned)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256));
    if (dtype == HS_DTYPE_F32)
        launch_cross_entropy<false, float>(blocks, (hipStream_t)stream, (const float*)logits, (const long long*)target, classes, (long)pixels, total,
                                           (long long)ignore_index, nullptr, (void*)loss);
    else
        launch_cross_entropy<false, bf16_t>(blocks, (hipStream_t)stream, (const bf16_t*)logits, (const long long*)target, classes, (long)pixels, total,
                                            (long long)ignore_index, nullptr, (void*)loss);
    return launch_status();
}

extern "C" int hs_cross_entropy_typed_bwd(int32_t dtype, const void* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                                          int64_t ignore_index, const float* grad_loss, void* grad_logits, void* stream) {
    if (!logits || !target || !grad_loss || !grad_logits || batch <= 0 || classes <= 0 || pixels <= 0) return HS_ERR_BAD_ARG;
    if (dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16) return HS_ERR_BAD_ARG;
    const long total = (long)batch * pixels;
    const dim3 blocks((unsigned)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256));
    if (dtype == HS_DTYPE_F32)
        launch_cross_entropy<true, float>(blocks, (hipStream_t)stream, (const float*)logits, (const long long*)target, classes, (long)pixels, total,
                                          (long long)ignore_index, grad_loss, grad_logits);
    else
        launch_cross_entropy<true, bf16_t>(blocks, (hipStream_t)stream, (const bf16_t*)logits, (const long long*)target, classes, (long)pixels, total,
                                           (long long)ignore_index, grad_loss, grad_logits);
    return launch_status();
}

extern "C" int hs_cross_entropy_fwd(const float* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                                    int64_t ignore_index, float* loss, void* stream) {
    return hs_cross_entropy_typed_fwd(HS_DTYPE_F32, logits, target, batch, classes, pixels, ignore_index, loss, stream);
}

extern "C" int hs_cross_entropy_bwd(const float* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                                    int64_t ignore_index, const float* grad_loss, float* grad_logits, void* stream) {
    return hs_cross_entropy_typed_bwd(HS_DTYPE_F32, logits, target, batch, classes, pixels, ignore_index, grad_loss, grad_logits, stream);
}

// BootstrappedCrossEntropyLoss.forward as ONE entry (round 6; hyperseg/losses/bootstrapped_ce_loss.py:15-27): hs_cross_entropy_typed_fwd +
// hs_bootstrap_mean_of_batch_fwd (six launches), keeping what the one-launch adjoint below needs.  Same values bit for bit.
// (Clearing the selection's workspace inside the loss launch instead of by the memset launch was tried -- visits x20 - x22: the memset's 5 us go, but
//  the four launches that count into the workspace afterwards get slower by 4 us between them (bm_sums_kernel 7.0 -> 9.7), plain or write-through
//  stores alike: a wash, not in.)
extern "C" int hs_bootstrapped_ce_fwd(int32_t dtype, const void* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                                      int64_t ignore_index, int32_t k, float thresh, void* workspace, float* loss, float* out8, float* mean_out,
                                      void* stream) {
    if (!workspace || !out8 || !mean_out || k <= 0 || batch > 65535) return HS_ERR_BAD_ARG;
    if (pixels <= k || pixels > 0x7fffffffL) return HS_ERR_UNSUPPORTED;  // the reference indexes ranked[k]
    const int st = hs_cross_entropy_typed_fwd(dtype, logits, target, batch, classes, pixels, ignore_index, loss, stream);
    if (st != HS_OK) return st;
    return bootstrap_mean_fwd(loss, batch, (int32_t)pixels, k, thresh, workspace, out8, mean_out, stream);
}

// ... and its adjoint as ONE launch: grad_logits = (softmax - onehot) x the pixel's weight, the weight formed in the launch from the saved losses,
// the saved selection state (out8) and the one upstream gradient of the mean (hs_bootstrap_mean_of_batch_bwd + hs_cross_entropy_typed_bwd: two
// launches and an (N, HW) tensor between them).
extern "C" int hs_bootstrapped_ce_bwd(int32_t dtype, const void* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                                      int64_t ignore_index, const float* loss, const float* state8, const float* grad_mean, void* grad_logits,
                                      void* stream) {
    if (!logits || !target || !loss || !state8 || !grad_mean || !grad_logits || batch <= 0 || classes <= 0 || pixels <= 0) return HS_ERR_BAD_ARG;
    if (dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16) return HS_ERR_BAD_ARG;
    const long total = (long)batch * pixels;
    const dim3 blocks((unsigned)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256));
    if (dtype == HS_DTYPE_F32)
        launch_cross_entropy<true, float>(blocks, (hipStream_t)stream, (const float*)logits, (const long long*)target, classes, (long)pixels, total,
                                          (long long)ignore_index, loss, grad_logits, state8, grad_mean, 1.0f / (float)batch);
    else
        launch_cross_entropy<true, bf16_t>(blocks, (hipStream_t)stream, (const bf16_t*)logits, (const long long*)target, classes, (long)pixels, total,
                                           (long long)ignore_index, loss, grad_logits, state8, grad_mean, 1.0f / (float)batch);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------------------
// Adam for a whole parameter list in ONE launch (round 5; the training step's optimizer, hyperseg/train.py:185-188 builds torch.optim.Adam).
// torch's fused Adam is one launch too, but cuts the list into 65 536-element chunks -- a dozen workgroups for the decoder's ~0.7 M
// parameters, 25 us of a 0.88 ms step; here a workgroup takes 1024 elements (float4 per thread), ~700 workgroups.
// Arithmetic as torch's adam_math (fused_adam_utils.cuh), opmath float:  g += wd p (coupled) | p -= lr wd p (decoupled);
// m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps).
// The step count lives on the device (graph replay freezes kernel arguments): one word PER WORKGROUP, read and incremented by its own
// workgroup only (a shared word would be read by some workgroups after another had incremented it).
// ------------------------------------------------------------------------------------------------------------------------------
namespace hs {
constexpr int ADAM_MAX_TENSORS = 48, ADAM_BLOCK = 1024;
struct AdamTable {
    float* p[ADAM_MAX_TENSORS]; const float* g[ADAM_MAX_TENSORS]; float* m[ADAM_MAX_TENSORS]; float* v[ADAM_MAX_TENSORS];
    int first_block[ADAM_MAX_TENSORS + 1];       // workgroups [first_block[i], first_block[i + 1]) belong to tensor i
    long numel[ADAM_MAX_TENSORS];
    int n;
};

__global__ __launch_bounds__(256)
void adam_kernel(AdamTable t, const float* __restrict__ lr_dev, float lr_host, double b1d, double b2d, float eps, float wd, int decoupled,
                 int maximize, float* __restrict__ steps) {
    const __attribute__((address_space(4))) AdamTable* kt = (const __attribute__((address_space(4))) AdamTable*)__builtin_amdgcn_kernarg_segment_ptr();
    const int blk = (int)blockIdx.x;
    int i = 0;
    for (int q = 1; q < kt->n; ++q)
        if (blk >= kt->first_block[q]) i = q;
    const float step = steps[blk] + 1.0f;
    const float lr = lr_dev ? *lr_dev : lr_host;
    // the betas are doubles as in torch (1 - 0.999f is 1.3e-5 off 1 - 0.999), narrowed where torch narrows them
    const float b2 = (float)b2d, om1 = (float)(1.0 - b1d), om2 = (float)(1.0 - b2d);
    const float bc1 = (float)(1.0 - pow(b1d, (double)step)), bc2 = (float)(1.0 - pow(b2d, (double)step));
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
    const long n = kt->numel[i], e0 = (long)(blk - kt->first_block[i]) * ADAM_BLOCK + 4 * (long)threadIdx.x;
    float* __restrict__ p = kt->p[i]; const float* __restrict__ g = kt->g[i];
    float* __restrict__ m = kt->m[i]; float* __restrict__ v = kt->v[i];
    auto one = [&](float& pv, float gv, float& mv, float& vv) {
        if (maximize) gv = -gv;
        if (wd != 0.0f) { if (decoupled) pv -= lr * wd * pv; else gv += wd * pv; }
        { const float d = gv - mv; mv = om1 < 0.5f ? mv + om1 * d : gv - d * (1.0f - om1); }      // std::lerp(m, g, 1 - b1), as torch
        vv = b2 * vv + om2 * gv * gv;
        pv -= step_size * mv / (sqrtf(vv) / bc2_sqrt + eps);
    };
    if (e0 + 3 < n && ((((size_t)p) | ((size_t)g) | ((size_t)m) | ((size_t)v)) & 15) == 0) {
        float4 pv = *reinterpret_cast<const float4*>(p + e0), mv = *reinterpret_cast<const float4*>(m + e0), vv = *reinterpret_cast<const float4*>(v + e0);
        const float4 gv = *reinterpret_cast<const float4*>(g + e0);
        one(pv.x, gv.x, mv.x, vv.x); one(pv.y, gv.y, mv.y, vv.y); one(pv.z, gv.z, mv.z, vv.z); one(pv.w, gv.w, mv.w, vv.w);
        *reinterpret_cast<float4*>(p + e0) = pv; *reinterpret_cast<float4*>(m + e0) = mv; *reinterpret_cast<float4*>(v + e0) = vv;
    } else {
        for (long e = e0; e < n && e < e0 + 4; ++e) {
            float pv = p[e], mv = m[e], vv = v[e];
            one(pv, g[e], mv, vv);
            p[e] = pv; m[e] = mv; v[e] = vv;
        }
    }
    __syncthreads();                                           // every thread has read the step word
    if (threadIdx.x == 0) steps[blk] = step;
}
}  // namespace hs

extern "C" int64_t hs_adam_blocks(const int64_t* numel, int32_t n) {
    if (!numel || n <= 0 || n > hs::ADAM_MAX_TENSORS) return 0;
    int64_t b = 0;
    for (int i = 0; i < n; ++i) { if (numel[i] <= 0) return 0; b += (numel[i] + hs::ADAM_BLOCK - 1) / hs::ADAM_BLOCK; }
    return b < 0x7fffffff ? b : 0;
}

extern "C" int hs_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                            int32_t n, const float* lr_device, float lr, double beta1, double beta2, float eps, float weight_decay, int32_t decoupled,
                            int32_t maximize, float* steps, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !steps || n <= 0) return HS_ERR_BAD_ARG;
    if (n > hs::ADAM_MAX_TENSORS) return HS_ERR_UNSUPPORTED;
    hs::AdamTable t{};
    int b = 0;
    for (int i = 0; i < n; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] <= 0) return HS_ERR_BAD_ARG;
        t.p[i] = params[i]; t.g[i] = grads[i]; t.m[i] = exp_avg[i]; t.v[i] = exp_avg_sq[i]; t.numel[i] = (long)numel[i];
        t.first_block[i] = b;
        const int64_t nb = (numel[i] + hs::ADAM_BLOCK - 1) / hs::ADAM_BLOCK;
        if (nb + b >= 0x7fffffff) return HS_ERR_UNSUPPORTED;
        b += (int)nb;
    }
    t.first_block[n] = b; t.n = n;
    hipLaunchKernelGGL(hs::adam_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, t, lr_device, lr, beta1, beta2, eps,
                       weight_decay, decoupled, maximize, steps);
    return hs::launch_status();
}

// Backward of the dynamic patch-wise convolution (Op A / Op B), fp32.  The reference defines no backward (autograd
// differentiates F.pad / unfold / grouped conv2d / fold: SURVEY.md Appendix E); these are the two adjoints stated there:
//
//   per-patch weight gradient (BASELINE config 5's "per-patch weight-grad kernel")
//     dBank[p, ((o*cin_g + c)*k + ky)*k + kx] = sum_{(y,x) in patch p} dY[b,o,y,x] * Xpad[b, grp(o)*cin_g + c, y+ky-pad, x+kx-pad]
//   input gradient
//     dX[b,c,y,x] = sum over every PADDED coordinate (yp,xp) that the padding maps onto (y,x), every tap (ky,kx) and every
//                   output channel o of c's group:  W_{patch(y',x')}[o,c,ky,kx] * dY[b,o,y',x'],  (y',x') = (yp-ky+pad, xp-kx+pad)
//     -- the weights are those of the patch that owns the OUTPUT pixel, and reflect / replicate / circular padding
//     folds the halo gradients back onto their source pixels (adjoint of F.pad).
// Both kernels are straightforward (correctness first; forward is the tuned path): bwd_weight is one workgroup per patch
// with the patch's dY and padded X tiles in LDS, bwd_input one thread per input element.
#include "hs_common.h"

// dev A/B switches of round 6's row-count changes (the product build leaves them at 1)
#ifndef HS_DWT_FWD_RPT4
#define HS_DWT_FWD_RPT4 1
#endif
#ifndef HS_DWT_BWD_IN_RPT3
#define HS_DWT_BWD_IN_RPT3 1
#endif
#ifndef HS_DWT_BWD_W_CPW
#define HS_DWT_BWD_W_CPW 1              // 2 measured slower (21.8 -> 23.5 us at config 5's level 4, visit x10): left at 1
#endif

namespace hs {

struct ConvBwdArgs {
    const float* __restrict__ x;       // (B, cin, H, W)
    const float* __restrict__ dy;      // (B, cout, H, W)
    const float* __restrict__ bank;    // (P, ld)
    float* __restrict__ dx;            // (B, cin, H, W)
    float* __restrict__ dbank;         // (P, ld)
    long ld;
    int B, H, W, fh, fw, ph, pw, cin, cout, k, pad, pad_mode, groups, cin_g, cout_g;
    float inv_ph, inv_pw;              // 1 / ph, 1 / pw (div_by_inv: the image-level depthwise kernels)
};

int try_fast_fwd(int dtype, const void* x, const void* bank, long ld, int batch, int c_in, int H, int W, int fh, int fw, int c_out,
                 int k, int pad, int pad_mode, int groups, const float* scale, const float* shift, int act, void* y, hipStream_t stream);
int try_fast_bwd_in(int dtype, const void* dy, const void* bank, long ld, int batch, int c_in, int H, int W, int fh, int fw,
                    int c_out, int k, int pad, int pad_mode, int groups, void* dx, hipStream_t stream);
int try_fast_bwd_w(int dtype, const void* x, const void* dy, int batch, int c_in, int H, int W, int fh, int fw, int c_out, int k,
                   int pad, int pad_mode, int groups, void* dbank, long ld, hipStream_t stream);

// number of padded coordinates (in [-pad, n+pad)) that map onto index i, and the q-th of them
__device__ __forceinline__ int pad_aliases(int i, int n, int pad, int mode, int* out) {
    int cnt = 0;
    out[cnt++] = i;
    if (pad == 0 || mode == HS_PAD_ZEROS) return cnt;
    for (int p = -pad; p < 0; ++p)
        if (pad_index(p, n, mode) == i) out[cnt++] = p;
    for (int p = n; p < n + pad; ++p)
        if (pad_index(p, n, mode) == i) out[cnt++] = p;
    return cnt;
}

__global__ __launch_bounds__(256)
void patch_conv_bwd_input_kernel(ConvBwdArgs a) {
    const size_t total = (size_t)a.B * a.cin * a.H * a.W;
    const int kk = a.k * a.k;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = e % a.W; size_t r = e / a.W;
        const int y = r % a.H; r /= a.H;
        const int c = r % a.cin; const int b = r / a.cin;
        const int g = c / a.cin_g, cl = c - g * a.cin_g;
        int ys[8], xs[8];
        const int ny = pad_aliases(y, a.H, a.pad, a.pad_mode, ys);
        const int nx = pad_aliases(x, a.W, a.pad, a.pad_mode, xs);
        float acc = 0.0f;
        for (int iy = 0; iy < ny; ++iy)
            for (int ky = 0; ky < a.k; ++ky) {
                const int yo = ys[iy] - ky + a.pad;
                if (yo < 0 || yo >= a.H) continue;
                for (int ix = 0; ix < nx; ++ix)
                    for (int kx = 0; kx < a.k; ++kx) {
                        const int xo = xs[ix] - kx + a.pad;
                        if (xo < 0 || xo >= a.W) continue;
                        const int p = (b * a.fh + yo / a.ph) * a.fw + xo / a.pw;
                        const float* __restrict__ wp = a.bank + (size_t)p * a.ld + (size_t)cl * kk + ky * a.k + kx;
                        const float* __restrict__ dyp = a.dy + (((size_t)b * a.cout + g * a.cout_g) * a.H + yo) * a.W + xo;
                        for (int o = 0; o < a.cout_g; ++o)
                            acc = fmaf(wp[(size_t)(g * a.cout_g + o) * a.cin_g * kk], dyp[(size_t)o * a.H * a.W], acc);
                    }
            }
        a.dx[e] = acc;
    }
}

// One workgroup per (patch, block of output channels [o0, o0 + ob)); LDS: the block's dY tile [ob][ph*pw] and the padded
// X tile of the input channels those outputs read, [nx][(ph+2pad)*(pw+2pad)] (all of cin for groups == 1, whole groups
// otherwise: the host sizes ob so that the pair fits, so neither the channel count nor the tile size is limited by the
// 160 KiB of LDS any more); every thread owns bank rows of the block and reduces over the patch's pixels.
__global__ __launch_bounds__(256)
void patch_conv_bwd_weight_kernel(ConvBwdArgs a, int ob) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int patch = blockIdx.x;
    const int o0 = blockIdx.y * ob;
    const int on = min(ob, a.cout - o0);                       // output channels of this block
    const int j = patch % a.fw, i = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const int npix = a.ph * a.pw;
    const int HH = a.ph + 2 * a.pad, HW = a.pw + 2 * a.pad, tpos = HH * HW;
    const int c_lo = (o0 / a.cout_g) * a.cin_g;                // first input channel any output of the block reads
    const int c_hi = ((o0 + on - 1) / a.cout_g + 1) * a.cin_g;
    const int nx = c_hi - c_lo;
    float* dyl = lds;                          // [on][npix]
    float* xl = lds + (size_t)ob * npix;       // [nx][tpos]
    const int y0 = i * a.ph, x0 = j * a.pw;
    for (int e = threadIdx.x; e < on * npix; e += blockDim.x) {
        const int o = e / npix, pix = e - o * npix;
        const int u = pix / a.pw, v = pix - u * a.pw;
        dyl[e] = a.dy[(((size_t)b * a.cout + o0 + o) * a.H + y0 + u) * a.W + x0 + v];
    }
    for (int e = threadIdx.x; e < nx * tpos; e += blockDim.x) {
        const int c = e / tpos, pos = e - c * tpos;
        const int u = pos / HW, v = pos - u * HW;
        const int yy = pad_index(y0 + u - a.pad, a.H, a.pad_mode), xx = pad_index(x0 + v - a.pad, a.W, a.pad_mode);
        xl[e] = (yy >= 0 && xx >= 0) ? a.x[(((size_t)b * a.cin + c_lo + c) * a.H + yy) * a.W + xx] : 0.0f;
    }
    __syncthreads();
    const int kk = a.k * a.k;
    const int wrow = a.cin_g * kk;
    for (int idx = threadIdx.x; idx < on * wrow; idx += blockDim.x) {
        const int ol = idx / wrow; int r = idx - ol * wrow;
        const int kx = r % a.k; r /= a.k;
        const int ky = r % a.k; const int cl = r / a.k;
        const int o = o0 + ol;
        const int c = (o / a.cout_g) * a.cin_g + cl - c_lo;
        const float* dr = dyl + (size_t)ol * npix;
        const float* xr = xl + (size_t)c * tpos + ky * HW + kx;
        float acc = 0.0f;
        for (int u = 0; u < a.ph; ++u)
            for (int v = 0; v < a.pw; ++v) acc = fmaf(dr[u * a.pw + v], xr[u * HW + v], acc);
        a.dbank[(size_t)patch * a.ld + (size_t)o * wrow + (idx - ol * wrow)] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k = 1, groups = 1 on the f32 matrix cores (round 3): the two adjoints of a per-patch 1x1 convolution are per-patch GEMMs over the
// patch's pixels,   dW[o][c] = sum_px dY[o][px] X[c][px]   (K = pixels)   and   dX[c][px] = sum_o W[o][c] dY[o][px]   (N = pixels),
// and 7 of the 9 layers of a v1_0 decoder are of this kind.  v_mfma_f32_16x16x4_f32, operands straight from global memory: in NCHW a
// patch row is contiguous along the pixels, so a lane's four consecutive k (dW) are one 16-byte load -- MFMA j multiplies pixel set
// {16 s + 4 kgroup + j} of chunk s on both operands.  One workgroup per patch, the 4 waves split the pixel chunks (dW: partial tiles
// meet in LDS and are summed in wave order: deterministic) or the pixel tiles (dX).  The LDS-staged scalar kernels above stay for
// everything else (k > 1, groups, patches that are not a multiple of 16 pixels / 4 columns, very wide layers).
// ---------------------------------------------------------------------------------------------------------------------------------
using bw_f32x4 = __attribute__((ext_vector_type(4))) float;
// two adjacent elements as one aligned access (8 bytes fp32, 4 bytes bf16)
// Pair<T> (two adjacent elements, one aligned load / store): hs_common.h


// VEC: patches whose rows are whole 4-pixel groups on 16-byte boundaries and whose pixel count is a multiple of 16.  Otherwise (the
// (ph + 2) x (pw + 2) halo tiles that a train-mode v1_0 inverted residual feeds to its first 1x1 convolution: 18 x 18, 10 x 10) the
// lane's four pixels are four 4-byte loads with their own row / column, and pixels past the patch contribute zeros.
// T: storage type of the ACTIVATIONS and their gradients (float, or bf16_t: bf16 in memory, f32 products and sums, one rounding on
// store -- hs_common.h Store<T>); the bank and its gradient are fp32 either way (hs_patch_conv_train.hip's header).
// VEC = 2 (round 4): even patch and image widths -- the halo tiles -- read as aligned PAIRS (two 8-byte loads per lane, operand and chunk,
// instead of four 4-byte ones; a pair never straddles a patch row).
__device__ __forceinline__ float dwt_act(float z, int act) { return act == HS_ACT_RELU ? fmaxf(z, 0.f) : (act == HS_ACT_RELU6 ? fminf(fmaxf(z, 0.f), 6.f) : z); }
struct ConvBn {
    const float* __restrict__ partial;        // [cin][BN_CHUNKS][2], or null: mean / invstd are given (the weight gradient)
    const float* __restrict__ gamma; const float* __restrict__ beta;
    float* mean; float* invstd; float* running_mean; float* running_var; long long* counter;
    float eps, momentum, n;                   // n = elements per channel
    int act;
};
// BNL (round 5): the layer's input is act(BatchNorm(x)) with x RAW in memory -- normalised on use from the saved statistics (the weight
// gradient of patch_conv_bn_fwd_k1m_kernel's layer); a lane's NTI input channels are fixed, so their (scale, shift) sit in registers.
// four consecutive elements as one aligned load: 16 bytes of fp32, 8 bytes of bf16 (widened)
__device__ __forceinline__ bw_f32x4 quad_ld(const float* __restrict__ p) { return *reinterpret_cast<const bw_f32x4*>(p); }
__device__ __forceinline__ bw_f32x4 quad_ld(const bf16_t* __restrict__ p) {
    typedef unsigned q2u __attribute__((ext_vector_type(2)));
    const q2u v = *reinterpret_cast<const q2u*>(p);
    return bw_f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
}
template <int MT, int NTI, int VEC, typename T, bool BNL = false>
__global__ __launch_bounds__(256)
void patch_conv_bwd_weight_k1m_kernel(ConvBwdArgs a, ConvBn bn) {
    __shared__ __attribute__((aligned(16))) float red[4][MT * NTI][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int patch = blockIdx.x;
    const int pj = patch % a.fw, pi = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const size_t plane = (size_t)a.H * a.W;
    const T* __restrict__ dyp[MT];
    const T* __restrict__ xp[NTI];
    const size_t org = (size_t)(pi * a.ph) * a.W + pj * a.pw;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) dyp[mt] = (const T*)a.dy + ((size_t)b * a.cout + min(16 * mt + n, a.cout - 1)) * plane + org;
#pragma unroll
    for (int nt = 0; nt < NTI; ++nt) xp[nt] = (const T*)a.x + ((size_t)b * a.cin + min(16 * nt + n, a.cin - 1)) * plane + org;
    const int npix = a.ph * a.pw, nch = (npix + 15) >> 4;                // chunks of 16 pixels in patch-linear order
    float bng[NTI], bnb[NTI];
    if constexpr (BNL) {
#pragma unroll
        for (int nt = 0; nt < NTI; ++nt) {
            const int c = min(16 * nt + n, a.cin - 1);
            const float g = (bn.gamma ? bn.gamma[c] : 1.0f) * bn.invstd[c];
            bng[nt] = g; bnb[nt] = (bn.beta ? bn.beta[c] : 0.f) - bn.mean[c] * g;
        }
    }
    auto fetch = [&](int s, bw_f32x4 (&av)[MT], bw_f32x4 (&bv)[NTI]) {
        if constexpr (VEC == 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int l = 16 * s + 4 * kg + 2 * h, lc = min(l, npix - 2), u = div_by_inv(lc, a.inv_pw), v = lc - u * a.pw;   // npix even: a pair is live or not
                const size_t off = (size_t)u * a.W + v;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float p0, p1;
                    Pair<T>::ld(dyp[mt], off, p0, p1);                     // 8 bytes (fp32) / 4 bytes (bf16)
                    av[mt][2 * h] = p0; av[mt][2 * h + 1] = p1;             // (the pixels past the patch are masked in products())
                }
#pragma unroll
                for (int nt = 0; nt < NTI; ++nt) {
                    float p0, p1;
                    Pair<T>::ld(xp[nt], off, p0, p1);
                    bv[nt][2 * h] = p0; bv[nt][2 * h + 1] = p1;
                }
            }
        } else if constexpr (VEC == 1) {
            const int l = min(16 * s + 4 * kg, npix - 4), u = div_by_inv(l, a.inv_pw), v = l - u * a.pw;   // pw % 4 == 0: the lane's 4 pixels are one row segment (clamped: chunks past the end are fetched, never used)
            const size_t off = (size_t)u * a.W + v;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = quad_ld(dyp[mt] + off);        // 16 bytes (fp32) / 8 bytes (bf16: round 6 -- pairs before)
#pragma unroll
            for (int nt = 0; nt < NTI; ++nt) bv[nt] = quad_ld(xp[nt] + off);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int l = 16 * s + 4 * kg + j, lc = min(l, npix - 1), u = div_by_inv(lc, a.inv_pw), v = lc - u * a.pw;
                const size_t off = (size_t)u * a.W + v;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt][j] = Store<T>::ld(dyp[mt], off);     // clamped address; masked by a multiply in products()
#pragma unroll
                for (int nt = 0; nt < NTI; ++nt) bv[nt][j] = Store<T>::ld(xp[nt], off);
            }
        }
    };
    bw_f32x4 acc[MT][NTI];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTI; ++nt) acc[mt][nt] = bw_f32x4{0.f, 0.f, 0.f, 0.f};
    // (the mask of the last, partial chunk is applied HERE, on use: as a multiply inside fetch() it made the pipeline below wait for a
    //  chunk's loads as soon as they were issued)
    auto products = [&](int s, const bw_f32x4 (&av)[MT], const bw_f32x4 (&bv)[NTI]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float live = (VEC == 1 || 16 * s + 4 * kg + j < npix) ? 1.0f : 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float am = VEC == 1 ? av[mt][j] : av[mt][j] * live;
#pragma unroll
                for (int nt = 0; nt < NTI; ++nt) {
                    float xv = bv[nt][j];
                    if constexpr (BNL) xv = dwt_act(fmaf(xv, bng[nt], bnb[nt]), bn.act);      // (recomputed per output tile: MT <= 2 on this route)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(am, xv, acc[mt][nt], 0, 0, 0);
                }
            }
        }
    };
    // two operand buffers, every fetch of the steady path unconditional (a chunk past the end reads the patch's last pixels again and is
    // never multiplied): with `if (s < nch) fetch(...)` the compiler had folded the two halves back into one load -> wait -> 24 products
    // loop -- no request in flight under the products, one exposed round trip per chunk (k1m_pixel_stream below has the same story)
    bw_f32x4 a0[MT], b0[NTI], a1[MT], b1[NTI];
    int s = wave;
    fetch(s, a0, b0);
    fetch(s + 4, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    for (; s + 8 < nch; s += 8) {
        products(s, a0, b0);
        fetch(s + 8, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        products(s + 4, a1, b1);
        fetch(s + 12, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (s < nch) products(s, a0, b0);
    if (s + 4 < nch) products(s + 4, a1, b1);
    // ---- the four partial tiles meet in LDS; element (tile, lane, r) = D row 4 (lane / 16) + r, column lane % 16
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTI; ++nt)
            *reinterpret_cast<bw_f32x4*>(&red[wave][mt * NTI + nt][4 * lane]) = acc[mt][nt];
    __syncthreads();
    float* __restrict__ dst = a.dbank + (size_t)patch * a.ld;
    for (int e = tid; e < MT * NTI * 256; e += 256) {
        const int tile = e >> 8, w = e & 255, ln = w >> 2, r = w & 3;
        const int o = 16 * (tile / NTI) + 4 * (ln >> 4) + r, c = 16 * (tile % NTI) + (ln & 15);
        if (o < a.cout && c < a.cin)
            dst[(size_t)o * a.cin + c] = ((red[0][tile][w] + red[1][tile][w]) + red[2][tile][w]) + red[3][tile][w];
    }
}

// The pixel stream both of these kernels are: out[m][px] = epi(sum_k A[m][k] in[k][px]) over one patch, A (this patch's weights, MT x KQ
// fragments) in registers, the waves taking pixel tiles in turn with the next tile's operands in flight.  PX = 1: a lane owns pixel
// 16 t + n of tile t.  PX = 2 (round 4; even patch and image widths): a lane owns the ADJACENT pixels 32 t + 2 n and + 1 of a 32-pixel
// super-tile -- one 8-byte load per input row and one 8-byte store per output row instead of two 4-byte ones (16 lanes cover 128
// contiguous bytes instead of 64), the even and the odd pixels going through the matrix cores as two tiles.  Same products in the same
// order per pixel: bit-identical to PX = 1.  `aw` arrives as LOADED (clamped addresses); its reduction slots k >= kin are zeroed here, after
// the first two tiles' requests have left (the stream reads a repeated row there): masks on either operand ahead of those requests made the
// compiler wait for everything before the second tile's loads were issued -- one exposed round trip per workgroup.
struct K1mIdentity { __device__ __forceinline__ float operator()(float v, int, int) const { return v; } };
// PRE (round 5): applied to every loaded input value ON USE, pre(value, q, j) for reduction row k = 16 q + 4 kg + j -- a training-mode
// BatchNorm + activation normalised on load (patch_conv_bn_fwd_k1m_kernel); applied inside tile(), never in fetch() (see above).
template <int MT, int KQ, int PX, typename T, typename EPI, typename PRE = K1mIdentity>
__device__ __forceinline__ void k1m_pixel_stream(const ConvBwdArgs& a, float (&aw)[MT][KQ][4], const T* __restrict__ src, T* __restrict__ dst,
                                                 int kin, int mout, size_t plane, int n, int kg, int wave, EPI epi, PRE pre = PRE{}) {
    constexpr int TP = 16 * PX;                                         // pixels per (super-)tile
    const int npix = a.ph * a.pw, ntile = (npix + TP - 1) / TP;
    auto fetch = [&](int t, float (&bv)[PX][KQ][4], size_t& off) {
        const int l = min(TP * t + PX * n, npix - PX), u = div_by_inv(l, a.inv_pw), v = l - u * a.pw;     // past the patch: live pixels, not stored
        off = (size_t)u * a.W + v;
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * q + 4 * kg + j;
                const size_t at = (size_t)min(k, kin - 1) * plane + off;
                // rows past kin repeat the last row: the caller's A fragments are ZERO there (a select on the loaded value here made the
                // compiler wait for this tile's loads before it issued the next tile's: one exposed round trip per workgroup)
                if constexpr (PX == 2) Pair<T>::ld(src, at, bv[0][q][j], bv[1][q][j]);
                else bv[0][q][j] = Store<T>::ld(src, at);
            }
    };
    auto tile = [&](int t, const float (&bl)[PX][KQ][4], size_t off) {
        const bool live = TP * t + PX * n < npix;
        float bv[PX][KQ][4];                                            // (identity PRE: the copy folds away)
#pragma unroll
        for (int h = 0; h < PX; ++h)
#pragma unroll
            for (int q = 0; q < KQ; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[h][q][j] = pre(bl[h][q][j], q, j);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bw_f32x4 acc[PX];
#pragma unroll
            for (int h = 0; h < PX; ++h) {
                acc[h] = bw_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < KQ; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt][q][j], bv[h][q][j], acc[h], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * mt + 4 * kg + r;
                if (m < mout && live) {
                    if constexpr (PX == 2) Pair<T>::st(dst, (size_t)m * plane + off, epi(acc[0][r], mt, r), epi(acc[1][r], mt, r));
                    else Store<T>::st(dst, (size_t)m * plane + off, epi(acc[0][r], mt, r));
                }
            }
        }
    };
    // Software pipeline over two operand buffers.  Every fetch on the steady path is UNCONDITIONAL (a tile index past the end reads the
    // patch's last pixels again -- fetch clamps -- and is never stored): a fetch under `if (t < ntile)` is a control-flow join at which the
    // compiler's wait-count bookkeeping assumes the shorter path, i.e. waits for the newest loads instead of the oldest -- vmcnt(0) before
    // the first product with the second tile's requests in the queue (ISA of visit r5d's build).
    float b0[PX][KQ][4], b1[PX][KQ][4];
    size_t o0 = 0, o1 = 0;
    int t = wave;
    // @stamp 0
    fetch(t, b0, o0);
    fetch(t + 4, b1, o1);
    __builtin_amdgcn_sched_barrier(0);
    // @stamp 1
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) aw[mt][q][j] = 16 * q + 4 * kg + j < kin ? aw[mt][q][j] : 0.0f;
    // @stamp 2
    for (; t + 8 < ntile; t += 8) {                                     // tiles t and t + 4 exist, t + 8 does: both buffers are refilled
        tile(t, b0, o0);
        // @stamp 3 + (t >> 2 < 10 ? t >> 2 : 10)
        fetch(t + 8, b0, o0);
        __builtin_amdgcn_sched_barrier(0);
        tile(t + 4, b1, o1);
        fetch(t + 12, b1, o1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // @stamp 20
    if (t < ntile) tile(t, b0, o0);
    if (t + 4 < ntile) tile(t + 4, b1, o1);
    // @stamp 24
}

template <int CT, int KQ, int PX, typename T>
__global__ __launch_bounds__(256)
void patch_conv_bwd_input_k1m_kernel(ConvBwdArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int patch = blockIdx.x;
    const int pj = patch % a.fw, pi = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const size_t plane = (size_t)a.H * a.W;
    // A[i = input channel][k = output channel] = W[o][c]: this patch's bank row, o = 16 q + 4 kg + j, c = 16 ct + n (clamped: the
    // rows / columns beyond the layer multiply zeros of B or land in rows that are never stored)
    // The patch's weights go through LDS: ONE coalesced pass of the workgroup over the bank row, then every lane picks its fragment
    // elements with ds_read.  Picked straight from global memory -- 24 loads per lane, each instruction 64 scattered dwords, every wave of
    // the workgroup fetching the same fragments again -- they held the CU's address path for ~8 k cycles before the first pixel request
    // got through (tools/k1m_phase_times.py, visit r5e: 8.2 k of a workgroup's 25 k cycles at config 5's level 4, 4.8 k of 12.8 k at level 3).
    __shared__ float wl[16 * KQ * 16 * CT];
    const float* __restrict__ wp = a.bank + (size_t)patch * a.ld;
    for (int e = tid; e < a.cout * a.cin; e += 256) wl[e] = wp[e];
    __syncthreads();
    float aw[CT][KQ][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                aw[ct][q][j] = wl[min(16 * q + 4 * kg + j, a.cout - 1) * a.cin + min(16 * ct + n, a.cin - 1)];
    const size_t org = (size_t)(pi * a.ph) * a.W + pj * a.pw;
    const T* __restrict__ dyb = (const T*)a.dy + (size_t)b * a.cout * plane + org;
    T* __restrict__ dxb = (T*)a.dx + (size_t)b * a.cin * plane + org;
    k1m_pixel_stream<CT, KQ, PX, T>(a, aw, dyb, dxb, a.cout, a.cin, plane, n, kg, wave, [](float v, int, int) { return v; });
}

// The forward of the same layer on a PLAIN input tensor (what the autograd path calls: no fused stage-input prologue), same structure
// as the input-gradient kernel with the bank row read untransposed:  y[o][px] = sum_c W[o][c] x[c][px].  a.dy = x (cin channels),
// a.dx = y (cout channels); BatchNorm affine + activation optional.
// AFFINE: a BatchNorm affine + activation follows in the epilogue.  The training path passes none -- and with the choice left to run time
// (`scale ? scale[o] : 1`, `apply_act(v, act)`) the compiler had built twelve serialised load-and-wait blocks at the top of the kernel and
// ~10 branch blocks around every stored value: 4600 lines of ISA, the launch 2x what its bytes need (visit r5c, ISA read with
// tools/isa_phases.py's recipe).  The template removes both; with AFFINE the rows are loaded unconditionally and ReLU / ReLU6 are one clamp
// whose bounds depend on `act` (swish, which no reference decoder uses, keeps its branch).
template <int MT, int KQ, int PX, bool AFFINE, typename T>
__global__ __launch_bounds__(256)
void patch_conv_fwd_k1m_kernel(ConvBwdArgs a, const float* __restrict__ scale, const float* __restrict__ shift, int act) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int patch = blockIdx.x;
    const int pj = patch % a.fw, pi = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const size_t plane = (size_t)a.H * a.W;
    __shared__ float wl[16 * MT * 16 * KQ];                             // the patch's weights, staged by one coalesced pass (see the input-gradient kernel)
    const float* __restrict__ wp = a.bank + (size_t)patch * a.ld;
    for (int e = tid; e < a.cout * a.cin; e += 256) wl[e] = wp[e];
    __syncthreads();
    float aw[MT][KQ][4];                                                // A[i = output channel][k = input channel]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                aw[mt][q][j] = wl[min(16 * mt + n, a.cout - 1) * a.cin + min(16 * q + 4 * kg + j, a.cin - 1)];
    const size_t org = (size_t)(pi * a.ph) * a.W + pj * a.pw;
    const T* __restrict__ xb = (const T*)a.dy + (size_t)b * a.cin * plane + org;
    T* __restrict__ yb = (T*)a.dx + (size_t)b * a.cout * plane + org;
    if constexpr (AFFINE) {
        float sc[MT][4], sh[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = min(16 * mt + 4 * kg + r, a.cout - 1);
                sc[mt][r] = scale[o]; sh[mt][r] = shift[o];
            }
        const float hi = act == HS_ACT_RELU6 ? 6.0f : __builtin_inff();
        const bool plain = act == HS_ACT_NONE, swish = act == HS_ACT_SWISH;
        k1m_pixel_stream<MT, KQ, PX, T>(a, aw, xb, yb, a.cin, a.cout, plane, n, kg, wave, [&](float v, int mt, int r) {
            const float z = fmaf(v, sc[mt][r], sh[mt][r]);
            return plain ? z : (swish ? swishf(z) : fminf(fmaxf(z, 0.0f), hi));       // ReLU / ReLU6 as apply_act states them
        });
    } else {
        k1m_pixel_stream<MT, KQ, PX, T>(a, aw, xb, yb, a.cin, a.cout, plane, n, kg, wave, [](float v, int, int) { return v; });
    }
}

// Training-mode BatchNorm + activation of the INPUT normalised on load (round 5: BatchNorm2 + ReLU6 in front of the train-mode inverted
// residual's last 1 x 1 layer, hyperseg_v1_0.py:361-370, without the normalised copy of the hidden map): the forward above on the RAW
// input.  `bn` as in dw_tiles_fwd_kernel: with `partial` (hs_bn_train_stats_fwd's slice sums) the workgroup finalises every input
// channel's statistics while the patch's weights are staged -- thread c sums channel c's 32 slice pairs exactly as bn_apply_kernel does --
// and patch 0 of frame 0 stores mean / invstd and updates the running estimates; the lanes then keep (scale, shift) of THEIR reduction
// rows in registers: 3 vector operations per loaded value against the MT matrix instructions it feeds.
template <int KQ, typename T>
__device__ __forceinline__ void conv_bn_rows(const ConvBn& n, const T* __restrict__ x, size_t plane, int cin, bool writer, float* gb /* LDS [2][16 KQ] */) {
    const int c = threadIdx.x;
    if (c < 16 * KQ) {
        float g = 0.f, bb = 0.f;
        if (c < cin) {
            float mean, invstd;
            if (n.partial) {
                const float shift = Store<T>::ld(x, (size_t)c * plane);           // the statistics' shift: the channel's first element (frame 0)
                float s = 0.f, q = 0.f;
                for (int i = 0; i < BN_CHUNKS; ++i) { s += n.partial[((size_t)c * BN_CHUNKS + i) * 2]; q += n.partial[((size_t)c * BN_CHUNKS + i) * 2 + 1]; }
                const float md = s / n.n, var = fmaxf(q / n.n - md * md, 0.f);
                mean = md + shift; invstd = rsqrtf(var + n.eps);
                if (writer) {
                    n.mean[c] = mean; n.invstd[c] = invstd;
                    if (n.running_mean) {
                        n.running_mean[c] = (1.f - n.momentum) * n.running_mean[c] + n.momentum * mean;
                        n.running_var[c] = (1.f - n.momentum) * n.running_var[c] + n.momentum * (n.n > 1.f ? var * n.n / (n.n - 1.f) : var);
                    }
                    if (n.counter && c == 0) *n.counter += 1;
                }
            } else {
                mean = n.mean[c]; invstd = n.invstd[c];
            }
            g = n.gamma ? n.gamma[c] * invstd : invstd;
            bb = (n.beta ? n.beta[c] : 0.f) - mean * g;
        }
        gb[c] = g; gb[16 * KQ + c] = bb;
    }
}

template <int MT, int KQ, int PX, typename T>
__global__ __launch_bounds__(256)
void patch_conv_bn_fwd_k1m_kernel(ConvBwdArgs a, ConvBn bn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int patch = blockIdx.x;
    const int pj = patch % a.fw, pi = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const size_t plane = (size_t)a.H * a.W;
    __shared__ float wl[16 * MT * 16 * KQ];
    __shared__ float gb[2 * 16 * KQ];
    const float* __restrict__ wp = a.bank + (size_t)patch * a.ld;
    for (int e = tid; e < a.cout * a.cin; e += 256) wl[e] = wp[e];
    conv_bn_rows<KQ, T>(bn, (const T*)a.dy, plane, a.cin, patch == 0, gb);
    __syncthreads();
    float aw[MT][KQ][4], sc[KQ][4], sh[KQ][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                aw[mt][q][j] = wl[min(16 * mt + n, a.cout - 1) * a.cin + min(16 * q + 4 * kg + j, a.cin - 1)];
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[q][j] = gb[16 * q + 4 * kg + j]; sh[q][j] = gb[16 * KQ + 16 * q + 4 * kg + j]; }
    const size_t org = (size_t)(pi * a.ph) * a.W + pj * a.pw;
    const T* __restrict__ xb = (const T*)a.dy + (size_t)b * a.cin * plane + org;
    T* __restrict__ yb = (T*)a.dx + (size_t)b * a.cout * plane + org;
    const int act = bn.act;
    k1m_pixel_stream<MT, KQ, PX, T>(a, aw, xb, yb, a.cin, a.cout, plane, n, kg, wave, [](float v, int, int) { return v; },
                                    [&](float v, int q, int j) { return dwt_act(fmaf(v, sc[q][j], sh[q][j]), act); });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Depthwise 3 x 3 with per-patch taps and ZERO padding on a plain tensor, forward and both adjoints (round 3): the middle layer of a
// train-mode v1_0 inverted residual, which runs on the image of halo tiles laid side by side (hyperseg_v1_0.py _run_train: patches of
// (ph + 2) x (pw + 2) = 18 x 18 / 10 x 10 pixels).  Vector code with the image-level structure the operator has: one thread per output
// element, its nine neighbours from L1 / L2 and its patch's nine taps as uniform-ish loads -- the generic kernels (LDS-staged per patch,
// any k / groups / padding) took 102 / 79 / ~60 us per launch at config 5 for 83 M multiply-adds.
// ---------------------------------------------------------------------------------------------------------------------------------
// MODE 0: y = conv(x, K).  MODE 1: dx = adjoint wrt x (the taps of the patch that owns the OUTPUT pixel, mirrored).
template <int MODE, typename T>
__global__ __launch_bounds__(256)
void patch_dw3_kernel(ConvBwdArgs a, const T* __restrict__ src, T* __restrict__ dst) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int plane_id = blockIdx.z, c = plane_id % a.cin, b = plane_id / a.cin;
    if (x >= a.W || y >= a.H) return;
    const T* __restrict__ sp = src + (size_t)plane_id * a.H * a.W;
    const float* __restrict__ bank = a.bank;
    float acc = 0.0f;
    if constexpr (MODE == 0) {
        const float* __restrict__ kp = bank + (size_t)((b * a.fh + y / a.ph) * a.fw + x / a.pw) * a.ld + c * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                const bool in = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                const float v = Store<T>::ld(sp, (size_t)min(max(yy, 0), a.H - 1) * a.W + min(max(xx, 0), a.W - 1));
                acc = fmaf(kp[ky * 3 + kx], in ? v : 0.0f, acc);
            }
    } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yo = y - ky + 1, xo = x - kx + 1;              // the output pixel this input feeds through tap (ky, kx)
                const bool in = yo >= 0 && yo < a.H && xo >= 0 && xo < a.W;
                const int yc = min(max(yo, 0), a.H - 1), xc = min(max(xo, 0), a.W - 1);
                const float g = Store<T>::ld(sp, (size_t)yc * a.W + xc);
                const float w = bank[(size_t)((b * a.fh + yc / a.ph) * a.fw + xc / a.pw) * a.ld + c * 9 + ky * 3 + kx];
                acc = fmaf(w, in ? g : 0.0f, acc);
            }
    }
    Store<T>::st(dst, (size_t)plane_id * a.H * a.W + (size_t)y * a.W + x, acc);
}

// The same two operators with TWO horizontally adjacent elements per thread (even patch and image widths: every use in the decoders) and no
// integer division (round 4; the one-element form above stays for odd widths).  A thread's 3 x 4 neighbourhood is 9 loads (the middle pair
// of a row is one 8-byte -- bf16: 4-byte -- load) instead of 18, its patch row / column come from div_by_inv, and a neighbour's patch is the
// own one +- 1 decided from the position inside the tile: the one-element adjoint did 18 divisions by run-time values per thread (~450 of its
// ~520 vector instructions; 52 us per launch at config 5 against 30 for the forward, which does two).  Same fma chain per element (tap-major,
// ky then kx): bit-identical results.
template <int MODE, typename T>
__global__ __launch_bounds__(256)
void patch_dw3_pair_kernel(ConvBwdArgs a, const T* __restrict__ src, T* __restrict__ dst) {
    const int x0 = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int plane_id = blockIdx.z;
    if (x0 >= a.W || y >= a.H) return;
    const int b = div_by_inv(plane_id, 1.0f / (float)a.cin), c = plane_id - b * a.cin;
    const T* __restrict__ sp = src + (size_t)plane_id * a.H * a.W;
    const int ty = div_by_inv(y, a.inv_ph), u = y - ty * a.ph, tx = div_by_inv(x0, a.inv_pw), v = x0 - tx * a.pw;   // v even, v + 1 < pw
    // ---- the 3 x 4 neighbourhood: rows y - 1 .. y + 1, columns x0 - 1 .. x0 + 2, zero outside the image (clamped loads, masked by multiplies)
    float s[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        const bool row_in = yy >= 0 && yy < a.H;
        const size_t rb = (size_t)min(max(yy, 0), a.H - 1) * a.W;
        float m0, m1;
        Pair<T>::ld(sp, rb + x0, m0, m1);
        const float l = Store<T>::ld(sp, rb + max(x0 - 1, 0)), rr = Store<T>::ld(sp, rb + min(x0 + 2, a.W - 1));
        s[r][0] = (row_in && x0 > 0) ? l : 0.0f;                           // (loads unconditional, selects afterwards)
        s[r][1] = row_in ? m0 : 0.0f; s[r][2] = row_in ? m1 : 0.0f;
        s[r][3] = (row_in && x0 + 2 < a.W) ? rr : 0.0f;
    }
    const float* __restrict__ bank = a.bank;
    float acc0 = 0.0f, acc1 = 0.0f;
    if constexpr (MODE == 0) {
        const float* __restrict__ kp = bank + (size_t)((b * a.fh + ty) * a.fw + tx) * a.ld + c * 9;
        float kv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) kv[t] = kp[t];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                acc0 = fmaf(kv[ky * 3 + kx], s[ky][kx], acc0);
                acc1 = fmaf(kv[ky * 3 + kx], s[ky][kx + 1], acc1);
            }
    } else {
        // input pixel (y, x0 + j) feeds output (y - ky + 1, x0 + j - kx + 1) through tap (ky, kx) of the patch that owns THAT output:
        // row of s = 2 - ky, column = j - kx + 2; the owner's tile row / column = own +- 1 at the tile's border
        const int pr[3] = {min(ty + (u == a.ph - 1 ? 1 : 0), a.fh - 1), ty, max(ty - (u == 0 ? 1 : 0), 0)};            // ky = 0, 1, 2
        // columns of s 0 .. 3 = image columns x0 - 1 .. x0 + 2
        const int pc[4] = {max(tx - (v == 0 ? 1 : 0), 0), tx, tx, min(tx + (v + 2 == a.pw ? 1 : 0), a.fw - 1)};
        float w0[9], w1[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const size_t row = (size_t)(b * a.fh + pr[ky]) * a.fw;
                w0[ky * 3 + kx] = bank[(row + pc[2 - kx]) * a.ld + c * 9 + ky * 3 + kx];
                w1[ky * 3 + kx] = bank[(row + pc[3 - kx]) * a.ld + c * 9 + ky * 3 + kx];
            }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                acc0 = fmaf(w0[ky * 3 + kx], s[2 - ky][2 - kx], acc0);
                acc1 = fmaf(w1[ky * 3 + kx], s[2 - ky][3 - kx], acc1);
            }
    }
    Pair<T>::st(dst, (size_t)plane_id * a.H * a.W + (size_t)y * a.W + x0, acc0, acc1);
}

// dK[patch][c][ky][kx] = sum over the patch's pixels of dY[c][y][x] X[c][y + ky - 1][x + kx - 1] (zero outside the image): one wave per
// (patch, channel), lanes over the pixels, nine accumulators, DPP wave sums.
template <typename T>
__global__ __launch_bounds__(256)
void patch_dw3_bwd_weight_kernel(ConvBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int patch = blockIdx.x, c = blockIdx.y * 4 + wave;
    if (c >= a.cin) return;
    const int pj = patch % a.fw, pi = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const size_t plane = (size_t)a.H * a.W;
    const T* __restrict__ xp = (const T*)a.x + ((size_t)b * a.cin + c) * plane;
    const T* __restrict__ gp = (const T*)a.dy + ((size_t)b * a.cin + c) * plane;
    const int y0 = pi * a.ph, x0 = pj * a.pw, npix = a.ph * a.pw;
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = lane; l < npix; l += 64) {
        const int u = l / a.pw, v = l - u * a.pw, y = y0 + u, x = x0 + v;
        const float g = Store<T>::ld(gp, (size_t)y * a.W + x);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                const bool in = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                const float val = Store<T>::ld(xp, (size_t)min(max(yy, 0), a.H - 1) * a.W + min(max(xx, 0), a.W - 1));
                acc[ky * 3 + kx] = fmaf(g, in ? val : 0.0f, acc[ky * 3 + kx]);
            }
    }
    float* __restrict__ dst = a.dbank + (size_t)patch * a.ld + c * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float s = wave_sum64(acc[t]);
        if (lane == 0) dst[t] = s;
    }
}

// dX of a k = 1, groups = 1 layer on patches of fewer than 16 pixels (levels 0 and 1 of a v1_0 decoder: 1 and 4 pixels): one workgroup per
// patch, a lane owns an input channel c and reads W[o][c] for o = 0 .. cout - 1 -- consecutive lanes, consecutive addresses -- against the
// patch's dY[o][px] (the same address in every lane: one broadcast load).  The element-wise kernel at the top of this file walked the bank
// with a stride of cin per thread and divided twice per element: 24.7 us for the 7.8 MB bank of config 5's level 1.
template <typename T>
__global__ __launch_bounds__(128)
void patch_conv_bwd_input_tiny_kernel(ConvBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tiny_dy[];     // [cout][16]: the patch's dY, zero past its pixels
    const int patch = blockIdx.x;
    const int pj = patch % a.fw, pi = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const size_t plane = (size_t)a.H * a.W;
    const size_t org = (size_t)(pi * a.ph) * a.W + pj * a.pw;
    const int npix = a.ph * a.pw;                                        // <= 15
    const float* __restrict__ wp = a.bank + (size_t)patch * a.ld;
    const T* __restrict__ dyb = (const T*)a.dy + (size_t)b * a.cout * plane + org;
    T* __restrict__ dxb = (T*)a.dx + (size_t)b * a.cin * plane + org;
    for (int e = threadIdx.x; e < a.cout * 16; e += 128) {
        const int o = e >> 4, q = e & 15, l = min(q, npix - 1), u = div_by_inv(l, a.inv_pw), v = l - u * a.pw;
        const float val = Store<T>::ld(dyb, (size_t)o * plane + (size_t)u * a.W + v);
        tiny_dy[e] = q < npix ? val : 0.0f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.cin; c += 128) {
        float acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
        for (int o0 = 0; o0 < a.cout; o0 += 8) {                         // eight weight loads in flight, dY broadcast from LDS
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = wp[(size_t)min(o0 + j, a.cout - 1) * a.cin + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (o0 + j < a.cout) {                                   // (uniform)
                    const bw_f32x4* d4 = reinterpret_cast<const bw_f32x4*>(tiny_dy + 16 * (o0 + j));
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bw_f32x4 d = d4[g];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[4 * g + r] = fmaf(w[j], d[r], acc[4 * g + r]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 15; ++q)
            if (q < npix) {
                const int u = div_by_inv(q, a.inv_pw), v = q - u * a.pw;
                Store<T>::st(dxb, (size_t)c * plane + (size_t)u * a.W + v, acc[q]);
            }
    }
}

// dW of the same tiny-patch layers: dW[o][c] = sum over the patch's <= 15 pixels of dY[o][px] X[c][px], one workgroup per patch, both operands
// staged in LDS (row stride 17: consecutive c fall on consecutive banks), every thread a run of consecutive bank columns -- coalesced
// stores, one reciprocal multiplication per output (the general kernel below divides three times per output and stages a padded tile).
template <typename T>
__global__ __launch_bounds__(256)
void patch_conv_bwd_weight_tiny_kernel(ConvBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tiny_w[];      // [cout][17] dY | [cin][17] X
    const int patch = blockIdx.x;
    const int pj = patch % a.fw, pi = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const size_t plane = (size_t)a.H * a.W;
    const size_t org = (size_t)(pi * a.ph) * a.W + pj * a.pw;
    const int npix = a.ph * a.pw;
    float* __restrict__ dyl = tiny_w;
    float* __restrict__ xl = tiny_w + (size_t)a.cout * 17;
    const T* __restrict__ dyb = (const T*)a.dy + (size_t)b * a.cout * plane + org;
    const T* __restrict__ xb = (const T*)a.x + (size_t)b * a.cin * plane + org;
    for (int e = threadIdx.x; e < (a.cout + a.cin) * 16; e += 256) {
        const int row = e >> 4, q = e & 15, l = min(q, npix - 1), u = div_by_inv(l, a.inv_pw), v = l - u * a.pw;
        const size_t off = (size_t)u * a.W + v;
        const bool is_dy = row < a.cout;                                 // (both loads unconditional from clamped rows; one is kept)
        const float gd = Store<T>::ld(dyb, (size_t)min(row, a.cout - 1) * plane + off);
        const float gx = Store<T>::ld(xb, (size_t)min(max(row - a.cout, 0), a.cin - 1) * plane + off);
        if (q < 15) tiny_w[(size_t)row * 17 + q] = q < npix ? (is_dy ? gd : gx) : 0.0f;
    }
    __syncthreads();
    const float inv_cin = 1.0f / (float)a.cin;
    float* __restrict__ dst = a.dbank + (size_t)patch * a.ld;
    for (int idx = threadIdx.x; idx < a.cout * a.cin; idx += 256) {
        const int o = div_by_inv(idx, inv_cin), c = idx - o * a.cin;
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 15; ++q)
            if (q < npix) acc = fmaf(dyl[o * 17 + q], xl[c * 17 + q], acc);      // (uniform bound; pixel order as the general kernel's)
        dst[idx] = acc;
    }
}

// ... and the weight gradient with two adjacent pixels per lane (even patch width): 10 loads per pair instead of 20, rows from div_by_inv.
template <typename T>
__global__ __launch_bounds__(256)
void patch_dw3_bwd_weight_pair_kernel(ConvBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int patch = blockIdx.x, c = blockIdx.y * 4 + wave;
    if (c >= a.cin) return;
    const int pj = patch % a.fw, pi = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const size_t plane = (size_t)a.H * a.W;
    const T* __restrict__ xp = (const T*)a.x + ((size_t)b * a.cin + c) * plane;
    const T* __restrict__ gp = (const T*)a.dy + ((size_t)b * a.cin + c) * plane;
    const int y0 = pi * a.ph, x0 = pj * a.pw, hw = a.pw >> 1, npair = a.ph * hw;
    const float inv_hw = 2.0f * a.inv_pw;
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = lane; l < npair; l += 64) {
        const int u = div_by_inv(l, inv_hw), v = 2 * (l - u * hw), y = y0 + u, x = x0 + v;
        float g0, g1;
        Pair<T>::ld(gp, (size_t)y * a.W + x, g0, g1);
        float s[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            const bool row_in = yy >= 0 && yy < a.H;
            const size_t rb = (size_t)min(max(yy, 0), a.H - 1) * a.W;
            float m0, m1;
            Pair<T>::ld(xp, rb + x, m0, m1);
            const float lf = Store<T>::ld(xp, rb + max(x - 1, 0)), rr = Store<T>::ld(xp, rb + min(x + 2, a.W - 1));
            s[r][0] = (row_in && x > 0) ? lf : 0.0f;
            s[r][1] = row_in ? m0 : 0.0f; s[r][2] = row_in ? m1 : 0.0f;
            s[r][3] = (row_in && x + 2 < a.W) ? rr : 0.0f;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                acc[ky * 3 + kx] = fmaf(g0, s[ky][kx], acc[ky * 3 + kx]);
                acc[ky * 3 + kx] = fmaf(g1, s[ky][kx + 1], acc[ky * 3 + kx]);
            }
    }
    float* __restrict__ dst = a.dbank + (size_t)patch * a.ld + c * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float sum = wave_sum64(acc[t]);
        if (lane == 0) dst[t] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The middle layer of a train-mode v1_0 inverted residual AS THE REFERENCE STATES IT (round 4): a VALID depthwise 3 x 3 on every halo tile,
//     y[b][c][i ph + u][j pw + v] = sum_{ky,kx} K[patch (b,i,j)][c][ky][kx] t[b][c][i (ph+2) + u + ky][j (pw+2) + v + kx]
// from the image of halo tiles (B, C, fh (ph+2), fw (pw+2)) straight to the block's (B, C, H, W) map.  Rounds 3-4 ran it as a zero-padded
// convolution over the whole tile image followed by a gather that drops every tile's ring (hs_tile_interior_*): 27 % more outputs than are
// kept, a 37 MB intermediate written and read again at config 5, an adjoint that had to look up the NEIGHBOURING patches' taps for ring
// positions whose gradient is zero by construction, and two extra launches per direction.  Here everything is tile-local: two adjacent
// elements per thread (even pw), aligned 8-byte loads, no bounds tests in the forward, the patch's own nine taps in the adjoint.  The fma
// chains run tap-major (ky, then kx) like patch_dw3_*: the kept values are bit-identical to the two-launch route.
// ---------------------------------------------------------------------------------------------------------------------------------
struct DwtArgs {
    const float* __restrict__ bank;    // (P, ld): taps of patch p, channel c at [p ld + 9 c + 3 ky + kx]
    float* __restrict__ dbank;
    long ld;
    int B, C, H, W, fh, fw, ph, pw;
    float inv_ph, inv_pw, inv_ph2, inv_pw2, inv_c;
    int pm;                            // layout of the tiles: 0 = image of tiles (B, C, fh (ph+2), fw (pw+2)), 1 = patch-major (B fh fw, C, ph+2, pw+2)
};
// BatchNorm (training mode) + activation applied to the TILES ON LOAD (round 5): the block's first 1x1 layer writes its raw output, a
// statistics pass leaves {sum, sum of squares} per channel and slice, and this layer normalises what it reads -- the normalised copy
// (37 MB at config 5's level 4: bn_apply's read + write were the two largest passes of the step) is never written.  Same arithmetic
// as hs_train_aux.hip's bn_apply_kernel: y = act(fma(x, gamma invstd, beta - mean gamma invstd)).
struct DwtBn {
    const float* __restrict__ partial;        // forward: [C][BN_CHUNKS][2] from hs_bn_train_stats_fwd; null: mean / invstd are given
    const float* __restrict__ gamma; const float* __restrict__ beta;
    float* mean; float* invstd;               // forward: written (channel c by the workgroup of plane (0, c), block (0, 0)); backward: read
    float* running_mean; float* running_var; long long* counter;
    float eps, momentum, n;                   // n = elements per channel of the tile tensor
    long shift_stride;                        // elements between the first elements of consecutive channels (the statistics' shift)
    int act;
};

// origin of tile (i, j), channel c of frame b, and the distance between its rows
__device__ __forceinline__ size_t dwt_tile(const DwtArgs& a, int b, int c, int i, int j, int& row_stride) {
    if (a.pm) {
        row_stride = a.pw + 2;
        return ((((size_t)b * a.fh + i) * a.fw + j) * a.C + c) * (size_t)((a.ph + 2) * (a.pw + 2));
    }
    row_stride = a.fw * (a.pw + 2);
    return (((size_t)b * a.C + c) * (size_t)(a.fh * (a.ph + 2)) + (size_t)i * (a.ph + 2)) * row_stride + (size_t)j * (a.pw + 2);
}

// RPT: output rows per thread (1, or 2 where the patch height is even -- the two rows then share one patch and three of the four tile
// rows they read: 8 pair loads and 16 normalisations for 4 outputs instead of 12 and 24, half as many workgroups to finalise the
// statistics in; round 5: 20.8 us with one row at config 5's level 4 against 13.8 without the normalisation).
template <typename T, bool BN, int RPT = 1>
__global__ __launch_bounds__(256)
void dw_tiles_fwd_kernel(DwtArgs a, DwtBn n, const T* __restrict__ t, T* __restrict__ y) {
    const int x0 = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)), yy = RPT * (blockIdx.y * 4 + (threadIdx.x >> 6));
    const int plane_id = blockIdx.z;
    const int b = div_by_inv(plane_id, a.inv_c), c = plane_id - b * a.C;
    if (x0 >= a.W || yy >= a.H) return;
    const int i = div_by_inv(yy, a.inv_ph), u = yy - i * a.ph, j = div_by_inv(x0, a.inv_pw), v = x0 - j * a.pw;
    int TW;
    const T* __restrict__ tp = t + dwt_tile(a, b, c, i, j, TW) + (size_t)u * TW + v;
    const float* __restrict__ kp = a.bank + (size_t)((b * a.fh + i) * a.fw + j) * a.ld + c * 9;
    float s[2 + RPT][4], kv[9];
#pragma unroll
    for (int r = 0; r < 2 + RPT; ++r) {
        Pair<T>::ld(tp, (size_t)r * TW, s[r][0], s[r][1]);
        Pair<T>::ld(tp, (size_t)r * TW + 2, s[r][2], s[r][3]);
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) kv[q] = kp[q];
    // (the statistics are finalised HERE, with the tile and tap requests already out: at the top of the kernel the 64 scalar loads and
    //  their serial sums sat in front of every workgroup's first vector load -- 23.8 -> 20.8 us for this launch, visits r5v19 / r5v20)
    float g = 1.0f, bb = 0.0f;
    if constexpr (BN) {
        // mean / invstd of channel c from the 32 slice sums (uniform over the workgroup: scalar loads), exactly as bn_apply_kernel forms them
        const float shift = Store<T>::ld(t, (size_t)c * n.shift_stride);
        float ps = 0.f, pq = 0.f;
        for (int ci = 0; ci < BN_CHUNKS; ++ci) { ps += n.partial[((size_t)c * BN_CHUNKS + ci) * 2]; pq += n.partial[((size_t)c * BN_CHUNKS + ci) * 2 + 1]; }
        const float md = ps / n.n, var = fmaxf(pq / n.n - md * md, 0.f), mean = md + shift, invstd = rsqrtf(var + n.eps);
        g = n.gamma ? n.gamma[c] * invstd : invstd;
        bb = (n.beta ? n.beta[c] : 0.f) - mean * g;
        if (b == 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {          // one writer per channel (never out of range: x0 = yy = 0)
            n.mean[c] = mean; n.invstd[c] = invstd;
            if (n.running_mean) {
                n.running_mean[c] = (1.f - n.momentum) * n.running_mean[c] + n.momentum * mean;
                n.running_var[c] = (1.f - n.momentum) * n.running_var[c] + n.momentum * (n.n > 1.f ? var * n.n / (n.n - 1.f) : var);
            }
            if (n.counter && c == 0) *n.counter += 1;
        }
#pragma unroll
        for (int r = 0; r < 2 + RPT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) s[r][q] = dwt_act(fmaf(s[r][q], g, bb), n.act);
    }
#pragma unroll
    for (int ro = 0; ro < RPT; ++ro) {
        float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                acc0 = fmaf(kv[ky * 3 + kx], s[ro + ky][kx], acc0);
                acc1 = fmaf(kv[ky * 3 + kx], s[ro + ky][kx + 1], acc1);
            }
        Pair<T>::st(y, ((size_t)plane_id * a.H + yy + ro) * a.W + x0, acc0, acc1);
    }
}

// dt[tile position (U, V)] = sum_{ky,kx} K[ky][kx] dy[U - ky][V - kx] over the patch's own outputs (0 <= U - ky < ph, 0 <= V - kx < pw)
// RPT: tile rows per thread (2 where the patch height is even: rows U, U + 1 of one tile share three of the four output rows they read and
// the nine taps -- 8 pair loads + 9 tap loads for 4 values instead of 12 + 18, half the workgroups; round 6: the one-row form ran at
// 2.1 TB/s at config 5's level 4, 1.2 TB/s at level 3).  Per value the same fma chain (ky, then kx) as before: bit-identical.
// STATS (round 6): the launch also leaves the two sums BatchNorm1's adjoint needs -- per channel, sum of d and of d x_hat over the tile tensor,
// d = dt act'(z), from the RAW tiles `xt` and the saved statistics (n: gamma, beta, mean, invstd, act) -- as one {s, q} pair per workgroup in
// `partial` [c][workgroup of the channel][2]; bn_bwd_apply_np_kernel combines a channel's pairs in workgroup order.  The statistics pass of
// the two-launch adjoint (a read of both tile tensors: 15.6 us at config 5's level 4) is gone; d is formed from the value as STORED (bf16: after
// its rounding), as the separate pass read it.
__device__ __forceinline__ float dwt_stored(float v, float*) { return v; }
__device__ __forceinline__ float dwt_stored(float v, bf16_t*) { bf16_t r; Store<bf16_t>::st(&r, 0, v); return Store<bf16_t>::ld(&r, 0); }
template <typename T, int RPT = 1, bool STATS = false>
__global__ __launch_bounds__(256)
void dw_tiles_bwd_in_kernel(DwtArgs a, const T* __restrict__ dy, T* __restrict__ dt, DwtBn n, const T* __restrict__ xt, float* __restrict__ partial) {
    __shared__ float red[4][2];
    const int TW = a.fw * (a.pw + 2), TH = a.fh * (a.ph + 2);
    const int X0 = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)), Y = RPT * (blockIdx.y * 4 + (threadIdx.x >> 6));
    const int plane_id = blockIdx.z;
    const bool live = X0 < TW && Y < TH;
    if (!STATS && !live) return;
    const int b = div_by_inv(plane_id, a.inv_c), c = plane_id - b * a.C;
    float s_d = 0.0f, s_q = 0.0f;
    if (live) {
        const int i = div_by_inv(Y, a.inv_ph2), U = Y - i * (a.ph + 2), j = div_by_inv(X0, a.inv_pw2), V0 = X0 - j * (a.pw + 2);
        const T* __restrict__ gp = dy + ((size_t)plane_id * a.H + (size_t)i * a.ph) * a.W + (size_t)j * a.pw;       // the patch's (0, 0) output
        const float* __restrict__ kp = a.bank + (size_t)((b * a.fh + i) * a.fw + j) * a.ld + c * 9;
        // rows U - 2 .. U + RPT - 1, columns V0 - 2 .. V0 + 1 of the patch's outputs: two aligned pairs per row, each wholly inside or wholly outside
        const bool left = V0 >= 2, right = V0 <= a.pw - 2;
        const int cl = left ? V0 - 2 : 0, cr = right ? V0 : 0;
        float g[2 + RPT][4], kv[9], xv[RPT][2];
#pragma unroll
        for (int r = 0; r < 2 + RPT; ++r) {                           // r <-> output row U - 2 + r
            const int ur = U - 2 + r;
            const bool row_in = ur >= 0 && ur < a.ph;
            const size_t rb = (size_t)min(max(ur, 0), a.ph - 1) * a.W;
            float l0, l1, r0, r1;
            Pair<T>::ld(gp, rb + cl, l0, l1);
            Pair<T>::ld(gp, rb + cr, r0, r1);
            g[r][0] = (row_in && left) ? l0 : 0.0f; g[r][1] = (row_in && left) ? l1 : 0.0f;
            g[r][2] = (row_in && right) ? r0 : 0.0f; g[r][3] = (row_in && right) ? r1 : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) kv[q] = kp[q];
        int RS;
        const size_t torg = dwt_tile(a, b, c, i, j, RS);
        if constexpr (STATS) {
#pragma unroll
            for (int ro = 0; ro < RPT; ++ro) Pair<T>::ld(xt, torg + (size_t)(U + ro) * RS + V0, xv[ro][0], xv[ro][1]);
        }
#pragma unroll
        for (int ro = 0; ro < RPT; ++ro) {
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {                      // output (U + ro - ky, V - kx): row index 2 + ro - ky, column index V - kx - (V0 - 2)
                    acc0 = fmaf(kv[ky * 3 + kx], g[2 + ro - ky][2 - kx], acc0);
                    acc1 = fmaf(kv[ky * 3 + kx], g[2 + ro - ky][3 - kx], acc1);
                }
            Pair<T>::st(dt, torg + (size_t)(U + ro) * RS + V0, acc0, acc1);
            if constexpr (STATS) {
                const float mean = n.mean[c], invstd = n.invstd[c], gm = n.gamma ? n.gamma[c] : 1.f, bt = n.beta ? n.beta[c] : 0.f;
                const float accs[2] = {dwt_stored(acc0, (T*)nullptr), dwt_stored(acc1, (T*)nullptr)};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float xh = (xv[ro][e] - mean) * invstd, z = fmaf(xh, gm, bt);
                    const float gr = n.act == HS_ACT_RELU ? (z > 0.f ? 1.f : 0.f) : (n.act == HS_ACT_RELU6 ? ((z > 0.f && z < 6.f) ? 1.f : 0.f) : 1.f);
                    const float d = accs[e] * gr;
                    s_d += d; s_q = fmaf(d, xh, s_q);
                }
            }
        }
    }
    if constexpr (STATS) {
        s_d = wave_sum64(s_d); s_q = wave_sum64(s_q);
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[wave][0] = s_d; red[wave][1] = s_q; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const size_t np = (size_t)a.B * gridDim.y * gridDim.x, widx = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            partial[((size_t)c * np + widx) * 2] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
            partial[((size_t)c * np + widx) * 2 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        }
    }
}

// dK[patch][c][ky][kx] = sum over the patch's outputs (u, v) of dy[u][v] t[u + ky][v + kx]: one wave per (patch, channel), a lane owns output pairs
// RPT: output rows per lane and iteration (2 where the patch height is even: 16 tile values normalised for 4 output gradients instead of
// 24; both the plain and the BN form take the same RPT for a shape, so they stay bit-equal to each other)
// ROWS (round 6): a patch of <= 16 element pairs per thread pass (8 x 8 patches at RPT = 2: config 5's level 3) left 48 of a wave's 64 lanes
// idle and paid nine 64-lane reductions per (patch, channel); with ROWS each row of 16 lanes takes a channel of its own -- four channels per
// wave, a 16-lane reduction (rowsum16: the very sums wave_sum64 forms from a wave whose other rows hold zeros, so dbank is bit-identical).
// CPW (round 6): channels per wave in the 64-lane form -- a wave requests one channel's ten pairs, reduces and retires (28 k waves of one memory
// round trip each at config 5's level 4); two channels per wave (both sets of requests out before the first product) measured SLOWER: see
// HS_DWT_BWD_W_CPW.
template <typename T, bool BN, int RPT = 1, bool ROWS = false, int CPW = 1>
__global__ __launch_bounds__(256)
void dw_tiles_bwd_w_kernel(DwtArgs a, DwtBn n, const T* __restrict__ t, const T* __restrict__ dy) {
    static_assert(!ROWS || CPW == 1, "ROWS: a channel per row of 16 lanes");
    constexpr int LPC = ROWS ? 16 : 64;                              // lanes per channel
    const int lane = threadIdx.x & (LPC - 1), wave = threadIdx.x >> 6;
    const int patch = blockIdx.x, cw = ROWS ? (blockIdx.y * 4 + wave) * 4 + (int)((threadIdx.x & 63) >> 4) : (blockIdx.y * 4 + wave) * CPW;
    if (!ROWS && cw >= a.C) return;
    const int j = patch % a.fw, i = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    bool live[CPW];                                                  // a channel past the last works on the last one and writes nothing
    int c[CPW], TW = 0;
    float g[CPW], bb[CPW];
    const T* __restrict__ tp[CPW];
    const T* __restrict__ gp[CPW];
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        live[k] = cw + k < a.C;
        c[k] = live[k] ? cw + k : a.C - 1;
        g[k] = 1.0f; bb[k] = 0.0f;
        if constexpr (BN) {                                          // the saved statistics: the tiles are normalised on load, as in the forward
            const float mean = n.mean[c[k]], invstd = n.invstd[c[k]];
            g[k] = n.gamma ? n.gamma[c[k]] * invstd : invstd;
            bb[k] = (n.beta ? n.beta[c[k]] : 0.f) - mean * g[k];
        }
        tp[k] = t + dwt_tile(a, b, c[k], i, j, TW);
        gp[k] = dy + (((size_t)b * a.C + c[k]) * a.H + (size_t)i * a.ph) * a.W + (size_t)j * a.pw;
    }
    const int hw = a.pw >> 1, npair = (a.ph / RPT) * hw;
    const float inv_hw = 2.0f * a.inv_pw;
    float acc[CPW][9];
#pragma unroll
    for (int k = 0; k < CPW; ++k)
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[k][q] = 0.f;
    for (int l = lane; l < npair; l += LPC) {
        const int ur = div_by_inv(l, inv_hw), u = RPT * ur, v = 2 * (l - ur * hw);
        float g0[CPW][RPT], g1[CPW][RPT], s[CPW][2 + RPT][4];
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
#pragma unroll
            for (int ro = 0; ro < RPT; ++ro) Pair<T>::ld(gp[k], (size_t)(u + ro) * a.W + v, g0[k][ro], g1[k][ro]);
#pragma unroll
            for (int r = 0; r < 2 + RPT; ++r) {
                Pair<T>::ld(tp[k], (size_t)(u + r) * TW + v, s[k][r][0], s[k][r][1]);
                Pair<T>::ld(tp[k], (size_t)(u + r) * TW + v + 2, s[k][r][2], s[k][r][3]);
            }
        }
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            if constexpr (BN) {
#pragma unroll
                for (int r = 0; r < 2 + RPT; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) s[k][r][q] = dwt_act(fmaf(s[k][r][q], g[k], bb[k]), n.act);
            }
#pragma unroll
            for (int ro = 0; ro < RPT; ++ro)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        acc[k][ky * 3 + kx] = fmaf(g0[k][ro], s[k][ro + ky][kx], acc[k][ky * 3 + kx]);
                        acc[k][ky * 3 + kx] = fmaf(g1[k][ro], s[k][ro + ky][kx + 1], acc[k][ky * 3 + kx]);
                    }
        }
    }
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        float* __restrict__ dst = a.dbank + (size_t)patch * a.ld + c[k] * 9;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const float sum = ROWS ? rowsum16(acc[k][q]) : wave_sum64(acc[k][q]);
            if (lane == 0 && live[k]) dst[q] = sum;
        }
    }
}

static int dwt_args(DwtArgs& a, int dtype, const void* p, const void* q, long ld, int B, int C, int H, int W, int fh, int fw, int pm) {
    if (!p || !q || B <= 0 || C <= 0 || H <= 0 || W <= 0 || fh <= 0 || fw <= 0 || ld < 9L * C) return HS_ERR_BAD_ARG;
    if (dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16) return HS_ERR_BAD_ARG;
    if (H % fh || W % fw) return HS_ERR_NOT_DIVISIBLE;
    const int ph = H / fh, pw = W / fw;
    // pairs: even patch width (then W and the tile image's width are even too); 8-byte aligned tensors; div_by_inv's range
    if ((pw & 1) || ((((size_t)p) | ((size_t)q)) & 7) || (long)B * C > 65535 || H + 2 * fh >= (1 << 21) || W + 2 * fw >= (1 << 21)) return HS_ERR_UNSUPPORTED;
    a = DwtArgs{nullptr, nullptr, ld, B, C, H, W, fh, fw, ph, pw, 1.0f / (float)ph, 1.0f / (float)pw, 1.0f / (float)(ph + 2), 1.0f / (float)(pw + 2),
                1.0f / (float)C, pm ? 1 : 0};
    return HS_OK;
}

}  // namespace hs

using namespace hs;

// ---- the matrix-core / image-level forms above behind three storage-type-aware launchers, shared by the fp32 entry points below and by
// hs_patch_conv_plain_* (hs_patch_conv_train.hip).  0 = launched, 1 = not covered (the caller goes on to its generic kernel).
static void fast_args(ConvBwdArgs& a, const void* bank, long ld, int batch, int c_in, int H, int W, int fh, int fw, int c_out) {
    a = ConvBwdArgs{};
    a.bank = (const float*)bank; a.ld = ld; a.B = batch; a.H = H; a.W = W; a.fh = fh; a.fw = fw; a.ph = H / fh; a.pw = W / fw;
    a.cin = c_in; a.cout = c_out; a.k = 1; a.groups = 1; a.cin_g = c_in; a.cout_g = c_out;
    a.inv_ph = 1.0f / (float)a.ph; a.inv_pw = 1.0f / (float)a.pw;
}
// the pair forms: even patch and image widths (a pair never straddles a tile or a row), 8-byte aligned planes, div_by_inv's range
static bool dw3_pairs(const ConvBwdArgs& a, const void* p, const void* q) {
    return (a.pw & 1) == 0 && (a.W & 1) == 0 && ((((size_t)p) | ((size_t)q)) & 7) == 0 && a.H < (1 << 21) && a.W < (1 << 21) && (long)a.B * a.cin < (1 << 21);
}
#define HS_T2(dtype, F32, BF16) do { if ((dtype) == HS_DTYPE_F32) { F32; } else { BF16; } } while (0)

int hs::try_fast_fwd(int dtype, const void* x, const void* bank, long ld, int batch, int c_in, int H, int W, int fh, int fw, int c_out,
                     int k, int pad, int pad_mode, int groups, const float* scale, const float* shift, int act, void* y, hipStream_t stream) {
    if (H % fh || W % fw || (dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16)) return 1;
    ConvBwdArgs a;
    fast_args(a, bank, ld, batch, c_in, H, W, fh, fw, c_out);
    if (k == 3 && pad == 1 && pad_mode == HS_PAD_ZEROS && groups == c_in && c_in == c_out && !scale && act == HS_ACT_NONE &&
        (long)batch * c_in <= 65535) {                                  // plain depthwise 3x3 (the autograd path's middle layer)
        if (dw3_pairs(a, x, y)) {
            const dim3 grid2((W / 2 + 63) / 64, (H + 3) / 4, batch * c_in);
            HS_T2(dtype, hipLaunchKernelGGL((patch_dw3_pair_kernel<0, float>), grid2, dim3(256), 0, stream, a, (const float*)x, (float*)y),
                         hipLaunchKernelGGL((patch_dw3_pair_kernel<0, bf16_t>), grid2, dim3(256), 0, stream, a, (const bf16_t*)x, (bf16_t*)y));
            return launch_status();
        }
        const dim3 grid((W + 63) / 64, (H + 3) / 4, batch * c_in);
        HS_T2(dtype, hipLaunchKernelGGL((patch_dw3_kernel<0, float>), grid, dim3(256), 0, stream, a, (const float*)x, (float*)y),
                     hipLaunchKernelGGL((patch_dw3_kernel<0, bf16_t>), grid, dim3(256), 0, stream, a, (const bf16_t*)x, (bf16_t*)y));
        return launch_status();
    }
    if (k != 1 || groups != 1 || a.ph * a.pw < 64) return 1;
    a.dy = (const float*)x; a.dx = (float*)y;
    const int mt = (c_out + 15) / 16, kq = (c_in + 15) / 16;
    const dim3 grid((unsigned)(batch * fh * fw));
    // even widths, aligned planes, patches of at least four super-tiles: two adjacent pixels per lane (measured at config 5, visit r5c: 23.8 -> 23.1
    // and 13.3 -> 12.1 us on the 324- / 256-pixel patches, 7.4 -> 9.5 us on the 16-pixel ones, where half of a super-tile's lanes idle)
    const bool px2 = dw3_pairs(a, x, y) && a.ph * a.pw >= 128;
    const bool affine = scale != nullptr || act != HS_ACT_NONE;
    if (affine && (!scale || !shift)) return 1;                         // an activation without the affine rows: the generic kernel
#define HS_FW_L(MTV, KQV, PXV, AFV) HS_T2(dtype, hipLaunchKernelGGL((patch_conv_fwd_k1m_kernel<MTV, KQV, PXV, AFV, float>), grid, dim3(256), 0, stream, a, scale, shift, act), \
                                                 hipLaunchKernelGGL((patch_conv_fwd_k1m_kernel<MTV, KQV, PXV, AFV, bf16_t>), grid, dim3(256), 0, stream, a, scale, shift, act))
#define HS_FW(MTV, KQV) if (mt == MTV && kq == KQV) { \
        if (px2 && affine) HS_FW_L(MTV, KQV, 2, true); else if (px2) HS_FW_L(MTV, KQV, 2, false); \
        else if (affine) HS_FW_L(MTV, KQV, 1, true); else HS_FW_L(MTV, KQV, 1, false); \
        return launch_status(); }
    HS_FW(1, 1) HS_FW(1, 2) HS_FW(1, 3) HS_FW(1, 4) HS_FW(2, 1) HS_FW(2, 2) HS_FW(2, 3) HS_FW(2, 4)
    HS_FW(3, 1) HS_FW(3, 2) HS_FW(3, 3) HS_FW(3, 4) HS_FW(4, 1) HS_FW(4, 2) HS_FW(4, 3)
#undef HS_FW
#undef HS_FW_L
    return 1;
}

int hs::try_fast_bwd_in(int dtype, const void* dy, const void* bank, long ld, int batch, int c_in, int H, int W, int fh, int fw,
                        int c_out, int k, int pad, int pad_mode, int groups, void* dx, hipStream_t stream) {
    if (H % fh || W % fw || (dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16)) return 1;
    ConvBwdArgs a;
    fast_args(a, bank, ld, batch, c_in, H, W, fh, fw, c_out);
    a.dy = (const float*)dy; a.dx = (float*)dx;
    if (k == 3 && pad == 1 && pad_mode == HS_PAD_ZEROS && groups == c_in && c_in == c_out && (long)batch * c_in <= 65535) {
        if (dw3_pairs(a, dy, dx)) {
            const dim3 grid2((W / 2 + 63) / 64, (H + 3) / 4, batch * c_in);
            HS_T2(dtype, hipLaunchKernelGGL((patch_dw3_pair_kernel<1, float>), grid2, dim3(256), 0, stream, a, (const float*)dy, (float*)dx),
                         hipLaunchKernelGGL((patch_dw3_pair_kernel<1, bf16_t>), grid2, dim3(256), 0, stream, a, (const bf16_t*)dy, (bf16_t*)dx));
            return launch_status();
        }
        const dim3 grid((W + 63) / 64, (H + 3) / 4, batch * c_in);
        HS_T2(dtype, hipLaunchKernelGGL((patch_dw3_kernel<1, float>), grid, dim3(256), 0, stream, a, (const float*)dy, (float*)dx),
                     hipLaunchKernelGGL((patch_dw3_kernel<1, bf16_t>), grid, dim3(256), 0, stream, a, (const bf16_t*)dy, (bf16_t*)dx));
        return launch_status();
    }
    if (k == 1 && groups == 1 && a.ph * a.pw < 16 && c_out <= 1024) {
        const dim3 gridt((unsigned)(batch * fh * fw));
        HS_T2(dtype, hipLaunchKernelGGL(patch_conv_bwd_input_tiny_kernel<float>, gridt, dim3(128), (size_t)c_out * 64, stream, a),
                     hipLaunchKernelGGL(patch_conv_bwd_input_tiny_kernel<bf16_t>, gridt, dim3(128), (size_t)c_out * 64, stream, a));
        return launch_status();
    }
    if (k != 1 || groups != 1 || a.ph * a.pw < 16) return 1;
    const int ct = (c_in + 15) / 16, kq = (c_out + 15) / 16;
    const dim3 grid((unsigned)(batch * fh * fw));
    const bool px2 = dw3_pairs(a, dy, dx) && a.ph * a.pw >= 128;
#define HS_BI(CTV, KQV) if (ct == CTV && kq == KQV) { \
        if (px2) HS_T2(dtype, hipLaunchKernelGGL((patch_conv_bwd_input_k1m_kernel<CTV, KQV, 2, float>), grid, dim3(256), 0, stream, a), \
                              hipLaunchKernelGGL((patch_conv_bwd_input_k1m_kernel<CTV, KQV, 2, bf16_t>), grid, dim3(256), 0, stream, a)); \
        else HS_T2(dtype, hipLaunchKernelGGL((patch_conv_bwd_input_k1m_kernel<CTV, KQV, 1, float>), grid, dim3(256), 0, stream, a), \
                          hipLaunchKernelGGL((patch_conv_bwd_input_k1m_kernel<CTV, KQV, 1, bf16_t>), grid, dim3(256), 0, stream, a)); \
        return launch_status(); }
    HS_BI(1, 1) HS_BI(1, 2) HS_BI(1, 3) HS_BI(1, 4) HS_BI(2, 1) HS_BI(2, 2) HS_BI(2, 3) HS_BI(2, 4)
    HS_BI(3, 1) HS_BI(3, 2) HS_BI(3, 3) HS_BI(3, 4) HS_BI(4, 1) HS_BI(4, 2) HS_BI(4, 3) HS_BI(6, 1) HS_BI(6, 2)
#undef HS_BI
    return 1;
}

int hs::try_fast_bwd_w(int dtype, const void* x, const void* dy, int batch, int c_in, int H, int W, int fh, int fw, int c_out, int k,
                       int pad, int pad_mode, int groups, void* dbank, long ld, hipStream_t stream) {
    if (H % fh || W % fw || (dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16)) return 1;
    ConvBwdArgs a;
    fast_args(a, nullptr, ld, batch, c_in, H, W, fh, fw, c_out);
    a.x = (const float*)x; a.dy = (const float*)dy; a.dbank = (float*)dbank;
    if (k == 3 && pad == 1 && pad_mode == HS_PAD_ZEROS && groups == c_in && c_in == c_out && (c_in + 3) / 4 <= 65535) {
        const dim3 grid((unsigned)(batch * fh * fw), (c_in + 3) / 4);
        if (dw3_pairs(a, x, dy)) {
            HS_T2(dtype, hipLaunchKernelGGL(patch_dw3_bwd_weight_pair_kernel<float>, grid, dim3(256), 0, stream, a),
                         hipLaunchKernelGGL(patch_dw3_bwd_weight_pair_kernel<bf16_t>, grid, dim3(256), 0, stream, a));
            return launch_status();
        }
        HS_T2(dtype, hipLaunchKernelGGL(patch_dw3_bwd_weight_kernel<float>, grid, dim3(256), 0, stream, a),
                     hipLaunchKernelGGL(patch_dw3_bwd_weight_kernel<bf16_t>, grid, dim3(256), 0, stream, a));
        return launch_status();
    }
    if (k == 1 && groups == 1 && a.ph * a.pw < 16 && (long)c_out * c_in < (1 << 21) && (size_t)(c_out + c_in) * 17 * 4 <= 64 * 1024) {
        const dim3 gridt((unsigned)(batch * fh * fw));
        const size_t lds = (size_t)(c_out + c_in) * 17 * 4;
        HS_T2(dtype, hipLaunchKernelGGL(patch_conv_bwd_weight_tiny_kernel<float>, gridt, dim3(256), lds, stream, a),
                     hipLaunchKernelGGL(patch_conv_bwd_weight_tiny_kernel<bf16_t>, gridt, dim3(256), lds, stream, a));
        return launch_status();
    }
    if (k != 1 || groups != 1 || a.ph * a.pw < 16) return 1;
    const bool vec = (a.pw & 3) == 0 && ((a.ph * a.pw) & 15) == 0 && (W & 3) == 0 &&
                     ((((size_t)x) | ((size_t)dy)) & (dtype == HS_DTYPE_F32 ? 15 : 7)) == 0;           // four elements per load, either storage type
    const int mt = (c_out + 15) / 16, nt = (c_in + 15) / 16;
    const dim3 grid((unsigned)(batch * fh * fw));
    const bool pairs = !vec && (a.pw & 1) == 0 && (W & 1) == 0 && ((((size_t)x) | ((size_t)dy)) & 7) == 0;       // either storage type
#define HS_BW(MTV, NTV) if (mt == MTV && nt == NTV) { \
        if (vec) HS_T2(dtype, hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 1, float>), grid, dim3(256), 0, stream, a, ConvBn{}), \
                              hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 1, bf16_t>), grid, dim3(256), 0, stream, a, ConvBn{})); \
        else if (pairs) HS_T2(dtype, hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 2, float>), grid, dim3(256), 0, stream, a, ConvBn{}), \
                                     hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 2, bf16_t>), grid, dim3(256), 0, stream, a, ConvBn{})); \
        else HS_T2(dtype, hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 0, float>), grid, dim3(256), 0, stream, a, ConvBn{}), \
                          hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 0, bf16_t>), grid, dim3(256), 0, stream, a, ConvBn{})); \
        return launch_status(); }
    HS_BW(1, 1) HS_BW(1, 2) HS_BW(1, 3) HS_BW(1, 4) HS_BW(2, 1) HS_BW(2, 2) HS_BW(2, 3) HS_BW(2, 4) HS_BW(3, 1) HS_BW(3, 2) HS_BW(4, 1) HS_BW(4, 2)
#undef HS_BW
    return 1;
}
#undef HS_T2

static int fill_bwd(ConvBwdArgs& a, const float* x, const float* dy, const float* bank, int64_t ld, int32_t batch,
                    int32_t c_in, int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad,
                    int32_t pad_mode, int32_t groups) {
    if (!dy || batch <= 0 || c_in <= 0 || c_out <= 0 || H <= 0 || W <= 0 || fh <= 0 || fw <= 0 || k <= 0 || groups <= 0)
        return HS_ERR_BAD_ARG;
    if (2 * pad != k - 1) return HS_ERR_UNSUPPORTED;
    if (pad > 3) return HS_ERR_UNSUPPORTED;
    if (H % fh != 0 || W % fw != 0) return HS_ERR_NOT_DIVISIBLE;
    if (c_in % groups != 0 || c_out % groups != 0) return HS_ERR_BAD_ARG;
    if (pad_mode < HS_PAD_ZEROS || pad_mode > HS_PAD_CIRCULAR) return HS_ERR_BAD_ARG;
    a.x = x; a.dy = dy; a.bank = bank; a.ld = (long)ld;
    a.B = batch; a.H = H; a.W = W; a.fh = fh; a.fw = fw; a.ph = H / fh; a.pw = W / fw;
    a.cin = c_in; a.cout = c_out; a.k = k; a.pad = pad; a.pad_mode = pad_mode; a.groups = groups;
    a.cin_g = c_in / groups; a.cout_g = c_out / groups;
    if (ld < (int64_t)c_out * a.cin_g * k * k) return HS_ERR_BAD_ARG;
    return HS_OK;
}

extern "C" int hs_patch_conv_bwd_input(const float* dy, const float* bank, int64_t ld, int32_t batch, int32_t c_in,
                                       int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad,
                                       int32_t pad_mode, int32_t groups, float* dx, void* stream) {
    ConvBwdArgs a;
    int st = fill_bwd(a, nullptr, dy, bank, ld, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups);
    if (st != HS_OK) return st;
    if (!bank || !dx) return HS_ERR_BAD_ARG;
    a.dx = dx; a.dbank = nullptr;
    {
        const int r = try_fast_bwd_in(HS_DTYPE_F32, dy, bank, (long)ld, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups, dx,
                                      (hipStream_t)stream);
        if (r != 1) return r;
    }
    const size_t total = (size_t)batch * c_in * H * W;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(patch_conv_bwd_input_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return launch_status();
}

extern "C" int hs_patch_conv_bwd_weight(const float* x, const float* dy, int32_t batch, int32_t c_in, int32_t H, int32_t W,
                                        int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad, int32_t pad_mode,
                                        int32_t groups, float* dbank, int64_t ld, void* stream) {
    ConvBwdArgs a;
    int st = fill_bwd(a, x, dy, nullptr, ld, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups);
    if (st != HS_OK) return st;
    if (!x || !dbank) return HS_ERR_BAD_ARG;
    a.dx = nullptr; a.dbank = dbank;
    {
        const int r = try_fast_bwd_w(HS_DTYPE_F32, x, dy, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups, dbank, (long)ld,
                                     (hipStream_t)stream);
        if (r != 1) return r;
    }
    // output-channel block: as many channels (whole groups when the convolution is grouped) as fit beside the X tile
    const size_t npix = (size_t)a.ph * a.pw, tpos = (size_t)(a.ph + 2 * pad) * (a.pw + 2 * pad);
    int ob = 0;
    size_t lds = 0;
    for (const size_t budget : {(size_t)64 * 1024, (size_t)150 * 1024}) {
        if (groups == 1) {
            const size_t xb = (size_t)c_in * tpos * sizeof(float);
            if (xb + npix * sizeof(float) > budget) continue;
            ob = (int)((budget - xb) / (npix * sizeof(float)));
            ob = ob > c_out ? c_out : ob;
            lds = xb + (size_t)ob * npix * sizeof(float);
        } else {
            const size_t per_group = ((size_t)a.cout_g * npix + (size_t)a.cin_g * tpos) * sizeof(float);
            if (per_group > budget) continue;
            int gb = (int)(budget / per_group);
            gb = gb > groups ? groups : gb;
            ob = gb * a.cout_g;
            lds = (size_t)gb * per_group;
        }
        break;
    }
    if (ob <= 0) return HS_ERR_LDS;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};
        const int e = allow_full_lds((const void*)patch_conv_bwd_weight_kernel, done);
        if (e != HS_OK) return e;
    }
    const unsigned nblk = (unsigned)((c_out + ob - 1) / ob);
    hipLaunchKernelGGL(patch_conv_bwd_weight_kernel, dim3((unsigned)(batch * fh * fw), nblk), dim3(256), lds,
                       (hipStream_t)stream, a, ob);
    return launch_status();
}

// ---- valid depthwise 3 x 3 on halo tiles (dw_tiles_*): the middle layer of a train-mode v1_0 inverted residual, tile image <-> (B, C, H, W)
extern "C" int hs_dw_tiles_fwd(int32_t dtype, const void* tiled, const float* bank, int64_t ld, int32_t batch, int32_t channels, int32_t H,
                               int32_t W, int32_t fh, int32_t fw, void* y, int32_t patch_major, void* stream) {
    DwtArgs a;
    const int st = dwt_args(a, dtype, tiled, y, (long)ld, batch, channels, H, W, fh, fw, patch_major);
    if (st != HS_OK) return st;
    if (!bank || (((size_t)bank) & 3)) return HS_ERR_BAD_ARG;
    a.bank = bank;
    const dim3 grid((W / 2 + 63) / 64, (H + 3) / 4, batch * channels);
    if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_fwd_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, a, DwtBn{}, (const float*)tiled, (float*)y);
    else hipLaunchKernelGGL((dw_tiles_fwd_kernel<bf16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, a, DwtBn{}, (const bf16_t*)tiled, (bf16_t*)y);
    return launch_status();
}

// elements per channel of the tile tensor and the distance between the first elements of consecutive channels, in either layout
static void dwt_bn_geometry(const DwtArgs& a, DwtBn& n) {
    const long tile = (long)(a.ph + 2) * (a.pw + 2);
    n.n = (float)((long)a.B * a.fh * a.fw * tile);
    n.shift_stride = a.pm ? tile : (long)a.fh * a.fw * tile;
}

// hs_dw_tiles_fwd with training-mode BatchNorm + activation applied to the tiles ON LOAD: `bn_partial` = hs_bn_train_stats_fwd's
// workspace for the tile tensor seen as (B', C, pixels) -- patch-major: B' = batch fh fw frames of (ph + 2)(pw + 2) pixels; image of
// tiles: B' = batch frames of fh (ph + 2) x fw (pw + 2) pixels.  Writes save_mean / save_invstd, updates the running estimates and the
// step counter exactly as hs_bn_act_train_fwd does.
extern "C" int hs_dw_tiles_bn_fwd(int32_t dtype, const void* tiled, const float* bn_partial, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, int32_t act, float* save_mean,
                                  float* save_invstd, int64_t* num_batches_tracked, const float* bank, int64_t ld, int32_t batch,
                                  int32_t channels, int32_t H, int32_t W, int32_t fh, int32_t fw, void* y, int32_t patch_major, void* stream) {
    DwtArgs a;
    const int st = dwt_args(a, dtype, tiled, y, (long)ld, batch, channels, H, W, fh, fw, patch_major);
    if (st != HS_OK) return st;
    if (!bank || (((size_t)bank) & 3) || !bn_partial || !save_mean || !save_invstd || ((running_mean != nullptr) != (running_var != nullptr)) ||
        act < HS_ACT_NONE || act > HS_ACT_RELU6 || eps < 0.f) return HS_ERR_BAD_ARG;
    a.bank = bank;
    DwtBn n{bn_partial, gamma, beta, save_mean, save_invstd, running_mean, running_var, (long long*)num_batches_tracked, eps, momentum, 0.f, 0, act};
    dwt_bn_geometry(a, n);
    if ((a.ph & 3) == 0 && HS_DWT_FWD_RPT4) {       // four output rows per thread (round 6): six tile rows for four outputs rows instead of eight
        const dim3 grid4((W / 2 + 63) / 64, (H / 4 + 3) / 4, batch * channels);
        if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_fwd_kernel<float, true, 4>), grid4, dim3(256), 0, (hipStream_t)stream, a, n, (const float*)tiled, (float*)y);
        else hipLaunchKernelGGL((dw_tiles_fwd_kernel<bf16_t, true, 4>), grid4, dim3(256), 0, (hipStream_t)stream, a, n, (const bf16_t*)tiled, (bf16_t*)y);
        return launch_status();
    }
    if ((a.ph & 1) == 0) {                 // two output rows per thread: they share a patch
        const dim3 grid2((W / 2 + 63) / 64, (H / 2 + 3) / 4, batch * channels);
        if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_fwd_kernel<float, true, 2>), grid2, dim3(256), 0, (hipStream_t)stream, a, n, (const float*)tiled, (float*)y);
        else hipLaunchKernelGGL((dw_tiles_fwd_kernel<bf16_t, true, 2>), grid2, dim3(256), 0, (hipStream_t)stream, a, n, (const bf16_t*)tiled, (bf16_t*)y);
        return launch_status();
    }
    const dim3 grid((W / 2 + 63) / 64, (H + 3) / 4, batch * channels);
    if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_fwd_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream, a, n, (const float*)tiled, (float*)y);
    else hipLaunchKernelGGL((dw_tiles_fwd_kernel<bf16_t, true>), grid, dim3(256), 0, (hipStream_t)stream, a, n, (const bf16_t*)tiled, (bf16_t*)y);
    return launch_status();
}

// rows per thread of hs_dw_tiles_bwd_in*: 3 where the tile height allows, else 2 (even patches), else 1
static int dwt_bwd_in_rows(const DwtArgs& a) { return ((a.ph + 2) % 3 == 0 && HS_DWT_BWD_IN_RPT3) ? 3 : ((a.ph & 1) == 0 ? 2 : 1); }
static dim3 dwt_bwd_in_grid(const DwtArgs& a, int rpt) {
    return dim3((a.fw * (a.pw + 2) / 2 + 63) / 64, (a.fh * (a.ph + 2) / rpt + 3) / 4, a.B * a.C);
}
template <bool STATS>
static int dwt_bwd_in_launch(int dtype, const DwtArgs& a, const void* dy, void* dtiled, const DwtBn& n, const void* xt, float* partial, void* stream) {
    const int rpt = dwt_bwd_in_rows(a);
    const dim3 grid = dwt_bwd_in_grid(a, rpt);
#define HS_DWT_BI(RPT) \
    if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_bwd_in_kernel<float, RPT, STATS>), grid, dim3(256), 0, (hipStream_t)stream, a, (const float*)dy, (float*)dtiled, n, (const float*)xt, partial); \
    else hipLaunchKernelGGL((dw_tiles_bwd_in_kernel<bf16_t, RPT, STATS>), grid, dim3(256), 0, (hipStream_t)stream, a, (const bf16_t*)dy, (bf16_t*)dtiled, n, (const bf16_t*)xt, partial);
    if (rpt == 3) { HS_DWT_BI(3) } else if (rpt == 2) { HS_DWT_BI(2) } else { HS_DWT_BI(1) }
#undef HS_DWT_BI
    return launch_status();
}

extern "C" int hs_dw_tiles_bwd_in(int32_t dtype, const void* dy, const float* bank, int64_t ld, int32_t batch, int32_t channels, int32_t H,
                                  int32_t W, int32_t fh, int32_t fw, void* dtiled, int32_t patch_major, void* stream) {
    DwtArgs a;
    const int st = dwt_args(a, dtype, dy, dtiled, (long)ld, batch, channels, H, W, fh, fw, patch_major);
    if (st != HS_OK) return st;
    if (!bank) return HS_ERR_BAD_ARG;
    a.bank = bank;
    return dwt_bwd_in_launch<false>(dtype, a, dy, dtiled, DwtBn{}, nullptr, nullptr, stream);
}

// hs_dw_tiles_bwd_in for a layer whose tiles are normalised on load (hs_dw_tiles_bn_fwd), leaving -- besides the gradient of the normalised tiles
// in `dtiled` -- the two sums of BatchNorm1's adjoint over the tile tensor as one pair per workgroup in `partial` (round 6):
// hs_dw_tiles_bn_bwd_in_partials() pairs per channel, combined by hs_bn_act_train_bwd_apply.  `tiled`: the RAW tiles (the forward's input).
extern "C" int64_t hs_dw_tiles_bn_bwd_in_partials(int32_t batch, int32_t H, int32_t W, int32_t fh, int32_t fw) {
    if (batch <= 0 || H <= 0 || W <= 0 || fh <= 0 || fw <= 0 || H % fh || W % fw) return 0;
    DwtArgs a{};
    a.B = batch; a.C = 1; a.fh = fh; a.fw = fw; a.ph = H / fh; a.pw = W / fw;
    const dim3 g = dwt_bwd_in_grid(a, dwt_bwd_in_rows(a));
    return (int64_t)batch * g.y * g.x;
}
extern "C" int hs_dw_tiles_bn_bwd_in(int32_t dtype, const void* dy, const float* bank, int64_t ld, const void* tiled, const float* gamma,
                                     const float* beta, const float* save_mean, const float* save_invstd, int32_t act, int32_t batch,
                                     int32_t channels, int32_t H, int32_t W, int32_t fh, int32_t fw, void* dtiled, float* partial,
                                     int32_t patch_major, void* stream) {
    DwtArgs a;
    const int st = dwt_args(a, dtype, dy, dtiled, (long)ld, batch, channels, H, W, fh, fw, patch_major);
    if (st != HS_OK) return st;
    if (!bank || !tiled || !save_mean || !save_invstd || !partial || act < HS_ACT_NONE || act > HS_ACT_RELU6 || (((size_t)tiled) & 7)) return HS_ERR_BAD_ARG;
    a.bank = bank;
    DwtBn n{nullptr, gamma, beta, const_cast<float*>(save_mean), const_cast<float*>(save_invstd), nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0, act};
    return dwt_bwd_in_launch<true>(dtype, a, dy, dtiled, n, tiled, partial, stream);
}

extern "C" int hs_dw_tiles_bwd_w(int32_t dtype, const void* tiled, const void* dy, int32_t batch, int32_t channels, int32_t H, int32_t W,
                                 int32_t fh, int32_t fw, float* dbank, int64_t ld, int32_t patch_major, void* stream) {
    DwtArgs a;
    const int st = dwt_args(a, dtype, tiled, dy, (long)ld, batch, channels, H, W, fh, fw, patch_major);
    if (st != HS_OK) return st;
    if (!dbank) return HS_ERR_BAD_ARG;
    a.dbank = dbank;
    const dim3 grid((unsigned)(batch * fh * fw), (channels + 3) / 4);
    if ((a.ph & 1) == 0 && (a.ph / 2) * (a.pw / 2) <= 16) {                  // a channel per row of 16 lanes
        const dim3 gridr((unsigned)(batch * fh * fw), (channels + 15) / 16);
        if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<float, false, 2, true>), gridr, dim3(256), 0, (hipStream_t)stream, a, DwtBn{}, (const float*)tiled, (const float*)dy);
        else hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<bf16_t, false, 2, true>), gridr, dim3(256), 0, (hipStream_t)stream, a, DwtBn{}, (const bf16_t*)tiled, (const bf16_t*)dy);
    } else if ((a.ph & 1) == 0) {
        const dim3 gridc((unsigned)(batch * fh * fw), (channels + 4 * HS_DWT_BWD_W_CPW - 1) / (4 * HS_DWT_BWD_W_CPW));
        if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<float, false, 2, false, HS_DWT_BWD_W_CPW>), gridc, dim3(256), 0, (hipStream_t)stream, a, DwtBn{}, (const float*)tiled, (const float*)dy);
        else hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<bf16_t, false, 2, false, HS_DWT_BWD_W_CPW>), gridc, dim3(256), 0, (hipStream_t)stream, a, DwtBn{}, (const bf16_t*)tiled, (const bf16_t*)dy);
    } else if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, a, DwtBn{}, (const float*)tiled, (const float*)dy);
    else hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<bf16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, a, DwtBn{}, (const bf16_t*)tiled, (const bf16_t*)dy);
    return launch_status();
}

// hs_dw_tiles_bwd_w on the RAW tiles of hs_dw_tiles_bn_fwd: the same on-load normalisation from the saved statistics
extern "C" int hs_dw_tiles_bn_bwd_w(int32_t dtype, const void* tiled, const void* dy, const float* gamma, const float* beta, const float* save_mean,
                                    const float* save_invstd, int32_t act, int32_t batch, int32_t channels, int32_t H, int32_t W, int32_t fh,
                                    int32_t fw, float* dbank, int64_t ld, int32_t patch_major, void* stream) {
    DwtArgs a;
    const int st = dwt_args(a, dtype, tiled, dy, (long)ld, batch, channels, H, W, fh, fw, patch_major);
    if (st != HS_OK) return st;
    if (!dbank || !save_mean || !save_invstd || act < HS_ACT_NONE || act > HS_ACT_RELU6) return HS_ERR_BAD_ARG;
    a.dbank = dbank;
    DwtBn n{nullptr, gamma, beta, const_cast<float*>(save_mean), const_cast<float*>(save_invstd), nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0, act};
    const dim3 grid((unsigned)(batch * fh * fw), (channels + 3) / 4);
    if ((a.ph & 1) == 0 && (a.ph / 2) * (a.pw / 2) <= 16) {                  // a channel per row of 16 lanes
        const dim3 gridr((unsigned)(batch * fh * fw), (channels + 15) / 16);
        if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<float, true, 2, true>), gridr, dim3(256), 0, (hipStream_t)stream, a, n, (const float*)tiled, (const float*)dy);
        else hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<bf16_t, true, 2, true>), gridr, dim3(256), 0, (hipStream_t)stream, a, n, (const bf16_t*)tiled, (const bf16_t*)dy);
    } else if ((a.ph & 1) == 0) {
        const dim3 gridc((unsigned)(batch * fh * fw), (channels + 4 * HS_DWT_BWD_W_CPW - 1) / (4 * HS_DWT_BWD_W_CPW));
        if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<float, true, 2, false, HS_DWT_BWD_W_CPW>), gridc, dim3(256), 0, (hipStream_t)stream, a, n, (const float*)tiled, (const float*)dy);
        else hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<bf16_t, true, 2, false, HS_DWT_BWD_W_CPW>), gridc, dim3(256), 0, (hipStream_t)stream, a, n, (const bf16_t*)tiled, (const bf16_t*)dy);
    } else if (dtype == HS_DTYPE_F32) hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream, a, n, (const float*)tiled, (const float*)dy);
    else hipLaunchKernelGGL((dw_tiles_bwd_w_kernel<bf16_t, true>), grid, dim3(256), 0, (hipStream_t)stream, a, n, (const bf16_t*)tiled, (const bf16_t*)dy);
    return launch_status();
}

// ---- round 5: the train-mode inverted residual's BatchNorm2 + ReLU6 normalised ON LOAD by its last 1 x 1 layer (no normalised copy of
// the hidden map): forward on the raw input + hs_bn_train_stats_fwd's slice sums, weight gradient on the raw input + saved statistics.
// k = 1, groups = 1, c_out <= 32, c_in <= 64, patches of >= 64 pixels; anything else: HS_ERR_UNSUPPORTED (the caller normalises first).
#define HS_T2(dtype, F32, BF16) do { if ((dtype) == HS_DTYPE_F32) { F32; } else { BF16; } } while (0)
extern "C" int hs_patch_conv_bn_fwd(int32_t dtype, const void* x, const float* bn_partial, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, int32_t act, float* save_mean,
                                    float* save_invstd, int64_t* num_batches_tracked, const float* bank, int64_t ld, int32_t batch,
                                    int32_t c_in, int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t c_out, void* y, void* stream) {
    if (!x || !y || !bank || !bn_partial || !save_mean || !save_invstd || batch <= 0 || c_in <= 0 || c_out <= 0 || H <= 0 || W <= 0 || fh <= 0 || fw <= 0)
        return HS_ERR_BAD_ARG;
    if ((dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16) || ((running_mean != nullptr) != (running_var != nullptr)) ||
        act < HS_ACT_NONE || act > HS_ACT_RELU6 || eps < 0.f || ld < (int64_t)c_in * c_out) return HS_ERR_BAD_ARG;
    if (H % fh || W % fw) return HS_ERR_NOT_DIVISIBLE;
    ConvBwdArgs a;
    fast_args(a, bank, (long)ld, batch, c_in, H, W, fh, fw, c_out);
    const int mt = (c_out + 15) / 16, kq = (c_in + 15) / 16;
    if (a.ph * a.pw < 64 || mt > 2 || kq > 4 || (long)batch * fh * fw > 0x7fffffffL) return HS_ERR_UNSUPPORTED;
    a.dy = (const float*)x; a.dx = (float*)y;
    ConvBn bn{bn_partial, gamma, beta, save_mean, save_invstd, running_mean, running_var, (long long*)num_batches_tracked, eps, momentum,
              (float)((double)batch * H * W), act};
    const dim3 grid((unsigned)(batch * fh * fw));
    hipStream_t s = (hipStream_t)stream;
    const bool px2 = dw3_pairs(a, x, y) && a.ph * a.pw >= 128;
#define HS_FB_L(MTV, KQV, PXV) HS_T2(dtype, hipLaunchKernelGGL((patch_conv_bn_fwd_k1m_kernel<MTV, KQV, PXV, float>), grid, dim3(256), 0, s, a, bn), \
                                            hipLaunchKernelGGL((patch_conv_bn_fwd_k1m_kernel<MTV, KQV, PXV, bf16_t>), grid, dim3(256), 0, s, a, bn))
#define HS_FB(MTV, KQV) if (mt == MTV && kq == KQV) { if (px2) HS_FB_L(MTV, KQV, 2); else HS_FB_L(MTV, KQV, 1); return launch_status(); }
    HS_FB(1, 1) HS_FB(1, 2) HS_FB(1, 3) HS_FB(1, 4) HS_FB(2, 1) HS_FB(2, 2) HS_FB(2, 3) HS_FB(2, 4)
#undef HS_FB
#undef HS_FB_L
    return HS_ERR_UNSUPPORTED;
}

extern "C" int hs_patch_conv_bn_bwd_w(int32_t dtype, const void* x, const void* dy, const float* gamma, const float* beta, const float* save_mean,
                                      const float* save_invstd, int32_t act, int32_t batch, int32_t c_in, int32_t H, int32_t W, int32_t fh,
                                      int32_t fw, int32_t c_out, float* dbank, int64_t ld, void* stream) {
    if (!x || !dy || !dbank || !save_mean || !save_invstd || batch <= 0 || c_in <= 0 || c_out <= 0 || H <= 0 || W <= 0 || fh <= 0 || fw <= 0)
        return HS_ERR_BAD_ARG;
    if ((dtype != HS_DTYPE_F32 && dtype != HS_DTYPE_BF16) || act < HS_ACT_NONE || act > HS_ACT_RELU6 || ld < (int64_t)c_in * c_out) return HS_ERR_BAD_ARG;
    if (H % fh || W % fw) return HS_ERR_NOT_DIVISIBLE;
    ConvBwdArgs a;
    fast_args(a, nullptr, (long)ld, batch, c_in, H, W, fh, fw, c_out);
    a.x = (const float*)x; a.dy = (const float*)dy; a.dbank = dbank;
    const int mt = (c_out + 15) / 16, nt = (c_in + 15) / 16;
    if (a.ph * a.pw < 16 || mt > 2 || nt > 4) return HS_ERR_UNSUPPORTED;
    ConvBn bn{nullptr, gamma, beta, const_cast<float*>(save_mean), const_cast<float*>(save_invstd), nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, act};
    const bool vec = (a.pw & 3) == 0 && ((a.ph * a.pw) & 15) == 0 && (W & 3) == 0 && ((((size_t)x) | ((size_t)dy)) & (dtype == HS_DTYPE_F32 ? 15 : 7)) == 0;
    const bool pairs = !vec && (a.pw & 1) == 0 && (W & 1) == 0 && ((((size_t)x) | ((size_t)dy)) & 7) == 0;
    const dim3 grid((unsigned)(batch * fh * fw));
    hipStream_t s = (hipStream_t)stream;
#define HS_BWB(MTV, NTV) if (mt == MTV && nt == NTV) { \
        if (vec) HS_T2(dtype, hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 1, float, true>), grid, dim3(256), 0, s, a, bn), \
                              hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 1, bf16_t, true>), grid, dim3(256), 0, s, a, bn)); \
        else if (pairs) HS_T2(dtype, hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 2, float, true>), grid, dim3(256), 0, s, a, bn), \
                                     hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 2, bf16_t, true>), grid, dim3(256), 0, s, a, bn)); \
        else HS_T2(dtype, hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 0, float, true>), grid, dim3(256), 0, s, a, bn), \
                          hipLaunchKernelGGL((patch_conv_bwd_weight_k1m_kernel<MTV, NTV, 0, bf16_t, true>), grid, dim3(256), 0, s, a, bn)); \
        return launch_status(); }
    HS_BWB(1, 1) HS_BWB(1, 2) HS_BWB(1, 3) HS_BWB(1, 4) HS_BWB(2, 1) HS_BWB(2, 2) HS_BWB(2, 3) HS_BWB(2, 4)
#undef HS_BWB
    return HS_ERR_UNSUPPORTED;
}
#undef HS_T2

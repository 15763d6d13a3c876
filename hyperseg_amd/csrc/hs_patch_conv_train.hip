// Training-path kernels of the dynamic patch-wise convolution with a STORAGE type parameter: fp32, or bf16 storage with
// fp32 accumulation (BASELINE config 5: "training loop ... bf16").  Plain tensors in and out -- the training route builds
// the stage input with stock differentiable ops (hyperseg_amd/autograd.py) -- so these take (B, C, H, W) activations
// instead of a fused prologue:
//     hs_patch_conv_plain_fwd     y      = patch_conv(x, bank)                      (Op A / Op B, meta_patch.py:35-57)
//     hs_patch_conv_plain_bwd_in  dx     = adjoint w.r.t. x (padding folded back)   
//     hs_patch_conv_plain_bwd_w   dbank  = per-patch weight gradient
// The reference has no reduced-precision path at all; bf16 here means: ACTIVATIONS and their gradients are READ and
// WRITTEN as bf16 (half the HBM bytes of the fp32 kernels, which is what these memory-bound kernels pay for), every product and
// sum is fp32, results are rounded to nearest-even once on store.  The BANK and its gradient are fp32 in either case (round 4):
// the bank is the output of signal2weights and its gradient the input of that layer's adjoint, both fp32 ("master weights"), it
// is a few per cent of a launch's bytes, and storing it as bf16 cost a cast launch per layer and direction (30 of the config-5
// step's 147 launches).
#include "hs_common.h"

namespace hs {

struct PlainArgs {
    const void* x; const void* bank; const void* dy;
    void* y; void* dx; void* dbank;
    long ld;
    int B, H, W, fh, fw, ph, pw, cin, cout, k, pad, pad_mode, groups, cin_g, cout_g;
    int TH, TW, tiles_y, tiles_x, w_stride, ob;
};

// ---- forward: one workgroup = one tile of one patch; bank and halo tile staged in LDS as fp32 ------------------------
template <typename T>
__global__ __launch_bounds__(256)
void plain_fwd_kernel(PlainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const T* __restrict__ x = (const T*)a.x; const float* __restrict__ bank = (const float*)a.bank; T* __restrict__ y = (T*)a.y;
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    const int tx_i = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty_i = blk % a.tiles_y; blk /= a.tiles_y;
    const int patch = blk;
    const int j = patch % a.fw, i = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const int kk = a.k * a.k, wrow = a.cin_g * kk;
    const int y0 = i * a.ph + ty_i * a.TH, x0 = j * a.pw + tx_i * a.TW;
    const int th = min(a.TH, (i + 1) * a.ph - y0), tw = min(a.TW, (j + 1) * a.pw - x0);
    const int HH = a.TH + 2 * a.pad, HW = a.TW + 2 * a.pad, tpos = HH * HW;
    float* wl = lds;                                       // [cout][w_stride]
    float* xl = lds + (size_t)a.cout * a.w_stride;         // [cin][tpos]
    const size_t wbase = (size_t)patch * a.ld;
    for (int e = tid; e < a.cout * wrow; e += 256) {
        const int o = e / wrow, r = e - o * wrow;
        wl[o * a.w_stride + r] = bank[wbase + e];
    }
    for (int e = tid; e < a.cin * tpos; e += 256) {
        const int c = e / tpos, pos = e - c * tpos;
        const int u = pos / HW, v = pos - u * HW;
        const int yy = pad_index(y0 + u - a.pad, a.H, a.pad_mode), xx = pad_index(x0 + v - a.pad, a.W, a.pad_mode);
        xl[e] = (yy >= 0 && xx >= 0) ? Store<T>::ld(x, (((size_t)b * a.cin + c) * a.H + yy) * a.W + xx) : 0.0f;
    }
    __syncthreads();
    const int npix = a.TH * a.TW;
    for (int idx = tid; idx < a.cout * npix; idx += 256) {
        const int o = idx / npix, pix = idx - o * npix;
        const int u = pix / a.TW, v = pix - u * a.TW;
        if (u >= th || v >= tw) continue;
        const int g = o / a.cout_g;
        const float* wr = wl + o * a.w_stride;
        const float* xr = xl + (size_t)g * a.cin_g * tpos + u * HW + v;
        float acc = 0.0f;
        for (int c = 0; c < a.cin_g; ++c)
            for (int ky = 0; ky < a.k; ++ky)
                for (int kx = 0; kx < a.k; ++kx)
                    acc = fmaf(wr[(c * a.k + ky) * a.k + kx], xr[c * tpos + ky * HW + kx], acc);
        Store<T>::st(y, (((size_t)b * a.cout + o) * a.H + (y0 + u)) * a.W + (x0 + v), acc);
    }
}

// ---- input gradient: one workgroup = one tile of one patch's INPUT pixels.  The gradient of an input pixel gathers from
// the outputs within k/2 of it, each weighted with the filters of the patch that owns the OUTPUT pixel; the dY halo tile
// (outputs around the input tile) is staged in LDS, the up-to-9 banks involved are read through L1/L2.  Padding is
// handled by aliasing: an input pixel also receives the gradients of the padded coordinates that map onto it. -----------
__device__ __forceinline__ int pad_aliases_of(int i, int n, int pad, int mode, int* out) {
    int cnt = 0;
    out[cnt++] = i;
    if (pad == 0 || mode == HS_PAD_ZEROS) return cnt;
    for (int p = -pad; p < 0; ++p)
        if (pad_index(p, n, mode) == i) out[cnt++] = p;
    for (int p = n; p < n + pad; ++p)
        if (pad_index(p, n, mode) == i) out[cnt++] = p;
    return cnt;
}

template <typename T>
__global__ __launch_bounds__(256)
void plain_bwd_in_kernel(PlainArgs a) {
    const T* __restrict__ dy = (const T*)a.dy; const float* __restrict__ bank = (const float*)a.bank; T* __restrict__ dx = (T*)a.dx;
    const size_t total = (size_t)a.B * a.cin * a.H * a.W;
    const int kk = a.k * a.k;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int xq = e % a.W; size_t r = e / a.W;
        const int yq = r % a.H; r /= a.H;
        const int c = r % a.cin; const int b = r / a.cin;
        const int g = c / a.cin_g, cl = c - g * a.cin_g;
        int ys[8], xs[8];
        const int ny = pad_aliases_of(yq, a.H, a.pad, a.pad_mode, ys);
        const int nx = pad_aliases_of(xq, a.W, a.pad, a.pad_mode, xs);
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
                        const size_t wb = (size_t)p * a.ld + (size_t)cl * kk + ky * a.k + kx;
                        const size_t db = (((size_t)b * a.cout + g * a.cout_g) * a.H + yo) * a.W + xo;
                        for (int o = 0; o < a.cout_g; ++o)
                            acc = fmaf(bank[wb + (size_t)(g * a.cout_g + o) * a.cin_g * kk],
                                       Store<T>::ld(dy, db + (size_t)o * a.H * a.W), acc);
                    }
            }
        Store<T>::st(dx, e, acc);
    }
}

// ---- weight gradient: one workgroup per (patch, block of output channels); dY tile and padded X tile in LDS (fp32) ------
template <typename T>
__global__ __launch_bounds__(256)
void plain_bwd_w_kernel(PlainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const T* __restrict__ x = (const T*)a.x; const T* __restrict__ dy = (const T*)a.dy; float* __restrict__ dbank = (float*)a.dbank;
    const int patch = blockIdx.x;
    const int o0 = blockIdx.y * a.ob;
    const int on = min(a.ob, a.cout - o0);
    const int j = patch % a.fw, i = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);
    const int npix = a.ph * a.pw;
    const int HH = a.ph + 2 * a.pad, HW = a.pw + 2 * a.pad, tpos = HH * HW;
    const int c_lo = (o0 / a.cout_g) * a.cin_g;
    const int c_hi = ((o0 + on - 1) / a.cout_g + 1) * a.cin_g;
    const int nx = c_hi - c_lo;
    float* dyl = lds;                          // [on][npix]
    float* xl = lds + (size_t)a.ob * npix;     // [nx][tpos]
    const int y0 = i * a.ph, x0 = j * a.pw;
    for (int e = threadIdx.x; e < on * npix; e += 256) {
        const int o = e / npix, pix = e - o * npix;
        const int u = pix / a.pw, v = pix - u * a.pw;
        dyl[e] = Store<T>::ld(dy, (((size_t)b * a.cout + o0 + o) * a.H + y0 + u) * a.W + x0 + v);
    }
    for (int e = threadIdx.x; e < nx * tpos; e += 256) {
        const int c = e / tpos, pos = e - c * tpos;
        const int u = pos / HW, v = pos - u * HW;
        const int yy = pad_index(y0 + u - a.pad, a.H, a.pad_mode), xx = pad_index(x0 + v - a.pad, a.W, a.pad_mode);
        xl[e] = (yy >= 0 && xx >= 0) ? Store<T>::ld(x, (((size_t)b * a.cin + c_lo + c) * a.H + yy) * a.W + xx) : 0.0f;
    }
    __syncthreads();
    const int kk = a.k * a.k, wrow = a.cin_g * kk;
    for (int idx = threadIdx.x; idx < on * wrow; idx += 256) {
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
        dbank[(size_t)patch * a.ld + (size_t)o * wrow + (idx - ol * wrow)] = acc;
    }
}

// the matrix-core / image-level forms of hs_patch_conv_bwd.hip (k = 1 / groups = 1, depthwise 3x3 with zero padding), both storage types
int try_fast_fwd(int dtype, const void* x, const void* bank, long ld, int batch, int c_in, int H, int W, int fh, int fw, int c_out,
                 int k, int pad, int pad_mode, int groups, const float* scale, const float* shift, int act, void* y, hipStream_t stream);
int try_fast_bwd_in(int dtype, const void* dy, const void* bank, long ld, int batch, int c_in, int H, int W, int fh, int fw,
                    int c_out, int k, int pad, int pad_mode, int groups, void* dx, hipStream_t stream);
int try_fast_bwd_w(int dtype, const void* x, const void* dy, int batch, int c_in, int H, int W, int fh, int fw, int c_out, int k,
                   int pad, int pad_mode, int groups, void* dbank, long ld, hipStream_t stream);

static int fill_plain(PlainArgs& a, int64_t ld, int32_t batch, int32_t c_in, int32_t H, int32_t W, int32_t fh, int32_t fw,
                      int32_t c_out, int32_t k, int32_t pad, int32_t pad_mode, int32_t groups) {
    if (batch <= 0 || c_in <= 0 || c_out <= 0 || H <= 0 || W <= 0 || fh <= 0 || fw <= 0 || k <= 0 || groups <= 0)
        return HS_ERR_BAD_ARG;
    if (2 * pad != k - 1 || pad > 3) return HS_ERR_UNSUPPORTED;
    if (H % fh != 0 || W % fw != 0) return HS_ERR_NOT_DIVISIBLE;
    if (c_in % groups != 0 || c_out % groups != 0) return HS_ERR_BAD_ARG;
    if (pad_mode < HS_PAD_ZEROS || pad_mode > HS_PAD_CIRCULAR) return HS_ERR_BAD_ARG;
    if (pad_mode == HS_PAD_REFLECT && (pad >= H || pad >= W)) return HS_ERR_BAD_ARG;
    a.ld = (long)ld;
    a.B = batch; a.H = H; a.W = W; a.fh = fh; a.fw = fw; a.ph = H / fh; a.pw = W / fw;
    a.cin = c_in; a.cout = c_out; a.k = k; a.pad = pad; a.pad_mode = pad_mode; a.groups = groups;
    a.cin_g = c_in / groups; a.cout_g = c_out / groups;
    if (ld < (int64_t)c_out * a.cin_g * k * k) return HS_ERR_BAD_ARG;
    return HS_OK;
}

}  // namespace hs

namespace hs {
int launch_conv1x1_bf16(const void* x, int batch, int cin, int H, int W, int fh, int fw, const float* bank, long ld, int c_out, void* y,
                        hipStream_t stream);                                                         // hs_patch_conv.hip
}
#ifndef HS_PLAIN_K1_BF16
#define HS_PLAIN_K1_BF16 1
#endif
using namespace hs;

#define HS_BY_DTYPE(dtype, CALL_F32, CALL_BF16) \
    if ((dtype) == HS_DTYPE_F32) { CALL_F32; } else if ((dtype) == HS_DTYPE_BF16) { CALL_BF16; } else return HS_ERR_BAD_ARG;

extern "C" int hs_patch_conv_plain_fwd(int32_t dtype, const void* x, const void* bank, int64_t ld, int32_t batch, int32_t c_in,
                                       int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad,
                                       int32_t pad_mode, int32_t groups, void* y, void* stream) {
    PlainArgs a{};
    int st = fill_plain(a, ld, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups);
    if (st != HS_OK) return st;
    if (!x || !bank || !y) return HS_ERR_BAD_ARG;
    {
        const int r = try_fast_fwd(dtype, x, bank, (long)ld, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups, nullptr, nullptr,
                                   HS_ACT_NONE, y, (hipStream_t)stream);
        if (r != 1) return r;
    }
    if (dtype == HS_DTYPE_BF16 && k == 1 && pad == 0 && groups == 1 && HS_PLAIN_K1_BF16) {     // tiny patches in bf16: the inference path's k = 1 form, typed (round 6)
        const int r = launch_conv1x1_bf16(x, batch, c_in, H, W, fh, fw, (const float*)bank, (long)ld, c_out, y, (hipStream_t)stream);
        if (r != 1) return r;
    }
    a.x = x; a.bank = bank; a.y = y;
    const int wrow = a.cin_g * k * k;
    a.w_stride = wrow | 1;
    const size_t budget = 96 * 1024;
    const size_t wbytes = (size_t)c_out * a.w_stride * sizeof(float);
    if (wbytes + (size_t)c_in * (1 + 2 * pad) * (1 + 2 * pad) * sizeof(float) > 150 * 1024) return HS_ERR_LDS;
    a.TW = a.pw > 64 ? 64 : a.pw;
    a.TH = a.ph > 64 ? 64 : a.ph;
    auto tile_bytes = [&](int th, int tw) { return wbytes + (size_t)c_in * (th + 2 * pad) * (tw + 2 * pad) * sizeof(float); };
    while (tile_bytes(a.TH, a.TW) > budget && (a.TH > 1 || a.TW > 1)) {
        if (a.TH >= a.TW && a.TH > 1) a.TH = (a.TH + 1) / 2; else a.TW = (a.TW + 1) / 2;
    }
    a.tiles_y = (a.ph + a.TH - 1) / a.TH;
    a.tiles_x = (a.pw + a.TW - 1) / a.TW;
    const size_t lds = tile_bytes(a.TH, a.TW);
    const long blocks = (long)batch * fh * fw * a.tiles_y * a.tiles_x;
    static std::atomic<unsigned long long> done32{0}, done16{0};
    if (lds > 64 * 1024) {
        const int e = dtype == HS_DTYPE_F32 ? allow_full_lds((const void*)plain_fwd_kernel<float>, done32)
                                            : allow_full_lds((const void*)plain_fwd_kernel<bf16_t>, done16);
        if (e != HS_OK) return e;
    }
    HS_BY_DTYPE(dtype,
        hipLaunchKernelGGL(plain_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a),
        hipLaunchKernelGGL(plain_fwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a))
    return launch_status();
}

extern "C" int hs_patch_conv_plain_bwd_in(int32_t dtype, const void* dy, const void* bank, int64_t ld, int32_t batch,
                                          int32_t c_in, int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t c_out, int32_t k,
                                          int32_t pad, int32_t pad_mode, int32_t groups, void* dx, void* stream) {
    PlainArgs a{};
    int st = fill_plain(a, ld, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups);
    if (st != HS_OK) return st;
    if (!dy || !bank || !dx) return HS_ERR_BAD_ARG;
    {
        const int r = try_fast_bwd_in(dtype, dy, bank, (long)ld, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups, dx,
                                      (hipStream_t)stream);
        if (r != 1) return r;
    }
    a.dy = dy; a.bank = bank; a.dx = dx;
    const size_t total = (size_t)batch * c_in * H * W;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 32768 ? 32768 : (total + 255) / 256);
    HS_BY_DTYPE(dtype,
        hipLaunchKernelGGL(plain_bwd_in_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a),
        hipLaunchKernelGGL(plain_bwd_in_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a))
    return launch_status();
}

extern "C" int hs_patch_conv_plain_bwd_w(int32_t dtype, const void* x, const void* dy, int32_t batch, int32_t c_in, int32_t H,
                                         int32_t W, int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad,
                                         int32_t pad_mode, int32_t groups, void* dbank, int64_t ld, void* stream) {
    PlainArgs a{};
    int st = fill_plain(a, ld, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups);
    if (st != HS_OK) return st;
    if (!x || !dy || !dbank) return HS_ERR_BAD_ARG;
    {
        const int r = try_fast_bwd_w(dtype, x, dy, batch, c_in, H, W, fh, fw, c_out, k, pad, pad_mode, groups, dbank, (long)ld,
                                     (hipStream_t)stream);
        if (r != 1) return r;
    }
    a.x = x; a.dy = dy; a.dbank = dbank;
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
    a.ob = ob;
    static std::atomic<unsigned long long> done32{0}, done16{0};
    if (lds > 64 * 1024) {
        const int e = dtype == HS_DTYPE_F32 ? allow_full_lds((const void*)plain_bwd_w_kernel<float>, done32)
                                            : allow_full_lds((const void*)plain_bwd_w_kernel<bf16_t>, done16);
        if (e != HS_OK) return e;
    }
    dim3 grid((unsigned)(batch * fh * fw), (unsigned)((c_out + ob - 1) / ob));
    HS_BY_DTYPE(dtype,
        hipLaunchKernelGGL(plain_bwd_w_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, a),
        hipLaunchKernelGGL(plain_bwd_w_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, a))
    return launch_status();
}

// MetaConv2d with the reference's FULL argument set (hyperseg/models/layers/meta_conv.py:141-186): per-sample weights,
// non-square kernels, stride, dilation, any padding amount and mode, groups.  The reference runs it as one grouped
// F.conv2d over the batch folded into the channels (after F.pad for the non-zero padding modes).
//
// No reference configuration instantiates a MetaConv2d outside "same" padding / stride 1 / dilation 1 (those go through
// hs_patch_conv_fwd's LDS-tiled kernels); this entry point exists so that the class is a drop-in for every argument the
// reference accepts.  Plain design: one workgroup = a strip of output pixels of one (sample, output channel); that
// channel's filter (cin/groups x kh x kw floats, shared by the whole strip) goes through LDS, the input is read through
// L1/L2 (neighbouring lanes read neighbouring pixels), one fmaf chain per output in (c, ky, kx) order.
#include "hs_common.h"

namespace hs {

struct MetaConvArgs {
    const float* __restrict__ x; const float* __restrict__ w; long ldw;
    const float* __restrict__ scale; const float* __restrict__ shift; int act;
    float* __restrict__ y;
    int Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, mode, groups, Ho, Wo;
};

__global__ __launch_bounds__(256)
void meta_conv_kernel(MetaConvArgs a) {
    extern __shared__ float flt[];                          // [cin/groups][kh][kw] of (b, o)
    const int o = blockIdx.y, b = blockIdx.z;
    const int cg = a.Cin / a.groups, nf = cg * a.kh * a.kw;
    const float* __restrict__ wrow = a.w + (size_t)b * a.ldw + (size_t)o * nf;
    for (int i = threadIdx.x; i < nf; i += 256) flt[i] = wrow[i];
    __syncthreads();
    const int g = o / (a.Cout / a.groups);
    const float* __restrict__ xb = a.x + ((size_t)b * a.Cin + (size_t)g * cg) * a.H * a.W;
    const float sc = a.scale ? a.scale[o] : 1.0f, sh = a.shift ? a.shift[o] : 0.0f;
    float* __restrict__ yo_ = a.y + ((size_t)b * a.Cout + o) * a.Ho * a.Wo;
    const int npix = a.Ho * a.Wo;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        const int yo = p / a.Wo, xo = p - yo * a.Wo;
        float acc = 0.0f;
        for (int c = 0; c < cg; ++c) {
            const float* __restrict__ xc = xb + (size_t)c * a.H * a.W;
            for (int ky = 0; ky < a.kh; ++ky) {
                const int iy = pad_index(yo * a.sh - a.ph + ky * a.dh, a.H, a.mode);
                for (int kx = 0; kx < a.kw; ++kx) {
                    const int ix = pad_index(xo * a.sw - a.pw + kx * a.dw, a.W, a.mode);
                    const float v = (iy >= 0 && ix >= 0) ? xc[(size_t)iy * a.W + ix] : 0.0f;
                    acc = fmaf(flt[(c * a.kh + ky) * a.kw + kx], v, acc);
                }
            }
        }
        yo_[p] = apply_act(a.scale || a.shift ? fmaf(acc, sc, sh) : acc, a.act);
    }
}

// ---- backward of the general form (round 4; zero padding -- the other modes pad explicitly with F.pad on the Python side, whose
// adjoint autograd already has).  Gather forms, no atomics, deterministic.
//   dX[b,c,iy,ix] = sum_{o in group(c)} sum_{ky,kx} W[b,o,c',ky,kx] dY[b,o,oy,ox]   where oy sh - pt + ky dh == iy, likewise x
//   dW[b,m]       = sum_{oy,ox} dY[b,o,oy,ox] X[b, g cin_g + c', oy sh - pt + ky dh, ox sw - pl + kx dw]     m = ((o cin_g + c') kh + ky) kw + kx
struct MetaConvBwdArgs {
    const float* __restrict__ x; const float* __restrict__ w; long ldw; const float* __restrict__ dy;
    float* __restrict__ dx; float* __restrict__ dwt; long lddw;
    int Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, groups, Ho, Wo;
};

__global__ __launch_bounds__(256)
void meta_conv_bwd_input_kernel(MetaConvBwdArgs a) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int cg = a.Cin / a.groups, og = a.Cout / a.groups, g = c / cg, cl = c - g * cg;
    const int nf = cg * a.kh * a.kw;
    const float* __restrict__ wb = a.w + (size_t)b * a.ldw;
    const float* __restrict__ dyb = a.dy + ((size_t)b * a.Cout + (size_t)g * og) * a.Ho * a.Wo;
    float* __restrict__ dxc = a.dx + ((size_t)b * a.Cin + c) * a.H * a.W;
    const int npix = a.H * a.W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        const int iy = p / a.W, ix = p - iy * a.W;
        float acc = 0.0f;
        for (int ky = 0; ky < a.kh; ++ky) {
            const int ty = iy + a.ph - ky * a.dh;
            if (ty < 0 || ty % a.sh != 0) continue;
            const int oy = ty / a.sh;
            if (oy >= a.Ho) continue;
            for (int kx = 0; kx < a.kw; ++kx) {
                const int tx = ix + a.pw - kx * a.dw;
                if (tx < 0 || tx % a.sw != 0) continue;
                const int ox = tx / a.sw;
                if (ox >= a.Wo) continue;
                for (int o = 0; o < og; ++o)
                    acc = fmaf(wb[(size_t)(g * og + o) * nf + (cl * a.kh + ky) * a.kw + kx], dyb[((size_t)o * a.Ho + oy) * a.Wo + ox], acc);
            }
        }
        dxc[p] = acc;
    }
}

__global__ __launch_bounds__(64)
void meta_conv_bwd_weight_kernel(MetaConvBwdArgs a) {
    const int m = blockIdx.x, b = blockIdx.y;              // one wave per weight of one sample
    const int cg = a.Cin / a.groups, og = a.Cout / a.groups;
    const int kx = m % a.kw, ky = (m / a.kw) % a.kh, cl = (m / (a.kw * a.kh)) % cg, o = m / (a.kw * a.kh * cg);
    const int g = o / og;
    const float* __restrict__ xc = a.x + ((size_t)b * a.Cin + (size_t)g * cg + cl) * a.H * a.W;
    const float* __restrict__ dyo = a.dy + ((size_t)b * a.Cout + o) * a.Ho * a.Wo;
    float acc = 0.0f;
    const int npix = a.Ho * a.Wo;
    for (int p = threadIdx.x; p < npix; p += 64) {
        const int oy = p / a.Wo, ox = p - oy * a.Wo;
        const int iy = oy * a.sh - a.ph + ky * a.dh, ix = ox * a.sw - a.pw + kx * a.dw;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) acc = fmaf(dyo[p], xc[(size_t)iy * a.W + ix], acc);
    }
    acc = wave_sum64(acc);
    if (threadIdx.x == 0) a.dwt[(size_t)b * a.lddw + m] = acc;
}

}  // namespace hs

using namespace hs;

extern "C" int hs_meta_conv_bwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W, const float* w, int64_t ldw,
                                int32_t c_out, int32_t kh, int32_t kw, int32_t stride_h, int32_t stride_w, int32_t pad_top,
                                int32_t pad_bottom, int32_t pad_left, int32_t pad_right, int32_t dil_h, int32_t dil_w, int32_t groups,
                                const float* dy, float* dx, float* dw, int64_t lddw, void* stream) {
    if (!x || !w || !dy || (!dx && !dw) || batch <= 0 || c_in <= 0 || c_out <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride_h <= 0 ||
        stride_w <= 0 || pad_top < 0 || pad_bottom < 0 || pad_left < 0 || pad_right < 0 || dil_h <= 0 || dil_w <= 0 || groups <= 0) return HS_ERR_BAD_ARG;
    if (c_in % groups != 0 || c_out % groups != 0) return HS_ERR_BAD_ARG;
    const long nf = (long)(c_in / groups) * kh * kw;
    if (ldw < nf * c_out || (dw && lddw < nf * c_out)) return HS_ERR_BAD_ARG;
    const int eh = H + pad_top + pad_bottom - dil_h * (kh - 1) - 1, ew = W + pad_left + pad_right - dil_w * (kw - 1) - 1;
    if (eh < 0 || ew < 0) return HS_ERR_BAD_ARG;
    if (batch > 65535 || c_in > 65535 || nf * c_out > 2147483647L) return HS_ERR_UNSUPPORTED;
    MetaConvBwdArgs a;
    a.x = x; a.w = w; a.ldw = ldw; a.dy = dy; a.dx = dx; a.dwt = dw; a.lddw = lddw;
    a.Cin = c_in; a.H = H; a.W = W; a.Cout = c_out; a.kh = kh; a.kw = kw; a.sh = stride_h; a.sw = stride_w;
    a.ph = pad_top; a.pw = pad_left; a.dh = dil_h; a.dw = dil_w; a.groups = groups; a.Ho = eh / stride_h + 1; a.Wo = ew / stride_w + 1;
    if (dx) {
        const long npix = (long)H * W;
        const unsigned strips = (unsigned)((npix + 255) / 256 > 1024 ? 1024 : (npix + 255) / 256);
        hipLaunchKernelGGL(meta_conv_bwd_input_kernel, dim3(strips, c_in, batch), dim3(256), 0, (hipStream_t)stream, a);
        const int st = launch_status();
        if (st != HS_OK) return st;
    }
    if (dw) {
        hipLaunchKernelGGL(meta_conv_bwd_weight_kernel, dim3((unsigned)(nf * c_out), batch), dim3(64), 0, (hipStream_t)stream, a);
        return launch_status();
    }
    return HS_OK;
}

extern "C" int hs_meta_conv_fwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W, const float* w, int64_t ldw,
                                int32_t c_out, int32_t kh, int32_t kw, int32_t stride_h, int32_t stride_w, int32_t pad_top,
                                int32_t pad_bottom, int32_t pad_left, int32_t pad_right, int32_t dil_h, int32_t dil_w,
                                int32_t pad_mode, int32_t groups,
                                const hs_epilogue* ep, float* y, void* stream) {
    if (!x || !w || !y || batch <= 0 || c_in <= 0 || c_out <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride_h <= 0 ||
        stride_w <= 0 || pad_top < 0 || pad_bottom < 0 || pad_left < 0 || pad_right < 0 || dil_h <= 0 || dil_w <= 0 || groups <= 0) return HS_ERR_BAD_ARG;
    if (c_in % groups != 0 || c_out % groups != 0) return HS_ERR_BAD_ARG;
    if (pad_mode < HS_PAD_ZEROS || pad_mode > HS_PAD_CIRCULAR) return HS_ERR_BAD_ARG;
    // F.pad's own preconditions: reflect needs pad < size, circular pad <= size
    const int pmax_h = pad_top > pad_bottom ? pad_top : pad_bottom, pmax_w = pad_left > pad_right ? pad_left : pad_right;
    if (pad_mode == HS_PAD_REFLECT && (pmax_h >= H || pmax_w >= W)) return HS_ERR_BAD_ARG;
    if (pad_mode == HS_PAD_CIRCULAR && (pmax_h > H || pmax_w > W)) return HS_ERR_BAD_ARG;
    const long nf = (long)(c_in / groups) * kh * kw;
    if (ldw < nf * c_out) return HS_ERR_BAD_ARG;
    const int eh = H + pad_top + pad_bottom - dil_h * (kh - 1) - 1, ew = W + pad_left + pad_right - dil_w * (kw - 1) - 1;
    if (eh < 0 || ew < 0) return HS_ERR_BAD_ARG;
    const int Ho = eh / stride_h + 1, Wo = ew / stride_w + 1;
    if (batch > 65535 || c_out > 65535) return HS_ERR_UNSUPPORTED;
    if (nf * (long)sizeof(float) > 64 * 1024) return HS_ERR_LDS;
    MetaConvArgs a;
    a.x = x; a.w = w; a.ldw = ldw; a.y = y;
    a.scale = ep ? ep->scale : nullptr; a.shift = ep ? ep->shift : nullptr; a.act = ep ? ep->act : HS_ACT_NONE;
    if (a.scale && !a.shift) return HS_ERR_BAD_ARG;
    a.Cin = c_in; a.H = H; a.W = W; a.Cout = c_out; a.kh = kh; a.kw = kw; a.sh = stride_h; a.sw = stride_w;
    a.ph = pad_top; a.pw = pad_left; a.dh = dil_h; a.dw = dil_w; a.mode = pad_mode; a.groups = groups; a.Ho = Ho; a.Wo = Wo;
    const long npix = (long)Ho * Wo;
    const unsigned strips = (unsigned)((npix + 255) / 256 > 1024 ? 1024 : (npix + 255) / 256);
    hipLaunchKernelGGL(meta_conv_kernel, dim3(strips, c_out, batch), dim3(256), (size_t)nf * sizeof(float), (hipStream_t)stream, a);
    return launch_status();
}

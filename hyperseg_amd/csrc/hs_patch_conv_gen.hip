// Op A with the filter bank generated INSIDE the consumer ("the bank never exists"):
//   y = act(bn(patch_conv_k1(stage_input, signal2weights(signal)[:, :hp])))
// -- HyperPatchNoPadding.forward + the BatchNorm / ReLU that follow it (hyperseg_v1_0.py:473-498, 728-760; unify:
// WeightLayer 287-309 + HyperPatchNoPadding 483-494) in ONE launch, for the coarse k = 1 levels whose patches are 1-16
// pixels.  There the bank (hp floats per patch: 21 KB at level 0) dwarfs the activations, so writing it to HBM
// (hs_signal2weights_multi_fwd) and reading it back (hs_patch_conv_fwd) IS the level: 2 x 18.4 MB of the decoder's
// traffic at HyperSeg-M, and two dependent launches of ~8 us each for < 1 % of its FLOPs.
//
// One workgroup = 16 consecutive patches x a block of OB output channels:
//   1. its rows [o0*cin, (o0+OB)*cin) of the bank for the 16 patches are produced as an f32-MFMA GEMM
//        bankT[row][patch] = sum_k Wsw_t[k][row] * S[patch][grp(row)*K + k]
//      (A = the transposed grouped-1x1 weight, 64-byte coalesced runs, L2-resident and shared by all patch tiles;
//       B = 16 patches x 4 signal channels) straight into LDS -- a 16-row tile that straddles a group boundary is two
//      MFMA passes with the rows of the other group zeroed;
//   2. the stage input cat(coords, skip, bilinear(prev)) of the 16 patches is generated into LDS;
//   3. thread = (output channel, pixel): dot product over cin with the bank column of the pixel's patch, folded
//      BatchNorm + activation, coalesced store.
// Per level that is one memory round trip (signal slice, weight rows, inputs) instead of bank-out / bank-in.
#include "hs_common.h"

namespace hs {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int GEN_THREADS = 256;
constexpr int GEN_PT = 16;                 // patches per workgroup (one MFMA N tile)
constexpr int GEN_LDB = GEN_PT + 1;        // bankT row stride (floats): odd -> the D-tile stores spread over the banks

struct ConvGenArgs {
    StageIn in;
    int fh, fw, ph, pw;
    const float* __restrict__ signal;
    int c_signal, signal_index, cs_g, rows_per_group, wc;
    const float* __restrict__ wsw_t;       // (cs_g, wc)
    int cin, cout, ob;                     // ob = output channels per workgroup
    const float* __restrict__ scale;
    const float* __restrict__ shift;
    int act;
    float* __restrict__ y;
    int n_patches;
    int pt;                                // patches per workgroup: 4, 8 or 16 (the MFMA N tile is 16 wide; columns >= pt idle)
    int lg_pw, lg_ppx, lg_nq;              // log2 of patch width, pixels per patch, pixels per workgroup (powers of two)
};

// Coalesced copy global -> LDS with NB loads per thread in flight (n elements, element e of the source is src(e)).
template <int NB, typename F>
__device__ __forceinline__ void stage_to_lds(float* __restrict__ dst, int n, int tid, F&& src) {
    for (int e0 = tid; e0 < n; e0 += GEN_THREADS * NB) {
        float v[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) v[q] = src(min(e0 + q * GEN_THREADS, n - 1));
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int e = e0 + q * GEN_THREADS;
            if (e < n) dst[e] = v[q];
        }
    }
}

__global__ __launch_bounds__(GEN_THREADS)
void patch_conv1x1_gen_kernel(ConvGenArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int ptab[GEN_PT][3];                    // (batch index, first row, first column) of the tile's patches
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lk = lane >> 4;
    const int PT = a.pt;
    const int p0 = blockIdx.x * PT;
    const int o0 = blockIdx.y * a.ob;
    const int on = min(a.ob, a.cout - o0);
    const int cin = a.cin;
    const int nrows = on * cin;                        // bank rows of this block: [n_lo, n_lo + nrows)
    const int n_lo = o0 * cin;
    const int nq = 1 << a.lg_nq;                       // pixels of the patch tile
    const int K = a.cs_g, KP = (K + 3) & ~3;           // signal channels per group, padded to whole k-steps
    const int g_lo = n_lo / a.rows_per_group, g_hi = (n_lo + nrows - 1) / a.rows_per_group;
    const int ng = g_hi - g_lo + 1;
    const int rows_p = (nrows + 15) & ~15;
    float* bankT = lds;                                // [ob*cin][GEN_LDB]
    float* xl = bankT + (size_t)a.ob * cin * GEN_LDB;  // [cin][nq]
    float* wl = xl + (size_t)cin * nq;                 // [KP][rows_p]   weight rows of the block, k-major
    float* sl = wl + (size_t)KP * (((size_t)a.ob * cin + 15) & ~(size_t)15);   // [ng][KP][16]  signal of the touched groups
    const int grid_sz = a.fh * a.fw;
    if (tid < GEN_PT) {
        const int p = min(p0 + min(tid, PT - 1), a.n_patches - 1);
        const int b = p / grid_sz, ij = p - b * grid_sz;
        const int i = ij / a.fw, j = ij - i * a.fw;
        ptab[tid][0] = b; ptab[tid][1] = i * a.ph; ptab[tid][2] = j * a.pw;
    }
    __syncthreads();

    // ---- every HBM / L2 load of the workgroup, 16 per thread in flight: weight rows, signal slice, stage input ----------
    stage_to_lds<16>(wl, KP * rows_p, tid, [&](int e) {
        const int k = e / rows_p, r = e - k * rows_p;
        return (k < K && r < nrows) ? a.wsw_t[(size_t)k * a.wc + n_lo + r] : 0.0f;
    });
    stage_to_lds<4>(sl, ng * KP * GEN_PT, tid, [&](int e) {
        const int pl = e & (GEN_PT - 1), kg = e >> 4;
        const int g = kg / KP, k = kg - g * KP;
        const int p = p0 + pl;
        const int pc = min(p, a.n_patches - 1);
        const int b = pc / grid_sz, ij = pc - b * grid_sz;
        return (k < K && pl < PT && p < a.n_patches)
                   ? a.signal[((size_t)b * a.c_signal + a.signal_index + (size_t)(g_lo + g) * K + k) * grid_sz + ij] : 0.0f;
    });
    stage_to_lds<8>(xl, cin << a.lg_nq, tid, [&](int e) {
        const int c = e >> a.lg_nq, qq = e & (nq - 1);
        const int pl = qq >> a.lg_ppx, pix = qq & ((1 << a.lg_ppx) - 1);
        const int u = pix >> a.lg_pw, vv = pix & ((1 << a.lg_pw) - 1);
        return stage_value(a.in, ptab[pl][0], c, stage_pos(a.in, ptab[pl][1] + u, ptab[pl][2] + vv));
    });
    __syncthreads();

    // ---- 1. bank rows of the block for the tile's patches: f32 MFMA, operands from LDS, D into LDS ---------------------
    const int ksteps = KP >> 2;
    const int ntiles = rows_p >> 4;
    for (int t = wave; t < ntiles; t += 4) {
        const int r0 = t * 16;                                     // first row of the tile inside the block
        const int g_first = (n_lo + r0) / a.rows_per_group;
        const int g_last = (n_lo + min(r0 + 15, nrows - 1)) / a.rows_per_group;
        const bool straddle = g_last != g_first;                   // uniform
        const int my_g = (n_lo + min(r0 + lrow, nrows - 1)) / a.rows_per_group;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int g = g_first; g <= g_last; ++g) {                  // 1 pass, or 2 when the tile straddles a group boundary
            const float* sg = sl + (size_t)(g - g_lo) * KP * GEN_PT;
            for (int ks = 0; ks < ksteps; ++ks) {
                float av = wl[(ks * 4 + lk) * rows_p + r0 + lrow];
                if (straddle && my_g != g) av = 0.0f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sg[(ks * 4 + lk) * GEN_PT + lrow], acc, 0, 0, 0);
            }
        }
        // D: lane holds rows 4*lk + r of patch lrow
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = r0 + 4 * lk + r;
            if (rl < nrows) bankT[rl * GEN_LDB + lrow] = acc[r];
        }
    }
    __syncthreads();

    // ---- 3. outputs ---------------------------------------------------------------------------------------------------
    const int total = on << a.lg_nq;
    for (int idx = tid; idx < total; idx += GEN_THREADS) {
        const int ol = idx >> a.lg_nq, q = idx & (nq - 1);
        const int pl = q >> a.lg_ppx, pix = q & ((1 << a.lg_ppx) - 1);
        const float* br = bankT + (size_t)ol * cin * GEN_LDB + pl;
        const float* xr = xl + q;
        float acc0 = 0.0f, acc1 = 0.0f;                // two chains: the dot product is latency-bound otherwise
        int c = 0;
        for (; c + 1 < cin; c += 2) {
            acc0 = fmaf(br[c * GEN_LDB], xr[c << a.lg_nq], acc0);
            acc1 = fmaf(br[(c + 1) * GEN_LDB], xr[(c + 1) << a.lg_nq], acc1);
        }
        if (c < cin) acc0 = fmaf(br[c * GEN_LDB], xr[c << a.lg_nq], acc0);
        float acc = acc0 + acc1;
        if (pl < PT && p0 + pl < a.n_patches) {
            const int o = o0 + ol;
            if (a.scale) acc = fmaf(acc, a.scale[o], a.shift[o]);
            acc = apply_act(acc, a.act);
            const int u = pix >> a.lg_pw, v = pix & ((1 << a.lg_pw) - 1);
            a.y[(((size_t)ptab[pl][0] * a.cout + o) * a.in.H + (ptab[pl][1] + u)) * a.in.W + (ptab[pl][2] + v)] = acc;
        }
    }
}

}  // namespace hs

using namespace hs;

extern "C" int hs_patch_conv_gen_fwd(const hs_stage_input* in, int32_t fh, int32_t fw, const float* signal,
                                     int32_t c_signal, const hs_s2w_layer* layer, int32_t c_out, const hs_epilogue* ep,
                                     float* y, void* stream) {
    ConvGenArgs a;
    int st = make_stage(in, &a.in);
    if (st != HS_OK) return st;
    if (!signal || !layer || !layer->wsw_t || !y || fh <= 0 || fw <= 0 || c_out <= 0 || c_signal <= 0) return HS_ERR_BAD_ARG;
    if (in->H % fh != 0 || in->W % fw != 0) return HS_ERR_NOT_DIVISIBLE;
    if (layer->groups <= 0 || layer->signal_channels % layer->groups != 0 || layer->wc % layer->groups != 0) return HS_ERR_BAD_ARG;
    if (layer->signal_index < 0 || layer->signal_index + layer->signal_channels > c_signal) return HS_ERR_BAD_ARG;
    a.fh = fh; a.fw = fw; a.ph = in->H / fh; a.pw = in->W / fw;
    a.cin = a.in.cin(); a.cout = c_out;
    if ((int64_t)a.cin * c_out > layer->rows || layer->rows > layer->wc) return HS_ERR_BAD_ARG;
    a.signal = signal; a.c_signal = c_signal; a.signal_index = layer->signal_index;
    a.cs_g = layer->signal_channels / layer->groups; a.rows_per_group = layer->wc / layer->groups; a.wc = layer->wc;
    a.wsw_t = layer->wsw_t;
    a.scale = ep ? ep->scale : nullptr; a.shift = ep ? ep->shift : nullptr; a.act = ep ? ep->act : HS_ACT_NONE;
    if (a.scale && !a.shift) return HS_ERR_BAD_ARG;
    a.y = y;
    a.n_patches = in->batch * fh * fw;
    const int ppx = a.ph * a.pw;
    if (ppx > 64) return HS_ERR_UNSUPPORTED;                   // the coarse levels only: larger patches amortise their bank
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    if (lg2(a.ph) < 0 || lg2(a.pw) < 0) return HS_ERR_UNSUPPORTED;      // index arithmetic by shifts
    // patches per workgroup: 16 for the 1..4-pixel patches, fewer for larger ones (the input tile is what costs there)
    a.pt = ppx <= 4 ? 16 : (ppx <= 8 ? 8 : 4);
    a.lg_pw = lg2(a.pw); a.lg_ppx = lg2(ppx); a.lg_nq = a.lg_ppx + lg2(a.pt);
    // output-channel block: as many channels as keep {bank tile, input tile, staged weight rows, signal slice} inside
    // 64 KB, and at least ~256 workgroups
    const size_t KP = ((size_t)a.cs_g + 3) & ~(size_t)3;
    const size_t xbytes = ((size_t)a.cin << a.lg_nq) * sizeof(float);
    auto need = [&](int ob) {
        const size_t rows = (size_t)ob * a.cin, rows_p = (rows + 15) & ~(size_t)15;
        const size_t ng = rows / a.rows_per_group + 2;
        return xbytes + (rows * GEN_LDB + KP * rows_p + ng * KP * GEN_PT) * sizeof(float);
    };
    if (need(1) > 150 * 1024) return HS_ERR_LDS;
    const int tiles = (a.n_patches + a.pt - 1) / a.pt;
    int ob = c_out;
    while (ob > 1 && need(ob) > 64 * 1024) --ob;
    while (ob > 1 && (long)tiles * ((c_out + ob - 1) / ob) < 256) ob = (ob + 1) / 2;
    a.ob = ob;
    const size_t lds = need(ob);
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};
        const int e = allow_full_lds((const void*)patch_conv1x1_gen_kernel, done);
        if (e != HS_OK) return e;
    }
    dim3 grid((unsigned)tiles, (unsigned)((c_out + ob - 1) / ob));
    hipLaunchKernelGGL(patch_conv1x1_gen_kernel, grid, dim3(GEN_THREADS), lds, (hipStream_t)stream, a);
    return launch_status();
}

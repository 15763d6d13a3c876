// Op A / Op B: dynamic patch-wise k x k convolution with a fused stage-input prologue
// (coords + skip + bilinear-resized previous level, image-level padding) and a fused
// BatchNorm-affine + activation epilogue.  One workgroup = one tile of one patch:
//   1. the patch's filter bank (contiguous ld floats in HBM) is staged in LDS once,
//   2. the input tile (with halo) is generated into LDS through the prologue,
//   3. each thread produces outputs (o, pixel) with pixel fastest, so LDS reads of the tile are
//      conflict-free and the weight reads are broadcasts; stores are row-contiguous.
// These levels (k=1 levels 0-2 of HyperSeg-M/S, and every v0_1 conv) are bound by the bank read:
// bytes/patch = hp*4, FLOPs/patch = 2*hp*ph*pw.
#include <type_traits>
#include "hs_common.h"
#include "hs_s2w_blocked.h"
#include <cstddef>

namespace hs {

struct ConvArgs {
    StageIn in;
    int fh, fw, ph, pw;
    const float* __restrict__ bank;
    long ld;
    int cout, k, pad, pad_mode, groups, cin_g, cout_g;
    const float* __restrict__ scale;
    const float* __restrict__ shift;
    int act;
    float* __restrict__ y;
    int TH, TW, tiles_y, tiles_x;   // output tile and tiles per patch
    int w_stride;                   // LDS row stride of the staged bank (odd => conflict-free)
};

constexpr int CONV_THREADS = 256;

__global__ __launch_bounds__(CONV_THREADS)
void patch_conv_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    const int tx_i = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty_i = blk % a.tiles_y; blk /= a.tiles_y;
    const int patch = blk;                       // (b*fh + i)*fw + j
    const int j = patch % a.fw, i = (patch / a.fw) % a.fh, b = patch / (a.fw * a.fh);

    const int cin = a.in.cin();
    const int kk = a.k * a.k;
    const int wrow = a.cin_g * kk;               // weights per output channel
    const int y0 = i * a.ph + ty_i * a.TH, x0 = j * a.pw + tx_i * a.TW;
    const int th = min(a.TH, (i + 1) * a.ph - y0), tw = min(a.TW, (j + 1) * a.pw - x0);
    const int HH = a.TH + 2 * a.pad, HW = a.TW + 2 * a.pad, tpos = HH * HW;

    float* wl = lds;                                       // [cout][w_stride]
    float* xl = lds + (size_t)a.cout * a.w_stride;         // [cin][tpos]

    // 1. filter bank -> LDS (coalesced; rows re-strided to an odd stride)
    const float* wp = a.bank + (size_t)patch * a.ld;
    const int hp = a.cout * wrow;
    for (int e = tid; e < hp; e += CONV_THREADS) {
        const int o = e / wrow, r = e - o * wrow;
        wl[o * a.w_stride + r] = wp[e];
    }
    // 2. input tile with halo through the fused prologue
    for (int pos = tid; pos < tpos; pos += CONV_THREADS) {
        const int u = pos / HW, v = pos - u * HW;
        const int yy = pad_index(y0 + u - a.pad, a.in.H, a.pad_mode);
        const int xx = pad_index(x0 + v - a.pad, a.in.W, a.pad_mode);
        const StagePos sp = stage_pos(a.in, yy, xx);
        for (int c = 0; c < cin; ++c) xl[c * tpos + pos] = stage_value(a.in, b, c, sp);
    }
    __syncthreads();

    // 3. outputs
    const int npix = a.TH * a.TW;
    const int total = a.cout * npix;
    for (int idx = tid; idx < total; idx += CONV_THREADS) {
        const int o = idx / npix, pix = idx - o * npix;
        const int u = pix / a.TW, v = pix - u * a.TW;
        if (u >= th || v >= tw) continue;
        const int g = o / a.cout_g;
        const float* wr = wl + o * a.w_stride;
        const float* xr = xl + (size_t)g * a.cin_g * tpos + u * HW + v;
        float acc = 0.0f;
        if (a.k == 1) {
            for (int c = 0; c < a.cin_g; ++c) acc = fmaf(wr[c], xr[c * tpos], acc);
        } else {
            for (int c = 0; c < a.cin_g; ++c)
                for (int ky = 0; ky < a.k; ++ky)
                    for (int kx = 0; kx < a.k; ++kx)
                        acc = fmaf(wr[(c * a.k + ky) * a.k + kx], xr[c * tpos + ky * HW + kx], acc);
        }
        if (a.scale) acc = fmaf(acc, a.scale[o], a.shift[o]);
        acc = apply_act(acc, a.act);
        a.y[(((size_t)b * a.cout + o) * a.in.H + (y0 + u)) * a.in.W + (x0 + v)] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// Op A fast path (k = 1, whole patch per workgroup): the levels where a patch is 1..16 pixels and the
// launch is nothing but "read the bank once".  The bank is copied to LDS with 16-byte loads (all of a
// thread's loads in flight together), the (c, pixel) elements of the stage input are spread over all
// 256 threads, and when a patch has fewer than 256 outputs the dot products are split SPLIT ways across
// adjacent lanes and reduced with DPP shuffles, so level 0 (64 outputs of length 82) still uses 4 waves.
// ------------------------------------------------------------------------------------------
struct Conv1Args {
    StageIn in;
    int fh, fw, ph, pw;
    const float* __restrict__ bank;
    long ld;
    int cout, groups, cin_g, cout_g;
    const float* __restrict__ scale;
    const float* __restrict__ shift;
    int act;
    float* __restrict__ y;
    int split;        // power of two <= 64: lanes cooperating on one output
};

// PWL >= 0: square patches of edge 2^PWL (the decoder's levels 0-2 are 1, 2 and 4 pixels; the launch's split is then a compile-time
// value too): the index arithmetic -- e -> (channel, pixel), pixel -> (row, column), output -> (o, pixel) -- is shifts and masks
// instead of run-time integer divisions, which were a third of this kernel's ~970 vector instructions per wave
// (profiles/round3_pmc_k1_and_s2w.txt).  PWL = -1: any patch size.
// TS (round 6): storage type of the input (given as the stage's `skip`) and of the output: float everywhere but the bf16 training step's k = 1
// levels (launch_conv1x1_bf16: no coordinates, no previous level, no epilogue), whose generic kernel was 6 us per launch slower than this one.
template <int PWL, typename TS = float>
__device__ __forceinline__ void conv1x1_body(const Conv1Args& a, const int patch, float* __restrict__ lds) {
    const int ph_ = PWL >= 0 ? (1 << PWL) : a.ph, pw_ = PWL >= 0 ? (1 << PWL) : a.pw;
    const int tid = threadIdx.x;
    const int pib = patch / a.fw, j = patch - pib * a.fw, b = pib / a.fh, i = pib - b * a.fh;
    const int cin = a.in.cin();
    const int npix = ph_ * pw_;
    const int hp = a.cout * a.cin_g;
    const int hp4 = (hp + 3) & ~3;
    float* wl = lds;                 // [cout][cin_g], natural order
    float* xl = lds + hp4;           // [cin][npix]

    // 1+2. all HBM loads of the workgroup in flight together: the bank (16-byte loads, up to 8 per thread) AND the
    //      stage-input elements (c, pixel); only then the LDS stores.  One memory round trip per workgroup.
    // (a clang vector type, not HIP's float4 struct: struct copies are memcpys, and an array of them stays in scratch memory
    // once a scheduling barrier sits between the copy in and the copy out)
    typedef float bank4 __attribute__((ext_vector_type(4)));
    const bank4* __restrict__ src = reinterpret_cast<const bank4*>(a.bank + (size_t)patch * a.ld);
    bank4* dst = reinterpret_cast<bank4*>(wl);
    const int n4 = hp4 >> 2;                          // ld is a multiple of 4 and >= hp: the tail read stays in-row
    const int y0 = i * ph_, x0 = j * pw_;
    const int total_x = cin * npix;
    // Stage-input elements without a branch around any load: stage_value() picks coordinate / skip / previous-level by
    // channel, lanes of one wave differ in that, and every arm's loads were waited for where the arms join (6 serialised
    // round trips, only 4 of the 38 loads in flight at the first wait; tools/isa_phases.py).  Here the skip sample and the
    // previous level's taps are BOTH requested from clamped channels (wave-uniform branches only), the unused one is
    // multiplied by 0 and the value is their sum -- same operations in the same order as stage_value() otherwise.
    float xv[4];
    bank4 wv[8];
    float sc_first = 1.0f, sh_first = 0.0f;
    {
        const StageIn& s = a.in;
        // pass 1: indices only
        int cc[4], yy[4], xx[4];
        float cv[4], ms[4], mp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = min(tid + q * CONV_THREADS, total_x - 1);
            int c = e / npix;
            const int pix = e - c * npix;
            const int u = pix / pw_, vv = pix - u * pw_;
            yy[q] = y0 + u; xx[q] = x0 + vv;
            cv[q] = 0.0f;
            bool is_coord = false;
            if (s.coords) {                                             // uniform
                is_coord = c < 2;
                cv[q] = c == 0 ? linspace_pm1(xx[q], s.W, s.step_x) : (c == 1 ? linspace_pm1(yy[q], s.H, s.step_y) : 0.0f);
                c -= 2;
            }
            ms[q] = (!is_coord && c < s.c_skip) ? 1.0f : 0.0f;
            mp[q] = (!is_coord && c >= s.c_skip) ? 1.0f : 0.0f;
            cc[q] = c;
        }
        // pass 2: the loads, four per wave-uniform arm, nothing used inside an arm
        float sk[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pt[4][4];
        float ly0[4], ly1[4], lx0[4], lx1[4];       // bilinear weights (plain arrays: a select between Tap structs went
                                                    // through scratch memory)
#pragma unroll
        for (int q = 0; q < 4; ++q) pt[q][0] = pt[q][1] = pt[q][2] = pt[q][3] = 0.0f;
        if (s.c_skip > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sk[q] = Store<TS>::ld(reinterpret_cast<const TS*>(s.skip), (((size_t)b * s.c_skip + min(max(cc[q], 0), s.c_skip - 1)) * s.H + yy[q]) * s.W + xx[q]);
        }
        // a same-resolution previous level is the bilinear form with both taps on the pixel and weights (1, 0): one arm
        const bool bilinear = s.prev_mode != HS_PREV_SAME;
        if (s.c_prev > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cp = min(max(cc[q] - s.c_skip, 0), s.c_prev - 1);
                const float* base = s.prev + ((size_t)b * s.c_prev + cp) * s.Hp * s.Wp;
                const Tap by = bilinear_tap(yy[q], s.scale_y, s.Hp), bx = bilinear_tap(xx[q], s.scale_x, s.Wp);
                const int y0i = bilinear ? by.i0 : yy[q], y1i = bilinear ? by.i1 : yy[q];
                const int x0i = bilinear ? bx.i0 : xx[q], x1i = bilinear ? bx.i1 : xx[q];
                ly0[q] = bilinear ? by.l0 : 1.0f; ly1[q] = bilinear ? by.l1 : 0.0f;
                lx0[q] = bilinear ? bx.l0 : 1.0f; lx1[q] = bilinear ? bx.l1 : 0.0f;
                const float* r0 = base + (size_t)y0i * s.Wp;
                const float* r1 = base + (size_t)y1i * s.Wp;
                pt[q][0] = r0[x0i]; pt[q][1] = r0[x1i]; pt[q][2] = r1[x0i]; pt[q][3] = r1[x1i];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) { ly0[q] = ly1[q] = lx0[q] = lx1[q] = 0.0f; }
        }
        // BN rows of this thread's first output (the only one at the decoder's shapes), with everything else
        if (a.scale) {
            const int o_first = min((tid / a.split) / npix, a.cout - 1);
            sc_first = a.scale[o_first]; sh_first = a.shift[o_first];
        }
        // the bank, requested behind the taps (32 registers: kept out of the index arithmetic above)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + q * CONV_THREADS;
            wv[q] = src[e < n4 ? e : n4 - 1];
        }
        __builtin_amdgcn_sched_barrier(0);      // every load above is issued before the first use below
        // pass 3: masks and the value
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float top = lx0[q] * (pt[q][0] * mp[q]) + lx1[q] * (pt[q][1] * mp[q]);
            const float bot = lx0[q] * (pt[q][2] * mp[q]) + lx1[q] * (pt[q][3] * mp[q]);
            const float pv = ly0[q] * top + ly1[q] * bot;
            xv[q] = (sk[q] * ms[q] + pv) + cv[q];
        }
    }
    // Unconditional LDS stores to the same clamped indices (surplus lanes rewrite the last element with the value they
    // loaded from it): behind an `if (e < n4)` the compiler SINKS each load into its store's branch and waits for it there --
    // five serialised round trips for the bank (tools/isa_phases.py) instead of the single one this kernel is built around.
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[min(tid + q * CONV_THREADS, n4 - 1)] = wv[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) xl[min(tid + q * CONV_THREADS, total_x - 1)] = xv[q];
    // remainders of very large banks / tiles (not reached by the decoder's own shapes)
    for (int e = tid + 8 * CONV_THREADS; e < n4; e += CONV_THREADS) dst[e] = src[e];
    for (int e = tid + 4 * CONV_THREADS; e < total_x; e += CONV_THREADS) {
        const int c = e / npix, pix = e - c * npix;
        const int u = pix / pw_, vv = pix - u * pw_;
        if constexpr (std::is_same<TS, float>::value) xl[e] = stage_value(a.in, b, c, stage_pos(a.in, y0 + u, x0 + vv));
        else xl[e] = Store<TS>::ld(reinterpret_cast<const TS*>(a.in.skip), (((size_t)b * a.in.c_skip + c) * a.in.H + (y0 + u)) * a.in.W + (x0 + vv));     // (typed form: skip only)
    }
    __syncthreads();

    // 3. outputs: SPLIT adjacent lanes share one (o, pixel)
    const int split = a.split;
    const int part = tid & (split - 1);
    const int per_pass = CONV_THREADS / split;
    const int total = a.cout * npix;
    for (int base = 0; base < total; base += per_pass) {
        const int idx = base + tid / split;
        const bool live = idx < total;
        const int o = live ? idx / npix : 0, pix = live ? idx - o * npix : 0;
        const int g = o / a.cout_g;
        const float* wr = wl + o * a.cin_g;
        const float* xr = xl + (size_t)g * a.cin_g * npix + pix;
        // four independent partial sums: the LDS reads of four steps are in flight together (a single dependent chain paid the
        // LDS latency once per channel: ~1.5 us of the launch at level 0); lanes of a split group then combine on the DPP path
        float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
        int c = part;
        for (; c + 3 * split < a.cin_g; c += 4 * split) {
            const float w0 = wr[c], w1 = wr[c + split], w2 = wr[c + 2 * split], w3 = wr[c + 3 * split];
            const float x0 = xr[c * npix], x1 = xr[(c + split) * npix], x2 = xr[(c + 2 * split) * npix], x3 = xr[(c + 3 * split) * npix];
            acc0 = fmaf(w0, x0, acc0); acc1 = fmaf(w1, x1, acc1); acc2 = fmaf(w2, x2, acc2); acc3 = fmaf(w3, x3, acc3);
        }
        for (; c < a.cin_g; c += split) acc0 = fmaf(wr[c], xr[c * npix], acc0);
        float acc = (acc0 + acc1) + (acc2 + acc3);
        if (split >= 2) acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xf, 0xf, false));   // lane ^ 1
        if (split >= 4) acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E, 0xf, 0xf, false));   // lane ^ 2
        for (int m = split >> 1; m >= 4; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (live && part == 0) {
            if (a.scale) {
                if (base == 0) acc = fmaf(acc, sc_first, sh_first);       // requested with the bank: no round trip here
                else acc = fmaf(acc, a.scale[o], a.shift[o]);
            }
            acc = apply_act(acc, a.act);
            const int u = pix / pw_, v = pix - u * pw_;
            Store<TS>::st(reinterpret_cast<TS*>(a.y), (((size_t)b * a.cout + o) * a.in.H + (i * ph_ + u)) * a.in.W + (j * pw_ + v), acc);
        }
    }
}

template <int PWL, typename TS = float>
__global__ __launch_bounds__(CONV_THREADS)
void patch_conv1x1_kernel(Conv1Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    conv1x1_body<PWL, TS>(a, (int)blockIdx.x, lds);
}

// Heterogeneous launch (round 4): workgroups [0, conv_blocks) are the k = 1 patch convolution's, the rest are blocked
// signal2weights workgroups producing the banks of LATER levels.  The k = 1 levels are one workgroup per patch walking a
// dependent chain (index arithmetic -> loads -> LDS -> barrier -> dot products -> store: ~6 us for ~2.5 us of traffic, the
// chip mostly idle); signal2weights only depends on the signal, so its blocks fill that idle time instead of standing in
// front of level 0 as their own 15 us launch.  (The same overlap as a forked HIP graph costs +100 us per frame on ROCm 7.2:
// profiles/round4_irc_prologue_vs_round3_same_box_and_side_stream_ab.txt.)
struct Conv1S2wArgs {
    Conv1Args c;
    int conv_blocks;
    S2bArgs s;
};

template <int PWL>
__global__ __launch_bounds__(CONV_THREADS)
void patch_conv1x1_s2w_kernel(Conv1S2wArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int blk = (int)blockIdx.x;
    if (blk < a.conv_blocks) {
        conv1x1_body<PWL>(a.c, blk, lds);
    } else {
        const __attribute__((address_space(4))) char* ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
        s2b_body((const __attribute__((address_space(4))) S2bArgs*)(ka + offsetof(Conv1S2wArgs, s)), blk - a.conv_blocks, lds);
    }
}

// Exact 2x bilinear upsample (align_corners=False): taps are {0.25, 0.75} with edge clamping.  One thread =
// 2 output rows x 4 output columns from a 3 x 4 input neighbourhood: two 16-byte stores per 12 cached loads.
// up2x_block is shared by the logits kernel and the fused argmax kernel so that both round identically.
__device__ __forceinline__ void up2x_block(const float* __restrict__ base, int Hi, int Wi, int yi, int q,
                                           float (&o0)[4], float (&o1)[4]) {
    const int xi = 2 * q;
    const int xm = xi > 0 ? xi - 1 : 0, xp = xi + 2 < Wi ? xi + 2 : Wi - 1;
    const int ym = yi > 0 ? yi - 1 : 0, yp = yi + 1 < Hi ? yi + 1 : Hi - 1;
    float in[3][4];
    const int ys[3] = {ym, yi, yp};
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const float* row = base + (size_t)ys[rr] * Wi;
        in[rr][0] = row[xm]; in[rr][1] = row[xi]; in[rr][2] = row[xi + 1]; in[rr][3] = row[xp];
    }
    // horizontal pass, same operation order as ATen: l0*a + l1*b with (l0, l1) = (0.25, 0.75) / (0.75, 0.25)
    float hz[3][4];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        hz[rr][0] = 0.25f * in[rr][0] + 0.75f * in[rr][1];
        hz[rr][1] = 0.75f * in[rr][1] + 0.25f * in[rr][2];
        hz[rr][2] = 0.25f * in[rr][1] + 0.75f * in[rr][2];
        hz[rr][3] = 0.75f * in[rr][2] + 0.25f * in[rr][3];
    }
    // ATen clamps the SOURCE index at 0 (lambda = 0 there): first output row/col equal the edge sample
    if (xi == 0) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) hz[rr][0] = 1.0f * in[rr][1] + 0.0f * in[rr][2];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        o0[c] = (yi == 0) ? (1.0f * hz[1][c] + 0.0f * hz[2][c]) : (0.25f * hz[0][c] + 0.75f * hz[1][c]);
        o1[c] = 0.75f * hz[1][c] + 0.25f * hz[2][c];
    }
}

__global__ __launch_bounds__(256)
void upsample2x_kernel(const float* __restrict__ x, int planes, int Hi, int Wi, float* __restrict__ y) {
    const int wq = Wi >> 1;                 // pairs of input columns
    const size_t n = (size_t)planes * Hi * wq;
    const int Wo = 2 * Wi;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int q = e % wq; size_t r = e / wq;
        const int yi = r % Hi; const size_t pl = r / Hi;
        float o0[4], o1[4];
        up2x_block(x + pl * Hi * Wi, Hi, Wi, yi, q, o0, o1);
        float* dst = y + (pl * 2 * Hi + 2 * yi) * Wo + 4 * q;
        // streaming stores: the full-resolution logits (39.8 MB at HyperSeg-M) are written once and not re-read by this
        // frame -- keep them from evicting the next frame's weights and features from L2 / MALL
        typedef float v4f __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4f{o0[0], o0[1], o0[2], o0[3]}, reinterpret_cast<v4f*>(dst));
        __builtin_nontemporal_store(v4f{o1[0], o1[1], o1[2], o1[3]}, reinterpret_cast<v4f*>(dst + Wo));
    }
}

// The same upsample with the class argmax taken in registers: uint8 masks instead of logits (M: 0.5 MB written instead
// of 39.8 MB, and no separate argmax pass re-reading them).  Ties resolve to the lowest class index.  Replaces
// F.interpolate + pred.argmax(1) (hyperseg_v1_0.py:250-251 + test.py:171 / test_fps.py:194).
// Four consecutive lanes share one 2x4 output block and split the classes among them (c = sub, sub + 4, ...): with one
// thread per block the launch is a single wave per SIMD walking 19 dependent load batches; this way it is four waves per
// SIMD with <= 5 classes (60 loads, one batch) each, combined with two shuffles (larger value wins, lower class on ties).
__global__ __launch_bounds__(256)
void upsample2x_argmax_kernel(const float* __restrict__ x, int B, int C, int Hi, int Wi, uint8_t* __restrict__ mask) {
    const int wq = Wi >> 1;
    const size_t n = (size_t)B * Hi * wq;
    const int Wo = 2 * Wi;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = (int)(t & 3);
    const size_t e0 = t >> 2;
    const size_t e = e0 < n ? e0 : n - 1;                    // surplus lanes shadow the last block (shuffles stay convergent)
    const int q = e % wq; size_t r = e / wq;
    const int yi = r % Hi; const size_t b = r / Hi;
    const float* __restrict__ xb = x + b * C * Hi * Wi;
    constexpr float NEG = -3.402823466e38f;
    float best0[4] = {NEG, NEG, NEG, NEG}, best1[4] = {NEG, NEG, NEG, NEG};
    int idx0[4] = {sub, sub, sub, sub}, idx1[4] = {sub, sub, sub, sub};
    for (int c0 = sub; c0 < C; c0 += 20) {
        float o0[5][4], o1[5][4];
#pragma unroll
        for (int u = 0; u < 5; ++u) {                        // 5 classes = 60 loads in flight
            const int c = min(c0 + 4 * u, C - 1);
            up2x_block(xb + (size_t)c * Hi * Wi, Hi, Wi, yi, q, o0[u], o1[u]);
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int c = c0 + 4 * u;
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (o0[u][i] > best0[i]) { best0[i] = o0[u][i]; idx0[i] = c; }
                    if (o1[u][i] > best1[i]) { best1[i] = o1[u][i]; idx1[i] = c; }
                }
            }
        }
    }
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v0 = __shfl_xor(best0[i], m, 64), v1 = __shfl_xor(best1[i], m, 64);
            const int j0 = __shfl_xor(idx0[i], m, 64), j1 = __shfl_xor(idx1[i], m, 64);
            if (v0 > best0[i] || (v0 == best0[i] && j0 < idx0[i])) { best0[i] = v0; idx0[i] = j0; }
            if (v1 > best1[i] || (v1 == best1[i] && j1 < idx1[i])) { best1[i] = v1; idx1[i] = j1; }
        }
    }
    if (sub == 0 && e0 < n) {
        uint8_t* dst = mask + (b * 2 * Hi + 2 * yi) * Wo + 4 * q;
        *reinterpret_cast<uchar4*>(dst) = make_uchar4(idx0[0], idx0[1], idx0[2], idx0[3]);
        *reinterpret_cast<uchar4*>(dst + Wo) = make_uchar4(idx1[0], idx1[1], idx1[2], idx1[3]);
    }
}

// ------------------------------------------------------------------------------------------
// TP / TO: storage of the previous level / of the result (fp32; bf16 on the training path under autocast: hs_stage_input_typed_fwd)
template <typename TP, typename TO>
__global__ void stage_input_kernel(StageIn s, TO* __restrict__ y) {
    const size_t n = (size_t)s.B * s.cin() * s.H * s.W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int x = e % s.W; size_t r = e / s.W;
        const int yy = r % s.H; r /= s.H;
        const int c = r % s.cin(); const int b = r / s.cin();
        Store<TO>::st(y, e, stage_value<TP>(s, b, c, stage_pos(s, yy, x)));
    }
}
// the same on a (64 x 4 pixels, plane) grid: no per-element division (three of them were most of the element-wise form's instructions;
// the training step launches it once per level: 5 x 8.4 -> us at config 5)
template <typename TP, typename TO>
__global__ __launch_bounds__(256)
void stage_input_plane_kernel(StageIn s, TO* __restrict__ y) {
    // four rows per thread (yy0, + 4, + 8, + 12), their taps requested before the first store (round 6: one element per thread ran at
    // 1.9 TB/s on config 5's level 4; two PLANES per thread on top -- the sampling positions shared -- measured slower on the small
    // levels, +1.7 us each, and no faster on the large one: visit x18)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), yy0 = blockIdx.y * 16 + (threadIdx.x >> 6);
    if (x >= s.W || yy0 >= s.H) return;
    const int plane = blockIdx.z, cin = s.cin();
    const int b = plane / cin, c = plane - b * cin;                       // (uniform)
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = stage_value<TP>(s, b, c, stage_pos(s, min(yy0 + 4 * r, s.H - 1), x));
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (yy0 + 4 * r < s.H) Store<TO>::st(y, ((size_t)plane * s.H + yy0 + 4 * r) * s.W + x, v[r]);
}

// Bilinear resize (align_corners=False).  One thread = 4 consecutive output pixels of a row; bilinear_row4 is shared by
// the logits kernel and the fused argmax kernel.
struct Row4 { Tap ty; Tap tx[4]; };
__device__ __forceinline__ Row4 row4_taps(int yo, int q, int Hi, int Wi, int Wo, float scale_y, float scale_x) {
    Row4 t;
    t.ty = bilinear_tap(yo, scale_y, Hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int xo = 4 * q + i;
        t.tx[i] = bilinear_tap(xo < Wo ? xo : Wo - 1, scale_x, Wi);
    }
    return t;
}
__device__ __forceinline__ void bilinear_row4(const float* __restrict__ plane, int Wi, const Row4& t, float (&out)[4]) {
    const float* r0 = plane + (size_t)t.ty.i0 * Wi;
    const float* r1 = plane + (size_t)t.ty.i1 * Wi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float top = t.tx[i].l0 * r0[t.tx[i].i0] + t.tx[i].l1 * r0[t.tx[i].i1];
        const float bot = t.tx[i].l0 * r1[t.tx[i].i0] + t.tx[i].l1 * r1[t.tx[i].i1];
        out[i] = t.ty.l0 * top + t.ty.l1 * bot;
    }
}

__global__ __launch_bounds__(256)
void upsample_bilinear_kernel(const float* __restrict__ x, int planes, int Hi, int Wi, int Ho, int Wo,
                              float scale_y, float scale_x, float* __restrict__ y) {
    const int wq = (Wo + 3) / 4;
    const size_t n = (size_t)planes * Ho * wq;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int q = e % wq; size_t r = e / wq;
        const int yo = r % Ho; const size_t pl = r / Ho;
        const Row4 t = row4_taps(yo, q, Hi, Wi, Wo, scale_y, scale_x);
        float out[4];
        bilinear_row4(x + pl * Hi * Wi, Wi, t, out);
        float* dst = y + (pl * Ho + yo) * Wo + 4 * q;
        if ((Wo & 3) == 0) {
            *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
            for (int i = 0; i < 4 && 4 * q + i < Wo; ++i) dst[i] = out[i];
        }
    }
}

// bf16 storage (the training path's final logits under autocast): the same taps and f32 arithmetic, one rounding on store
// The exact-2x case in bf16 storage (round 6: the training step's last launch under autocast -- the general kernel below, four 2-byte loads and a
// 2-byte store per output, took 21.9 us where the fp32 step's upsample2x_kernel takes 9.9): upsample2x_kernel's block of 2 x 4 outputs per
// thread from a 3 x 4 input neighbourhood, f32 arithmetic in the same operation order, each output row leaving as ONE 8-byte store.
__device__ __forceinline__ void up2x_block_bf16(const bf16_t* __restrict__ base, int Hi, int Wi, int yi, int q, float (&o0)[4], float (&o1)[4]) {
    const int xi = 2 * q;
    const int xm = xi > 0 ? xi - 1 : 0, xp = xi + 2 < Wi ? xi + 2 : Wi - 1;
    const int ym = yi > 0 ? yi - 1 : 0, yp = yi + 1 < Hi ? yi + 1 : Hi - 1;
    float in[3][4];
    const int ys[3] = {ym, yi, yp};
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const bf16_t* row = base + (size_t)ys[rr] * Wi;
        in[rr][0] = Store<bf16_t>::ld(row, xm); in[rr][1] = Store<bf16_t>::ld(row, xi);
        in[rr][2] = Store<bf16_t>::ld(row, xi + 1); in[rr][3] = Store<bf16_t>::ld(row, xp);
    }
    float hz[3][4];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        hz[rr][0] = 0.25f * in[rr][0] + 0.75f * in[rr][1];
        hz[rr][1] = 0.75f * in[rr][1] + 0.25f * in[rr][2];
        hz[rr][2] = 0.25f * in[rr][1] + 0.75f * in[rr][2];
        hz[rr][3] = 0.75f * in[rr][2] + 0.25f * in[rr][3];
    }
    if (xi == 0) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) hz[rr][0] = 1.0f * in[rr][1] + 0.0f * in[rr][2];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        o0[c] = (yi == 0) ? (1.0f * hz[1][c] + 0.0f * hz[2][c]) : (0.25f * hz[0][c] + 0.75f * hz[1][c]);
        o1[c] = 0.75f * hz[1][c] + 0.25f * hz[2][c];
    }
}
__device__ __forceinline__ void store4_bf16(bf16_t* __restrict__ dst, const float (&o)[4]) {      // dst 8-byte aligned
    bf16_t t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Store<bf16_t>::st(t, i, o[i]);
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    *reinterpret_cast<v2u*>(dst) = v2u{(unsigned)t[0].v | ((unsigned)t[1].v << 16), (unsigned)t[2].v | ((unsigned)t[3].v << 16)};
}
__global__ __launch_bounds__(256)
void upsample2x_bf16_kernel(const bf16_t* __restrict__ x, int planes, int Hi, int Wi, bf16_t* __restrict__ y) {
    const int wq = Wi >> 1;                 // pairs of input columns
    const size_t n = (size_t)planes * Hi * wq;
    const int Wo = 2 * Wi;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int q = e % wq; size_t r = e / wq;
        const int yi = r % Hi; const size_t pl = r / Hi;
        float o0[4], o1[4];
        up2x_block_bf16(x + pl * Hi * Wi, Hi, Wi, yi, q, o0, o1);
        bf16_t* dst = y + (pl * 2 * Hi + 2 * yi) * Wo + 4 * q;
        store4_bf16(dst, o0);
        store4_bf16(dst + Wo, o1);
    }
}

__global__ __launch_bounds__(256)
void upsample_bilinear_bf16_kernel(const bf16_t* __restrict__ x, int planes, int Hi, int Wi, int Ho, int Wo,
                                   float scale_y, float scale_x, bf16_t* __restrict__ y) {
    const int wq = (Wo + 3) / 4;
    const size_t n = (size_t)planes * Ho * wq;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int q = e % wq; size_t r = e / wq;
        const int yo = r % Ho; const size_t pl = r / Ho;
        const Row4 t = row4_taps(yo, q, Hi, Wi, Wo, scale_y, scale_x);
        const bf16_t* r0 = x + pl * Hi * Wi + (size_t)t.ty.i0 * Wi;
        const bf16_t* r1 = x + pl * Hi * Wi + (size_t)t.ty.i1 * Wi;
        bf16_t* dst = y + (pl * Ho + yo) * Wo + 4 * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float top = t.tx[i].l0 * Store<bf16_t>::ld(r0, t.tx[i].i0) + t.tx[i].l1 * Store<bf16_t>::ld(r0, t.tx[i].i1);
            const float bot = t.tx[i].l0 * Store<bf16_t>::ld(r1, t.tx[i].i0) + t.tx[i].l1 * Store<bf16_t>::ld(r1, t.tx[i].i1);
            if (4 * q + i < Wo) Store<bf16_t>::st(dst, i, t.ty.l0 * top + t.ty.l1 * bot);
        }
    }
}

__global__ __launch_bounds__(256)
void upsample_argmax_kernel(const float* __restrict__ x, int B, int C, int Hi, int Wi, int Ho, int Wo, float scale_y,
                            float scale_x, uint8_t* __restrict__ mask) {
    const int wq = (Wo + 3) / 4;
    const size_t n = (size_t)B * Ho * wq;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int q = e % wq; size_t r = e / wq;
    const int yo = r % Ho; const size_t b = r / Ho;
    const Row4 t = row4_taps(yo, q, Hi, Wi, Wo, scale_y, scale_x);
    const float* __restrict__ xb = x + b * C * Hi * Wi;
    float best[4];
    int idx[4] = {0, 0, 0, 0};
    bilinear_row4(xb, Wi, t, best);
#pragma unroll 6
    for (int c = 1; c < C; ++c) {
        float o[4];
        bilinear_row4(xb + (size_t)c * Hi * Wi, Wi, t, o);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (o[i] > best[i]) { best[i] = o[i]; idx[i] = c; }
    }
    uint8_t* dst = mask + (b * Ho + yo) * Wo + 4 * q;
    if ((Wo & 3) == 0) {
        *reinterpret_cast<uchar4*>(dst) = make_uchar4(idx[0], idx[1], idx[2], idx[3]);
    } else {
        for (int i = 0; i < 4 && 4 * q + i < Wo; ++i) dst[i] = (uint8_t)idx[i];
    }
}

int try_fast_fwd(int dtype, const void* x, const void* bank, long ld, int batch, int c_in, int H, int W, int fh, int fw, int c_out,
                 int k, int pad, int pad_mode, int groups, const float* scale, const float* shift, int act, void* y,
                 hipStream_t stream);                                                                            // hs_patch_conv_bwd.hip
int try_launch_k1m(const StageIn& si, int fh, int fw, const float* bank, long ld, int cin, int c_out,
                   const float* scale, const float* shift, int act, float* y, hipStream_t stream);     // hs_patch_conv_k1m.hip

}  // namespace hs

using namespace hs;

// The k = 1 whole-patch-per-workgroup form; ``co``: blocked signal2weights workgroups to run beside it (heterogeneous launch) or
// null.  1 = the form does not apply.
static int launch_conv1x1(const StageIn& si, int batch, int fh, int fw, int ph, int pw, const float* bank, long ld, int c_out, int groups,
                          int cin_g, int cout_g, const float* scale, const float* shift, int act, float* y, const S2bArgs* co,
                          hipStream_t stream) {
    const int cin = si.cin();
    const size_t hp4 = ((size_t)c_out * cin_g + 3) & ~(size_t)3;
    size_t lds1 = (hp4 + (size_t)cin * ph * pw) * sizeof(float);
    if (!(lds1 <= 96 * 1024 && (ld & 3) == 0 && ((uintptr_t)bank & 15) == 0 && ph * pw <= 4096)) return 1;
    Conv1S2wArgs g;
    Conv1Args& f = g.c;
    f.in = si; f.fh = fh; f.fw = fw; f.ph = ph; f.pw = pw; f.bank = bank; f.ld = ld;
    f.cout = c_out; f.groups = groups; f.cin_g = cin_g; f.cout_g = cout_g;
    f.scale = scale; f.shift = shift; f.act = act; f.y = y;
    const int outs = c_out * ph * pw;
    int split = 1;
    while (split < 64 && outs * split * 2 <= CONV_THREADS && split * 2 <= cin_g) split *= 2;
    f.split = split;
    const long conv_blocks = (long)batch * fh * fw;
    if (co) {
        g.conv_blocks = (int)conv_blocks;
        g.s = *co;
        if (lds1 < S2B_LDS_FLOATS * sizeof(float)) lds1 = S2B_LDS_FLOATS * sizeof(float);
    }
    const dim3 grid((unsigned)(conv_blocks + (co ? co->n_wg : 0)));
#define HS_K1_LAUNCH(PWL) do { \
        if (lds1 > 64 * 1024) { \
            static std::atomic<unsigned long long> done{0}, done_co{0}; \
            const int e = co ? allow_full_lds((const void*)patch_conv1x1_s2w_kernel<PWL>, done_co) \
                             : allow_full_lds((const void*)patch_conv1x1_kernel<PWL>, done); \
            if (e != HS_OK) return e; \
        } \
        if (co) hipLaunchKernelGGL(patch_conv1x1_s2w_kernel<PWL>, grid, dim3(CONV_THREADS), lds1, stream, g); \
        else hipLaunchKernelGGL(patch_conv1x1_kernel<PWL>, grid, dim3(CONV_THREADS), lds1, stream, f); \
        return launch_status(); } while (0)
    if (ph == pw && ph == 1) HS_K1_LAUNCH(0);
    if (ph == pw && ph == 2) HS_K1_LAUNCH(1);
    if (ph == pw && ph == 4) HS_K1_LAUNCH(2);
    if (ph == pw && ph == 8) HS_K1_LAUNCH(3);
    HS_K1_LAUNCH(-1);
#undef HS_K1_LAUNCH
}

// The same form for the bf16 training step's k = 1 levels (hs_patch_conv_plain_fwd, hs_patch_conv_train.hip): x (B, cin, H, W) and y (B, c_out, H, W)
// in bf16, fp32 bank, groups = 1, no epilogue.  1 = the form does not apply (the caller keeps its generic kernel).
namespace hs {
int launch_conv1x1_bf16(const void* x, int batch, int cin, int H, int W, int fh, int fw, const float* bank, long ld, int c_out, void* y,
                        hipStream_t stream) {
    const int ph = H / fh, pw = W / fw;
    const size_t hp4 = ((size_t)c_out * cin + 3) & ~(size_t)3;
    const size_t lds1 = (hp4 + (size_t)cin * ph * pw) * sizeof(float);
    if (!(lds1 <= 64 * 1024 && (ld & 3) == 0 && ((uintptr_t)bank & 15) == 0 && ph * pw <= 64 && cin >= 1)) return 1;
    Conv1Args f{};
    f.in.skip = reinterpret_cast<const float*>(x); f.in.prev = nullptr;
    f.in.B = batch; f.in.H = H; f.in.W = W; f.in.c_skip = cin; f.in.c_prev = 0; f.in.Hp = 1; f.in.Wp = 1; f.in.coords = 0; f.in.prev_mode = HS_PREV_NONE;
    f.in.step_x = f.in.step_y = 0.0f; f.in.scale_y = f.in.scale_x = 1.0f;
    f.fh = fh; f.fw = fw; f.ph = ph; f.pw = pw; f.bank = bank; f.ld = ld;
    f.cout = c_out; f.groups = 1; f.cin_g = cin; f.cout_g = c_out;
    f.scale = nullptr; f.shift = nullptr; f.act = HS_ACT_NONE; f.y = reinterpret_cast<float*>(y);
    const int outs = c_out * ph * pw;
    int split = 1;
    while (split < 64 && outs * split * 2 <= CONV_THREADS && split * 2 <= cin) split *= 2;
    f.split = split;
    const dim3 grid((unsigned)((long)batch * fh * fw));
    if (ph == pw && ph == 1) hipLaunchKernelGGL((patch_conv1x1_kernel<0, bf16_t>), grid, dim3(CONV_THREADS), lds1, stream, f);
    else if (ph == pw && ph == 2) hipLaunchKernelGGL((patch_conv1x1_kernel<1, bf16_t>), grid, dim3(CONV_THREADS), lds1, stream, f);
    else if (ph == pw && ph == 4) hipLaunchKernelGGL((patch_conv1x1_kernel<2, bf16_t>), grid, dim3(CONV_THREADS), lds1, stream, f);
    else if (ph == pw && ph == 8) hipLaunchKernelGGL((patch_conv1x1_kernel<3, bf16_t>), grid, dim3(CONV_THREADS), lds1, stream, f);
    else hipLaunchKernelGGL((patch_conv1x1_kernel<-1, bf16_t>), grid, dim3(CONV_THREADS), lds1, stream, f);
    return launch_status();
}
}  // namespace hs

extern "C" int hs_patch_conv_fwd(const hs_stage_input* in, int32_t fh, int32_t fw, const float* bank, int64_t ld,
                                 int32_t c_out, int32_t k, int32_t pad, int32_t pad_mode, int32_t groups,
                                 const hs_epilogue* ep, float* y, void* stream) {
    ConvArgs a;
    int st = make_stage(in, &a.in);
    if (st != HS_OK) return st;
    if (!bank || !y || fh <= 0 || fw <= 0 || c_out <= 0 || k <= 0 || groups <= 0 || pad < 0) return HS_ERR_BAD_ARG;
    if (2 * pad != k - 1) return HS_ERR_UNSUPPORTED;         // "same" convolutions only (every reference config)
    if (pad_mode < HS_PAD_ZEROS || pad_mode > HS_PAD_CIRCULAR) return HS_ERR_BAD_ARG;
    if (in->H % fh != 0 || in->W % fw != 0) return HS_ERR_NOT_DIVISIBLE;
    const int cin = a.in.cin();
    if (cin % groups != 0 || c_out % groups != 0) return HS_ERR_BAD_ARG;
    if (pad_mode == HS_PAD_REFLECT && (pad >= in->H || pad >= in->W)) return HS_ERR_BAD_ARG;
    a.fh = fh; a.fw = fw; a.ph = in->H / fh; a.pw = in->W / fw;
    a.bank = bank; a.ld = ld;
    a.cout = c_out; a.k = k; a.pad = pad; a.pad_mode = pad_mode; a.groups = groups;
    a.cin_g = cin / groups; a.cout_g = c_out / groups;
    if (ld < (int64_t)c_out * a.cin_g * k * k) return HS_ERR_BAD_ARG;
    a.scale = ep ? ep->scale : nullptr; a.shift = ep ? ep->shift : nullptr; a.act = ep ? ep->act : HS_ACT_NONE;
    if (a.scale && !a.shift) return HS_ERR_BAD_ARG;
    a.y = y;
    if (!a.in.coords && a.in.c_prev == 0) {       // plain input (the autograd path): matrix-core / image-level forms of hs_patch_conv_bwd.hip
        const int r = try_fast_fwd(HS_DTYPE_F32, a.in.skip, bank, (long)ld, a.in.B, cin, a.in.H, a.in.W, fh, fw, c_out, k, pad, pad_mode, groups,
                                   a.scale, a.shift, a.act, y, (hipStream_t)stream);
        if (r != 1) return r;
    }
    if (k == 1 && groups == 1) {           // batched tiny patches: the weight stream on the matrix cores (hs_patch_conv_k1m.hip)
        const int r = try_launch_k1m(a.in, fh, fw, bank, (long)ld, cin, c_out, a.scale, a.shift, a.act, y, (hipStream_t)stream);
        if (r != 1) return r;
    }
    if (k == 1) {
        const int r = launch_conv1x1(a.in, in->batch, fh, fw, a.ph, a.pw, bank, (long)ld, c_out, groups, a.cin_g, a.cout_g, a.scale, a.shift, a.act, y,
                                     nullptr, (hipStream_t)stream);
        if (r != 1) return r;
    }
    const int wrow = a.cin_g * k * k;
    a.w_stride = wrow | 1;
    // tile: whole patch if it fits the LDS budget, otherwise split (rows first, then columns)
    const size_t budget = 96 * 1024;
    const size_t wbytes = (size_t)c_out * a.w_stride * sizeof(float);
    if (wbytes + (size_t)cin * (1 + 2 * pad) * (1 + 2 * pad) * sizeof(float) > 150 * 1024) return HS_ERR_LDS;
    a.TW = a.pw > 64 ? 64 : a.pw;
    a.TH = a.ph > 64 ? 64 : a.ph;
    auto tile_bytes = [&](int th, int tw) { return wbytes + (size_t)cin * (th + 2 * pad) * (tw + 2 * pad) * sizeof(float); };
    while (tile_bytes(a.TH, a.TW) > budget && (a.TH > 1 || a.TW > 1)) {
        if (a.TH >= a.TW && a.TH > 1) a.TH = (a.TH + 1) / 2; else a.TW = (a.TW + 1) / 2;
    }
    a.tiles_y = (a.ph + a.TH - 1) / a.TH;
    a.tiles_x = (a.pw + a.TW - 1) / a.TW;
    const size_t lds = tile_bytes(a.TH, a.TW);
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};
        const int e = allow_full_lds((const void*)patch_conv_kernel, done);
        if (e != HS_OK) return e;
    }
    const long blocks = (long)in->batch * fh * fw * a.tiles_y * a.tiles_x;
    hipLaunchKernelGGL(patch_conv_kernel, dim3((unsigned)blocks), dim3(CONV_THREADS), lds, (hipStream_t)stream, a);
    return launch_status();
}

// hs_patch_conv_s2w_fwd: hs_patch_conv_fwd (k = 1, no padding) and hs_signal2weights_multi_fwd for ``layers`` in ONE launch (see
// patch_conv1x1_s2w_kernel).  HS_ERR_UNSUPPORTED when either half would not take the form this launch is built from (the caller
// then issues the two calls separately); nothing has been launched in that case.
#ifndef HS_K1M_MIN_PATCHES
#define HS_K1M_MIN_PATCHES 1024
#endif
extern "C" int hs_patch_conv_s2w_fwd(const hs_stage_input* in, int32_t fh, int32_t fw, const float* bank, int64_t ld, int32_t c_out,
                                     int32_t groups, const hs_epilogue* ep, float* y,
                                     const float* signal, int32_t batch, int32_t c_signal, int32_t sfh, int32_t sfw,
                                     const hs_s2w_layer* layers, int32_t n_layers, void* stream) {
    StageIn si;
    int st = make_stage(in, &si);
    if (st != HS_OK) return st;
    if (!bank || !y || !signal || !layers || fh <= 0 || fw <= 0 || c_out <= 0 || groups <= 0) return HS_ERR_BAD_ARG;
    if (n_layers <= 0 || n_layers > S2W_MAX_LAYERS || batch <= 0 || sfh <= 0 || sfw <= 0) return HS_ERR_BAD_ARG;
    if ((size_t)batch * c_signal * sfh * sfw >= (1ull << 31)) return HS_ERR_UNSUPPORTED;     // 32-bit element offsets
    if (in->H % fh != 0 || in->W % fw != 0) return HS_ERR_NOT_DIVISIBLE;
    const int cin = si.cin();
    if (cin % groups != 0 || c_out % groups != 0) return HS_ERR_BAD_ARG;
    const int cin_g = cin / groups, cout_g = c_out / groups, ph = in->H / fh, pw = in->W / fw;
    if (ld < (int64_t)c_out * cin_g) return HS_ERR_BAD_ARG;
    const float* scale = ep ? ep->scale : nullptr; const float* shift = ep ? ep->shift : nullptr;
    if (scale && !shift) return HS_ERR_BAD_ARG;
    if (!si.coords && si.c_prev == 0) return HS_ERR_UNSUPPORTED;                   // plain inputs have their own forms
    if (groups == 1 && ph == 2 && pw == 2 && (long)in->batch * fh * fw >= HS_K1M_MIN_PATCHES) return HS_ERR_UNSUPPORTED;   // the weight-stream form's
    for (int i = 0; i < n_layers; ++i) {
        const int chk = s2w_check_layer(layers[i], c_signal);
        if (chk != HS_OK) return chk;
    }
    S2bArgs co;
    if (s2b_fill_args(co, signal, batch, c_signal, sfh, sfw, layers, nullptr, n_layers) != 0) return HS_ERR_UNSUPPORTED;
    const int r = launch_conv1x1(si, in->batch, fh, fw, ph, pw, bank, (long)ld, c_out, groups, cin_g, cout_g, scale, shift,
                                 ep ? ep->act : HS_ACT_NONE, y, &co, (hipStream_t)stream);
    return r == 1 ? HS_ERR_UNSUPPORTED : r;
}

extern "C" int hs_stage_input_typed_fwd(const hs_stage_input* in, int32_t prev_dtype, int32_t out_dtype, void* y, void* stream) {
    StageIn s;
    int st = make_stage(in, &s);
    if (st != HS_OK) return st;
    if (!y) return HS_ERR_BAD_ARG;
    if ((prev_dtype != HS_DTYPE_F32 && prev_dtype != HS_DTYPE_BF16) || (out_dtype != HS_DTYPE_F32 && out_dtype != HS_DTYPE_BF16)) return HS_ERR_BAD_ARG;
    const size_t n = (size_t)s.B * s.cin() * s.H * s.W;
    const dim3 blocks((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256));
    hipStream_t q = (hipStream_t)stream;
    if ((long)s.B * s.cin() <= 65535) {
        const dim3 g2((s.W + 63) / 64, (s.H + 15) / 16, s.B * s.cin());            // four rows per thread
        if (prev_dtype == HS_DTYPE_F32 && out_dtype == HS_DTYPE_F32) hipLaunchKernelGGL((stage_input_plane_kernel<float, float>), g2, dim3(256), 0, q, s, (float*)y);
        else if (prev_dtype == HS_DTYPE_F32) hipLaunchKernelGGL((stage_input_plane_kernel<float, bf16_t>), g2, dim3(256), 0, q, s, (bf16_t*)y);
        else if (out_dtype == HS_DTYPE_F32) hipLaunchKernelGGL((stage_input_plane_kernel<bf16_t, float>), g2, dim3(256), 0, q, s, (float*)y);
        else hipLaunchKernelGGL((stage_input_plane_kernel<bf16_t, bf16_t>), g2, dim3(256), 0, q, s, (bf16_t*)y);
        return launch_status();
    }
    if (prev_dtype == HS_DTYPE_F32 && out_dtype == HS_DTYPE_F32) hipLaunchKernelGGL((stage_input_kernel<float, float>), blocks, dim3(256), 0, q, s, (float*)y);
    else if (prev_dtype == HS_DTYPE_F32) hipLaunchKernelGGL((stage_input_kernel<float, bf16_t>), blocks, dim3(256), 0, q, s, (bf16_t*)y);
    else if (out_dtype == HS_DTYPE_F32) hipLaunchKernelGGL((stage_input_kernel<bf16_t, float>), blocks, dim3(256), 0, q, s, (float*)y);
    else hipLaunchKernelGGL((stage_input_kernel<bf16_t, bf16_t>), blocks, dim3(256), 0, q, s, (bf16_t*)y);
    return launch_status();
}

extern "C" int hs_stage_input_fwd(const hs_stage_input* in, float* y, void* stream) {
    return hs_stage_input_typed_fwd(in, HS_DTYPE_F32, HS_DTYPE_F32, y, stream);
}

extern "C" int hs_upsample_argmax_fwd(const float* x, int32_t batch, int32_t channels, int32_t Hi, int32_t Wi,
                                      int32_t Ho, int32_t Wo, uint8_t* mask, void* stream) {
    if (!x || !mask || batch <= 0 || channels <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return HS_ERR_BAD_ARG;
    if (channels > 256) return HS_ERR_UNSUPPORTED;           // uint8 class indices
    if (Ho == 2 * Hi && Wo == 2 * Wi && (Wi & 1) == 0) {
        const size_t n2 = (size_t)batch * Hi * (Wi / 2) * 4;         // 4 lanes per 2x4 output block
        hipLaunchKernelGGL(upsample2x_argmax_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           x, batch, channels, Hi, Wi, mask);
        return launch_status();
    }
    const size_t n = (size_t)batch * Ho * ((Wo + 3) / 4);
    hipLaunchKernelGGL(upsample_argmax_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, batch, channels, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo, mask);
    return launch_status();
}

extern "C" int hs_upsample_bilinear_bf16_fwd(const void* x, int32_t batch, int32_t channels, int32_t Hi, int32_t Wi,
                                             int32_t Ho, int32_t Wo, void* y, void* stream) {
    if (!x || !y || batch <= 0 || channels <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return HS_ERR_BAD_ARG;
    if (Ho == 2 * Hi && Wo == 2 * Wi && (Wi & 1) == 0 && (((size_t)y) & 7) == 0) {          // rows of 4 k outputs: 8-byte stores
        const size_t n2 = (size_t)batch * channels * Hi * (Wi / 2);
        const unsigned blocks2 = (unsigned)((n2 + 255) / 256 > 8192 ? 8192 : (n2 + 255) / 256);
        hipLaunchKernelGGL(upsample2x_bf16_kernel, dim3(blocks2), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, batch * channels, Hi, Wi, (bf16_t*)y);
        return launch_status();
    }
    const size_t n = (size_t)batch * channels * Ho * ((Wo + 3) / 4);
    const unsigned blocks = (unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
    hipLaunchKernelGGL(upsample_bilinear_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, batch * channels, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo, (bf16_t*)y);
    return launch_status();
}

extern "C" int hs_upsample_bilinear_fwd(const float* x, int32_t batch, int32_t channels, int32_t Hi, int32_t Wi,
                                        int32_t Ho, int32_t Wo, float* y, void* stream) {
    if (!x || !y || batch <= 0 || channels <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return HS_ERR_BAD_ARG;
    if (Ho == 2 * Hi && Wo == 2 * Wi && (Wi & 1) == 0) {
        const size_t n2 = (size_t)batch * channels * Hi * (Wi / 2);
        const unsigned blocks2 = (unsigned)((n2 + 255) / 256 > 8192 ? 8192 : (n2 + 255) / 256);
        hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks2), dim3(256), 0, (hipStream_t)stream,
                           x, batch * channels, Hi, Wi, y);
        return launch_status();
    }
    const size_t n = (size_t)batch * channels * Ho * ((Wo + 3) / 4);
    const unsigned blocks = (unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       x, batch * channels, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo, y);
    return launch_status();
}

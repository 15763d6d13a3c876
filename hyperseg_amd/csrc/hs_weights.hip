// Filter-bank producers: signal2weights (grouped 1x1 conv -> patch-major bank), bank re-layout
// from the reference's channel-major weight tensor, and inference BatchNorm folding.
//
// HBM layout: bank[p*ld + m], p = (b*fh + i)*fw + j.  One patch's whole bank is one contiguous
// run of ld floats, so the stage kernels read it with full-line coalesced (or scalar-cache)
// loads; the reference instead keeps (B, hp, fh, fw) and pays a permute+reshape copy per level
// (hyperseg_v1_0.py:334-337, 491-492).
#include "hs_common.h"
#include "hs_s2w_blocked.h"

namespace hs {

// ------------------------------------------------------------------------------------------
// signal2weights, all levels of a decoder in ONE launch, on the f32 MATRIX cores.
//
// Per layer and group this is a dense GEMM  bank[p, n] = sum_k S[p, k] * Wsw[n, k]  (M = rows of the group,
// N = patches, K = Cs/G = 7..80) -- exactly the case the north star reserves MFMA for.  One wave owns a strip of
// 32 bank rows (two 16x16 tiles sharing the signal operand) and walks 16-patch tiles with
// v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fma chain, bit-compatible with the VALU form):
//   A[i = row][k]     lane l holds Wsw_t[k0 + (l>>4)][n0 + (l&15)]      64-byte coalesced runs, L2 resident
//   B[k][j = patch]   lane l holds S[b, idx + g*K + k0 + (l>>4), ij]    16 consecutive patches = 64 bytes
//   D[i][j]           lane l holds rows n0 + 4*(l>>4) + {0..3} of patch p0 + (l&15) -> one 16-byte store
// No LDS, no scalar-cache traffic; operand loads are independent of the accumulator chain so they stream.
// ------------------------------------------------------------------------------------------
constexpr int S2W_THREADS = 256;          // 4 waves
constexpr int S2W_STRIP = 32;             // bank rows per wave-task (2 MFMA row tiles)
constexpr int S2W_PT = 2;                 // 16-patch tiles per wave-task

struct S2wLayer {
    const float* __restrict__ wsw_t;
    float* __restrict__ bank;
    long ld;
    int signal_index, cs_g, rows_per_group, wc, rows;
    int strips_per_group, strip_begin;     // strips of this layer start at blockIdx.x == strip_begin
    int a_rs, a_ks;                        // weight element (n, k) at wsw_t[n * a_rs + k * a_ks]: (1, wc) transposed, (cs_g, 1) the conv's own layout
};
struct S2wArgs {
    const float* __restrict__ signal;
    int c_signal, grid_sz, n_patches, n_layers, n_strips;
    S2wLayer layer[S2W_MAX_LAYERS];
};

// One wave-task: a 32-row strip x S2W_PT 16-patch tiles, K padded to 4*KS.  Every operand load of the task is
// issued before the first MFMA (A: 2*KS registers, B: S2W_PT*KS registers), so the L2 latency is paid once per
// task and hidden across the ~8 waves per SIMD the grid provides.
template <int KS>
__device__ __forceinline__ void s2w_task(const float* __restrict__ wsw_t, const float* __restrict__ signal,
                                         float* __restrict__ bank, long ld, int cs_g, int rpg, int wc, int rows,
                                         int r0, int n0, size_t sig_base, int c_signal, int grid_sz, int n_patches,
                                         int tile0, int lane, float* __restrict__ stage, int a_rs, int a_ks) {
    const int lrow = lane & 15, lk = lane >> 4;
    const bool a0_ok = (r0 + lrow) < rpg, a1_ok = (r0 + 16 + lrow) < rpg;
    // 32-bit element offsets from the uniform base pointers (saddr + voffset addressing: one VGPR per load);
    // masked lanes are clamped so that they still read inside the arrays
    const unsigned na0 = (unsigned)min(n0 + lrow, wc - 1), na1 = (unsigned)min(n0 + 16 + lrow, wc - 1);
    float a0[KS], a1[KS], bv[S2W_PT][KS];
    bool p_ok[S2W_PT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const unsigned kc = (unsigned)min(ks * 4 + lk, cs_g - 1);
        a0[ks] = wsw_t[kc * (unsigned)a_ks + na0 * (unsigned)a_rs];
        a1[ks] = wsw_t[kc * (unsigned)a_ks + na1 * (unsigned)a_rs];
    }
#pragma unroll
    for (int t = 0; t < S2W_PT; ++t) {
        const int p = (tile0 + t) * 16 + lrow;
        p_ok[t] = p < n_patches;
        const int pc = p_ok[t] ? p : 0;
        const int bb = pc / grid_sz, ij = pc - bb * grid_sz;
        const unsigned sb = (unsigned)((size_t)bb * c_signal * grid_sz + sig_base + ij);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned kc = (unsigned)min(ks * 4 + lk, cs_g - 1);
            bv[t][ks] = signal[sb + kc * (unsigned)grid_sz];
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bool k_ok = (ks * 4 + lk) < cs_g;
        a0[ks] = (k_ok && a0_ok) ? a0[ks] : 0.0f;
        a1[ks] = (k_ok && a1_ok) ? a1[ks] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < S2W_PT; ++t) {
        if ((tile0 + t) * 16 >= n_patches) break;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool k_ok = (ks * 4 + lk) < cs_g;
            const float fb = (k_ok && p_ok[t]) ? bv[t][ks] : 0.0f;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ks], fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], fb, acc1, 0, 0, 0);
        }
        // D (lane l: rows 4*lk + r of patch lrow) -> wave-private LDS [patch][row] -> 128-byte row runs: each store
        // instruction writes two patches x 32 consecutive bank rows, whatever the alignment of the group start
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            stage[lrow * 33 + 4 * lk + r] = acc0[r];
            stage[lrow * 33 + 16 + 4 * lk + r] = acc1[r];
        }
        const int row = lane & 31;
        const bool row_ok = (r0 + row) < rpg && (n0 + row) < rows;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pl = 2 * q + (lane >> 5);
            const int p = (tile0 + t) * 16 + pl;
            const float v = stage[pl * 33 + row];
            if (row_ok && p < n_patches) bank[(size_t)p * ld + n0 + row] = v;
        }
    }
}

__global__ __launch_bounds__(S2W_THREADS)
void signal2weights_kernel(S2wArgs a) {
    // layer descriptor through the kernarg segment with a uniform index -> scalar loads, no register copies
    const __attribute__((address_space(4))) S2wArgs* ka =
        (const __attribute__((address_space(4))) S2wArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int vstrip = (int)blockIdx.x;
    if (vstrip >= ka->n_strips) return;
    int li = 0;
    for (int q = 1; q < ka->n_layers; ++q)
        if (vstrip >= ka->layer[q].strip_begin) li = q;
    const float* __restrict__ wsw_t = ka->layer[li].wsw_t;
    float* __restrict__ bank = ka->layer[li].bank;
    const long ld = ka->layer[li].ld;
    const int signal_index = ka->layer[li].signal_index, cs_g = ka->layer[li].cs_g;
    const int rpg = ka->layer[li].rows_per_group, wc = ka->layer[li].wc, rows = ka->layer[li].rows;
    const int spg = ka->layer[li].strips_per_group;
    const int strip = vstrip - ka->layer[li].strip_begin;
    const int grid_sz = ka->grid_sz, n_patches = ka->n_patches, c_signal = ka->c_signal;
    const float* __restrict__ signal = ka->signal;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = strip / spg;
    const int r0 = (strip - g * spg) * S2W_STRIP;            // first row of the strip inside the group
    const int n0 = g * rpg + r0;                             // natural row index
    const size_t sig_base = (size_t)(signal_index + g * cs_g) * grid_sz;
    const int tile0 = (blockIdx.y * 4 + wave) * S2W_PT;
    if (tile0 * 16 >= n_patches) return;
    const int ksteps = (cs_g + 3) >> 2;
    __shared__ float stage_all[4 * 16 * 33];
    float* stage = stage_all + wave * (16 * 33);
    const int a_rs = ka->layer[li].a_rs, a_ks = ka->layer[li].a_ks;
#define HS_S2W_CASE(KS) s2w_task<KS>(wsw_t, signal, bank, ld, cs_g, rpg, wc, rows, r0, n0, sig_base, c_signal, \
                                     grid_sz, n_patches, tile0, lane, stage, a_rs, a_ks)
    if (ksteps <= 2) HS_S2W_CASE(2);
    else if (ksteps <= 4) HS_S2W_CASE(4);
    else if (ksteps <= 8) HS_S2W_CASE(8);
    else if (ksteps <= 12) HS_S2W_CASE(12);
    else HS_S2W_CASE(20);                                    // K <= 80; larger K is rejected by the host wrapper
#undef HS_S2W_CASE
}

// ------------------------------------------------------------------------------------------
// signal2weights, blocked form (round 3): the same GEMMs with both operands staged through LDS by DMA.
//
// Round 2's kernel above gives every wave its own 32-row x 32-patch task and lets it fetch its operands itself: 4 KS
// per-lane dword gathers per wave (80 at K = 80), every wave of a workgroup re-reading the same weights -- 19-21 us for 4 us of
// matrix-core work (profiles/round2_bench_kernel_stats.csv), ~1.3 resident waves per SIMD.  Here a workgroup owns a 64-row x
// 64-patch block of one (layer, group):
//   A  the conv weight, re-laid ONCE per parameter version by hs_s2w_pack_fwd into the exact LDS image of the blocks
//      ([group][row block][k-step][row tile][k mod 4][16 rows], zero-padded): one linear DMA stream, 16 bytes per lane;
//   B  the signal (B, C, fh, fw): per k-step one DMA instruction, lane = (patch tile, k mod 4, 4 consecutive patches) reading
//      16 bytes -- the LDS image [k-step][patch tile][k mod 4][16 patches] makes every fragment read one conflict-free ds_read_b32;
//   K  in fills of S2B_KC k-steps (20 KB of LDS per workgroup: 8 workgroups per CU), accumulators live across fills;
//   D  through LDS ([patch][64 rows]) so that a store instruction writes one patch's 64 consecutive bank rows (256 bytes).
// Same arithmetic as the direct form: v_mfma_f32_16x16x4_f32, k ascending -- bit-identical banks (tests/test_hip_parity.py).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void signal2weights_blocked_kernel(S2bArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[S2B_LDS_FLOATS];
    s2b_body((const __attribute__((address_space(4))) S2bArgs*)__builtin_amdgcn_kernarg_segment_ptr(), (int)blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------
// The blocked form as a STREAM (round 5; VERDICT r4 #5a): fewer workgroups, each walking several 64-row x 64-patch blocks with the
// NEXT fill's LDS-DMA in flight while the current block is multiplied, staged and stored.
// In the one-block-per-workgroup launch above all 8832 waves are resident from the first cycle and in the same phase at the same
// time -- the DMA burst, the matrix-core phase and 32 MB of stores follow each other chip-wide (profiles/round3_pmc_k1_and_s2w.txt), 16 us
// where the stores alone need ~8.  Here a workgroup owns two operand buffers (2 x 20 KB: four workgroups per CU): per fill it waits
// for the fill to land, requests the next one (the next K chunk of the block, or the first chunk of its next block, blocks strided by
// the grid so that every workgroup gets heavy and light layers alike) into the other buffer, multiplies, and when the block is complete
// sends D out through the buffer it has just consumed.  Same blocks, same instruction, same k order: bit-identical banks.
// (vmcnt counts stores as well and is only ordered among operations of one kind, so the wait at the top is vmcnt(0): it also waits for
// the previous block's store acknowledgements -- by then the next fill has been in flight for a whole block.)
// ------------------------------------------------------------------------------------------
// MEASURED NEGATIVE (visit r5v11, profiles/round5_s2w_stream_negative.txt): 19.5 us against 16.3 at HyperSeg-M, 23.6 against 19.3 at S,
// whole frame 0.779 against 0.775 ms -- 736 workgroups x 3 blocks keep 12 waves per CU where the one-block form keeps 32, and the walk
// is wait -> barrier -> products -> barrier -> stage -> barrier -> stores per block with nothing of the SAME workgroup to overlap but the
// next fill.  Off in the product build (-DHS_S2B_STREAM=1 builds it; parity is the same tests either way).
#ifndef HS_S2B_STREAM
#define HS_S2B_STREAM 0
#endif
constexpr int S2B_STREAM_WG_PER_CU = 4;

struct S2bBlock {
    const float* ablk; float* bank; long ld;
    unsigned sig0;                         // this lane's signal element offset at k = 0
    int cs_g, KS, n0, rows_left, pb;       // rows_left: valid rows of the block from n0 on (<= 0: none for this lane's row)
};

__device__ __forceinline__ void s2b_decode(const __attribute__((address_space(4))) S2bArgs* ka, const int wg, const int lane, S2bBlock& b) {
    int li = 0;
    for (int q = 1; q < ka->n_layers; ++q)
        if (wg >= ka->layer[q].wg_begin) li = q;
    const int rpg = ka->layer[li].rpg, rows = ka->layer[li].rows, KS = ka->layer[li].ks, RB = ka->layer[li].rb;
    const int cs_g = ka->layer[li].cs_g, grid_sz = ka->grid_sz, PB = ka->pb;
    const int local = wg - ka->layer[li].wg_begin;
    const int grb = (int)s2b_div((unsigned)local, ka->m_pb), pb = local - grb * PB;
    const int g = grb / RB, rb = grb - g * RB;
    const int pt = lane >> 4, j4 = lane & 3;
    const int p4 = min(pb * S2B_PATCHES + 16 * pt + 4 * j4, ka->n_patches - 4);
    const int bb = (int)s2b_div((unsigned)p4, ka->m_grid), ij = p4 - bb * grid_sz;
    b.sig0 = (unsigned)((bb * ka->c_signal + ka->layer[li].signal_index + g * cs_g) * grid_sz + ij);
    b.ablk = ka->layer[li].blk + (size_t)((g * RB + rb) * KS) * 256;
    b.bank = ka->layer[li].bank; b.ld = ka->layer[li].ld;
    b.cs_g = cs_g; b.KS = KS; b.pb = pb;
    const int r0 = rb * S2B_ROWS;
    b.n0 = g * rpg + r0;
    b.rows_left = min(rpg - r0, rows - b.n0);
}

__device__ __forceinline__ void s2b_issue_fill(const __attribute__((address_space(4))) S2bArgs* ka, const S2bBlock& b, const int k0,
                                               float* __restrict__ A, const int wave, const int lane) {
    float* Bm = A + S2B_KC * 256;
    const int kn = min(S2B_KC, b.KS - k0), kq = (lane >> 2) & 3;
    const unsigned grid_sz = (unsigned)ka->grid_sz;
    for (int c = wave; c < kn; c += 4) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b.ablk + (size_t)(k0 + c) * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(A + c * 256), 16, 0, 0);
        const unsigned k = (unsigned)min(4 * (k0 + c) + kq, b.cs_g - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ka->signal + b.sig0 + k * grid_sz),
                                         (__attribute__((address_space(3))) void*)(Bm + c * 256), 16, 0, 0);
    }
}

__global__ __launch_bounds__(256)
void signal2weights_stream_kernel(S2bArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s2b_lds[];                        // 2 x S2B_LDS_FLOATS
    const __attribute__((address_space(4))) S2bArgs* ka = (const __attribute__((address_space(4))) S2bArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_wg = ka->n_wg, stride = (int)gridDim.x, n_patches = ka->n_patches;
    int blk = (int)blockIdx.x, k0 = 0, buf = 0;
    S2bBlock cur, nxt;
    s2b_decode(ka, blk, lane, cur);
    s2b_issue_fill(ka, cur, 0, s2b_lds, wave, lane);
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (;;) {
        const int kn = min(S2B_KC, cur.KS - k0);
        const bool last_chunk = k0 + S2B_KC >= cur.KS;                                     // uniform
        const int nblk = last_chunk ? blk + stride : blk, nk0 = last_chunk ? 0 : k0 + S2B_KC;
        const bool has_next = nblk < n_wg;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // this wave's part of the fill (and its earlier stores)
        __syncthreads();                                                                   // the whole fill; the other buffer is free
        if (has_next) {
            if (last_chunk) s2b_decode(ka, nblk, lane, nxt); else nxt = cur;
            s2b_issue_fill(ka, nxt, nk0, s2b_lds + (buf ^ 1) * S2B_LDS_FLOATS, wave, lane);
        }
        float* A = s2b_lds + buf * S2B_LDS_FLOATS;
        const float* Bm = A + S2B_KC * 256;
        for (int c = 0; c < kn; ++c) {
            const float av = A[c * 256 + wave * 64 + lane];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bm[c * 256 + t * 64 + lane], acc[t], 0, 0, 0);
        }
        if (last_chunk) {
            __syncthreads();                                                               // operands dead: the output block takes their place
            {
                const int j = lane & 15, q4 = lane >> 4;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        A[(16 * t + j) * (S2B_ROWS + 1) + 16 * wave + 4 * q4 + r] = acc[t][r];
                        acc[t][r] = 0.0f;
                    }
            }
            __syncthreads();
            const bool row_ok = lane < cur.rows_left;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int pl = wave * 16 + q, p = cur.pb * S2B_PATCHES + pl;
                const float v = A[pl * (S2B_ROWS + 1) + lane];
                if (row_ok && p < n_patches) cur.bank[(size_t)p * cur.ld + cur.n0 + lane] = v;
            }
        }
        if (!has_next) break;
        cur = nxt; blk = nblk; k0 = nk0; buf ^= 1;
    }
}

// hs_s2w_pack_fwd: (cs_g, wc) transposed conv weight -> the blocked kernel's LDS images, zero-padded
__global__ void s2w_pack_kernel(const float* __restrict__ wsw_t, int cs_g, int wc, int groups, int rb_n, int ks_n,
                                float* __restrict__ out, long total) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int i = (int)(e & 15), kq = (int)((e >> 4) & 3), rt = (int)((e >> 6) & 3);
    const long blkid = e >> 8;
    const int ks = (int)(blkid % ks_n);
    const long grb = blkid / ks_n;
    const int rb = (int)(grb % rb_n), g = (int)(grb / rb_n);
    const int rpg = wc / groups;
    const int r = rb * S2B_ROWS + rt * 16 + i, k = 4 * ks + kq;
    out[e] = (r < rpg && k < cs_g) ? wsw_t[(size_t)k * wc + (size_t)g * rpg + r] : 0.0f;
}

// ------------------------------------------------------------------------------------------
// bank_pack: (B, hp_total, fh, fw) channel-major -> patch-major, 32x32 LDS transpose tiles.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void bank_pack_kernel(const float* __restrict__ w, int hp_total, int grid_sz, int ch_offset,
                      int rows, float* __restrict__ bank, long ld) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int m_base = blockIdx.x * 32, q_base = blockIdx.y * 32;   // q = i*fw + j
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;         // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int m = m_base + r, q = q_base + tx;
        float v = 0.0f;
        if (m < rows && q < grid_sz) v = w[((size_t)b * hp_total + ch_offset + m) * grid_sz + q];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int q = q_base + r, m = m_base + tx;
        if (m < rows && q < grid_sz)
            bank[((size_t)b * grid_sz + q) * ld + m] = tile[tx][r];
    }
}

__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps,
                               int n, float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sc = gamma[i] / sqrtf(var[i] + eps);
    scale[i] = sc;
    shift[i] = beta[i] - mean[i] * sc;
}

}  // namespace hs

using namespace hs;

// floats of a layer's packed image
static long s2b_floats(int cs_g, int wc, int groups) {
    const int rpg = wc / groups;
    return (long)groups * ((rpg + S2B_ROWS - 1) / S2B_ROWS) * ((cs_g + 3) / 4) * 256;
}

extern "C" int64_t hs_s2w_pack_floats(int32_t signal_channels, int32_t groups, int32_t wc) {
    if (signal_channels <= 0 || groups <= 0 || wc <= 0 || signal_channels % groups != 0 || wc % groups != 0) return HS_ERR_BAD_ARG;
    return s2b_floats(signal_channels / groups, wc, groups);
}

extern "C" int hs_s2w_pack_fwd(const float* wsw_t, int32_t signal_channels, int32_t groups, int32_t wc, float* out, void* stream) {
    if (!wsw_t || !out) return HS_ERR_BAD_ARG;
    const int64_t n = hs_s2w_pack_floats(signal_channels, groups, wc);
    if (n < 0) return (int)n;
    const int cs_g = signal_channels / groups, rpg = wc / groups;
    hipLaunchKernelGGL(s2w_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wsw_t, cs_g, wc, groups,
                       (rpg + S2B_ROWS - 1) / S2B_ROWS, (cs_g + 3) / 4, out, (long)n);
    return launch_status();
}

// The blocked form: every layer brought its packed weights, patches come in whole 16-byte groups.  1 = not applicable.
static int launch_s2w_blocked(const float* signal, int batch, int c_signal, int fh, int fw, const hs_s2w_layer* layers,
                              const int* order, int n_layers, hipStream_t stream) {
    S2bArgs a;
    if (s2b_fill_args(a, signal, batch, c_signal, fh, fw, layers, order, n_layers) != 0) return 1;
    // the stream form where there is more than one block per workgroup slot to walk; small launches keep one block per workgroup
    static const int cus_of_device0 = [] {                                                 // (one process per GPU: every visible device is the same part)
        int dev = 0, cus = 0;
        return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
    }();
    const int slots = cus_of_device0 * S2B_STREAM_WG_PER_CU;
    if (HS_S2B_STREAM && a.n_wg > slots) {
        const int per = (a.n_wg + slots - 1) / slots;                                      // blocks per workgroup (the last ones one fewer)
        const int wgs = (a.n_wg + per - 1) / per;
        hipLaunchKernelGGL(signal2weights_stream_kernel, dim3((unsigned)wgs), dim3(256), (size_t)2 * S2B_LDS_FLOATS * sizeof(float), stream, a);
        return launch_status();
    }
    hipLaunchKernelGGL(signal2weights_blocked_kernel, dim3((unsigned)a.n_wg), dim3(256), 0, stream, a);
    return launch_status();
}

// ``native``: wsw_t points at the Conv2d weight in its OWN (wc, cs_g) layout (the training path: weights change every step, so
// neither the transposed copy nor the packed image of the blocked form exists); always the direct kernel then.
int hs::s2w_multi_launch(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                         const hs_s2w_layer* layers, int32_t n_layers, bool native, void* stream) {
    if (!signal || !layers || n_layers <= 0 || n_layers > S2W_MAX_LAYERS) return HS_ERR_BAD_ARG;
    if (batch <= 0 || fh <= 0 || fw <= 0) return HS_ERR_BAD_ARG;
    if ((size_t)batch * c_signal * fh * fw >= (1ull << 31)) return HS_ERR_UNSUPPORTED;     // 32-bit element offsets
    S2wArgs a;
    a.signal = signal; a.c_signal = c_signal; a.grid_sz = fh * fw; a.n_patches = batch * fh * fw; a.n_layers = n_layers;
    // heaviest layers (largest K) first, so the long wave-tasks do not form the tail of the launch
    int order[S2W_MAX_LAYERS];
    for (int i = 0; i < n_layers; ++i) order[i] = i;
    for (int i = 1; i < n_layers; ++i)
        for (int q = i; q > 0; --q) {
            const hs_s2w_layer &x = layers[order[q]], &y = layers[order[q - 1]];
#ifdef HS_S2B_LIGHT_FIRST
            if (x.groups > 0 && y.groups > 0 && x.signal_channels / x.groups < y.signal_channels / y.groups) {
#else
            if (x.groups > 0 && y.groups > 0 && x.signal_channels / x.groups > y.signal_channels / y.groups) {
#endif
                const int t = order[q]; order[q] = order[q - 1]; order[q - 1] = t;
            } else break;
        }
    int strips = 0;
    for (int i = 0; i < n_layers; ++i) {
        const hs_s2w_layer& l = layers[order[i]];
        { const int chk = s2w_check_layer(l, c_signal); if (chk != HS_OK) return chk; }
        S2wLayer& d = a.layer[i];
        d.wsw_t = l.wsw_t; d.bank = l.bank; d.ld = (long)l.ld;
        d.signal_index = l.signal_index;
        d.cs_g = l.signal_channels / l.groups; d.rows_per_group = l.wc / l.groups; d.wc = l.wc; d.rows = l.rows;
        d.strips_per_group = (d.rows_per_group + S2W_STRIP - 1) / S2W_STRIP;
        d.strip_begin = strips;
        d.a_rs = native ? d.cs_g : 1; d.a_ks = native ? 1 : l.wc;
        strips += d.strips_per_group * l.groups;
    }
    if (!native) {
        const int st = launch_s2w_blocked(signal, batch, c_signal, fh, fw, layers, order, n_layers, (hipStream_t)stream);
        if (st != 1) return st;
    }
    for (int i = n_layers; i < S2W_MAX_LAYERS; ++i) { a.layer[i] = a.layer[0]; a.layer[i].strip_begin = 0x7fffffff; }
    a.n_strips = strips;
    const int tiles = (a.n_patches + 15) / 16;
    dim3 grid(strips, (tiles + 4 * S2W_PT - 1) / (4 * S2W_PT));
    hipLaunchKernelGGL(signal2weights_kernel, grid, dim3(S2W_THREADS), 0, (hipStream_t)stream, a);
    return launch_status();
}

extern "C" int hs_signal2weights_multi_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                           const hs_s2w_layer* layers, int32_t n_layers, void* stream) {
    return s2w_multi_launch(signal, batch, c_signal, fh, fw, layers, n_layers, false, stream);
}

extern "C" int hs_signal2weights_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                     int32_t signal_index, int32_t signal_channels, int32_t groups,
                                     const float* wsw_t, int32_t wc, int32_t rows,
                                     float* bank, int64_t ld, void* stream) {
    hs_s2w_layer l;
    l.signal_index = signal_index; l.signal_channels = signal_channels; l.groups = groups;
    l.wsw_t = wsw_t; l.wc = wc; l.rows = rows; l.bank = bank; l.ld = ld; l.wsw_blk = nullptr;
    return hs_signal2weights_multi_fwd(signal, batch, c_signal, fh, fw, &l, 1, stream);
}

extern "C" int hs_bank_pack_fwd(const float* w, int32_t batch, int32_t hp_total, int32_t fh, int32_t fw,
                                int32_t ch_offset, int32_t rows, float* bank, int64_t ld, void* stream) {
    if (!w || !bank || batch <= 0 || fh <= 0 || fw <= 0 || rows <= 0 || ld < rows || ch_offset < 0) return HS_ERR_BAD_ARG;
    if (ch_offset + rows > hp_total) return HS_ERR_BAD_ARG;
    const int grid_sz = fh * fw;
    dim3 grid((rows + 31) / 32, (grid_sz + 31) / 32, batch);
    hipLaunchKernelGGL(bank_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       w, hp_total, grid_sz, ch_offset, rows, bank, (long)ld);
    return launch_status();
}

extern "C" int hs_bn_fold_fwd(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                              int32_t n, float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !mean || !var || !scale || !shift || n <= 0) return HS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, mean, var, eps, n, scale, shift);
    return launch_status();
}

extern "C" int hs_version(void) { return HS_ABI_VERSION; }
extern "C" const char* hs_build_info(void) { return "libhyperseg_hip gfx950 (CDNA4) fp32; built " __DATE__ " " __TIME__; }

// hs_decoder_chain_fwd: the decoder's three coarse k = 1 levels (patches of 1, 2 and 4 pixels; hyperseg_v1_0.py:221-253 with the
// blocks of :486-498, 728-760) and, optionally, the first inverted-residual level behind them (8 x 8-pixel patches, :281-376) as ONE
// launch -- one workgroup per grid cell that walks level 0 -> 1 -> 2 (-> 3) and hands its level outputs to the neighbouring cells'
// workgroups INSIDE the launch.
//
// Why: as separate launches these levels are dependent latency chains of 6-10 us each (index arithmetic -> bank + inputs from HBM ->
// LDS -> products -> store, then the graph edge) for 2.5 + 1 us of HBM traffic (profiles/round4_bench_kernel_stats.csv: 5.7 + 7.2 +
// 6.7 + 9.4 us; VERDICT r4 #1c).  Level l + 1 needs of level l only a one-pixel ring around its own cell (the 2x bilinear upsample of
// align_corners=False touches the 3 x 3 cells around a cell), everything else it reads -- its bank, its skip feature, its BatchNorm
// rows -- is known when the launch starts.  So here
//   * every HBM load of all levels is issued at the top of the kernel, ALL by LDS-DMA (16-byte pieces for the banks, 4-byte gathers
//     for skip pixels and BatchNorm rows): level 0's first, and level 0 starts as soon as ITS operands have landed (s_waitcnt
//     vmcnt(N) with the later levels' loads still in flight: their round trip hides under level 0);
//   * the products run on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32, an fma chain in ascending k) with both operands in
//     LDS: a level is a [c_out x c_in] . [c_in x pixels] product of one to four 16 x 16 tiles, the K range split over the waves where
//     there are fewer tiles than waves (partial tiles summed in a fixed order).  (Round 5's first version used the lane-split dot
//     products of patch_conv1x1_kernel: 2.4 / 3.8 / 3.2 k cycles for levels 0 / 1 / 2 of a 25 k-cycle workgroup,
//     profiles/round5_k1_chain_phase_cycles_and_kernel_times_v4.txt);
//   * a level's outputs go to the neighbours as 8-byte {value, tag} granules written by ONE agent-scope (sc1, write-through) store
//     each and polled by the consumer with agent-scope loads until the tag matches -- the data is the flag, no fence, no flag word
//     (MI355X_MICROARCH.md price list, handoff-1to1: 0.8-2.9 us per hop, against 1.5-1.9 us for a kernel boundary PLUS the next
//     launch's own load round trip); the rings are 4 KB, 3 KB and 2.5 KB per workgroup;
//   * the tag is a per-workspace generation number: a workgroup reads the tag ITS OWN first granule carries from the previous launch
//     (complete: that launch has ended) and publishes with tag + 1.  Every launch rewrites every granule exactly once, so all cells
//     agree on the generation without a host-side argument (kernel arguments are frozen under graph replay) and without a memset
//     node in front of the launch.  The workspace is the caller's, zero-filled once, bound to one (batch, grid) and one stream.
//   * the inverted residual (level 3, Op C on the cell's own 8 x 8 patch and its reflect halo): pw1 [hid x c_in] . [c_in x 100 halo
//     positions] and pw3 [c_out x hid] . [hid x 64 pixels] on the same matrix-core helper, the depthwise 3 x 3 between them on the
//     vector ALU (one thread = one hidden channel's output row: 30 inputs, 72 fmas), hidden activations in the LDS the dead banks
//     of levels 0-2 leave free.
// Correctness never depends on dispatch order or workgroup placement; PROGRESS needs every workgroup of the grid resident at once
// (a workgroup spins on its neighbours): the host checks the grid against the occupancy the runtime reports (capped at 5 workgroups per
// CU: the API answers one high when the SGPR file is the limit) and returns HS_ERR_UNSUPPORTED otherwise -- the caller then takes the
// per-level launches.  Every spin is bounded: a workgroup that waits longer than ~0.2 s raises the workspace's error word and goes on
// with what it has (wrong logits, no hang).
#include "hs_common.h"

namespace hs {

constexpr int KC_THREADS = 256, KC_WAVES = 4;
// LDS-DMA pieces (1 KB: one wave-instruction) every wave issues per bank: fixed counts, so that the vmcnt arithmetic below is a
// compile-time constant; pieces past a bank's end re-read its last 16 bytes into a dump area.  24 / 12 / 4 / 12 KB of bank at most.
constexpr int KC_P0 = 6, KC_P1 = 3, KC_P2 = 1, KC_P3 = 3;
constexpr int KC_SK3 = 3;                     // 4-byte gathers per thread for the inverted residual's skip halo (c_skip x 100 <= 768)
constexpr int KC_SPIN_LIMIT = 1 << 17;
constexpr int KC_HALO = 100, KC_LDT = 112;    // halo positions of an 8 x 8 patch, and their row stride (7 tiles of 16)

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
using kc4 = __attribute__((ext_vector_type(4))) float;
#define KC_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct KcLevel {
    const float* __restrict__ skip;      // (B, c_skip, fh << l, fw << l)
    const float* __restrict__ bank;      // patch-major, row stride ld
    const float* __restrict__ scale;     // epilogue (null: none)
    const float* __restrict__ shift;
    long ld;
    int c_skip, cout, cin, act;
    int pieces;                          // 1 KB pieces of the bank
    int kp;                              // c_in rounded up to whole k-steps of 4
    int nm, ksplit;                      // 16-row output tiles; K range split over that many waves (nm * ksplit <= 4)
    float step_x, step_y;                // linspace steps of the level's coordinate channels
};

struct KcIr {                            // the inverted residual behind level 2 (cout == 0: absent)
    const float* __restrict__ skip;
    const float* __restrict__ bank;
    const float* __restrict__ s1; const float* __restrict__ b1;
    const float* __restrict__ s2; const float* __restrict__ b2;
    const float* __restrict__ s3; const float* __restrict__ b3;
    long ld;
    int c_skip, cin, hid, cout, pieces, kp;
    float step_x, step_y;
};

struct KcArgs {
    KcLevel L[3];
    KcIr R;
    int B, fh, fw;
    u64* x0;             // [cells][c0] granules
    u64* x1;             // [cells][4][c1]
    u64* x2;             // [cells][16][c2]   (with the inverted residual)
    unsigned* err;
    float* __restrict__ y;
};

// LDS map (bytes).  rows[k]: 64-float arrays -- levels 0-2 scale / shift (0-5), inverted residual s1 b1 s2 b2 s3 b3 (6-11)
struct KcLds { int bank[4], dump, rows, xin0, sk1, sk2, sk3, genw, own0, own1, own2, part, tab, total;
               int nb0, xin1, nb1, xin2, alias1_end;            // inside bank 0's region once level 0 is done
               int h1, xt, nb2, h2, alias2_end; };              // inside banks 0-2 once level 2 is done

__host__ __device__ inline KcLds kc_lds_map(const KcLevel* L, const KcIr& R) {
    KcLds m;
    const bool ir = R.cout > 0;
    int o = 0;
    for (int l = 0; l < 3; ++l) { m.bank[l] = o; o += L[l].pieces * 1024; }
    m.bank[3] = o; o += ir ? R.pieces * 1024 : 0;
    m.dump = o; o += 1024;
    m.rows = o; o += (ir ? 12 : 6) * 64 * 4;
    m.xin0 = o; o += 256 * 4;                 // level 0's input vector; the 4-byte DMA writes whole waves
    m.sk1 = o; o += 256 * 4;                  // skip pixels of the later levels as they land (copied into the stage inputs later)
    m.sk2 = o; o += 256 * 4;
    m.sk3 = o; o += ir ? KC_SK3 * 256 * 4 : 0;
    m.genw = o; o += 64 * 4;
    // own0 / own1 (this cell's level-0 / level-1 outputs for its own next level): first written after every LDS-DMA of the workgroup
    // has landed -- without the inverted residual (whose operands are requested later) they can live in the dump area
    if (!ir && 256 + 4 * L[1].cout * 4 <= 1024) { m.own0 = m.dump; m.own1 = m.dump + 256; }
    else { m.own0 = o; o += 64 * 4; m.own1 = o; o += 4 * 64 * 4; }
    m.own2 = o; o += ir ? 16 * L[2].cout * 4 : 0;
    m.part = o; o += KC_WAVES * 256 * 4;      // partial 16 x 16 tiles of the K-split products
    m.tab = o; o += ir ? 20 * 8 * 4 : 0;      // per halo row / column of the inverted residual: bilinear taps into level 2's window, coordinate
    m.total = o;
    int q = m.bank[0];
    m.nb0 = q; q += 9 * L[0].cout * 4;
    m.xin1 = q; q += L[1].kp * 4 * 4;
    m.nb1 = q; q += 16 * L[1].cout * 4;
    m.xin2 = q; q += L[2].kp * 16 * 4;
    m.alias1_end = q;                         // host: <= bank[1]
    q = m.bank[0];
    m.h1 = q; q += ir ? R.hid * KC_LDT * 4 : 0;
    m.xt = q; m.h2 = q; q += ir ? R.kp * KC_LDT * 4 : 0;
    m.nb2 = q; q += ir ? 36 * L[2].cout * 4 : 0;
    m.alias2_end = q;                         // host: <= bank[3], and h2 (hid x 64 floats) <= xt + nb2
    return m;
}

__device__ __forceinline__ int kc_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One 16 x 16 tile of W . X over k-steps [ks0, ks1): A[row][k] = w[k] (the lane's row of W, already offset by the lane's k mod 4),
// B[k][col] = x[k * ldx] (likewise).  Rows of X past c_in are zero (the stage builders pad to whole k-steps), so what A reads past a
// row's end -- the next row, or the piece's padding: finite bank values -- does not matter.
__device__ __forceinline__ kc4 kc_mma(const float* __restrict__ w, const float* __restrict__ x, int ldx, int ks0, int ks1, kc4 acc) {
    int ks = ks0;
    for (; ks + 4 <= ks1; ks += 4) {          // four k-steps' operands in flight before the first product
        const float a0 = w[4 * ks], a1 = w[4 * ks + 4], a2 = w[4 * ks + 8], a3 = w[4 * ks + 12];
        const float b0 = x[4 * ks * ldx], b1 = x[(4 * ks + 4) * ldx], b2 = x[(4 * ks + 8) * ldx], b3 = x[(4 * ks + 12) * ldx];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
    }
    for (; ks < ks1; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * ks], x[4 * ks * ldx], acc, 0, 0, 0);
    return acc;
}

// A k = 1 level's products: tile (wave % nm) of the nm output-row tiles, K part (wave / nm) of ksplit; the partial tile goes to
// part[wave][16 rows][16 columns].  x: [kp][npix] (npix = 1, 4, 16: the tile's columns repeat the pixels).
__device__ __forceinline__ void kc_level_products(const KcLevel& lv, const float* __restrict__ wl, const float* __restrict__ x, int npix,
                                                  float* __restrict__ part, int wave, int lane) {
    const int lrow = lane & 15, lk = lane >> 4;
    const int tile = wave % lv.nm, kpart = wave / lv.nm;
    if (kpart >= lv.ksplit) return;                                              // (uniform: nm = 3 leaves a wave idle)
    const int kst = lv.kp >> 2, kper = (kst + lv.ksplit - 1) / lv.ksplit;
    const int ks0 = kpart * kper, ks1 = min(kst, ks0 + kper);
    const int row = min(tile * 16 + lrow, lv.cout - 1);
    kc4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = kc_mma(wl + row * lv.cin + lk, x + lk * npix + (lrow & (npix - 1)), npix, ks0, ks1, acc);
    float* p = part + wave * 256 + (4 * lk) * 16 + lrow;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r * 16] = acc[r];
}
// output (o, px) of a level from the partial tiles: K parts summed in ascending order
__device__ __forceinline__ float kc_level_output(const KcLevel& lv, const float* __restrict__ part, int o, int px) {
    const int tile = o >> 4, row = o & 15;
    float acc = part[tile * 256 + row * 16 + px];
    for (int kpart = 1; kpart < lv.ksplit; ++kpart) acc += part[(kpart * lv.nm + tile) * 256 + row * 16 + px];
    return acc;
}

// The granules at g[q] until every NEEDED one carries tag == gen (need[q] false: the slot is a filler -- g[q] is a valid address all
// the same, so that every load is issued unconditionally and all NQ are in flight together; behind `if (need)` each load sat in its
// own branch with a vmcnt(0) at the join).  Each wave polls its own granules and leaves when all of ITS lanes are served.
// false: the wait was abandoned (error word raised).
template <int NQ>
__device__ __forceinline__ bool kc_gather(gu64* (&g)[NQ], const bool (&need)[NQ], unsigned gen, unsigned (&v)[NQ], gu32* err, unsigned code) {
    for (int spins = 0;; ++spins) {
        u64 x[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) x[q] = __hip_atomic_load(g[q], KC_RLX_AGENT);
        bool ok = true;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            v[q] = (unsigned)x[q];
            ok &= !need[q] || (unsigned)(x[q] >> 32) == gen;
        }
        if (__all(ok)) return true;
        if (spins >= KC_SPIN_LIMIT) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, code, KC_RLX_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

template <bool IR>
__global__ __launch_bounds__(KC_THREADS)
void decoder_chain_kernel(KcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = a.fh, fw = a.fw;
    const int cell = (int)blockIdx.x;
    const int pib = cell / fw, j = cell - pib * fw, b = pib / fh, i = pib - b * fh;
    const int c0 = a.L[0].cout, c1 = a.L[1].cout, c2 = a.L[2].cout;
    const int cin0 = a.L[0].cin;
    const int cs0 = a.L[0].c_skip, cs1 = a.L[1].c_skip, cs2 = a.L[2].c_skip;
    const KcLds M = kc_lds_map(a.L, a.R);
    const float* wl0 = reinterpret_cast<const float*>(lds + M.bank[0]);
    const float* wl1 = reinterpret_cast<const float*>(lds + M.bank[1]);
    const float* wl2 = reinterpret_cast<const float*>(lds + M.bank[2]);
    float* rows = reinterpret_cast<float*>(lds + M.rows);
    float* xin0 = reinterpret_cast<float*>(lds + M.xin0);
    float* own0 = reinterpret_cast<float*>(lds + M.own0);
    float* own1 = reinterpret_cast<float*>(lds + M.own1);
    float* own2 = reinterpret_cast<float*>(lds + M.own2);
    float* part = reinterpret_cast<float*>(lds + M.part);
    float* nb0 = reinterpret_cast<float*>(lds + M.nb0);
    float* xin1 = reinterpret_cast<float*>(lds + M.xin1);
    float* nb1 = reinterpret_cast<float*>(lds + M.nb1);
    float* xin2 = reinterpret_cast<float*>(lds + M.xin2);
    // every shared word through GLOBAL (address space 1) agent-scope accesses, never flat ones
    gu64* const gx0 = (gu64*)a.x0;
    gu64* const gx1 = (gu64*)a.x1;
    gu64* const gx2 = (gu64*)a.x2;
    gu32* const gerr = (gu32*)a.err;
    const int H1 = 2 * fh, W1 = 2 * fw, H2 = 4 * fh, W2 = 4 * fw, H3 = 8 * fh, W3 = 8 * fw;
    // @stamp 0

    // ---------------------------------------------------------------- every HBM load of the workgroup, level 0's operands first
    // ALL of them by LDS-DMA: no load targets a register, so the only vmcnt waits in this kernel are the ones written below.  (With
    // register loads pending beside LDS-DMA the compiler's wait insertion assumes out-of-order completion and puts a vmcnt(0) in front
    // of the first use of any loaded register -- the two groups then become two serialised round trips.)  The counts per group are
    // compile-time constants PER WAVE (the BatchNorm rows are spread over the waves; the vmcnt arithmetic is per wave).
    auto dma_bank = [&](const float* bank, long ld, int region, int npieces, int q) {
        const int c = wave + KC_WAVES * q;                                       // (uniform)
        const unsigned last16 = (unsigned)ld * 4u - 16u;
        const unsigned char* gb = reinterpret_cast<const unsigned char*>(bank + (size_t)cell * (size_t)ld);
        const unsigned off = min((unsigned)(c * 1024 + lane * 16), last16);
        const int dst = c < npieces ? region + c * 1024 : M.dump;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + off),
                                         (__attribute__((address_space(3))) void*)(lds + dst), 16, 0, 0);
    };
    // one float per lane: lane l of the wave -> dst + 4 l
    auto dma_word = [&](const float* src, int dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + dst), 4, 0, 0);
    };
    auto dma_row = [&](const float* p, int n, int k) {                           // a 64-float row array (null: left to the unit-row fill below)
        dma_word(p ? p + min(lane, n - 1) : a.L[0].skip, p ? M.rows + k * 256 : M.dump);
    };
    // group 0: level 0's bank and its skip pixel (thread t -> input channel t: 2 coordinates first); wave 0 adds the BatchNorm rows and
    // the generation word
#pragma unroll
    for (int q = 0; q < KC_P0; ++q) dma_bank(a.L[0].bank, a.L[0].ld, M.bank[0], a.L[0].pieces, q);
    dma_word(a.L[0].skip + ((((size_t)b * cs0 + (size_t)kc_clamp(tid - 2, 0, cs0 - 1)) * fh + i) * fw + j), M.xin0 + wave * 256);
    if (wave == 0) {
        dma_row(a.L[0].scale, c0, 0);
        dma_row(a.L[0].shift, c0, 1);
        // the generation: the tag (upper half) of this cell's first granule as the PREVIOUS launch left it
        dma_word(reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.x0) + ((size_t)cell * c0 * 8 + 4)), M.genw);
    }
    __builtin_amdgcn_sched_barrier(0);
    // group 1: the later levels' banks, skip pixels and BatchNorm rows
#pragma unroll
    for (int q = 0; q < KC_P1; ++q) dma_bank(a.L[1].bank, a.L[1].ld, M.bank[1], a.L[1].pieces, q);
#pragma unroll
    for (int q = 0; q < KC_P2; ++q) dma_bank(a.L[2].bank, a.L[2].ld, M.bank[2], a.L[2].pieces, q);
    {
        const int c = kc_clamp(tid >> 2, 0, cs1 - 1), px = tid & 3;
        dma_word(a.L[1].skip + ((((size_t)b * cs1 + c) * H1 + (2 * i + (px >> 1))) * W1 + (2 * j + (px & 1))), M.sk1 + wave * 256);
    }
    {
        const int c = kc_clamp(tid >> 4, 0, cs2 - 1), px = tid & 15;
        dma_word(a.L[2].skip + ((((size_t)b * cs2 + c) * H2 + (4 * i + (px >> 2))) * W2 + (4 * j + (px & 3))), M.sk2 + wave * 256);
    }
    if (wave == 1) {
        dma_row(a.L[1].scale, c1, 2); dma_row(a.L[1].shift, c1, 3);
        dma_row(a.L[2].scale, c2, 4); dma_row(a.L[2].shift, c2, 5);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (IR) {
        // while group 0 is in flight: the inverted residual's tap table -- per halo row u / column v of the patch: the REFLECTED image
        // position's bilinear taps into level 2's 6 x 6 window {offset 0, offset 1, l0, l1} and its coordinate value
        if (tid < 20) {
            const bool isrow = tid < 10;
            const int k = isrow ? tid : tid - 10;
            const int p = isrow ? pad_index(8 * i + k - 1, H3, HS_PAD_REFLECT) : pad_index(8 * j + k - 1, W3, HS_PAD_REFLECT);
            const Tap t = bilinear_tap(p, 0.5f, isrow ? H2 : W2);
            float* d = reinterpret_cast<float*>(lds + M.tab) + tid * 8;
            const int base = isrow ? 4 * i - 1 : 4 * j - 1;
            d[0] = __int_as_float(t.i0 - base); d[1] = __int_as_float(t.i1 - base); d[2] = t.l0; d[3] = t.l1;
            d[4] = isrow ? linspace_pm1(p, H3, a.R.step_y) : linspace_pm1(p, W3, a.R.step_x);
        }
    }
    // @stamp 1
    // group 0 has landed when at most group 1's operations are outstanding (vector-memory operations complete in issue order).
    // Group 1 per wave: banks KC_P1 + KC_P2, skip gathers 2; wave 1 adds the four BatchNorm rows of levels 1 and 2.  (The inverted
    // residual's operands are requested later, behind level 0's publishing stores: at the top they delayed level 0 by ~2.7 k cycles of
    // DMA issue -- profiles/round5_chain_v2_*_v5.txt -- and they have until level 3 to land.)
    static_assert(KC_P1 + KC_P2 + 2 == 6, "the counts in the s_waitcnt's below");
    if (wave == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    // @stamp 2
    {   // level 0's input vector: the coordinates over what the gather put in slots 0 and 1, zeros up to the next whole k-step (each
        // slot was written by the DMA of the wave that owns the thread: ordered by that wave's wait)
        if (tid == 0) xin0[0] = linspace_pm1(j, fw, a.L[0].step_x);
        if (tid == 1) xin0[1] = linspace_pm1(i, fh, a.L[0].step_y);
        if (tid >= cin0 && tid < a.L[0].kp) xin0[tid] = 0.0f;
        if (!a.L[0].scale && tid < 64) { rows[tid] = 1.0f; rows[64 + tid] = 0.0f; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");              // group 1 stays in flight across the barrier
    const unsigned g_prev = *reinterpret_cast<const unsigned*>(lds + M.genw);
    unsigned gen = (unsigned)__builtin_amdgcn_readfirstlane((int)g_prev) + 1u;
    gen = gen ? gen : 1u;
    // @stamp 3

    // ---------------------------------------------------------------- level 0: c0 outputs of one pixel
    kc_level_products(a.L[0], wl0, xin0, 1, part, wave, lane);
    // group 1 (the later levels' operands) has had level 0's products to land: the wait for it goes HERE, in front of the publishing
    // stores -- behind them a vmcnt(0) would also wait for the stores' own write-through round trip (stores count in vmcnt on
    // gfx9-class hardware), which nobody in this workgroup needs
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (every wave's share of group 1 has landed; the barrier makes it the workgroup's, the partial tiles visible, level 0's bank free)
    if (tid < c0) {
        const float v = apply_act(fmaf(kc_level_output(a.L[0], part, tid, 0), rows[tid], rows[64 + tid]), a.L[0].act);
        own0[tid] = v;
        __hip_atomic_store(gx0 + (size_t)cell * c0 + tid, ((u64)gen << 32) | (u64)__float_as_uint(v), KC_RLX_AGENT);
    }
    // @stamp 4
    if constexpr (IR) {
        // the inverted residual's operands (bank, skip feature on the reflect halo, BatchNorm rows): requested now, they land while this
        // workgroup waits for its neighbours' level-0 outputs; every later __syncthreads() / poll wait covers them
#pragma unroll
        for (int q = 0; q < KC_P3; ++q) dma_bank(a.R.bank, a.R.ld, M.bank[3], a.R.pieces, q);
        {
            const float* skb = a.R.skip + (size_t)b * a.R.c_skip * H3 * W3;      // (uniform base; 32-bit element offsets below)
#pragma unroll
            for (int q = 0; q < KC_SK3; ++q) {
                const int e = min(tid + q * KC_THREADS, a.R.c_skip * KC_HALO - 1);
                const int c = e / KC_HALO, pos = e - c * KC_HALO, u = pos / 10, v = pos - u * 10;
                const int yy = pad_index(8 * i + u - 1, H3, HS_PAD_REFLECT), xx = pad_index(8 * j + v - 1, W3, HS_PAD_REFLECT);
                dma_word(skb + (unsigned)((c * H3 + yy) * W3 + xx), M.sk3 + (q * KC_THREADS + wave * 64) * 4);
            }
        }
        if (wave == 2) { dma_row(a.R.s1, a.R.hid, 6); dma_row(a.R.b1, a.R.hid, 7); dma_row(a.R.s2, a.R.hid, 8); dma_row(a.R.b2, a.R.hid, 9); }
        if (wave == 3) { dma_row(a.R.s3, a.R.cout, 10); dma_row(a.R.b3, a.R.cout, 11); }
    }
    if (tid < 64) {   // levels without an epilogue: unit rows (read two barriers further down)
        if (!a.L[1].scale) { rows[128 + tid] = 1.0f; rows[192 + tid] = 0.0f; }
        if (!a.L[2].scale) { rows[256 + tid] = 1.0f; rows[320 + tid] = 0.0f; }
    }
    // @stamp 5

    // ---------------------------------------------------------------- level 0 -> 1: the 3 x 3 cells around this one (clamped at the border)
    {
        constexpr int NQ = 2;                                                    // 8 c0 <= 512 granules
        gu64* g[NQ]; unsigned v[NQ]; int dst[NQ]; bool need[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = min(tid + q * KC_THREADS, 8 * c0 - 1);                 // surplus lanes shadow the last element
            need[q] = tid + q * KC_THREADS < 8 * c0;
            const int n8 = e / c0, ch = e - n8 * c0;
            const int n = n8 + (n8 >= 4 ? 1 : 0);
            const int ci = kc_clamp(i + n / 3 - 1, 0, fh - 1), cj = kc_clamp(j + n % 3 - 1, 0, fw - 1);
            g[q] = gx0 + ((size_t)((b * fh + ci) * fw + cj) * c0 + ch);
            dst[q] = n * c0 + ch;
            v[q] = 0u;
        }
        kc_gather<NQ>(g, need, gen, v, gerr, 1u);
        // @stamp 6
        __syncthreads();                                                         // own0 complete; level 0's bank region is free (nb0 / xin1 alias it)
#pragma unroll
        for (int q = 0; q < NQ; ++q) if (need[q]) nb0[dst[q]] = __uint_as_float(v[q]);
        if (tid < c0) nb0[4 * c0 + tid] = own0[tid];
        // level 1's input that does not depend on level 0: coordinates, the skip pixels, the zero rows up to a whole k-step
        if (tid < cs1 * 4) xin1[8 + tid] = reinterpret_cast<const float*>(lds + M.sk1)[tid];
        if (tid < 8) {
            const int px = tid & 3;
            xin1[tid] = tid < 4 ? linspace_pm1(2 * j + (px & 1), W1, a.L[1].step_x) : linspace_pm1(2 * i + (px >> 1), H1, a.L[1].step_y);
        }
        if (tid < (a.L[1].kp - a.L[1].cin) * 4) xin1[a.L[1].cin * 4 + tid] = 0.0f;
    }
    __syncthreads();
    // bilinear 2x of level 0 (align_corners=False; the operation order of hs_common.h's stage_value): c0 channels x 4 pixels
    for (int e = tid; e < c0 * 4; e += KC_THREADS) {
        const int cp = e >> 2, px = e & 3;
        const Tap ty = bilinear_tap(2 * i + (px >> 1), 0.5f, fh), tx = bilinear_tap(2 * j + (px & 1), 0.5f, fw);
        const float* r0 = nb0 + ((ty.i0 - i + 1) * 3) * c0 + cp;
        const float* r1 = nb0 + ((ty.i1 - i + 1) * 3) * c0 + cp;
        const int x0i = (tx.i0 - j + 1) * c0, x1i = (tx.i1 - j + 1) * c0;
        const float top = tx.l0 * r0[x0i] + tx.l1 * r0[x1i];
        const float bot = tx.l0 * r1[x0i] + tx.l1 * r1[x1i];
        xin1[(2 + cs1 + cp) * 4 + px] = ty.l0 * top + ty.l1 * bot;
    }
    __syncthreads();
    // @stamp 7

    // ---------------------------------------------------------------- level 1: c1 outputs x 4 pixels
    kc_level_products(a.L[1], wl1, xin1, 4, part, wave, lane);
    __syncthreads();
    for (int idx = tid; idx < c1 * 4; idx += KC_THREADS) {
        const int o = idx >> 2, px = idx & 3;
        const float v = apply_act(fmaf(kc_level_output(a.L[1], part, o, px), rows[128 + o], rows[192 + o]), a.L[1].act);
        own1[px * c1 + o] = v;
        __hip_atomic_store(gx1 + ((size_t)cell * 4 + px) * c1 + o, ((u64)gen << 32) | (u64)__float_as_uint(v), KC_RLX_AGENT);
    }
    // @stamp 8

    // ---------------------------------------------------------------- level 1 -> 2: the 4 x 4 level-1 pixels [2i - 1, 2i + 2] x [2j - 1, 2j + 2]
    {
        constexpr int NQ = 4;                                                    // 16 c1 <= 1024
        gu64* g[NQ]; unsigned v[NQ]; int dst[NQ]; int own[NQ]; bool need[NQ], live[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = min(tid + q * KC_THREADS, 16 * c1 - 1);
            live[q] = tid + q * KC_THREADS < 16 * c1;
            const int wp = e / c1, ch = e - wp * c1;
            const int yl = kc_clamp(2 * i - 1 + (wp >> 2), 0, H1 - 1), xl = kc_clamp(2 * j - 1 + (wp & 3), 0, W1 - 1);
            const int ci = yl >> 1, cj = xl >> 1, px = (yl & 1) * 2 + (xl & 1);
            dst[q] = wp * c1 + ch;
            own[q] = px * c1 + ch;
            need[q] = live[q] && !(ci == i && cj == j);                          // this cell's own pixels come from LDS
            g[q] = gx1 + (((size_t)((b * fh + ci) * fw + cj) * 4 + px) * c1 + ch);
            v[q] = 0u;
        }
        kc_gather<NQ>(g, need, gen, v, gerr, 2u);
        // @stamp 9
        __syncthreads();                                                         // own1 complete
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (live[q]) nb1[dst[q]] = need[q] ? __uint_as_float(v[q]) : own1[own[q]];
        }
        if (tid < cs2 * 16) xin2[32 + tid] = reinterpret_cast<const float*>(lds + M.sk2)[tid];
        if (tid < 32) {
            const int px = tid & 15;
            xin2[tid] = tid < 16 ? linspace_pm1(4 * j + (px & 3), W2, a.L[2].step_x) : linspace_pm1(4 * i + (px >> 2), H2, a.L[2].step_y);
        }
        if (tid < (a.L[2].kp - a.L[2].cin) * 16) xin2[a.L[2].cin * 16 + tid] = 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < c1 * 16; e += KC_THREADS) {
        const int cp = e >> 4, px = e & 15;
        const Tap ty = bilinear_tap(4 * i + (px >> 2), 0.5f, H1), tx = bilinear_tap(4 * j + (px & 3), 0.5f, W1);
        const float* r0 = nb1 + ((ty.i0 - (2 * i - 1)) * 4) * c1 + cp;
        const float* r1 = nb1 + ((ty.i1 - (2 * i - 1)) * 4) * c1 + cp;
        const int x0i = (tx.i0 - (2 * j - 1)) * c1, x1i = (tx.i1 - (2 * j - 1)) * c1;
        const float top = tx.l0 * r0[x0i] + tx.l1 * r0[x1i];
        const float bot = tx.l0 * r1[x0i] + tx.l1 * r1[x1i];
        xin2[(2 + cs2 + cp) * 16 + px] = ty.l0 * top + ty.l1 * bot;
    }
    __syncthreads();
    // @stamp 10

    // ---------------------------------------------------------------- level 2: c2 outputs x 16 pixels
    kc_level_products(a.L[2], wl2, xin2, 16, part, wave, lane);
    __syncthreads();
    for (int idx = tid; idx < c2 * 16; idx += KC_THREADS) {
        const int o = idx >> 4, px = idx & 15;
        const float v = apply_act(fmaf(kc_level_output(a.L[2], part, o, px), rows[256 + o], rows[320 + o]), a.L[2].act);
        if constexpr (IR) {
            own2[px * c2 + o] = v;
            __hip_atomic_store(gx2 + ((size_t)cell * 16 + px) * c2 + o, ((u64)gen << 32) | (u64)__float_as_uint(v), KC_RLX_AGENT);
        } else {
            a.y[(((size_t)b * c2 + o) * H2 + (4 * i + (px >> 2))) * W2 + (4 * j + (px & 3))] = v;      // plain stores: the next launch reads them
        }
    }
    // @stamp 11
    if constexpr (IR) {
        // ================================================================ the inverted residual on this cell's 8 x 8 patch (Op C)
        const int hid = a.R.hid, cin3 = a.R.cin, co3 = a.R.cout, cs3 = a.R.c_skip;
        const float* wl3 = reinterpret_cast<const float*>(lds + M.bank[3]);
        float* h1 = reinterpret_cast<float*>(lds + M.h1);
        float* xt = reinterpret_cast<float*>(lds + M.xt);
        float* nb2 = reinterpret_cast<float*>(lds + M.nb2);
        float* h2 = reinterpret_cast<float*>(lds + M.h2);
        const float* sk3 = reinterpret_cast<const float*>(lds + M.sk3);
        const float *s1 = rows + 6 * 64, *b1 = rows + 7 * 64, *s2 = rows + 8 * 64, *b2 = rows + 9 * 64, *s3 = rows + 10 * 64, *b3 = rows + 11 * 64;
        // ------------------------------------------------------------ level 2 -> 3: the 6 x 6 level-2 pixels [4i - 1, 4i + 4] x [4j - 1, 4j + 4]
        {
            constexpr int NQ = 3;                                                // 36 c2 <= 768
            gu64* g[NQ]; unsigned v[NQ]; int dst[NQ]; int own[NQ]; bool need[NQ], live[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int e = min(tid + q * KC_THREADS, 36 * c2 - 1);
                live[q] = tid + q * KC_THREADS < 36 * c2;
                const int wp = e / c2, ch = e - wp * c2;
                const int wr = wp / 6, wc = wp - wr * 6;
                const int yl = kc_clamp(4 * i - 1 + wr, 0, H2 - 1), xl = kc_clamp(4 * j - 1 + wc, 0, W2 - 1);
                const int ci = yl >> 2, cj = xl >> 2, px = (yl & 3) * 4 + (xl & 3);
                dst[q] = wp * c2 + ch;
                own[q] = px * c2 + ch;
                need[q] = live[q] && !(ci == i && cj == j);
                g[q] = gx2 + (((size_t)((b * fh + ci) * fw + cj) * 16 + px) * c2 + ch);
                v[q] = 0u;
            }
            kc_gather<NQ>(g, need, gen, v, gerr, 3u);
            // @stamp 12
            __syncthreads();                                                     // own2 complete; the banks of levels 0-2 are free (h1 / xt / nb2 alias them)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (live[q]) nb2[dst[q]] = need[q] ? __uint_as_float(v[q]) : own2[own[q]];
            }
            // the halo tile's rows that do not depend on level 2: coordinates, skip feature (both at the REFLECTED position: the block pads
            // the stage input, hyperseg_v1_0.py:339-345), zero rows up to a whole k-step; unit rows for absent BatchNorms
            const float* tab = reinterpret_cast<const float*>(lds + M.tab);
            for (int e = tid; e < 2 * KC_HALO; e += KC_THREADS) {
                const int c = e >= KC_HALO ? 1 : 0, pos = e - c * KC_HALO, u = pos / 10, vv = pos - u * 10;
                xt[c * KC_LDT + pos] = c == 0 ? tab[(10 + vv) * 8 + 4] : tab[u * 8 + 4];
            }
            for (int e = tid; e < cs3 * KC_HALO; e += KC_THREADS) {
                const int c = e / KC_HALO, pos = e - c * KC_HALO;
                xt[(2 + c) * KC_LDT + pos] = sk3[e];
            }
            for (int e = tid; e < (a.R.kp - cin3) * KC_LDT; e += KC_THREADS) xt[cin3 * KC_LDT + e] = 0.0f;
            if (tid < 64) {
                if (!a.R.s1) { rows[6 * 64 + tid] = 1.0f; rows[7 * 64 + tid] = 0.0f; }
                if (!a.R.s2) { rows[8 * 64 + tid] = 1.0f; rows[9 * 64 + tid] = 0.0f; }
                if (!a.R.s3) { rows[10 * 64 + tid] = 1.0f; rows[11 * 64 + tid] = 0.0f; }
            }
        }
        __syncthreads();
        // bilinear 2x of level 2 at the halo positions: thread = (position, channel group); the position's taps come from the table
        {
            const float* tab = reinterpret_cast<const float*>(lds + M.tab);
            for (int e = tid; e < c2 * KC_HALO; e += KC_THREADS) {
                const int cp = e / KC_HALO, pos = e - cp * KC_HALO, u = pos / 10, vv = pos - u * 10;
                const float* ty = tab + u * 8;
                const float* tx = tab + (10 + vv) * 8;
                const float* r0 = nb2 + (__float_as_int(ty[0]) * 6) * c2 + cp;
                const float* r1 = nb2 + (__float_as_int(ty[1]) * 6) * c2 + cp;
                const int x0i = __float_as_int(tx[0]) * c2, x1i = __float_as_int(tx[1]) * c2;
                const float top = tx[2] * r0[x0i] + tx[3] * r0[x1i];
                const float bot = tx[2] * r1[x0i] + tx[3] * r1[x1i];
                xt[(2 + cs3 + cp) * KC_LDT + pos] = ty[2] * top + ty[3] * bot;
            }
        }
        __syncthreads();
        // @stamp 13
        // ------------------------------------------------------------ pw1: h1 = relu6(bn1(W1 . x)) on the 100 halo positions (7 column tiles)
        // wave w takes column tiles w and w + 4; all of its (row tile, column tile) accumulators advance together through K -- up to
        // eight independent matrix-core chains per wave (one tile at a time was a chain of dependent products, 40+ cycles each, with
        // the LDS reads of every step exposed: 8.6 k cycles for 36 products per wave, profiles/round5_chain_v2_*_v5.txt)
        {
            const int lrow = lane & 15, lk = lane >> 4;
            const int kst = a.R.kp >> 2, mt = (hid + 15) >> 4;
            const int n0 = wave, n1 = wave + KC_WAVES;
            const bool two = n1 < KC_LDT / 16;                                   // (uniform)
            kc4 acc[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[q][m] = kc4{0.f, 0.f, 0.f, 0.f};
            const float* wr[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) wr[m] = wl3 + min(m * 16 + lrow, hid - 1) * cin3 + lk;
            const float* x0 = xt + lk * KC_LDT + n0 * 16 + lrow;
            const float* x1 = xt + lk * KC_LDT + (two ? n1 : n0) * 16 + lrow;
#pragma unroll 2
            for (int ks = 0; ks < kst; ++ks) {
                float av[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) av[m] = wr[m][4 * ks];
                const float b0 = x0[4 * ks * KC_LDT], b1 = x1[4 * ks * KC_LDT];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (m < mt) {
                        acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], b0, acc[0][m], 0, 0, 0);
                        if (two) acc[1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], b1, acc[1][m], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q == 0 || two) {
                    const int n = q == 0 ? n0 : n1;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int h = m * 16 + 4 * lk + r;
                            if (m < mt && h < hid) h1[h * KC_LDT + n * 16 + lrow] = fminf(fmaxf(fmaf(acc[q][m][r], s1[h], b1[h]), 0.0f), 6.0f);
                        }
                    }
                }
            }
        }
        __syncthreads();
        // @stamp 14
        // ------------------------------------------------------------ depthwise 3 x 3 (valid) + bn2 + relu6: thread = (hidden channel, output row)
        {
            const float* taps = wl3 + cin3 * hid;
            for (int e = tid; e < hid * 8; e += KC_THREADS) {
                const int h = e >> 3, r = e & 7;
                const float* src = h1 + h * KC_LDT + r * 10;
                float in[3][10], t[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) t[q] = taps[h * 9 + q];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 10; ++cc) in[rr][cc] = src[rr * 10 + cc];
                const float sc = s2[h], sh = b2[h];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    float acc = 0.0f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc = fmaf(t[ky * 3 + kx], in[ky][cc + kx], acc);
                    h2[h * 64 + r * 8 + cc] = fminf(fmaxf(fmaf(acc, sc, sh), 0.0f), 6.0f);
                }
            }
            // (h2 aliases the halo tile: every wave passed the barrier behind pw1, the tile is dead)
        }
        __syncthreads();
        // @stamp 15
        // ------------------------------------------------------------ pw3 + bn3: wave = 16 pixels (two patch rows), plain stores
        {
            const int lrow = lane & 15, lk = lane >> 4;
            const float* w3 = wl3 + cin3 * hid + 9 * hid;
            const int kst = hid >> 2, mt = (co3 + 15) >> 4;
            for (int m = 0; m < mt; ++m) {
                const int row = min(m * 16 + lrow, co3 - 1);
                kc4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = kc_mma(w3 + row * hid + lk, h2 + lk * 64 + wave * 16 + lrow, 64, 0, kst, acc);
                const int px = wave * 16 + lrow;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = m * 16 + 4 * lk + r;
                    if (o < co3) a.y[(((size_t)b * co3 + o) * H3 + (8 * i + (px >> 3))) * W3 + (8 * j + (px & 7))] = fmaf(acc[r], s3[o], b3[o]);
                }
            }
        }
    }
    // @stamp 24
}

// resident workgroups of the kernel on the current device: (CUs, per CU)
template <bool IR>
static int kc_residency(size_t lds, int* cus, int* per_cu) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    int n = 0, c = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, decoder_chain_kernel<IR>, KC_THREADS, lds);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    *cus = c; *per_cu = n;
    return HS_OK;
}

static int kc_plan(int batch, int fh, int fw, const hs_k1_level* lv, int n_levels, const hs_chain_ir_level* ir, KcArgs* out,
                   size_t* lds_bytes, int64_t* ws_bytes) {
    if (!lv || n_levels != 3 || batch <= 0 || fh <= 0 || fw <= 0) return n_levels == 3 ? HS_ERR_BAD_ARG : HS_ERR_UNSUPPORTED;
    KcArgs a;
    a.B = batch; a.fh = fh; a.fw = fw;
    int prev = 0;
    const int maxp[3] = {KC_P0 * KC_WAVES, KC_P1 * KC_WAVES, KC_P2 * KC_WAVES};
    for (int l = 0; l < 3; ++l) {
        const hs_k1_level& s = lv[l];
        if (!s.skip || !s.bank || s.c_skip <= 0 || s.c_out <= 0 || (s.scale && !s.shift)) return HS_ERR_BAD_ARG;
        KcLevel& d = a.L[l];
        d.skip = s.skip; d.bank = s.bank; d.scale = s.scale; d.shift = s.shift; d.ld = (long)s.ld;
        d.c_skip = s.c_skip; d.cout = s.c_out; d.cin = 2 + s.c_skip + prev; d.act = s.act;
        const long hp = (long)d.cout * d.cin;
        if (s.ld < hp) return HS_ERR_BAD_ARG;
        if ((s.ld & 3) != 0 || ((uintptr_t)s.bank & 15) != 0 || s.ld * 4 >= (1l << 31)) return HS_ERR_UNSUPPORTED;
        const int npix = 1 << (2 * l);
        d.pieces = (int)((hp * 4 + 1023) / 1024);
        d.kp = (d.cin + 3) & ~3;
        // (what the matrix-core A fragments read past the LAST row's end -- up to 3 floats -- lies inside the last piece or in the next
        // LDS region: finite either way only if it was written; one spare piece-tail float per missing k is required)
        if (d.pieces > maxp[l] || d.cout > 64 || s.c_skip * npix > KC_THREADS || hp + 3 > (long)d.pieces * 256) return HS_ERR_UNSUPPORTED;
        d.nm = (d.cout + 15) / 16;
        d.ksplit = d.nm == 1 ? 4 : (d.nm == 2 ? 2 : 1);
        const int H = fh << l, W = fw << l;
        d.step_x = W > 1 ? 2.0f / (float)(W - 1) : 0.0f;
        d.step_y = H > 1 ? 2.0f / (float)(H - 1) : 0.0f;
        prev = d.cout;
    }
    if (a.L[0].kp > 256 || 8 * a.L[0].cout > 2 * KC_THREADS || 16 * a.L[1].cout > 4 * KC_THREADS) return HS_ERR_UNSUPPORTED;
    KcIr& r = a.R;
    r = KcIr{};
    if (ir) {
        if (!ir->skip || !ir->bank || ir->c_skip <= 0 || ir->hidden <= 0 || ir->c_out <= 0) return HS_ERR_BAD_ARG;
        if ((ir->s1 && !ir->b1) || (ir->s2 && !ir->b2) || (ir->s3 && !ir->b3)) return HS_ERR_BAD_ARG;
        r.skip = ir->skip; r.bank = ir->bank; r.ld = (long)ir->ld;
        r.s1 = ir->s1; r.b1 = ir->b1; r.s2 = ir->s2; r.b2 = ir->b2; r.s3 = ir->s3; r.b3 = ir->b3;
        r.c_skip = ir->c_skip; r.cin = 2 + ir->c_skip + prev; r.hid = ir->hidden; r.cout = ir->c_out;
        const long hp = (long)r.cin * r.hid + 9l * r.hid + (long)r.hid * r.cout;
        if (ir->ld < hp) return HS_ERR_BAD_ARG;
        if ((ir->ld & 3) != 0 || ((uintptr_t)ir->bank & 15) != 0 || ir->ld * 4 >= (1l << 31)) return HS_ERR_UNSUPPORTED;
        r.pieces = (int)((hp * 4 + 1023) / 1024);
        r.kp = (r.cin + 3) & ~3;
        if (r.pieces > KC_P3 * KC_WAVES || r.hid > 64 || (r.hid & 3) != 0 || r.cout > 64 || r.c_skip * KC_HALO > KC_SK3 * KC_THREADS ||
            36 * a.L[2].cout > 3 * KC_THREADS || hp + 3 > (long)r.pieces * 256) return HS_ERR_UNSUPPORTED;
        const int H = fh << 3, W = fw << 3;
        r.step_x = 2.0f / (float)(W - 1); r.step_y = 2.0f / (float)(H - 1);
    }
    const KcLds M = kc_lds_map(a.L, a.R);
    if (M.alias1_end > M.bank[1]) return HS_ERR_UNSUPPORTED;                       // the level-1 / level-2 stage buffers must stay inside level 0's bank region
    if (ir && (M.alias2_end > M.bank[3] || r.hid * 64 * 4 > M.alias2_end - M.xt)) return HS_ERR_UNSUPPORTED;
    if (M.total > 64 * 1024) return HS_ERR_UNSUPPORTED;
    const long cells = (long)batch * fh * fw;
    if (cells * 16 * 64 >= (1l << 31)) return HS_ERR_UNSUPPORTED;
    if (out) *out = a;
    if (lds_bytes) *lds_bytes = (size_t)M.total;
    if (ws_bytes) *ws_bytes = 256 + 8 * cells * ((long)a.L[0].cout + 4l * a.L[1].cout + (ir ? 16l * a.L[2].cout : 0l));
    return HS_OK;
}

}  // namespace hs

using namespace hs;

extern "C" int64_t hs_decoder_chain_workspace(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels,
                                              const hs_chain_ir_level* ir) {
    int64_t bytes = 0;
    const int st = kc_plan(batch, fh, fw, levels, n_levels, ir, nullptr, nullptr, &bytes);
    return st == HS_OK ? bytes : (int64_t)st;
}

extern "C" int hs_decoder_chain_fwd(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels,
                                    const hs_chain_ir_level* ir, void* workspace, float* y, void* stream) {
    KcArgs a;
    size_t lds = 0;
    int64_t ws = 0;
    int st = kc_plan(batch, fh, fw, levels, n_levels, ir, &a, &lds, &ws);
    if (st != HS_OK) return st;
    if (!workspace || !y || ((uintptr_t)workspace & 15) != 0) return HS_ERR_BAD_ARG;
    // every workgroup spins on its neighbours: the whole grid has to be resident at once
    int cus = 0, per_cu = 0;
    st = ir ? kc_residency<true>(lds, &cus, &per_cu) : kc_residency<false>(lds, &cus, &per_cu);
    if (st != HS_OK) return st;
    // The occupancy query answers one workgroup per CU high when the SGPR file is what limits a 256-thread workgroup (admitted =
    // min(API, 8, 800 / (ceil(sgprs / 16) * 16 + 16)): MI355X_MICROARCH.md, residency); for any SGPR count a kernel can have that
    // bound is >= 5, and LDS / VGPR limits are reported exactly -- so min(API, 5) workgroups per CU is never more than the hardware
    // admits.  (These kernels: ~100 SGPRs, < 64 VGPRs; 32-63 KB of LDS is what limits them at the decoder's shapes: 2-5 per CU --
    // HyperSeg-S's 1152 cells need the 5: its workgroup is 32 512 bytes.)
    const long cells = (long)batch * fh * fw;
    const int admitted = per_cu < 5 ? per_cu : 5;
    if (admitted < 1 || cells > (long)admitted * cus) return HS_ERR_UNSUPPORTED;
    unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
    a.err = reinterpret_cast<unsigned*>(w);
    a.x0 = reinterpret_cast<u64*>(w + 256);
    a.x1 = a.x0 + cells * (long)a.L[0].cout;
    a.x2 = a.x1 + cells * 4l * a.L[1].cout;
    a.y = y;
    if (ir) hipLaunchKernelGGL(decoder_chain_kernel<true>, dim3((unsigned)cells), dim3(KC_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(decoder_chain_kernel<false>, dim3((unsigned)cells), dim3(KC_THREADS), lds, (hipStream_t)stream, a);
    return launch_status();
}

// the three k = 1 levels alone (round 5's first form of the entry point)
extern "C" int64_t hs_k1_chain_workspace(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels) {
    return hs_decoder_chain_workspace(batch, fh, fw, levels, n_levels, nullptr);
}
extern "C" int hs_k1_chain_fwd(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels, void* workspace,
                               float* y, void* stream) {
    return hs_decoder_chain_fwd(batch, fh, fw, levels, n_levels, nullptr, workspace, y, stream);
}

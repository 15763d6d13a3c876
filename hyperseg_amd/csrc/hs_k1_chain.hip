// hs_k1_chain_fwd: the decoder's three coarse k = 1 levels (patches of 1, 2 and 4 pixels; hyperseg_v1_0.py:221-253 with the
// blocks of :486-498, 728-760) as ONE launch -- one workgroup per grid cell that walks level 0 -> 1 -> 2 and hands its level outputs
// to the neighbouring cells' workgroups INSIDE the launch.
//
// Why: as three launches these levels are three dependent latency chains of ~6 us each (index arithmetic -> bank + inputs from HBM ->
// LDS -> dot products -> store, then the graph edge) for 2.5 us of HBM traffic (profiles/round4_bench_kernel_stats.csv: 5.7 + 7.2 +
// 6.7 us; VERDICT r4 #1c).  Level l + 1 needs of level l only a one-pixel ring around its own cell (the 2x bilinear upsample of
// align_corners=False touches the 3 x 3 cells around a cell), everything else it reads -- its bank, its skip feature, its BatchNorm
// rows -- is known when the launch starts.  So here
//   * every HBM load of all three levels is issued at the top of the kernel: the banks by LDS-DMA (16 bytes per lane straight into
//     LDS, no registers), level 0's first, and level 0 starts as soon as ITS operands have landed (s_waitcnt vmcnt(N) with the later
//     levels' loads still in flight: their round trip hides under level 0);
//   * a level's outputs go to the neighbours as 8-byte {value, tag} granules written by ONE agent-scope (sc1, write-through) store
//     each and polled by the consumer with agent-scope loads until the tag matches -- the data is the flag, no fence, no flag word
//     (MI355X_MICROARCH.md price list, handoff-1to1: 0.8-2.9 us per hop, against 1.5-1.9 us for a kernel boundary PLUS the next
//     launch's own load round trip); the 1- and 4-pixel rings are 4 KB and 3 KB per workgroup;
//   * the tag is a per-workspace generation number: a workgroup reads the tag ITS OWN first granule carries from the previous launch
//     (complete: that launch has ended) and publishes with tag + 1.  Every launch rewrites every granule exactly once, so all cells
//     agree on the generation without a host-side argument (kernel arguments are frozen under graph replay) and without a memset
//     node in front of the launch.  The workspace is the caller's, zero-filled once, bound to one (batch, grid) and one stream.
// Correctness never depends on dispatch order or workgroup placement; PROGRESS needs every workgroup of the grid resident at once
// (a workgroup spins on its neighbours): the host checks the grid against the occupancy the runtime reports (capped at 5 workgroups per
// CU: the API answers one high when the SGPR file is the limit) and returns HS_ERR_UNSUPPORTED otherwise -- the caller then takes the three
// hs_patch_conv_fwd launches.  Every spin is bounded: a workgroup that waits longer than ~0.2 s raises the workspace's error word and
// goes on with what it has (wrong logits, no hang).
#include "hs_common.h"

namespace hs {

constexpr int KC_THREADS = 256, KC_WAVES = 4;
// LDS-DMA pieces (1 KB: one wave-instruction) every wave issues per bank: fixed counts, so that the vmcnt arithmetic below is a
// compile-time constant; pieces past a bank's end re-read its last 16 bytes into a dump area.  24 / 12 / 4 KB of bank at most.
constexpr int KC_P0 = 6, KC_P1 = 3, KC_P2 = 1;
constexpr int KC_SPIN_LIMIT = 1 << 17;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define KC_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct KcLevel {
    const float* __restrict__ skip;      // (B, c_skip, fh << l, fw << l)
    const float* __restrict__ bank;      // patch-major, row stride ld
    const float* __restrict__ scale;     // epilogue (null: none)
    const float* __restrict__ shift;
    long ld;
    int c_skip, cout, cin, act;
    int split;                           // lanes cooperating on one output (power of two)
    int pieces;                          // 1 KB pieces of the bank
    float step_x, step_y;                // linspace steps of the level's coordinate channels
};

struct KcArgs {
    KcLevel L[3];
    int B, fh, fw;
    u64* x0;             // [cells][c0] granules
    u64* x1;             // [cells][4][c1]
    unsigned* err;
    float* __restrict__ y;
};

struct KcLds { int bank[3], dump, sc[3], sh[3], xin0, sk1, sk2, genw, own0, own1, nb0, xin1, nb1, xin2, total; };

__host__ __device__ inline KcLds kc_lds_map(const int* pieces, const int* cin, const int* cout) {
    KcLds m;
    int o = 0;
    for (int l = 0; l < 3; ++l) { m.bank[l] = o; o += pieces[l] * 1024; }
    m.dump = o; o += 1024;
    for (int l = 0; l < 3; ++l) { m.sc[l] = o; o += 64 * 4; m.sh[l] = o; o += 64 * 4; }
    m.xin0 = o; o += 256 * 4;                 // level 0's input vector; the 4-byte DMA writes whole waves
    m.sk1 = o; o += 256 * 4;                  // skip pixels of levels 1 and 2 as they land (copied into xin1 / xin2 later)
    m.sk2 = o; o += 256 * 4;
    m.genw = o; o += 64 * 4;
    m.own0 = o; o += 64 * 4;
    m.own1 = o; o += 4 * 64 * 4;
    m.total = o;
    // dead-bank aliases: level 0's bank region is free once every wave has finished level 0's dot products
    int q = m.bank[0];
    m.nb0 = q; q += 9 * cout[0] * 4;
    m.xin1 = q; q += cin[1] * 4 * 4;
    m.nb1 = q; q += 16 * cout[1] * 4;
    m.xin2 = q; q += cin[2] * 16 * 4;
    return m;          // the host checks q <= bank[1] (the aliases stay inside level 0's region)
}

__device__ __forceinline__ int kc_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// acc over c = part, part + split, ... of w[c] * x[c * npix]; four partial sums (independent LDS reads in flight), then the lanes of a
// split group combine on the DPP path (split <= 4) -- the same scheme as patch_conv1x1_kernel
__device__ __forceinline__ float kc_dot(const float* __restrict__ wr, const float* __restrict__ xr, int cin, int npix, int part, int split) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int c = part;
    for (; c + 3 * split < cin; c += 4 * split) {
        const float w0 = wr[c], w1 = wr[c + split], w2 = wr[c + 2 * split], w3 = wr[c + 3 * split];
        const float x0 = xr[c * npix], x1 = xr[(c + split) * npix], x2 = xr[(c + 2 * split) * npix], x3 = xr[(c + 3 * split) * npix];
        a0 = fmaf(w0, x0, a0); a1 = fmaf(w1, x1, a1); a2 = fmaf(w2, x2, a2); a3 = fmaf(w3, x3, a3);
    }
    for (; c < cin; c += split) a0 = fmaf(wr[c], xr[c * npix], a0);
    float acc = (a0 + a1) + (a2 + a3);
    if (split >= 2) acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xf, 0xf, false));   // lane ^ 1
    if (split >= 4) acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E, 0xf, 0xf, false));   // lane ^ 2
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

__global__ __launch_bounds__(KC_THREADS)
void k1_chain_kernel(KcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = a.fh, fw = a.fw;
    const int cell = (int)blockIdx.x;
    const int pib = cell / fw, j = cell - pib * fw, b = pib / fh, i = pib - b * fh;
    const int c0 = a.L[0].cout, c1 = a.L[1].cout, c2 = a.L[2].cout;
    const int cin0 = a.L[0].cin, cin1 = a.L[1].cin, cin2 = a.L[2].cin;
    const int cs0 = a.L[0].c_skip, cs1 = a.L[1].c_skip, cs2 = a.L[2].c_skip;
    const int pieces[3] = {a.L[0].pieces, a.L[1].pieces, a.L[2].pieces};
    const int cins[3] = {cin0, cin1, cin2}, couts[3] = {c0, c1, c2};
    const KcLds M = kc_lds_map(pieces, cins, couts);
    float* wl0 = reinterpret_cast<float*>(lds + M.bank[0]);
    float* wl1 = reinterpret_cast<float*>(lds + M.bank[1]);
    float* wl2 = reinterpret_cast<float*>(lds + M.bank[2]);
    float* xin0 = reinterpret_cast<float*>(lds + M.xin0);
    float* own0 = reinterpret_cast<float*>(lds + M.own0);
    float* own1 = reinterpret_cast<float*>(lds + M.own1);
    float* nb0 = reinterpret_cast<float*>(lds + M.nb0);
    float* xin1 = reinterpret_cast<float*>(lds + M.xin1);
    float* nb1 = reinterpret_cast<float*>(lds + M.nb1);
    float* xin2 = reinterpret_cast<float*>(lds + M.xin2);
    // every shared word through GLOBAL (address space 1) agent-scope accesses, never flat ones
    gu64* const gx0 = (gu64*)a.x0;
    gu64* const gx1 = (gu64*)a.x1;
    gu32* const gerr = (gu32*)a.err;
    // @stamp 0

    // ---------------------------------------------------------------- every HBM load of the workgroup, level 0's operands first
    // ALL of them by LDS-DMA (16-byte pieces for the banks, 4-byte gathers for skip pixels, BatchNorm rows and the generation word):
    // no load targets a register, so the only vmcnt waits in this kernel are the two written below.  (With register loads pending
    // beside LDS-DMA the compiler's wait insertion assumes out-of-order completion and puts a vmcnt(0) in front of the first use of
    // any loaded register -- the two groups then become two serialised round trips.)  Every wave issues the SAME number of
    // operations per group (a wave with nothing to fetch aims at the dump area): the vmcnt arithmetic is per wave.
    auto dma_bank = [&](const KcLevel& lv, int region, int npieces, int q) {
        const int c = wave + KC_WAVES * q;                                       // (uniform)
        const unsigned last16 = (unsigned)lv.ld * 4u - 16u;
        const unsigned char* gb = reinterpret_cast<const unsigned char*>(lv.bank + (size_t)cell * (size_t)lv.ld);
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
    const int H1 = 2 * fh, W1 = 2 * fw, H2 = 4 * fh, W2 = 4 * fw;
    auto dma_rows = [&](const KcLevel& lv, int l) {                              // BatchNorm scale / shift: wave 0 fetches, the others aim at the dump
        const float* sp = lv.scale ? lv.scale : lv.skip;                         // (no epilogue: any readable address; the rows are then set to 1 / 0)
        const float* hp = lv.scale ? lv.shift : lv.skip;
        const int idx = lv.scale ? min(lane, lv.cout - 1) : 0;
        dma_word(sp + idx, wave == 0 ? M.sc[l] : M.dump);
        dma_word(hp + idx, wave == 0 ? M.sh[l] : M.dump);
    };
    // group 0: level 0's bank, its skip pixel (thread t -> input channel t: 2 coordinates first), its BatchNorm rows, the generation word
#pragma unroll
    for (int q = 0; q < KC_P0; ++q) dma_bank(a.L[0], M.bank[0], pieces[0], q);
    dma_word(a.L[0].skip + ((((size_t)b * cs0 + (size_t)kc_clamp(tid - 2, 0, cs0 - 1)) * fh + i) * fw + j), M.xin0 + wave * 256);
    dma_rows(a.L[0], 0);
    // the generation: the tag (upper half) of this cell's first granule as the PREVIOUS launch left it
    dma_word(reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.x0) + ((size_t)cell * c0 * 8 + 4)), wave == 0 ? M.genw : M.dump);
    __builtin_amdgcn_sched_barrier(0);
    // group 1: the later levels' banks, skip pixels and BatchNorm rows
#pragma unroll
    for (int q = 0; q < KC_P1; ++q) dma_bank(a.L[1], M.bank[1], pieces[1], q);
#pragma unroll
    for (int q = 0; q < KC_P2; ++q) dma_bank(a.L[2], M.bank[2], pieces[2], q);
    {
        const int c = kc_clamp(tid >> 2, 0, cs1 - 1), px = tid & 3;
        dma_word(a.L[1].skip + ((((size_t)b * cs1 + c) * H1 + (2 * i + (px >> 1))) * W1 + (2 * j + (px & 1))), M.sk1 + wave * 256);
    }
    {
        const int c = kc_clamp(tid >> 4, 0, cs2 - 1), px = tid & 15;
        dma_word(a.L[2].skip + ((((size_t)b * cs2 + c) * H2 + (4 * i + (px >> 2))) * W2 + (4 * j + (px & 3))), M.sk2 + wave * 256);
    }
    dma_rows(a.L[1], 1);
    dma_rows(a.L[2], 2);
    __builtin_amdgcn_sched_barrier(0);
    // @stamp 1
    // group 0 has landed when at most group 1's operations are outstanding (vector-memory operations complete in issue order)
    static_assert(KC_P1 + KC_P2 + 2 + 4 == 10, "the count in the s_waitcnt below");
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    // @stamp 2
    {   // level 0's input vector: the coordinates over what the gather put in slots 0 and 1 (wave 0 wrote them itself: ordered by its wait)
        if (tid == 0) xin0[0] = linspace_pm1(j, fw, a.L[0].step_x);
        if (tid == 1) xin0[1] = linspace_pm1(i, fh, a.L[0].step_y);
        float* sc = reinterpret_cast<float*>(lds + M.sc[0]);
        float* sh = reinterpret_cast<float*>(lds + M.sh[0]);
        if (!a.L[0].scale && tid < 64) { sc[tid] = 1.0f; sh[tid] = 0.0f; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");              // group 1 stays in flight across the barrier
    const unsigned g_prev = *reinterpret_cast<const unsigned*>(lds + M.genw);
    unsigned gen = (unsigned)__builtin_amdgcn_readfirstlane((int)g_prev) + 1u;
    gen = gen ? gen : 1u;
    // @stamp 3

    // ---------------------------------------------------------------- level 0: c0 outputs of one pixel
    {
        const int split = a.L[0].split, part = tid & (split - 1), per_pass = KC_THREADS / split;
        const float* sc = reinterpret_cast<const float*>(lds + M.sc[0]);
        const float* sh = reinterpret_cast<const float*>(lds + M.sh[0]);
        for (int base = 0; base < c0; base += per_pass) {
            const int o = base + tid / split;
            const bool live = o < c0;
            const int oo = live ? o : 0;
            float acc = kc_dot(wl0 + oo * cin0, xin0, cin0, 1, part, split);
            acc = apply_act(fmaf(acc, sc[oo], sh[oo]), a.L[0].act);
            // group 1 (the later levels' operands) has had level 0's whole duration to land: the wait for it goes HERE, in front of the
            // publishing stores -- behind them a vmcnt(0) would also wait for the stores' own write-through round trip (stores count in
            // vmcnt on gfx9-class hardware), which nobody in this workgroup needs
            if (base + per_pass >= c0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (live && part == 0) {
                own0[o] = acc;
                __hip_atomic_store(gx0 + (size_t)cell * c0 + o, ((u64)gen << 32) | (u64)__float_as_uint(acc), KC_RLX_AGENT);
            }
        }
    }
    // @stamp 4
    // every wave's share of group 1 has landed (waited for above); the barrier makes it the workgroup's, and level 0's bank region free
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // @stamp 5
    if (tid < 64) {   // levels without an epilogue: unit rows (read after the next barrier)
        if (!a.L[1].scale) { reinterpret_cast<float*>(lds + M.sc[1])[tid] = 1.0f; reinterpret_cast<float*>(lds + M.sh[1])[tid] = 0.0f; }
        if (!a.L[2].scale) { reinterpret_cast<float*>(lds + M.sc[2])[tid] = 1.0f; reinterpret_cast<float*>(lds + M.sh[2])[tid] = 0.0f; }
    }

    // ---------------------------------------------------------------- level 0 -> 1: the 3 x 3 cells around this one (clamped at the border)
    {
        constexpr int NQ = 3;                                                    // 8 c0 <= 768 granules
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
#pragma unroll
        for (int q = 0; q < NQ; ++q) if (need[q]) nb0[dst[q]] = __uint_as_float(v[q]);
        if (tid < c0) nb0[4 * c0 + tid] = own0[tid];
        // level 1's input that does not depend on level 0: coordinates and the skip pixels
        if (tid < cs1 * 4) xin1[8 + tid] = reinterpret_cast<const float*>(lds + M.sk1)[tid];
        if (tid < 8) {
            const int px = tid & 3;
            xin1[tid] = tid < 4 ? linspace_pm1(2 * j + (px & 1), W1, a.L[1].step_x) : linspace_pm1(2 * i + (px >> 1), H1, a.L[1].step_y);
        }
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
    {
        const int split = a.L[1].split, part = tid & (split - 1), per_pass = KC_THREADS / split;
        const float* sc = reinterpret_cast<const float*>(lds + M.sc[1]);
        const float* sh = reinterpret_cast<const float*>(lds + M.sh[1]);
        const int total = c1 * 4;
        for (int base = 0; base < total; base += per_pass) {
            const int idx = base + tid / split;
            const bool live = idx < total;
            const int o = live ? idx >> 2 : 0, px = idx & 3;
            float acc = kc_dot(wl1 + o * cin1, xin1 + px, cin1, 4, part, split);
            acc = apply_act(fmaf(acc, sc[o], sh[o]), a.L[1].act);
            if (live && part == 0) {
                own1[px * c1 + o] = acc;
                __hip_atomic_store(gx1 + ((size_t)cell * 4 + px) * c1 + o, ((u64)gen << 32) | (u64)__float_as_uint(acc), KC_RLX_AGENT);
            }
        }
    }
    __syncthreads();

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
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (live[q]) nb1[dst[q]] = need[q] ? __uint_as_float(v[q]) : own1[own[q]];
        }
        if (tid < cs2 * 16) xin2[32 + tid] = reinterpret_cast<const float*>(lds + M.sk2)[tid];
        if (tid < 32) {
            const int px = tid & 15;
            xin2[tid] = tid < 16 ? linspace_pm1(4 * j + (px & 3), W2, a.L[2].step_x) : linspace_pm1(4 * i + (px >> 2), H2, a.L[2].step_y);
        }
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
    // ---------------------------------------------------------------- level 2: c2 outputs x 16 pixels, plain stores (the next launch reads them)
    {
        const int split = a.L[2].split, part = tid & (split - 1), per_pass = KC_THREADS / split;
        const float* sc = reinterpret_cast<const float*>(lds + M.sc[2]);
        const float* sh = reinterpret_cast<const float*>(lds + M.sh[2]);
        const int total = c2 * 16;
        for (int base = 0; base < total; base += per_pass) {
            const int idx = base + tid / split;
            const bool live = idx < total;
            const int o = live ? idx >> 4 : 0, px = idx & 15;
            float acc = kc_dot(wl2 + o * cin2, xin2 + px, cin2, 16, part, split);
            acc = apply_act(fmaf(acc, sc[o], sh[o]), a.L[2].act);
            if (live && part == 0)
                a.y[(((size_t)b * c2 + o) * H2 + (4 * i + (px >> 2))) * W2 + (4 * j + (px & 3))] = acc;
        }
    }
    // @stamp 24
}

// resident workgroups of this kernel on the current device, (CUs, per CU); cached per device (write-once, idempotent)
static int kc_residency(size_t lds, int* cus, int* per_cu) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    int n = 0, c = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k1_chain_kernel, KC_THREADS, lds);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    *cus = c; *per_cu = n;
    return HS_OK;
}

static int kc_plan(int batch, int fh, int fw, const hs_k1_level* lv, int n_levels, KcArgs* out, size_t* lds_bytes, int64_t* ws_bytes) {
    if (!lv || n_levels != 3 || batch <= 0 || fh <= 0 || fw <= 0) return n_levels == 3 ? HS_ERR_BAD_ARG : HS_ERR_UNSUPPORTED;
    KcArgs a;
    a.B = batch; a.fh = fh; a.fw = fw;
    int prev = 0;
    const int maxp[3] = {KC_P0 * KC_WAVES, KC_P1 * KC_WAVES, KC_P2 * KC_WAVES};
    int pieces[3], cin[3], cout[3];
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
        if (d.pieces > maxp[l] || d.cout > 64 || s.c_skip * npix > KC_THREADS) return HS_ERR_UNSUPPORTED;
        int split = 1;
        while (split < 4 && d.cout * npix * split * 2 <= KC_THREADS && split * 2 <= d.cin) split *= 2;
        d.split = split;
        const int H = fh << l, W = fw << l;
        d.step_x = W > 1 ? 2.0f / (float)(W - 1) : 0.0f;
        d.step_y = H > 1 ? 2.0f / (float)(H - 1) : 0.0f;
        pieces[l] = d.pieces; cin[l] = d.cin; cout[l] = d.cout;
        prev = d.cout;
    }
    if (8 * cout[0] > 3 * KC_THREADS || 16 * cout[1] > 4 * KC_THREADS) return HS_ERR_UNSUPPORTED;
    const KcLds M = kc_lds_map(pieces, cin, cout);
    if (M.xin2 + cin[2] * 16 * 4 > M.bank[1]) return HS_ERR_UNSUPPORTED;          // the aliases must stay inside level 0's bank region
    if (M.total > 64 * 1024) return HS_ERR_UNSUPPORTED;
    const long cells = (long)batch * fh * fw;
    if (cells * 4 * 64 >= (1l << 31)) return HS_ERR_UNSUPPORTED;
    if (out) *out = a;
    if (lds_bytes) *lds_bytes = (size_t)M.total;
    if (ws_bytes) *ws_bytes = 256 + 8 * cells * ((long)cout[0] + 4l * cout[1]);
    return HS_OK;
}

}  // namespace hs

using namespace hs;

extern "C" int64_t hs_k1_chain_workspace(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels) {
    int64_t bytes = 0;
    const int st = kc_plan(batch, fh, fw, levels, n_levels, nullptr, nullptr, &bytes);
    return st == HS_OK ? bytes : (int64_t)st;
}

extern "C" int hs_k1_chain_fwd(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels, void* workspace,
                               float* y, void* stream) {
    KcArgs a;
    size_t lds = 0;
    int64_t ws = 0;
    int st = kc_plan(batch, fh, fw, levels, n_levels, &a, &lds, &ws);
    if (st != HS_OK) return st;
    if (!workspace || !y || ((uintptr_t)workspace & 15) != 0) return HS_ERR_BAD_ARG;
    // every workgroup spins on its neighbours: the whole grid has to be resident at once
    int cus = 0, per_cu = 0;
    st = kc_residency(lds, &cus, &per_cu);
    if (st != HS_OK) return st;
    // The occupancy query answers one workgroup per CU high when the SGPR file is what limits a 256-thread workgroup (admitted =
    // min(API, 8, 800 / (ceil(sgprs / 16) * 16 + 16)): MI355X_MICROARCH.md, residency); for any SGPR count a kernel can have that
    // bound is >= 5, and LDS / VGPR limits are reported exactly -- so min(API, 5) workgroups per CU is never more than the hardware
    // admits.  (This kernel: ~100 SGPRs, 48 VGPRs; 29-44 KB of LDS is what limits it at the decoder's shapes: 3-5 per CU.)
    const long cells = (long)batch * fh * fw;
    const int admitted = per_cu < 5 ? per_cu : 5;
    if (admitted < 1 || cells > (long)admitted * cus) return HS_ERR_UNSUPPORTED;
    unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
    a.err = reinterpret_cast<unsigned*>(w);
    a.x0 = reinterpret_cast<u64*>(w + 256);
    a.x1 = reinterpret_cast<u64*>(w + 256 + 8 * cells * (long)a.L[0].cout);
    a.y = y;
    hipLaunchKernelGGL(k1_chain_kernel, dim3((unsigned)cells), dim3(KC_THREADS), lds, (hipStream_t)stream, a);
    return launch_status();
}

// The squeeze-excite gate of an MBConv block computed by the TAIL of the launch that pools (round 5; efficientnet.py:106-111).
//
// Rounds 1-4: [depthwise + pool partials] -> [SE gate: one launch of 5.6 us on 13 blocks, two launches of 4.9 + 4.7 us on 10] -> [project
// GEMM]: 168 us of a 773 us HyperSeg-M frame went into 33 launches that move < 1.3 MB each and are pure dependent latency (the launch edge,
// the partials' round trip, two matrix-vector products).  Here the LAST T workgroups of the pooling launch ("tails") finish the gate
// themselves, so the block is [depthwise + pool + gate] -> [project GEMM] and those launches do not exist:
//
//   every workgroup   reads the generation word g of its batch element at its start, and publishes each pool partial as an 8-byte
//                     granule {value, tag = g + 1} with an agent-scope store (no fence, no read-modify-write: granule = data + flag);
//   tail s (of T)     owns L = ceil(C / T) channels: polls THEIR partial granules (bounded spin), means -> LDS;
//                     zp[j] = sum_{c in slice} w1[j][c] mean[c] for every squeezed channel j, published as granules zg[s][j];
//                     polls all T x Csq of those, z[j] = swish(b1[j] + sum_s zp_s[j]) in slice order (deterministic);
//                     gate[c] = sigmoid(b2[c] + sum_j w2t[j][c] z[j]) for its own channels, stored plainly (the project GEMM is the
//                     next launch); tail 0 then stores g + 1 into the generation word (visible to the next launch).
//
// Forward progress: only tails ever wait, on work that no tail holds back -- the other workgroups wait for nothing, and the T <= 32
// tails fit the chip together many times over, so the waits end whatever order the dispatcher picks.  Every wait is bounded
// (SE_SPIN_LIMIT polls): a tail that gives up raises the workspace's error word and stores NaN gates (loud, not a hang).
// The workspace (generation words, error word, both granule arrays) must start zeroed and belong to ONE launch at a time:
// hyperseg_amd.functional.ExclusiveWorkspaces hands one to every stream / captured graph.
//
// Tried before and rejected (see hs_encoder.hip): device-wide barriers with fences + atomics (9-19 us per block), every excite
// workgroup re-deriving the whole squeeze (30-170 us on the wide blocks).  What differs here: no fence and no atomic RMW on the
// producers' path, and the two matrix-vector products are SPLIT across the tails (<= 16 KB of weights and granules each) instead of
// replicated -- the measured cost of a tagged-granule hand-off in hs_k1_chain.hip is ~1 us, and a tail pays two.
#pragma once
#include "hs_common.h"

namespace hs {

typedef unsigned long long se_u64;
typedef __attribute__((address_space(1))) se_u64 se_gu64;
typedef __attribute__((address_space(1))) unsigned se_gu32;
#define SE_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int SE_SPIN_LIMIT = 1 << 17;
constexpr int SE_MAX_TAILS = 32, SE_MAX_L = 64, SE_MAX_CSQ = 96, SE_MAX_STAGE = 2048, SE_STAGE = 2560;
// LDS of a tail (floats): stage (partials [L][nblk + 1], later the z shares [T][Csq]) | mean [64] | z [SE_MAX_CSQ] | gate partial sums
// [4][64] | this slice's reduce weights [Csq][64] | expand weights [Csq][64]  -- 17 KB at Csq = 10, 53 KB at Csq = 80
__host__ __device__ inline int se_lds_floats(int Csq) { return SE_STAGE + 64 + SE_MAX_CSQ + 4 * 64 + 2 * 64 * Csq; }

struct SeTail {                           // by value in the kernarg segment; ws == nullptr: plain pool partials, no tail
    const float* w1; const float* b1;     // reduce conv (Csq, C), (Csq)
    const float* w2t; const float* b2;    // expand conv TRANSPOSED (Csq, C), (C)
    float* gate;                          // (B, C)
    float* z_out;                         // (B, Csq) or null
    se_u64* ws;                           // [B] generation words | [1] error word | zg [B][T][Csq] | pg [B][C][nblk]
    int B, C, Csq, nblk, T, L;
    float inv_hw;
};

__host__ __device__ inline size_t se_ws_zg(const SeTail& t) { return (size_t)t.B + 1; }
__host__ __device__ inline size_t se_ws_pg(const SeTail& t) { return se_ws_zg(t) + (size_t)t.B * t.T * t.Csq; }
__host__ __device__ inline size_t se_ws_words(const SeTail& t) { return se_ws_pg(t) + (size_t)t.B * t.C * t.nblk; }

// channels per tail and the number of tails for (C, nblk, workgroups per batch element); false: this shape keeps the SE launches
inline bool se_tail_plan(int C, int Csq, int nblk, long wgs_per_batch, int& T, int& L) {
    if (C <= 0 || Csq <= 0 || Csq > SE_MAX_CSQ || nblk <= 0 || nblk > SE_MAX_STAGE / 4) return false;
    L = SE_MAX_STAGE / nblk;
    if (L > SE_MAX_L) L = SE_MAX_L;       // <= 64 channels per tail: one lane per channel, weight rows of 256 bytes
    L &= ~3;
    if (L < 4) return false;
    T = (C + L - 1) / L;
    if (T > SE_MAX_TAILS || (long)T * Csq > SE_STAGE || T > wgs_per_batch) return false;
    return true;
}

// hs_se_tail (include/hyperseg_hip.h) -> SeTail for a launch whose batch element has `nblk` partials per channel and `wgs` workgroups
inline int make_se_tail(const hs_se_tail* in, int batch, int channels, int nblk, long wgs, int pixels, SeTail& t) {
    if (!in->w_reduce || !in->b_reduce || !in->w_expand_t || !in->b_expand || !in->gate || !in->workspace) return HS_ERR_BAD_ARG;
    if ((((size_t)in->workspace) & 7) != 0) return HS_ERR_BAD_ARG;
    int T, L;
    if (!se_tail_plan(channels, in->c_squeezed, nblk, wgs, T, L)) return HS_ERR_UNSUPPORTED;
    t.w1 = in->w_reduce; t.b1 = in->b_reduce; t.w2t = in->w_expand_t; t.b2 = in->b_expand;
    t.gate = in->gate; t.z_out = in->squeezed; t.ws = (se_u64*)in->workspace;
    t.B = batch; t.C = channels; t.Csq = in->c_squeezed; t.nblk = nblk; t.T = T; t.L = L;
    t.inv_hw = 1.0f / (float)pixels;
    return HS_OK;
}

__device__ __forceinline__ unsigned se_tag(const SeTail& t, int b) {         // at the START of every workgroup of batch element b
    return (unsigned)__hip_atomic_load((se_gu64*)t.ws + b, SE_RLX_AGENT) + 1u;
}
__device__ __forceinline__ void se_publish(se_u64* base, size_t idx, float v, unsigned tag) {
    __hip_atomic_store((se_gu64*)base + idx, ((se_u64)tag << 32) | (se_u64)__float_as_uint(v), SE_RLX_AGENT);
}

// NQ granules per thread, all in flight together (addresses always valid; need[q] false: filler), until every needed one carries `tag`
template <int NQ>
__device__ __forceinline__ bool se_gather(se_gu64* (&g)[NQ], const bool (&need)[NQ], unsigned tag, float (&v)[NQ]) {
    for (int spins = 0;; ++spins) {
        se_u64 x[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) x[q] = __hip_atomic_load(g[q], SE_RLX_AGENT);
        bool ok = true;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            v[q] = __uint_as_float((unsigned)x[q]);
            ok &= !need[q] || (unsigned)(x[q] >> 32) == tag;
        }
        if (__all(ok)) return true;
        if (spins >= SE_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

// Called by EVERY workgroup of the launch after its last partial has been published; `widx` = the workgroup's index among the
// `total` workgroups of batch element b (any fixed order), `lds` = se_lds_floats(Csq) floats nobody else uses any more.  All threads of
// the workgroup call it together (it has barriers); blockDim.x must be a multiple of 64.
//
// A tail is a chain of dependent round trips, so everything that does not depend on another workgroup is requested FIRST: the slice's
// reduce / expand weight rows go to LDS by LDS-DMA (no registers, all rows in flight; first version: loaded in batches of 8 rows where
// they were needed -- 2 + Csq / 16 more round trips, 6-12 us per tail, visit r5v7), the biases to registers; each of the two waits
// (partials, z shares) is ONE gather spread over all threads.
__device__ __forceinline__ void se_tail_run(const SeTail& t, int b, long widx, long total, unsigned tag, float* lds) {
    const long s_l = total - 1 - widx;
    if (s_l >= t.T) return;                                          // uniform: not a tail
    const int s = (int)s_l;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: row addresses below stay in scalar registers
    const int C = t.C, Csq = t.Csq, nblk = t.nblk, T = t.T;
    const int c0 = s * t.L, Ls = min(t.L, C - c0);                   // this tail's channels [c0, c0 + Ls); 1 <= Ls <= 64 by the plan
    float* stage = lds;
    float* mean = lds + SE_STAGE;                                    // [64], zero past Ls
    float* zs = mean + 64;                                           // [Csq]
    float* red = zs + SE_MAX_CSQ;                                    // [<= 4][64] partial gate sums
    float* w1s = red + 4 * 64;                                       // [Csq][64]: row j = w1[j][c0 + lane]
    float* w2s = w1s + 64 * Csq;                                     // [Csq][64]: row j = w2t[j][c0 + lane]
    se_gu64* zg = (se_gu64*)t.ws + se_ws_zg(t) + (size_t)b * T * Csq;
    se_gu64* pg = (se_gu64*)t.ws + se_ws_pg(t) + ((size_t)b * C + c0) * nblk;
    bool good = true;

    // ---- 0. weights of the slice -> LDS (lane l of a row -> its word l; lanes past the slice re-read its last channel, times mean 0)
    const int cl = min(lane, Ls - 1);
    for (int j = wave; j < Csq; j += nw) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t.w1 + (size_t)j * C + c0 + cl),
                                         (__attribute__((address_space(3))) void*)(w1s + j * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t.w2t + (size_t)j * C + c0 + cl),
                                         (__attribute__((address_space(3))) void*)(w2s + j * 64), 4, 0, 0);
    }
    const float b1v = t.b1[min(tid, Csq - 1)], b2v = t.b2[c0 + cl];

    // ---- 1. the slice's pool partials -> means
    const int units = Ls * nblk;                                     // <= SE_MAX_STAGE, contiguous granules
    for (int e0 = tid; e0 < units; e0 += nthr * 8) {
        se_gu64* g[8]; bool need[8]; float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int e = e0 + q * nthr; need[q] = e < units; g[q] = pg + min(e, units - 1); }
        good &= se_gather<8>(g, need, tag, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * nthr;
            if (e < units) { const int ch = e / nblk; stage[ch * (nblk + 1) + (e - ch * nblk)] = v[q]; }
        }
    }
    __syncthreads();
    for (int ch = tid; ch < 64; ch += nthr) {
        const float* run = stage + ch * (nblk + 1);
        float a = 0.0f;
        if (ch < Ls)
            for (int i = 0; i < nblk; ++i) a += run[i];
        mean[ch] = a * t.inv_hw;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's weight rows have landed
    __syncthreads();

    // ---- 2. this slice's share of every squeezed channel: wave = row j, lane = channel
    {
        const float m = mean[lane];
        for (int j = wave; j < Csq; j += nw) {
            const float p = wave_sum64(w1s[j * 64 + lane] * m);
            if (lane == 0) se_publish((se_u64*)t.ws, se_ws_zg(t) + ((size_t)b * T + s) * Csq + j, p, tag);
        }
    }

    // ---- 3. every slice's share -> z (slice order: the same sum in every tail)
    const int nz = T * Csq;                                          // <= SE_STAGE
    for (int e0 = tid; e0 < nz; e0 += nthr * 8) {
        se_gu64* g[8]; bool need[8]; float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int e = e0 + q * nthr; need[q] = e < nz; g[q] = zg + min(e, nz - 1); }
        good &= se_gather<8>(g, need, tag, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int e = e0 + q * nthr; if (e < nz) stage[e] = v[q]; }
    }
    good = __syncthreads_and(good);                                  // one verdict for the workgroup
    for (int j = tid; j < Csq; j += nthr) {
        float a = 0.0f;
        for (int s2 = 0; s2 < T; ++s2) a += stage[s2 * Csq + j];
        const float z = swishf(a + (j == tid ? b1v : t.b1[j]));
        zs[j] = z;
        if (s == 0 && t.z_out) t.z_out[(size_t)b * Csq + j] = z;
    }
    __syncthreads();

    // ---- 4. the slice's gates: thread = (channel, wave's share of the squeezed channels), shares summed in wave order
    {
        const int nparts = min(nw, 4);
        const int jq = (Csq + nparts - 1) / nparts, ja = wave * jq, jb = min(ja + jq, Csq);
        float acc = 0.0f;
        if (wave < nparts) {
            for (int j = ja; j < jb; ++j) acc = fmaf(w2s[j * 64 + lane], zs[j], acc);
            red[wave * 64 + lane] = acc;
        }
        __syncthreads();
        if (wave == 0 && lane < Ls) {
            float a = red[lane];
            for (int p = 1; p < nparts; ++p) a += red[p * 64 + lane];
            const float gv = sigmoidf_fast(a + b2v);
            t.gate[(size_t)b * C + c0 + lane] = good ? gv : __int_as_float(0x7fc00000);
        }
    }
    if (!good && tid == 0) __hip_atomic_store((se_gu32*)(t.ws + t.B), 1u, SE_RLX_AGENT);
    if (s == 0 && tid == 0) __hip_atomic_store((se_gu64*)t.ws + b, (se_u64)tag, SE_RLX_AGENT);     // the next launch's generation
}

}  // namespace hs

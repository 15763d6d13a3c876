// Op C -- the v1_0 / unify inverted residual (hyperseg_v1_0.py:328-376, hyperseg_v1_0_unify.py:330-389) -- on the f16
// matrix cores with split operands, round-3 form.  Same arithmetic idea as round 2's hs_patch_ir_split.hip (every f32
// operand is scaled by a power of two and split, x * 2^e = hi + lo + r with |r| <= 2^-22 |x * 2^e|, a product is
// ah*bh + al*bh + ah*bl on v_mfma_f32_16x16x32_f16 accumulating in f32), rebuilt around what the round-2 counters said
// (profiles/round2_pmc_ir_split_M_level4.txt: 4128 vector instructions per wave, 29 % of the LDS cycles bank conflicts, the
// grid limited to 2 waves per SIMD, 23 k of a workgroup's 51 k cycles in its prologue):
//
//   regions     16 pixels wide x RH = 2 * NW rows (NW = 4 waves: 16 x 8 half-patch regions, 1024 workgroups at HyperSeg-M
//               level 4 = 4 per CU, all resident; <= 128 VGPRs and <= 40 KB of LDS are the budget that allows it).  Regions
//               of one patch are consecutive block indices on one XCD, so the patch's bank is fetched into one L2 once.
//   weights     the patch's WHOLE bank is split once per workgroup, in the prologue, into LDS in the bank's own order
//               (W1 hi | W1 lo | W3 hi | W3 lo, taps as f32 with BN2's scale folded in): every row's scale is the exact
//               row maximum (a 16-lane DPP max), so there is no running exponent and no accumulator rescale, the chunk
//               loop carries no global load, no vmcnt wait and no operand staging, and the LDS image is exactly as large
//               as the bank (rows past the last hidden channel read finite neighbours and meet zero BN rows).
//   B operand   K order = [skip run | previous-level run] per lane group (4 consecutive channels each), coordinates in a
//               one-MFMA tail where the lane group picks the product.  The previous level's low-res window sits in LDS as
//               [position][channel], so one ds_read_b128 fetches a tap of the lane's 4 channels; the input scale is per
//               halo POSITION (a B column): max over the lane's values and two v_permlane swaps, no 16-lane reduction.
//   h1 / h2     h1 planes (f32) in slot order 4 (c & 3) + (c >> 2) with plane stride == 16 (mod 64) floats and an
//               18-float row: pw1's ds_write_b32 and the depthwise stage's ds_read_b64 (lane = half | row_lo << 1 |
//               channel << 3 | row_hi << 5) are both conflict-free in tools/lds_conflicts.py's model of the gfx950 LDS.
//               h2 (two f16 planes per hidden channel) has no padding: region row t of plane p sits in 32-byte slot
//               t ^ ((p & 3) + 4 (p >> 3)), which spreads the 8 planes one ds_read_b64_tr_b16 touches over all banks.
//   depthwise   thread = (hidden channel, output row, 8-pixel half row): 30 inputs, 72 independent v_fma_f32 in
//               tap-major order (round 2's v_pk_fma chains were dependent three deep), BN2 folded into taps + initial value.
//
// Any channel split with c_skip <= 4 SPL, c_prev <= 4 PPL (<= 16 each), c_out <= 32 and hid <= 96 is served: channel counts
// are run-time values, the templates only size register arrays -- there is no table of literal shapes.
#include "hs_ir_common.h"


namespace hs {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
typedef __fp16 half4tr __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define HS_IRC_LDS_H4(p) ((__attribute__((address_space(3))) half4tr*)(p))
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr float IRC_H2_SCALE = 4096.0f;          // h2 = relu6(.) in [0, 6] -> [0, 24576] before the f16 split

// ---- geometry shared by the kernel, the launcher and the host-side layout check (hs_irc_layout) ----
template <int RH_, int NW_ = 4> struct IrcGeom {
    static constexpr int NW = NW_, NTHR = 64 * NW;
    static constexpr int RH = RH_, RW = 16;                    // region
    static constexpr int HH = RH + 2, HWD = RW + 2;            // halo grid
    static constexpr int NPOS = HH * HWD;
    static constexpr int NT1 = (NPOS + 15) / 16;               // pw1 position tiles
    static constexpr int J1 = (NT1 + NW - 1) / NW;
    static constexpr int J3 = RH / NW;                         // pw3 pixel tiles (= region rows) per wave
    static constexpr int CS = HWD;                             // h1 row stride (floats)
    static constexpr int PS1 = ((NPOS - 16 + 63) / 64) * 64 + 16;   // h1 plane stride: smallest value >= NPOS that is == 16 (mod 64)
    static constexpr int H1_FLOATS = 16 * PS1;
    static constexpr int H2_PLANE = RH * RW;                   // halfs per (piece, hidden channel)
    static constexpr int H2_HALFS = 2 * 16 * H2_PLANE;
    static constexpr int PWH = RH / 2 + 2, PWW = RW / 2 + 2;   // low-res window of the previous level
};
static_assert(IrcGeom<8>::PS1 == 208 && IrcGeom<16>::PS1 == 336 && IrcGeom<16, 8>::PS1 == 336, "h1 plane strides the conflict model was run on");

__host__ __device__ inline int irc_h1_slot(int c) { return 4 * (c & 3) + (c >> 2); }
__host__ __device__ inline int irc_h2_swz(int p) { return (p & 3) + 4 * (p >> 3); }

struct IrcArgs {
    StageIn in;
    int fh, fw, ph, pw;
    const float* __restrict__ bank;
    long ld;
    int hid, cout;
    const float* __restrict__ s1; const float* __restrict__ b1;
    const float* __restrict__ s2; const float* __restrict__ b2;
    const float* __restrict__ s3; const float* __restrict__ b3;
    float* __restrict__ y;
    int sub_y, sub_x;            // regions per patch
    int prio;                    // 0 | 1 | 2 | 3: whose turn it is on a CU (see the kernel)
    unsigned m_nsub, m_subx, m_fw, m_fh;      // magic_of(sub_y * sub_x), (sub_x), (fw), (fh)
};

// LDS map (bytes) of one workgroup; the same function sizes the launch
struct IrcLds {
    int taps, w1h, w1l, w3h, w3l, s1f, b1, b2s, s3f, b3, h1, h2, raw, raw_rows, red, total;
    int cinp, hidp, HP, CP, raw_chunks;
};
__host__ __device__ inline IrcLds irc_lds_map(int cin, int hid, int cout, int mt3, int h1_floats, int h2_halfs) {
    IrcLds m;
    m.cinp = (cin + 1) & ~1; m.hidp = (hid + 1) & ~1;
    m.HP = (hid + 15) & ~15; m.CP = 16 * mt3;
    int o = 0;
    m.taps = o; o += hid * 9 * 4;
    m.w1h = o; o += hid * m.cinp * 2;
    m.w1l = o; o += hid * m.cinp * 2;
    m.w3h = o; o += cout * m.hidp * 2;
    m.w3l = o; o += cout * m.hidp * 2;
    o += 64;                                             // zeros: what the last rows' overruns read
    o = (o + 15) & ~15;
    m.s1f = o; o += m.HP * 4;
    m.b1 = o; o += m.HP * 4;
    m.b2s = o; o += m.HP * 4;
    m.s3f = o; o += m.CP * 4;
    m.b3 = o; o += m.CP * 4;
    o = (o + 15) & ~15;
    m.h1 = o; o += h1_floats * 4;
    m.h2 = o; o += h2_halfs * 2;
    // the raw f32 bank lands here by LDS-DMA (whole 1 KB pieces) WHILE the tiles occupy h1 | h2 (round 4: the landing zone used to
    // alias the scratch, which chained bank wait -> split -> tiles -> B fragments one after the other)
    o = (o + 1023) & ~1023;
    m.raw_chunks = ((cin * hid + 9 * hid + hid * cout) * 4 + 1023) >> 10;
    m.raw = o; o += m.raw_chunks * 1024;
    m.raw_rows = o; o += (2 * m.HP + m.CP) * 4;           // BatchNorm scales as loaded (s1 | s2 | s3), consumed by the split
    m.red = o; o += 16 * 4;                               // per-wave maxima of the two matrices
    m.total = o;
    return m;
}

// max over the 16 lanes of a DPP row of non-negative floats, as integers (one v_max_u32_dpp per step)
__device__ __forceinline__ unsigned rowmax16_u(unsigned x) {
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));     // row_half_mirror
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));     // row_mirror
    return x;
}
__device__ __forceinline__ unsigned absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
// biased exponent eb of a non-negative m given as bits (m < 2^(eb - 126)), clamped so that 2^(141 - eb) and 2^(eb - 141)
// are normal floats; m * 2^(141 - eb) < 2^15
__device__ __forceinline__ int irc_exp_of(unsigned mbits) { return min(max((int)(mbits >> 23), 27), 254); }
__device__ __forceinline__ float irc_scale_of(int eb) { return __uint_as_float((unsigned)(268 - eb) << 23); }
__device__ __forceinline__ float irc_inv_scale_of(int eb) { return __uint_as_float((unsigned)(eb - 14) << 23); }

// base[elem] through a 32-bit byte offset from a uniform base: saddr + voffset addressing, one VALU (the shift) per load
__device__ __forceinline__ float ldg(const float* __restrict__ base, unsigned elem) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(elem << 2));
}
// x / d for x * d < 2^32 with the host's magic m = 2^32 / d + 1 (d > 1; m = 0 stands for d = 1)
__device__ __forceinline__ unsigned div_magic(unsigned x, unsigned m) { return m ? __umulhi(x, m) : x; }
inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / d + 1); }

// (hi, lo) f16 pieces of a * s and b * s, each packed pair in one register: 2 VALU per element (v_fma_mix* convert on the
// way out; the residual a * s - hi is exact in f32).  hi = rne16(a * s), lo = rne16(a * s - hi).
__device__ __forceinline__ void split2(float a, float b, float s, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l) : "v"(a), "v"(b), "v"(s));
    hi = h; lo = l;
}

// NW_ = 8 (RH = 16 only): the same region on EIGHT waves -- two workgroups per CU then give every SIMD four waves to interleave
// (the phases of this kernel are chains of dependent LDS round trips and vector instructions: at two waves per SIMD nothing covers
// a wave's stalls, and both workgroups of a CU run the same phase at the same time).  Per-wave tile counts halve (pw1 6 -> 3,
// depthwise 2 -> 1 row blocks, pw3 4 -> 2), the register budget is 128.
template <int SPL, int PPL, int MT3, int RH_, int NW_ = 4>
__global__ __launch_bounds__(64 * NW_, NW_ == 8 ? 4 : (RH_ == 8 ? 4 : 2))
void patch_irc_kernel(IrcArgs a) {
    using G = IrcGeom<RH_, NW_>;
    static_assert(NW_ == 4 || (NW_ == 8 && RH_ == 16), "eight waves: one 16 x 16 region");
    constexpr int NW = G::NW;
    constexpr int NTHR = G::NTHR, RH = G::RH, RW = G::RW, HWD = G::HWD, NPOS = G::NPOS;
    constexpr int NT1 = G::NT1, J1 = G::J1, J3 = G::J3, CS = G::CS, PS1 = G::PS1;
    constexpr int PWH = G::PWH, PWW = G::PWW, CPW = 4 * PPL;
    constexpr int CP = 16 * MT3;
    static_assert(SPL + PPL <= 8 && (PPL == 2 || PPL == 4) && (SPL == 1 || SPL == 2 || SPL == 4), "K slots of a lane group");
    static_assert(PWH * PWW * CPW <= G::H1_FLOATS, "the previous level's window aliases h1");

    const int hid = a.hid, cout = a.cout;
    const int cskip = a.in.c_skip, cprev = a.in.c_prev, cin = 2 + cskip + cprev;
    const IrcLds L = irc_lds_map(cin, hid, cout, MT3, G::H1_FLOATS, G::H2_HALFS);
    const int HP = L.HP, cinp = L.cinp, hidp = L.hidp;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* taps = reinterpret_cast<float*>(lds_raw + L.taps);
    _Float16* w1h = reinterpret_cast<_Float16*>(lds_raw + L.w1h);
    _Float16* w3h = reinterpret_cast<_Float16*>(lds_raw + L.w3h);
    const int w1_piece = hid * cinp, w3_piece = cout * hidp;       // halfs between the hi and the lo image
    float* s1f = reinterpret_cast<float*>(lds_raw + L.s1f);
    float* b1l = reinterpret_cast<float*>(lds_raw + L.b1);
    float* b2s = reinterpret_cast<float*>(lds_raw + L.b2s);
    float* s3f = reinterpret_cast<float*>(lds_raw + L.s3f);
    float* b3l = reinterpret_cast<float*>(lds_raw + L.b3);
    float* h1 = reinterpret_cast<float*>(lds_raw + L.h1);
    _Float16* h2 = reinterpret_cast<_Float16*>(lds_raw + L.h2);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably uniform: scalar branches, scalar address bases
    const int lrow = lane & 15, lk = lane >> 4;
    // @stamp 0
    // Whose turn it is on a CU (round 6, visits r6v9 / r6v10).  The hardware arbitrates oldest wave first: of the two workgroups a CU holds,
    // the one that started first lives 40.0 k cycles and the one that started 49 cycles later 50.3 k, and the launch lasts as long as the
    // slower one.  The wave's slot on its SIMD (HW_ID.WAVE_ID: 0 for the first resident workgroup, 1 for the second; a newcomer inherits
    // the slot that was freed) tells the two apart without any memory.  a.prio: 0 = leave it to the hardware; 1 / 3 = the two take turns
    // at raised priority per chunk / per stage (equal lifetimes, 48.4 k / 49.5 k: the CU's throughput is NOT conserved -- turns cost
    // 8 % of it -- so the launch gains 1-3 %, not the 10 % the imbalance suggested); 2 = the younger workgroup first (best when the
    // launch runs several generations of workgroups per CU: HyperSeg-S 46.1 -> 44.5 us).
    const int wslot = (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 4) & 1u);        // HW_REG_HW_ID, bits [3:0] = wave slot
    const int prio = a.prio;
    if (prio == 2 && wslot) __builtin_amdgcn_s_setprio(2);
#define HS_IRC_TURN(phase) do { if (prio & 1) { if (((phase) + wslot) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); } } while (0)
    HS_IRC_TURN(0);
    // blocks go round-robin to the 8 XCDs: relabel so that every XCD owns a contiguous range -- the regions of one patch
    // (consecutive indices) then share one L2 for the patch's bank and their common halo rows
    int blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
    const int nsub = a.sub_y * a.sub_x;
    const int patch = (int)div_magic((unsigned)blk, a.m_nsub), sub = blk - patch * nsub;
    const int sy = (int)div_magic((unsigned)sub, a.m_subx), sx = sub - sy * a.sub_x;
    const int pib = (int)div_magic((unsigned)patch, a.m_fw), pj = patch - pib * a.fw;
    const int b = (int)div_magic((unsigned)pib, a.m_fh), pi = pib - b * a.fh;
    const int y0 = pi * a.ph + sy * RH, x0 = pj * a.pw + sx * RW;
    const int H = a.in.H, W = a.in.W;
    const unsigned plane = (unsigned)H * (unsigned)W;
    const float* __restrict__ bank = a.bank + (size_t)patch * (size_t)a.ld;
    const int off_kd = cin * hid, off_w3 = off_kd + 9 * hid;

    // ================================================ prologue ==========================================================
    // Round-3 order:  all loads (bank DMA + tiles) | B | split the bank LDS -> LDS | B | tiles -> scratch | B | B fragments | B --
    // 24.6 k of a workgroup's 52 k cycles, every phase waiting for the one before (profiles/round3_irc_phase_cycles_a6_rh16.txt:
    // 9.4 k for the chip-wide load burst, 5.5 k for the split, 4.4 k + 5.0 k for tiles and fragments).  Round 4:
    //   (a) tile loads (skip, previous level, BatchNorm rows) first and ALONE: 28 of the 45 KB a workgroup asks for;
    //   (b) tiles -> scratch, then the bank's LDS-DMA is issued into its OWN landing zone and is in flight under
    //   (c) the B fragments (the VALU-heaviest phase needs no weight);
    //   (d) the split, cheap form: ONE scale per matrix (W1, W3) instead of one per row -- the 2^15 window of the f16 pieces
    //       leaves 15 binades of room below the matrix maximum before a row loses f32-class accuracy (hs_ir_math in
    //       include/hyperseg_hip.h) -- so the pass is a stream of aligned quads: ds_read_b128, 4 x v_fma_mix pairs, two ds_write_b64,
    //       no per-row bookkeeping (round 3: 212 row jobs of <= 18 values, ~260 vector instructions per wave).
    //   scratch = h1 | h2 (dead until pw1(0)):
    //               SK[4][SKPL][SPL]  skip values of lane group kg at halo position pos (one ds_read of SPL floats, conflict-free:
    //               the planes are 0 (mod 64) dwords apart) | WN[4][WNP][PPL]  low-res window of the previous level, likewise |
    //               TY[HH][8], TX[HWD][8]  per halo row / column: bilinear taps {offset0, offset1, l0, l1} and the coordinate
    unsigned char* scr = lds_raw + L.h1;
    const int raw_chunks = L.raw_chunks;
    float* raw = reinterpret_cast<float*>(lds_raw + L.raw);
    float* raw_s1 = reinterpret_cast<float*>(lds_raw + L.raw_rows);
    float* raw_s2 = raw_s1 + HP;
    float* raw_s3 = raw_s2 + HP;
    unsigned* red = reinterpret_cast<unsigned*>(lds_raw + L.red);
    constexpr int SKP = 192, WNP = ((PWH * PWW + 63) / 64) * 64;
    constexpr int SKPL = ((NPOS + 63) / 64) * 64;
    static_assert(SKP == 192 && (RH != 8 || SKPL == SKP), "skip planes hold every halo position, 0 (mod 64) dwords apart");
    float* SK = reinterpret_cast<float*>(scr);
    float* WN = SK + 4 * SKPL * SPL;
    float* TY = WN + 4 * WNP * PPL;
    float* TX = TY + G::HH * 8;
    static_assert((4 * SKPL * SPL + 4 * WNP * PPL + (G::HH + HWD) * 8) * 4 <= G::H1_FLOATS * 4 + G::H2_HALFS * 2, "phase-2 scratch fits h1 | h2");

    // (1) skip tile.  Wave w loads the channels of lane group kg = w (w SPL .. w SPL + SPL - 1); lanes 0 .. 59 = (halo row u of a
    //     pass of 10 rows, 16-byte segment sg of columns x0 - 4 + 4 sg ..): the channel is wave-uniform (scalar base), the per-lane
    //     offset is computed once per pass, and a thread ends up with an SPL-channel x 4-column block -- what phase 2 stores as one
    //     SPL-float vector per halo position.  Segments that would leave the image are clamped inside it: their halo column then
    //     comes from the reflected interior column (phase 2).
    // wave w loads for lane group kg = w & 3; with eight waves the two passes of the skip tile go to the two halves of the workgroup
    const int kgw = wave & 3, wv_hi = wave >> 2;
    constexpr int SKPASS_ALL = (G::HH + 9) / 10;                       // 1 (RH = 8) or 2 (RH = 16) passes of 10 halo rows
    constexpr int SKPASS = NW == 8 ? 1 : SKPASS_ALL;                   // passes per wave
    static_assert(NW == 4 || SKPASS_ALL == 2, "eight waves share two passes");
    f32x4 sk4[SKPASS][SPL];
    const int sk_ul = lane / 6, sk_sg = lane - 6 * sk_ul;              // row of the pass (0 .. 9 live), segment
    {
        const float* __restrict__ skb = a.in.skip + (size_t)b * cskip * plane;
        const int c0 = min(max(x0 - 4 + 4 * sk_sg, 0), W - 4);
#pragma unroll
        for (int pp = 0; pp < SKPASS; ++pp) {
            const int u = min((NW == 8 ? wv_hi : pp) * 10 + sk_ul, G::HH - 1);
            const int yy = pad_index(y0 + u - 1, H, HS_PAD_REFLECT);
            const unsigned off = (__umul24((unsigned)yy, (unsigned)W) + (unsigned)c0) << 2;
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const int ch = kgw * SPL + k;                          // uniform; no branch around the load: clamp + mask
                const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(skb + (size_t)min(ch, cskip - 1) * plane) + (size_t)off);
                const float mk = ch < cskip ? 1.0f : 0.0f;
                sk4[pp][k] = f32x4{v[0] * mk, v[1] * mk, v[2] * mk, v[3] * mk};
            }
        }
    }
    // (2) low-res window of the previous level: rows [ly0, ly0 + PWH) x cols [lx0, lx0 + PWW), clamped at the border.  Wave w loads
    //     the PPL channels of lane group w, one channel per instruction: lanes 0 .. 4 PWH - 1 = (row r, 16-byte segment of columns
    //     x0 / 2 - 4 + 4 s ..).
    const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;
    static_assert(4 * PWH <= 64, "a window channel fits one wave");
    f32x4 pw4[PPL];
    const int pw_r = min(lane >> 2, PWH - 1), pw_s = lane & 3;
#pragma unroll
    for (int k = 0; k < PPL; ++k) pw4[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (NW == 4 || wave < 4) {
        const float* __restrict__ pvb = a.in.prev + (size_t)b * cprev * a.in.Hp * a.in.Wp;
        const int yy = min(max(ly0 + pw_r, 0), a.in.Hp - 1);
        const int c0 = min(max((x0 >> 1) - 4 + 4 * pw_s, 0), a.in.Wp - 4);
        const unsigned off = (__umul24((unsigned)yy, (unsigned)a.in.Wp) + (unsigned)c0) << 2;
        const size_t cpl = (size_t)a.in.Hp * a.in.Wp;
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int ch = kgw * PPL + k;                              // uniform (eight waves: the upper four re-load the same window; only the lower four store it)
            const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(pvb + (size_t)min(ch, cprev - 1) * cpl) + (size_t)off);
            const float mk = ch < cprev ? 1.0f : 0.0f;
            pw4[k] = f32x4{v[0] * mk, v[1] * mk, v[2] * mk, v[3] * mk};
        }
    }
    // (3) BatchNorm rows: thread t takes row t (zero beyond the real channels)
    static_assert(NTHR >= 96, "one pass over the BatchNorm rows");
    const float bn_mh = tid < hid ? 1.0f : 0.0f, bn_mo = tid < cout ? 1.0f : 0.0f;
    const unsigned bn_th = (unsigned)min(tid, hid - 1), bn_to = (unsigned)min(tid, cout - 1);
    const float r_s1 = ldg(a.s1, bn_th), r_b1 = ldg(a.b1, bn_th), r_s2 = ldg(a.s2, bn_th), r_b2 = ldg(a.b2, bn_th);
    const float r_s3 = ldg(a.s3, bn_to), r_b3 = ldg(a.b3, bn_to);                       // first use below the tap tables
    // (3b) the tap tables need no loaded value: built HERE, under the loads' round trip (round 6: they used to follow the first wait;
    //      ~100 dependent vector instructions with two divisions on wave 0, the wave every barrier then waits for)
    if (tid < G::HH + HWD) {          // per halo row: {row offset 0, row offset 1 (in window positions), l0, l1, coordinate y}; columns likewise
        const bool isrow = tid < G::HH;
        const int i = isrow ? tid : tid - G::HH;
        const int p = isrow ? pad_index(y0 + i - 1, H, HS_PAD_REFLECT) : pad_index(x0 + i - 1, W, HS_PAD_REFLECT);
        const Tap t = bilinear_tap(p, isrow ? a.in.scale_y : a.in.scale_x, isrow ? a.in.Hp : a.in.Wp);
        const int o0 = isrow ? (t.i0 - ly0) * PWW : t.i0 - lx0, o1 = isrow ? (t.i1 - ly0) * PWW : t.i1 - lx0;
        float* d = (isrow ? TY : TX) + i * 8;
        d[0] = __int_as_float(o0); d[1] = __int_as_float(o1); d[2] = t.l0; d[3] = t.l1;
        d[4] = isrow ? linspace_pm1(p, H, a.in.step_y) : linspace_pm1(p, W, a.in.step_x);
    }
    if (tid < 16) reinterpret_cast<unsigned*>(lds_raw + L.w3l + w3_piece * 2)[tid] = 0u;          // the 64 zero bytes
    __builtin_amdgcn_sched_barrier(0);
    if (tid < HP) { raw_s1[tid] = r_s1 * bn_mh; raw_s2[tid] = r_s2 * bn_mh; b1l[tid] = r_b1 * bn_mh; b2s[tid] = r_b2 * bn_mh * IRC_H2_SCALE; if (tid >= hid) s1f[tid] = 0.0f; }
    if (tid < CP) { raw_s3[tid] = r_s3 * bn_mo; b3l[tid] = r_b3 * bn_mo; if (tid >= cout) s3f[tid] = 0.0f; }
    // @stamp 1

    // ================================================ phase 2: tiles and tap tables into the scratch ====================
    {   // skip tile: the thread's SPL-channel x 4-column block -> one SPL-float vector per halo position of lane group kg = wave
        const bool left = sk_sg == 0, right = sk_sg == 5;
        const bool clamp_l = x0 == 0, clamp_r = x0 + RW == W;           // uniform: the region touches the image border
        using skv = __attribute__((ext_vector_type(SPL))) float;
#pragma unroll
        for (int pp = 0; pp < SKPASS; ++pp) {
            const int u = (NW == 8 ? wv_hi : pp) * 10 + sk_ul;
            if (sk_ul < 10 && u < G::HH) {
                float* d = SK + (kgw * SKPL + u * HWD + 4 * sk_sg - 3) * SPL;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int vv = 4 * sk_sg - 3 + i;
                    if (vv >= 0 && vv < HWD) {
                        // halo column 0 is image column x0 - 1 (element 3 of segment 0) or, reflected at the left border, column 1
                        // (element 1 of the clamped segment); halo column 17 likewise
                        const int src = (left && clamp_l) ? 1 : (right && clamp_r) ? 2 : i;
                        if constexpr (SPL == 1) d[i] = src == i ? sk4[pp][0][i] : (src == 1 ? sk4[pp][0][1] : sk4[pp][0][2]);
                        else {
                            skv o;
#pragma unroll
                            for (int k = 0; k < SPL; ++k) o[k] = src == i ? sk4[pp][k][i] : (src == 1 ? sk4[pp][k][1] : sk4[pp][k][2]);
                            *reinterpret_cast<skv*>(d + i * SPL) = o;
                        }
                    }
                }
            }
        }
    }
    if (lane < 4 * PWH && wave < 4) {
        using pwv = __attribute__((ext_vector_type(PPL))) float;
        float* d = WN + (wave * WNP + pw_r * PWW + 4 * pw_s - 3) * PPL;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 4 * pw_s - 3 + i;
            if (q >= 0 && q < PWW) {
                pwv o;
#pragma unroll
                for (int k = 0; k < PPL; ++k) o[k] = pw4[k][i];
                *reinterpret_cast<pwv*>(d + i * PPL) = o;
            }
        }
    }
    // (4) the patch's bank: LDS-DMA into its landing zone, 16 bytes per lane, whole 1 KB pieces (the tail piece re-reads the row's
    //     last 16 bytes).  Issued HERE -- every register load above has been consumed -- and in flight until the barrier after the
    //     B fragments; the barrier right below must not wait for it (a plain __syncthreads() would: it drains vmcnt).
    {
        const unsigned char* gb = reinterpret_cast<const unsigned char*>(bank);
        const unsigned last16 = (unsigned)a.ld * 4u - 16u;
        for (int c = wave; c < raw_chunks; c += NW) {
            const unsigned off = min((unsigned)(c * 1024 + lane * 16), last16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + off),
                                             (__attribute__((address_space(3))) void*)(lds_raw + L.raw + c * 1024), 16, 0, 0);
        }
    }
    // @stamp 2
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // tiles visible; the DMA stays in flight
    // @stamp 3
    HS_IRC_TURN(1);

    // ================================================ phase 3: the B fragments ==========================================
    // Per halo position the lane's SPL skip values, PPL bilinear previous-level values and (tail) the two coordinates;
    // scale = the position's maximum over all its channels -> 2^15.
    half8 bq[J1][3];                                       // [hi | lo | tail]
    float invb[J1];
    int hoff[J1];
    // Tiles in groups of GT: first every tile's tap-table rows, then every tile's skip vector and bilinear taps -- two LDS round
    // trips per group.  (Written tile by tile the compiler drained the LDS queue 19 times for 54 reads: visit r4b, 5.8 k cycles.)
    constexpr int GT = NW == 8 ? 2 : 3;
#pragma unroll
    for (int g0 = 0; g0 < J1; g0 += GT) {
        constexpr int SV = SPL == 4 ? 4 : (SPL == 2 ? 2 : 1);
        f32x4 ty[GT], tx[GT];
        float cy[GT], cx[GT];
        int pcs[GT];
        bool lives[GT];
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int jt = g0 + t;
            if (jt < J1) {
                const int pos = (wave + NW * jt) * 16 + lrow;
                lives[t] = pos < NPOS;                               // the last tile is short; tiles past NT1 are all dead
                pcs[t] = lives[t] ? pos : 0;
                const int u = pcs[t] / HWD, v = pcs[t] - u * HWD;
                hoff[jt] = lives[t] ? u * CS + v : NPOS;             // dead lanes store to the plane's padding
                ty[t] = *reinterpret_cast<const f32x4*>(TY + u * 8); tx[t] = *reinterpret_cast<const f32x4*>(TX + v * 8);
                cy[t] = TY[u * 8 + 4]; cx[t] = TX[v * 8 + 4];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 pv[GT][4];
        float sv[GT][SV];
        const float* pb = WN + lk * WNP * PPL;
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            if (g0 + t < J1) {
                const int r0 = __float_as_int(ty[t][0]), r1 = __float_as_int(ty[t][1]), q0 = __float_as_int(tx[t][0]), q1 = __float_as_int(tx[t][1]);
                if constexpr (SPL == 4) {
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(SK + (lk * SKPL + pcs[t]) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) sv[t][j] = s4[j];
                } else if constexpr (SPL == 2) {
                    using f32x2 = __attribute__((ext_vector_type(2))) float;
                    const f32x2 s2 = *reinterpret_cast<const f32x2*>(SK + (lk * SKPL + pcs[t]) * 2);
                    sv[t][0] = s2[0]; sv[t][1] = s2[1];
                } else {
                    sv[t][0] = SK[lk * SKPL + pcs[t]];
                }
                if constexpr (PPL == 4) {
                    pv[t][0] = *reinterpret_cast<const f32x4*>(pb + (r0 + q0) * 4); pv[t][1] = *reinterpret_cast<const f32x4*>(pb + (r0 + q1) * 4);
                    pv[t][2] = *reinterpret_cast<const f32x4*>(pb + (r1 + q0) * 4); pv[t][3] = *reinterpret_cast<const f32x4*>(pb + (r1 + q1) * 4);
                } else {
                    using f32x2 = __attribute__((ext_vector_type(2))) float;
                    const f32x2 p00 = *reinterpret_cast<const f32x2*>(pb + (r0 + q0) * 2), p01 = *reinterpret_cast<const f32x2*>(pb + (r0 + q1) * 2);
                    const f32x2 p10 = *reinterpret_cast<const f32x2*>(pb + (r1 + q0) * 2), p11 = *reinterpret_cast<const f32x2*>(pb + (r1 + q1) * 2);
                    pv[t][0] = f32x4{p00[0], p00[1], 0.f, 0.f}; pv[t][1] = f32x4{p01[0], p01[1], 0.f, 0.f};
                    pv[t][2] = f32x4{p10[0], p10[1], 0.f, 0.f}; pv[t][3] = f32x4{p11[0], p11[1], 0.f, 0.f};
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int jt = g0 + t;
            if (jt < J1) {
                const bool live = lives[t];
                const float w00 = ty[t][2] * tx[t][2], w01 = ty[t][2] * tx[t][3], w10 = ty[t][3] * tx[t][2], w11 = ty[t][3] * tx[t][3];
                float kv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) kv[j] = 0.0f;
#pragma unroll
                for (int j = 0; j < SV; ++j) kv[j] = sv[t][j];
#pragma unroll
                for (int j = 0; j < PPL; ++j) kv[SPL + j] = fmaf(w11, pv[t][3][j], fmaf(w10, pv[t][2][j], fmaf(w01, pv[t][1][j], w00 * pv[t][0][j])));
                const float cxl = live ? cx[t] : 0.0f, cyl = live ? cy[t] : 0.0f;
                unsigned m = max(absbits(cxl), absbits(cyl));
#pragma unroll
                for (int j = 0; j < SPL + PPL; ++j) { kv[j] = live ? kv[j] : 0.0f; m = max(m, absbits(kv[j])); }
                {   // maximum over the 4 lane groups that hold this position: lanes n, n + 16, n + 32, n + 48
                    auto r16 = __builtin_amdgcn_permlane16_swap(m, m, false, false);
                    m = max(r16[0], r16[1]);
                    auto r32 = __builtin_amdgcn_permlane32_swap(m, m, false, false);
                    m = max(r32[0], r32[1]);
                }
                const int eb = irc_exp_of(m);
                const float sc = irc_scale_of(eb);
                invb[jt] = irc_inv_scale_of(eb);
                // K slots of the lane group: [skip run (SPL) | previous-level run (PPL) | zeros]
                u32x4 qh, ql;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (2 * j < SPL + PPL) { unsigned hh, ll; split2(kv[2 * j], kv[2 * j + 1], sc, hh, ll); qh[j] = hh; ql[j] = ll; }
                    else { qh[j] = 0u; ql[j] = 0u; }
                }
                bq[jt][0] = __builtin_bit_cast(half8, qh);
                bq[jt][1] = __builtin_bit_cast(half8, ql);
                unsigned ch, cl;
                split2(cxl, cyl, sc, ch, cl);
                // tail: the lane group picks the product -- 0: ah * bh, 1: al * bh, 2: ah * bl, 3: nothing
                const u32x4 qt = {lk < 2 ? ch : (lk == 2 ? cl : 0u), 0u, 0u, 0u};
                bq[jt][2] = __builtin_bit_cast(half8, qt);
            }
        }
    }
    // @stamp 4
    // pw3: transpose-read addresses (halfs, relative to h2).  Lane i of a 16-lane group supplies the 4-pixel run (i & 3) of
    // plane (i >> 2) of its block and receives pixel i of the block's 4 planes.  Row t of plane p lives in 32-byte slot t ^ swz(p).
    int trp[2];                                            // plane base (halfs) of read rd = 0, 1; + the row slot per tile
    int trs[2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int p = 8 * (lk & 1) + 4 * rd + (lrow >> 2);
        trp[rd] = p * G::H2_PLANE + 4 * (lrow & 3);
        trs[rd] = irc_h2_swz(p);
    }
    const int zero_h = (int)((lds_raw + L.w3l + w3_piece * 2) - (lds_raw + L.h2)) / 2;    // the zero block, in halfs relative to h2
    f32x4 acc3[MT3][J3];
#pragma unroll
    for (int m = 0; m < MT3; ++m)
#pragma unroll
        for (int jt = 0; jt < J3; ++jt) acc3[m][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                       // the scratch is dead; every wave's DMA pieces have landed (vmcnt(0) + barrier)
    // @stamp 5
    HS_IRC_TURN(2);

    // ================================================ phase 1: split the bank, LDS -> LDS ==============================
    const int n1 = cin * hid, n3 = hid * cout;
    int eb1 = 0, eb3 = 0;
    if ((cin & 1) == 0 && (hid & 3) == 0) {
        // quads of 4 consecutive weights: W1 [0, n1) and W3 [off_w3, off_w3 + n3) are whole, 16-byte aligned quads here, and with an
        // even cin / hid the f16 images are the matrices' own row-major order (cinp = cin, hidp = hid)
        const int nq1 = n1 >> 2, nq = nq1 + (n3 >> 2);
        // QB quads per thread and round, ALL of a round's ds_read_b128 issued before the first use and kept in registers across
        // the maximum's barrier (a run-time loop of read -> use -> write round trips costs one LDS latency per quad: 4.7 k cycles
        // for 3.5 quads per thread in visit r4b)
        constexpr int QB = NW == 8 ? 2 : 4;
        using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
        for (int q0 = 0; q0 < nq; q0 += QB * NTHR) {                 // one round at every BASELINE shape (<= 1024 quads)
            f32x4 v[QB];
            unsigned m1 = 0, m3 = 0;
#pragma unroll
            for (int i = 0; i < QB; ++i) {
                const int q = min(q0 + tid + i * NTHR, nq - 1);
                v[i] = *reinterpret_cast<const f32x4*>(raw + (q < nq1 ? 4 * q : off_w3 + 4 * (q - nq1)));
            }
#pragma unroll
            for (int i = 0; i < QB; ++i) {
                const int q = q0 + tid + i * NTHR;
                const unsigned m = q < nq ? max(max(absbits(v[i][0]), absbits(v[i][1])), max(absbits(v[i][2]), absbits(v[i][3]))) : 0u;
                m1 = max(m1, q < nq1 ? m : 0u); m3 = max(m3, q < nq1 ? 0u : m);
            }
            if (q0 == 0) {                                           // the scales come from the first round's maxima ...
                m1 = rowmax16_u(m1); m3 = rowmax16_u(m3);
                m1 = max(max((unsigned)__builtin_amdgcn_readlane((int)m1, 0), (unsigned)__builtin_amdgcn_readlane((int)m1, 16)),
                         max((unsigned)__builtin_amdgcn_readlane((int)m1, 32), (unsigned)__builtin_amdgcn_readlane((int)m1, 48)));
                m3 = max(max((unsigned)__builtin_amdgcn_readlane((int)m3, 0), (unsigned)__builtin_amdgcn_readlane((int)m3, 16)),
                         max((unsigned)__builtin_amdgcn_readlane((int)m3, 32), (unsigned)__builtin_amdgcn_readlane((int)m3, 48)));
                if (nq > QB * NTHR) {                                // ... unless the matrices take more than one: a maximum pass first
                    for (int q = QB * NTHR + tid; q < nq; q += NTHR) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(raw + (q < nq1 ? 4 * q : off_w3 + 4 * (q - nq1)));
                        const unsigned m = max(max(absbits(w[0]), absbits(w[1])), max(absbits(w[2]), absbits(w[3])));
                        m1 = max(m1, q < nq1 ? m : 0u); m3 = max(m3, q < nq1 ? 0u : m);
                    }
                    m1 = rowmax16_u(m1); m3 = rowmax16_u(m3);
                    m1 = max(max((unsigned)__builtin_amdgcn_readlane((int)m1, 0), (unsigned)__builtin_amdgcn_readlane((int)m1, 16)),
                             max((unsigned)__builtin_amdgcn_readlane((int)m1, 32), (unsigned)__builtin_amdgcn_readlane((int)m1, 48)));
                    m3 = max(max((unsigned)__builtin_amdgcn_readlane((int)m3, 0), (unsigned)__builtin_amdgcn_readlane((int)m3, 16)),
                             max((unsigned)__builtin_amdgcn_readlane((int)m3, 32), (unsigned)__builtin_amdgcn_readlane((int)m3, 48)));
                }
                if (lane == 0) { red[2 * wave] = m1; red[2 * wave + 1] = m3; }
                __syncthreads();
            }
            unsigned g1 = 0, g3 = 0;
#pragma unroll
            for (int w4 = 0; w4 < NW / 2; ++w4) {                    // {m1, m3} of two waves per 16-byte read
                const u32x4 rr = *reinterpret_cast<const u32x4*>(red + 4 * w4);
                g1 = max(g1, max(rr[0], rr[2])); g3 = max(g3, max(rr[1], rr[3]));
            }
            eb1 = irc_exp_of(g1);
            eb3 = irc_exp_of(g3);
            const float sc1m = irc_scale_of(eb1), sc3m = irc_scale_of(eb3);
#pragma unroll
            for (int i = 0; i < QB; ++i) {
                const int q = q0 + tid + i * NTHR;
                const bool is1 = q < nq1;
                const int e = is1 ? 4 * q : 4 * (q - nq1);
                const float sc = is1 ? sc1m : sc3m;
                u32x2 hi, lo;
                { unsigned h, l; split2(v[i][0], v[i][1], sc, h, l); hi[0] = h; lo[0] = l; }
                { unsigned h, l; split2(v[i][2], v[i][3], sc, h, l); hi[1] = h; lo[1] = l; }
                if (q < nq) {
                    _Float16* d = (is1 ? w1h : w3h) + e;
                    *reinterpret_cast<u32x2*>(d) = hi;
                    *reinterpret_cast<u32x2*>(d + (is1 ? w1_piece : w3_piece)) = lo;
                }
            }
        }
        if (tid < hid) s1f[tid] = raw_s1[tid] * irc_inv_scale_of(eb1);
        if (tid < cout) s3f[tid] = raw_s3[tid] * (irc_inv_scale_of(eb3) * (1.0f / IRC_H2_SCALE));
    } else {
        // odd channel counts: one JOB per thread, a contiguous piece of one matrix row (a W1 row = 2 jobs, a W3 row = 4, <= 18 values
        // each), per-row scale from the jobs' maxima combined over the 2 / 4 neighbouring lanes on the DPP path -- the round-3 form,
        // which also writes the zero pad column of an odd row length
        constexpr int MAXP = 12;                                       // value pairs per job: cin <= 34 -> 9, hid <= 96 -> 12
        const int ca = (((cin + 1) >> 1) + 1) & ~1, cb = (((hid + 3) >> 2) + 1) & ~1;   // piece lengths (even, rounded up): W1 rows in 2, W3 rows in 4
        const int j1n = 2 * hid, j3b = (j1n + 3) & ~3, jn = j3b + 4 * cout;
        for (int job = tid; job < ((jn + 3) & ~3); job += NTHR) {      // whole quads enter together (the DPP steps read neighbours)
            const bool is1 = job < j1n, is3 = job >= j3b && job < jn;
            const int row = is1 ? job >> 1 : (job - j3b) >> 2, part = is1 ? job & 1 : (job - j3b) & 3;
            const int len_row = is1 ? cin : hid, piece = is1 ? ca : cb;
            const int c_lo = part * piece, cnt = (is1 || is3) ? min(max(len_row - c_lo, 0), piece) : 0;
            const float* rp = raw + (is1 ? row * cin : off_w3 + row * hid) + c_lo;
            float v[2 * MAXP];
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < 2 * MAXP; ++i) {
                v[i] = i < cnt ? rp[i] : 0.0f;                         // LDS reads; a predicated read costs nothing extra
                m = max(m, absbits(v[i]));
            }
            m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false));          // lane ^ 1: the row's other W1 piece
            const unsigned m4 = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false));   // lane ^ 2: W3 rows span a quad
            const int eb = irc_exp_of(is1 ? m : m4);
            const float sc = irc_scale_of(eb);
            _Float16* dbase = is1 ? w1h + row * cinp + c_lo : w3h + row * hidp + c_lo;      // even half index: 4-byte aligned
            unsigned* dh = reinterpret_cast<unsigned*>(dbase);
            unsigned* dl = reinterpret_cast<unsigned*>(dbase + (is1 ? w1_piece : w3_piece));
            const int npair = (min(cnt, (is1 ? cinp : hidp) - c_lo) + 1) >> 1;     // the pad column of an odd row length is written as zero
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                unsigned hi, lo;
                split2(v[2 * i], v[2 * i + 1], sc, hi, lo);
                if (i < npair) { dh[i] = hi; dl[i] = lo; }
            }
            if (is1 && part == 0) s1f[row] = raw_s1[row] * irc_inv_scale_of(eb);
            if (is3 && part == 0) s3f[row] = raw_s3[row] * (irc_inv_scale_of(eb) * (1.0f / IRC_H2_SCALE));
        }
    }
    {   // taps with BN2's scale folded in: 9 hid <= 864 values, the rounds' LDS reads issued together
        constexpr int TB = NW == 8 ? 2 : 4;
        float tv[TB], ts[TB];
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            const int e = min(tid + i * NTHR, 9 * hid - 1);
            tv[i] = raw[off_kd + e]; ts[i] = raw_s2[(unsigned)e / 9u];
        }
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            const int e = tid + i * NTHR;
            if (e < 9 * hid) taps[e] = tv[i] * (ts[i] * IRC_H2_SCALE);
        }
        static_assert(TB * NTHR >= 9 * 96, "one round covers the taps of the widest block");
    }
    // @stamp 6
    __syncthreads();                                       // the f16 images are complete: h1 / h2 from here on
    // @stamp 7
    HS_IRC_TURN(3);

    // depthwise thread map: lane = half | row_lo << 1 | channel-of-the-wave << 3 | row_hi << 5; channel = wave + 4 j; with RH = 16
    // a thread takes rows r and r + 8 of its channel (same taps).  This is the map the conflict model was run on
    // (tools/lds_conflicts.py; rows + 8 shift every lane by 144 floats == 16 (mod 64): the same bank picture).
    // eight waves: waves w and w + 4 share the channels of wave w & 3 and take rows 0-7 / 8-15 (the + 8 rows of the four-wave form)
    constexpr int DWI = NW == 8 ? 1 : RH / 8;
    const int dw_half = lane & 1, dw_row = (((lane >> 1) & 3) | ((lane >> 5) << 2)) + (NW == 8 ? 8 * wv_hi : 0), dw_j = (lane >> 3) & 3;
    const int dw_c = kgw + 4 * dw_j;                       // hidden channel of the chunk; its h1 slot is 4 * (wave & 3) + j
    const int dw_sw = irc_h2_swz(dw_c);
    const float* dw_src = h1 + (4 * kgw + dw_j) * PS1 + dw_row * CS + 8 * dw_half;
    _Float16* dw_dst = h2 + dw_c * G::H2_PLANE + 8 * dw_half;

    // ================================================ stages ==========================================================
    auto stage_pw1 = [&](int h0) {
        // A fragments in the bank's own order: row (h0 + lrow), K slots of lane group lk
        const _Float16* ar = w1h + (h0 + lrow) * cinp;
        u32x4 ah = {0u, 0u, 0u, 0u}, al = {0u, 0u, 0u, 0u};
        {
            const _Float16* ps = ar + 2 + lk * SPL;                    // skip run
            const _Float16* pp = ar + 2 + cskip + lk * PPL;            // previous-level run
            _Float16 eh[8], el[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { eh[j] = (_Float16)0.0f; el[j] = (_Float16)0.0f; }
#pragma unroll
            for (int j = 0; j < SPL; ++j) { eh[j] = ps[j]; el[j] = ps[w1_piece + j]; }
#pragma unroll
            for (int j = 0; j < PPL; ++j) { eh[SPL + j] = pp[j]; el[SPL + j] = pp[w1_piece + j]; }
            half8 vh, vl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { vh[j] = eh[j]; vl[j] = el[j]; }
            ah = __builtin_bit_cast(u32x4, vh); al = __builtin_bit_cast(u32x4, vl);
        }
        // tail: coordinates are bank columns 0, 1; lane group 1 multiplies a lo
        const unsigned at0 = *reinterpret_cast<const unsigned*>(ar + (lk == 1 ? w1_piece : 0));
        const half8 a_hi = __builtin_bit_cast(half8, ah), a_lo = __builtin_bit_cast(half8, al);
        const half8 a_t = __builtin_bit_cast(half8, u32x4{at0, 0u, 0u, 0u});
        const f32x4 sc1 = *reinterpret_cast<const f32x4*>(s1f + h0 + 4 * lk);
        const f32x4 sh1 = *reinterpret_cast<const f32x4*>(b1l + h0 + 4 * lk);
#pragma unroll
        for (int jt = 0; jt < J1; ++jt) {
            if ((NT1 % NW == 0) || jt < J1 - 1 || wave < NT1 - NW * (J1 - 1)) {      // uniform: the last round may be short
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_t, bq[jt][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, bq[jt][0], acc, 0, 0, 0);       // lo * hi
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bq[jt][1], acc, 0, 0, 0);       // hi * lo
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bq[jt][0], acc, 0, 0, 0);       // hi * hi
                const float ib = invb[jt];
                float* dst = h1 + lk * PS1 + hoff[jt];                                               // channel 4 lk + r -> slot 4 r + lk
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[4 * r * PS1] = __builtin_amdgcn_fmed3f(fmaf(acc[r], sc1[r] * ib, sh1[r]), 0.0f, 6.0f);
            }
        }
    };
    // depthwise 3x3 + bn2 + relu6 (x 2^12): h1 -> the two f16 planes of h2.  Taps carry s2 * 2^12, the sum starts at b2 * 2^12.
    auto stage_dw = [&](int h0) {
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        const float* kb = taps + (h0 + dw_c) * 9;
        // every LDS read of the stage -- 9 taps, the BN2 offset, 15 DWI row pieces -- in flight before the first FMA (written row by
        // row the compiler waited for the LDS 16 times per stage: with both workgroups of a CU in the same phase nobody covers that)
        float k[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) k[q] = kb[q];
        const float init = b2s[h0 + dw_c];
        f32x2 rv[DWI][3][5];
#pragma unroll
        for (int it = 0; it < DWI; ++it)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int q = 0; q < 5; ++q) rv[it][ky][q] = *reinterpret_cast<const f32x2*>(dw_src + (8 * it + ky) * CS + 2 * q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < DWI; ++it) {
            float o[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) o[v] = init;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float rowv[10];
#pragma unroll
                for (int q = 0; q < 5; ++q) { rowv[2 * q] = rv[it][ky][q][0]; rowv[2 * q + 1] = rv[it][ky][q][1]; }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int v = 0; v < 8; ++v) o[v] = fmaf(k[ky * 3 + kx], rowv[v + kx], o[v]);
            }
            u32x4 hi, lo;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float c0 = __builtin_amdgcn_fmed3f(o[2 * v], 0.0f, 6.0f * IRC_H2_SCALE);
                const float c1 = __builtin_amdgcn_fmed3f(o[2 * v + 1], 0.0f, 6.0f * IRC_H2_SCALE);
                unsigned hh, ll;
                split2(c0, c1, 1.0f, hh, ll);
                hi[v] = hh; lo[v] = ll;
            }
            _Float16* d = dw_dst + (((dw_row + 8 * it) ^ dw_sw) * RW);
            *reinterpret_cast<u32x4*>(d) = hi;
            *reinterpret_cast<u32x4*>(d + 16 * G::H2_PLANE) = lo;
        }
    };
    // pw3: acc3 += W3[:, chunk] . h2;  K = 16 hidden channels x 3 products over two MFMAs:
    //   [ah*bh(0..7) | ah*bh(8..15) | al*bh(0..7) | al*bh(8..15)]  and  [ah*bl(0..7) | ah*bl(8..15) | 0 | 0] (zeros on the B side)
    auto stage_pw3 = [&](int h0) {
        half8 a3[MT3];
#pragma unroll
        for (int m = 0; m < MT3; ++m) {
            const int o = min(16 * m + lrow, cout - 1);
            const _Float16* p = w3h + (lk >> 1) * w3_piece + o * hidp + h0 + 8 * (lk & 1);        // even half index: 4-byte aligned
            const unsigned* p32 = reinterpret_cast<const unsigned*>(p);
            a3[m] = __builtin_bit_cast(half8, u32x4{p32[0], p32[1], p32[2], p32[3]});
        }
        half4tr xr[J3][4];                                             // the stage's 4 J3 transpose reads, all issued before the first MFMA
#pragma unroll
        for (int jt = 0; jt < J3; ++jt) {
            const int t = wave + NW * jt;                               // region row
            const int s0 = ((t ^ trs[0]) * RW), s1 = ((t ^ trs[1]) * RW);
            xr[jt][0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(HS_IRC_LDS_H4(h2 + trp[0] + s0));
            xr[jt][1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(HS_IRC_LDS_H4(h2 + trp[1] + s1));
            const int l0 = lk < 2 ? 16 * G::H2_PLANE + trp[0] + s0 : zero_h, l1 = lk < 2 ? 16 * G::H2_PLANE + trp[1] + s1 : zero_h;
            xr[jt][2] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(HS_IRC_LDS_H4(h2 + l0));
            xr[jt][3] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(HS_IRC_LDS_H4(h2 + l1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < J3; ++jt) {
            half8 bh, bl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bh[e] = (_Float16)xr[jt][0][e]; bh[4 + e] = (_Float16)xr[jt][1][e];
                bl[e] = (_Float16)xr[jt][2][e]; bl[4 + e] = (_Float16)xr[jt][3][e];
            }
#pragma unroll
            for (int m = 0; m < MT3; ++m) {
                // operand roles swapped (round 6): D[pixel][channel] -- a lane ends up with 4 CONSECUTIVE PIXELS (4 lk .. 4 lk + 3) of ONE
                // output channel (16 m + lrow), i.e. one 16-byte store per row where D[channel][pixel] cost four 4-byte stores
                acc3[m][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, a3[m], acc3[m][jt], 0, 0, 0);
                acc3[m][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, a3[m], acc3[m][jt], 0, 0, 0);
            }
        }
    };

    // ---- chunk loop:  pw1(0) | B | dw(c) | B | pw3(c), pw1(c + 1) | B | ... : no global memory traffic inside ----
    stage_pw1(0);
    __syncthreads();
    // @stamp 9
    for (int h0 = 0; h0 < HP; h0 += 16) {
        HS_IRC_TURN(prio == 3 ? (h0 >> 3) : (h0 >> 4));
        stage_dw(h0);
        // @stamp 10 + 4 * (h0 < 32 ? h0 / 16 : 2)
        __syncthreads();
        // @stamp 11 + 4 * (h0 < 32 ? h0 / 16 : 2)
        if (prio == 3) HS_IRC_TURN((h0 >> 3) + 1);
        stage_pw3(h0);
        // @stamp 12 + 4 * (h0 < 32 ? h0 / 16 : 2)
        if (h0 + 16 < HP) {
            stage_pw1(h0 + 16);
            __syncthreads();
            // @stamp 13 + 4 * (h0 < 32 ? h0 / 16 : 2)
        }
    }

    __builtin_amdgcn_s_setprio(0);
    // ---- epilogue: bn3 + store ----
    float* __restrict__ yb = a.y + (size_t)b * cout * plane;
    // the epilogue is store-ISSUE bound (visit r6v3: a quarter of the store instructions, timing only, 26.5 -> 25.8 us): with the swapped
    // operand roles of pw3 a lane holds pixels 4 lk .. 4 lk + 3 of channel 16 m + lrow -- one BatchNorm row per lane and m, and
    // MT3 J3 16-byte stores per lane where the D[channel][pixel] layout issued 4 MT3 J3 dword stores
#pragma unroll
    for (int m = 0; m < MT3; ++m) {
        const int o = m * 16 + lrow;
        if (o < cout) {
            const float sc = s3f[o], sh = b3l[o];
            float* __restrict__ yo = yb + (size_t)((unsigned)o * plane + (unsigned)((y0 + wave) * W + x0 + 4 * lk));
#pragma unroll
            for (int jt = 0; jt < J3; ++jt)
                *reinterpret_cast<f32x4*>(yo + (unsigned)(NW * jt * W)) =
                    f32x4{fmaf(acc3[m][jt][0], sc, sh), fmaf(acc3[m][jt][1], sc, sh), fmaf(acc3[m][jt][2], sc, sh), fmaf(acc3[m][jt][3], sc, sh)};
        }
    }
    // @stamp 24
}

template <int SPL, int PPL, int MT3, int RH_, int NW_ = 4>
static int launch_irc(IrcArgs& a, hipStream_t stream) {
    using G = IrcGeom<RH_, NW_>;
    const int cin = 2 + a.in.c_skip + a.in.c_prev;
    const IrcLds L = irc_lds_map(cin, a.hid, a.cout, MT3, G::H1_FLOATS, G::H2_HALFS);
    if (L.total > 160 * 1024) return 1;
    if (a.y == nullptr) return HS_OK;                         // route query (hs_patch_ir_route): covered, nothing is launched
    if (L.total > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};
        const int e = allow_full_lds((const void*)patch_irc_kernel<SPL, PPL, MT3, RH_, NW_>, done);
        if (e != HS_OK) return e;
    }
    a.sub_y = a.ph / G::RH; a.sub_x = a.pw / G::RW;
    a.m_nsub = magic_of((unsigned)(a.sub_y * a.sub_x)); a.m_subx = magic_of((unsigned)a.sub_x);
    a.m_fw = magic_of((unsigned)a.fw); a.m_fh = magic_of((unsigned)a.fh);
    const long blocks = (long)a.in.B * a.fh * a.fw * a.sub_y * a.sub_x;
    {   // one generation of workgroups (<= 2 per CU): turns per stage; several generations: the younger workgroup first (visit r6v11)
        static const int forced = [] { const char* e = getenv("HS_IRC_PRIO"); return e ? atoi(e) : -1; }();      // dev A/B knob
        a.prio = forced >= 0 ? forced : (blocks > 2 * 256 ? 2 : 3);
    }
    hipLaunchKernelGGL((patch_irc_kernel<SPL, PPL, MT3, RH_, NW_>), dim3((unsigned)blocks), dim3(64 * G::NW), (size_t)L.total, stream, a);
    return launch_status();
}

// Returns 1 when the shape is outside what the kernel covers (the caller then takes the exact-f32 kernels).
int try_launch_irc(const StageIn& in, int fh, int fw, const float* bank, long ld, int hid, int c_out,
                   const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                   float* y, hipStream_t stream) {
    IrcArgs a;
    a.in = in; a.fh = fh; a.fw = fw; a.ph = in.H / fh; a.pw = in.W / fw;
    a.bank = bank; a.ld = ld; a.hid = hid; a.cout = c_out;
    a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3; a.y = y;
    const int cs = in.c_skip, cp = in.c_prev;
    if (!in.coords || in.prev_mode != HS_PREV_BILINEAR || cs <= 0 || cp <= 0) return 1;
    if (in.Hp * 2 != in.H || in.Wp * 2 != in.W) return 1;      // the LDS window assumes the exact 2x pyramid
    if (in.H >= 32768 || in.W >= 32768 || (size_t)in.H * in.W >= (1u << 24)) return 1;      // packed (yy << 16 | xx) positions; 24-bit multiplies
    if (a.pw % 16 != 0 || a.ph % 8 != 0) return 1;
    if (cs > 16 || cp > 16 || c_out > 32 || hid > 96 || hid < 2) return 1;
    // 32-bit element offsets from uniform bases
    if ((size_t)in.H * in.W * (size_t)(c_out > cs ? c_out : cs) >= (1u << 30) || (size_t)cp * in.Hp * in.Wp >= (1u << 30)) return 1;
    {   // rows past the last hidden channel are read unmasked and must meet finite f16 data: the W1 hi image overruns into the lo
        // image, the lo image into W3 (irc_lds_map); the raw bank has its own landing zone (sized by irc_lds_map, checked in launch_irc)
        const int cin = 2 + cs + cp, cinp = (cin + 1) & ~1, hidp = (hid + 1) & ~1, HP = (hid + 15) & ~15;
        if ((HP - hid) * cinp > hid * cinp || (HP - hid) * cinp > 2 * c_out * hidp + 32) return 1;
        if (ld * 4 < 16 || (ld & 3) != 0 || ((size_t)bank & 15) != 0) return 1;      // the bank DMA moves 16-byte pieces of 16-byte aligned rows (base included)
        if (y && ((size_t)y & 15) != 0) return 1;                                     // the epilogue stores 16-byte quads
    }
    // region height: 16 rows (a workgroup of 4 fat waves per 16 x 16 region: the per-workgroup work -- splitting the patch's bank,
    // tile shuffles, index arithmetic -- is paid once per 256 pixels) whenever the patch allows, else 8
    const bool tall = a.ph % 16 == 0;
#ifndef HS_IRC_NW
#define HS_IRC_NW 4          // waves per 16 x 16 region (dev A/B knob: tools/build_variants.py irc_nw8)
#endif
#define HS_IRC_CASE(SPL, PPL) \
    if (cs <= 4 * SPL && cp <= 4 * PPL) { \
        if (tall) return c_out <= 16 ? launch_irc<SPL, PPL, 1, 16, HS_IRC_NW>(a, stream) : launch_irc<SPL, PPL, 2, 16, HS_IRC_NW>(a, stream); \
        return c_out <= 16 ? launch_irc<SPL, PPL, 1, 8>(a, stream) : launch_irc<SPL, PPL, 2, 8>(a, stream); \
    }
    HS_IRC_CASE(1, 2)      // <= 4 skip + <= 8 previous-level channels
    HS_IRC_CASE(1, 4)      // CamVid-S level 4 (4 + 16)
    HS_IRC_CASE(2, 4)      // level-3 shapes on wide patches (6 + 16)
    HS_IRC_CASE(4, 2)      // HyperSeg-S level 4 (16 + 8)
    HS_IRC_CASE(4, 4)      // HyperSeg-M level 4 (16 + 16) and everything else up to 16 + 16
#undef HS_IRC_CASE
    return 1;
}

}  // namespace hs

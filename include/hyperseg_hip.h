/*
 * hyperseg_hip.h -- C ABI of libhyperseg_hip.so: the MI355X (gfx950) kernels of the HyperSeg
 * decoder hot path (dynamic patch-wise convolution + inter-stage glue).
 *
 * The reference (YuvalNirkin/hyperseg) has no FFI layer: its boundary for this path is the
 * nn.Module API of hyperseg/models/layers/{meta_sequential,meta_conv,meta_patch}.py and the
 * HyperPatch* classes of hyperseg/models/hyperseg_v*.py.  The host-side mirror of that API
 * lives in hyperseg_amd/models/ and binds exactly these entry points with ctypes
 * (hyperseg_amd/_hip.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - All tensors are fp32, NCHW, contiguous, device pointers BORROWED for the duration of the
 *    call; the library allocates nothing, keeps no global mutable state, never synchronises the
 *    device, and enqueues on the caller's stream (hipStream_t passed as void*).  Re-entrant.
 *  - Every function returns 0 on success, a negative hs_status on rejected arguments (nothing
 *    enqueued), or a positive hipError_t if the launch itself failed.
 *  - "grid" = the (fh, fw) weight grid; a level of resolution (H, W) has patches of
 *    ph = H/fh by pw = W/fw pixels; patch (b, i, j) has linear index p = (b*fh + i)*fw + j.
 *  - A *bank* is the per-patch filter bank in PATCH-MAJOR layout: bank[p*ld + m], m < rows.
 *    Row order is chosen by the producer (hs_signal2weights_fwd / hs_bank_pack_fwd) through a
 *    row map; the consumers document the order they expect.
 *  - A *stage input* is the channel concatenation the reference builds with torch.cat
 *    (hyperseg_v1_0.py:231-240): [2 coord channels | skip feature | previous level output,
 *    optionally bilinearly resized]; it is generated on the fly, never materialised.
 */
#ifndef HYPERSEG_HIP_H
#define HYPERSEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HS_ABI_VERSION 1

typedef enum {
    HS_OK = 0,
    HS_ERR_BAD_ARG = -1,       /* null pointer / non-positive size / inconsistent shapes */
    HS_ERR_NOT_DIVISIBLE = -2, /* H % fh != 0 or W % fw != 0 (hyperseg_v1_0.py:490-494 view error) */
    HS_ERR_UNSUPPORTED = -3,   /* shape outside what the kernels were instantiated for */
    HS_ERR_LDS = -4            /* tile does not fit the 160 KiB LDS */
} hs_status;

/* Epilogue activation.  The reference's decoder uses ReLU / ReLU6 only (hyperseg_v1_0.py:115, 283, 729); SWISH (x * sigmoid(x),
 * efficientnet_utils.py:58-79) is what the encoder-side entry points take, and the patch-convolution epilogues accept it too. */
typedef enum { HS_ACT_NONE = 0, HS_ACT_RELU = 1, HS_ACT_RELU6 = 2, HS_ACT_SWISH = 3 } hs_act;
typedef enum { HS_PAD_ZEROS = 0, HS_PAD_REFLECT = 1, HS_PAD_REPLICATE = 2, HS_PAD_CIRCULAR = 3 } hs_pad_mode;
typedef enum { HS_PREV_NONE = 0, HS_PREV_SAME = 1, HS_PREV_BILINEAR = 2 } hs_prev_mode;

/* Stage input: replaces F.interpolate + 2x torch.cat + get_image_coordinates
 * (hyperseg_v1_0.py:203-240; v0_1: hyperseg_v0_1.py:186-198, 240-246). */
typedef struct {
    const float* skip;   /* (B, c_skip, H, W) */
    const float* prev;   /* (B, c_prev, Hp, Wp) or NULL */
    int32_t batch, H, W;
    int32_t c_skip, c_prev;
    int32_t Hp, Wp;      /* resolution of prev; bilinear (align_corners=False) to (H, W) if prev_mode == 2 */
    int32_t coords;      /* 1: prepend x in [-1,1] over W (ch 0) and y over H (ch 1), linspace endpoints inclusive */
    int32_t prev_mode;   /* hs_prev_mode */
} hs_stage_input;

/* Per-channel affine applied after a convolution: y = act(conv * scale + shift).  Inference
 * BatchNorm folded by hs_bn_fold_fwd; scale == NULL means identity. */
typedef struct {
    const float* scale;
    const float* shift;
    int32_t act;         /* hs_act */
} hs_epilogue;

/* ABI version / build info (sanity check for the ctypes binding). */
int hs_version(void);
const char* hs_build_info(void);

/* a9: "hypernetwork head emits per-patch weights" -- grouped 1x1 conv, bias-free.
 * Replaces signal2weights(...)[:, :hp] (hyperseg_v1_0.py:479-484, 321-326;
 * hyperseg_v1_0_unify.py:302-309) AND the permute/reshape copy that follows it.
 *   bank[p*ld + n] = sum_k wsw_t[k*wc + n] * signal[b, signal_index + g(n)*cs_g + k, i, j],
 *   n < rows <= wc,  g(n) = n / (wc / groups).  Runs on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
 * wsw_t is the Conv2d weight (wc, cs_g, 1, 1) TRANSPOSED to (cs_g, wc) (done once by the host). */
typedef struct hs_s2w_layer hs_s2w_layer;
int hs_signal2weights_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                          int32_t signal_index, int32_t signal_channels, int32_t groups,
                          const float* wsw_t, int32_t wc, int32_t rows,
                          float* bank, int64_t ld, void* stream);

/* The same for up to 8 signal2weights layers (all levels of a decoder) in ONE launch: every layer reads its own
 * slice of the same signal and writes its own bank. */
struct hs_s2w_layer {
    int32_t signal_index, signal_channels, groups;
    const float* wsw_t;        /* (signal_channels/groups, wc) */
    int32_t wc;
    int32_t rows;
    float* bank;
    int64_t ld;
    const float* wsw_blk;      /* optional: the same weight packed by hs_s2w_pack_fwd (hs_s2w_pack_floats floats).  When EVERY
                                * layer brings one and fh * fw is a multiple of 4, the launch takes the blocked form: operands
                                * staged through LDS by DMA, 64-row x 64-patch blocks; results are bit-identical. */
};
/* Packs the transposed conv weight of one signal2weights layer into the blocked kernel's operand images (once per parameter
 * version; wsw_t as above).  hs_s2w_pack_floats = the size of `out` in floats. */
int64_t hs_s2w_pack_floats(int32_t signal_channels, int32_t groups, int32_t wc);
int hs_s2w_pack_fwd(const float* wsw_t, int32_t signal_channels, int32_t groups, int32_t wc, float* out, void* stream);
int hs_signal2weights_multi_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                                const hs_s2w_layer* layers, int32_t n_layers, void* stream);

/* Re-layout of a reference-layout weight tensor (B, hp_total, fh, fw) (channel-major, as
 * MetaPatch.forward / HyperPatchInvertedResidual receive it: meta_patch.py:49,
 * hyperseg_v1_0_unify.py:335-349) into a patch-major bank:
 *   bank[p*ld + m] = w[b, ch_offset + m, i, j]. */
int hs_bank_pack_fwd(const float* w, int32_t batch, int32_t hp_total, int32_t fh, int32_t fw,
                     int32_t ch_offset, int32_t rows, float* bank, int64_t ld, void* stream);

/* Inference BatchNorm folding: scale = gamma / sqrt(var + eps), shift = beta - mean * scale,
 * for n channels (nn.BatchNorm2d eval semantics; eps 1e-5 in every reference module). */
int hs_bn_fold_fwd(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                   int32_t n, float* scale, float* shift, void* stream);

/* Op A / Op B: dynamic patch-wise k x k convolution with image-level padding, stride 1,
 * dilation 1.  Replaces MetaPatch.forward + MetaConv2d.forward (meta_patch.py:35-57,
 * meta_conv.py:163-186), HyperPatchNoPadding.forward (hyperseg_v1_0.py:486-498) and
 * HyperPatch.forward (hyperseg_v1_0.py:543-557), plus the BatchNorm/activation modules that
 * follow them in make_*_patch_conv2d_block.  With fh = fw = 1 it is MetaConv2d itself.
 * Bank row order (natural): m = ((o*cin_g + c)*k + ky)*k + kx, cin_g = cin/groups.
 *   y[b,o,y,x] = act(scale[o] * sum W[m] * in_pad[b, grp(o)*cin_g + c, y+ky-pad, x+kx-pad] + shift[o]) */
int hs_patch_conv_fwd(const hs_stage_input* in, int32_t fh, int32_t fw,
                      const float* bank, int64_t ld,
                      int32_t c_out, int32_t k, int32_t pad, int32_t pad_mode, int32_t groups,
                      const hs_epilogue* ep, float* y, void* stream);

/* hs_patch_conv_fwd with k = 1 (no padding) AND hs_signal2weights_multi_fwd for `layers` as ONE heterogeneous launch: the
 * first batch*fh*fw workgroups are the patch convolution's (one per patch), the rest are signal2weights blocks writing the
 * banks of LATER decoder levels (they depend on the signal only).  The k = 1 levels (hyperseg_v1_0.py:486-498) are
 * latency-bound launches that leave most of the chip idle; the bank producer (hyperseg_v1_0.py:473-484) fills that time
 * instead of preceding level 0 as its own launch.  Results are bit-identical to the two separate calls.
 * Returns HS_ERR_UNSUPPORTED -- nothing launched -- when either half would not take the form this launch is built from
 * (large patches, a layer without wsw_blk, unaligned operands, >= 1024 two-pixel patches): issue the two calls instead.
 * signal (batch, c_signal, sfh, sfw); every layer's bank must be distinct from `bank` and `y`. */
int hs_patch_conv_s2w_fwd(const hs_stage_input* in, int32_t fh, int32_t fw, const float* bank, int64_t ld,
                          int32_t c_out, int32_t groups, const hs_epilogue* ep, float* y,
                          const float* signal, int32_t batch, int32_t c_signal, int32_t sfh, int32_t sfw,
                          const hs_s2w_layer* layers, int32_t n_layers, void* stream);

/* The decoder's three coarse k = 1 levels (patches of 1 x 1, 2 x 2 and 4 x 4 pixels: MultiScaleDecoder.forward's first three
 * iterations, hyperseg_v1_0.py:221-253, each `cat(coords, skip, bilinear2x(prev))` -> HyperPatchNoPadding (:486-498) -> BatchNorm ->
 * ReLU, :728-760) as ONE launch: one workgroup per grid cell walks level 0 -> 1 -> 2, every bank / skip pixel / BatchNorm row of
 * all three levels requested at the top of the kernel, and a level's outputs handed to the NEIGHBOURING cells' workgroups inside
 * the launch (the 2x bilinear upsample reads a one-pixel ring around a cell) as 8-byte {value, generation} granules -- one
 * agent-scope store each, polled by the consumer until the generation matches.  Same values as three hs_patch_conv_fwd calls
 * (same operations per output; the dot products' summation order differs at rounding level).
 *   levels[l]: skip (batch, c_skip, fh << l, fw << l); bank of the level, patch-major, rows o * c_in + c with
 *              c_in = 2 + c_skip + c_out of the previous level (0 for l = 0), row stride ld (multiple of 4, 16-byte aligned);
 *              scale / shift / act: the epilogue (scale null: none).   y (batch, levels[2].c_out, 4 fh, 4 fw).
 *   workspace: hs_k1_chain_workspace() bytes, 16-byte aligned, ZERO-FILLED ONCE by the caller and then left alone: it carries the
 *              generation counter between calls (kernel arguments are frozen under graph replay, so the state lives in memory the
 *              kernel owns).  One workspace per (batch, fh, fw) and per stream: two launches must not share it concurrently.
 *              Its first 32-bit word is an error flag: non-zero after a launch whose workgroups gave up waiting for a neighbour
 *              (bounded spins: ~0.2 s) -- the results of that launch are invalid.
 * HS_ERR_UNSUPPORTED -- nothing launched -- unless n_levels == 3, every bank fits its LDS-DMA budget (24 / 12 / 4 KB),
 * c_out <= 64, c_skip * pixels <= 256 per level, AND the whole grid (batch * fh * fw workgroups) is resident at once on the
 * current device (workgroups wait for their neighbours): the caller then issues the three hs_patch_conv_fwd calls. */
typedef struct hs_k1_level {
    const float* skip; int32_t c_skip;
    const float* bank; int64_t ld;
    int32_t c_out;
    const float* scale; const float* shift; int32_t act;
} hs_k1_level;
int64_t hs_k1_chain_workspace(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels);   /* bytes, or a negative hs_status */
int hs_k1_chain_fwd(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels, void* workspace,
                    float* y, void* stream);

/* The same launch carrying the decoder's FIRST INVERTED-RESIDUAL level as well (HyperPatchInvertedResidual on 8 x 8-pixel patches,
 * hyperseg_v1_0.py:281-376; Op C of hs_patch_ir_fwd): behind level 2 every workgroup hands its 4 x 4 outputs to the neighbouring
 * cells, assembles the reflect halo tile of ITS patch (coords, skip, bilinear 2x of level 2) and runs pw1 -> BN1 -> ReLU6 ->
 * depthwise 3 x 3 -> BN2 -> ReLU6 -> pw3 -> BN3 with the hidden activations in LDS -- exact f32 (v_mfma_f32_16x16x4_f32 for the two
 * 1 x 1 layers, v_fma_f32 for the depthwise one).  ir == NULL: hs_k1_chain_fwd.
 *   ir: skip (batch, c_skip, 8 fh, 8 fw); bank rows [pw1: hidden x c_in | depthwise: hidden x 9 | pw3: c_out x hidden] with
 *       c_in = 2 + c_skip + levels[2].c_out, row stride ld; s1 / b1 ... s3 / b3: the three folded BatchNorms (a null scale: none).
 *       No residual connection (the reference adds one only when c_in == c_out, which no decoder level has).
 *   y (batch, ir->c_out, 8 fh, 8 fw).  workspace: hs_decoder_chain_workspace() bytes, otherwise as for hs_k1_chain_fwd.
 * HS_ERR_UNSUPPORTED in addition when hidden > 64 or not a multiple of 4, c_out > 64, c_skip x 100 > 768, levels[2].c_out > 21,
 * the bank is larger than 12 KB, or the workgroup's LDS (~63 KB at HyperSeg-M) would pass 64 KB. */
typedef struct hs_chain_ir_level {
    const float* skip; int32_t c_skip;
    const float* bank; int64_t ld;
    int32_t hidden, c_out;
    const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3;
} hs_chain_ir_level;
int64_t hs_decoder_chain_workspace(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels,
                                   const hs_chain_ir_level* ir);
int hs_decoder_chain_fwd(int32_t batch, int32_t fh, int32_t fw, const hs_k1_level* levels, int32_t n_levels,
                         const hs_chain_ir_level* ir, void* workspace, float* y, void* stream);

/* MetaConv2d.forward with the reference's FULL argument set (meta_conv.py:141-186): per-sample weights w (B, rows >=
 * c_out * c_in/groups * kh * kw, row stride ldw; natural order ((o*cin_g + c)*kh + ky)*kw + kx), non-square kernels, stride,
 * dilation, any padding amounts per side and mode (F.pad semantics for reflect / replicate / circular, zero padding
 * otherwise), groups; optional BatchNorm affine + activation.  y (B, c_out, Ho, Wo),
 * Ho = (H + pad_top + pad_bottom - dil_h (kh - 1) - 1) / stride_h + 1.  The four pad amounts are explicit because the
 * reference pads asymmetrically in its non-zero modes (it hands (ph, pw, ph, pw) to F.pad, i.e. left = top = ph, right =
 * bottom = pw: meta_conv.py:159, 175-176); the host mirror passes exactly that.  The "same"-padded, stride-1 convolutions
 * every reference configuration uses run faster through hs_patch_conv_fwd (fh = fw = 1); this one covers the rest. */
int hs_meta_conv_fwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W, const float* w, int64_t ldw,
                     int32_t c_out, int32_t kh, int32_t kw, int32_t stride_h, int32_t stride_w, int32_t pad_top,
                     int32_t pad_bottom, int32_t pad_left, int32_t pad_right, int32_t dil_h, int32_t dil_w,
                     int32_t pad_mode, int32_t groups, const hs_epilogue* ep, float* y, void* stream);

/* Backward of hs_meta_conv_fwd for ZERO padding (MetaConv2d is differentiable in the reference through ATen's conv2d,
 * meta_conv.py:163-186; the other padding modes pad explicitly first, and that pad's adjoint is the caller's): dx (B, c_in, H, W)
 * and / or dw (B, rows, row stride lddw), either may be null.  Gather forms, deterministic. */
int hs_meta_conv_bwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W, const float* w, int64_t ldw,
                     int32_t c_out, int32_t kh, int32_t kw, int32_t stride_h, int32_t stride_w, int32_t pad_top,
                     int32_t pad_bottom, int32_t pad_left, int32_t pad_right, int32_t dil_h, int32_t dil_w, int32_t groups,
                     const float* dy, float* dx, float* dw, int64_t lddw, void* stream);

/* Op A with the bank generated inside the consumer: signal2weights (grouped 1x1 conv of the signal, hs_s2w_layer minus its
 * bank / ld fields, which are ignored) + k = 1 dynamic patch convolution + BatchNorm affine + activation in ONE launch --
 * HyperPatchNoPadding.forward and the norm / activation modules behind it (hyperseg_v1_0.py:473-498, 728-760; unify:
 * WeightLayer hyperseg_v1_0_unify.py:287-309 + 483-494).  The filter bank is never written to HBM.  Conv groups == 1,
 * patches of at most 64 pixels (the coarse levels, where the bank is the level's whole traffic); otherwise
 * HS_ERR_UNSUPPORTED and the caller uses hs_signal2weights_*_fwd + hs_patch_conv_fwd. */
int hs_patch_conv_gen_fwd(const hs_stage_input* in, int32_t fh, int32_t fw, const float* signal, int32_t c_signal,
                          const hs_s2w_layer* layer, int32_t c_out, const hs_epilogue* ep, float* y, void* stream);

/* Op C: fused per-patch inverted residual of hyperseg_v1_0.py:328-376 (and its unify twin
 * hyperseg_v1_0_unify.py:330-389): reflect halo tile -> pw1 -> BN1 -> ReLU6 -> dw3x3 ->
 * BN2 -> ReLU6 -> pw3 -> BN3 (+ the stage input itself if residual != 0, which requires
 * cin == c_out: use_res_connect, hyperseg_v1_0.py:295, 372-376), one launch, hidden activations
 * never leave the CU.  Bank rows in the reference's flat order (hyperseg_v1_0.py:302-309):
 *   [0, cin*hid)               W1[h][c]
 *   [cin*hid, +9*hid)          K[h][ky][kx]
 *   [.., +hid*c_out)           W3[o][h]
 * bn1/bn2/bn3 are folded scale/shift pairs (activation fields ignored). */
int hs_patch_ir_fwd(const hs_stage_input* in, int32_t fh, int32_t fw,
                    const float* bank, int64_t ld, int32_t hidden, int32_t c_out,
                    const hs_epilogue* bn1, const hs_epilogue* bn2, const hs_epilogue* bn3,
                    int32_t residual, int32_t math /* hs_ir_math, below */, float* y, void* stream);

/* Op D: the inverted residual of hyperseg_v0_1.py:205-237 (HyperSeg-L) as ONE launch.  The reference runs it as three
 * IMAGE-level patch convolutions -- pw1 (MetaPatchConv2d k=1) + BN + ReLU6, depthwise 3x3 with reflect padding of the
 * hidden activation (MetaPatchConv2d k=3, groups=hidden) + BN + ReLU6, pw3 (k=1) + BN -- so the depthwise taps that
 * cross a patch border read hidden activations produced with the NEIGHBOURING patch's pw1 weights (unlike Op C, which
 * applies a patch's own weights to its whole halo tile).  The kernel recomputes that one-pixel ring with the
 * neighbours' weights instead of exchanging it through HBM.  Bank rows as for hs_patch_ir_fwd (the nested
 * MetaSequential's cumulative ranges, hyperseg_v0_1.py:226-237: pw1 | depthwise | pw3).  Only the decoder's fused form
 * (coords + skip + 2x-bilinear previous level) at the instantiated shapes; otherwise HS_ERR_UNSUPPORTED, and the caller
 * runs the block as three hs_patch_conv_fwd launches. */
int hs_patch_ir_v0_fwd(const hs_stage_input* in, int32_t fh, int32_t fw,
                       const float* bank, int64_t ld, int32_t hidden, int32_t c_out,
                       const hs_epilogue* bn1, const hs_epilogue* bn2, const hs_epilogue* bn3,
                       int32_t math /* hs_ir_math */, float* y, void* stream);
/* The same operator with a caller-owned workspace: where patches are 4 x 4 or 8 x 8 pixels (HyperSeg-L levels 2 and 3: every 16 x 16
 * region would span 4-16 patches plus a ring of 12-20 more owners) it runs exactly as the reference states it -- pw1 + BN + ReLU6
 * written once to the workspace (channels-last, hidden rounded up to 16), then depthwise + BN + ReLU6 + pw3 + BN -- two launches of
 * per-patch matrix-core GEMMs whose 16-pixel N dimension is the patch (hs_patch_ir_d2.hip).  hs_patch_ir_v0_workspace: the bytes that
 * form needs for this shape, 0 if the shape is not covered (then, or with workspace = NULL, this call equals hs_patch_ir_v0_fwd).
 * The workspace is scratch: nothing is kept in it between calls, and the library holds no state of its own. */
int64_t hs_patch_ir_v0_workspace(const hs_stage_input* in, int32_t fh, int32_t fw, int32_t hidden, int32_t c_out);
int hs_patch_ir_v0_ws_fwd(const hs_stage_input* in, int32_t fh, int32_t fw,
                          const float* bank, int64_t ld, int32_t hidden, int32_t c_out,
                          const hs_epilogue* bn1, const hs_epilogue* bn2, const hs_epilogue* bn3,
                          int32_t math /* hs_ir_math */, float* workspace, int64_t workspace_bytes, float* y, void* stream);

/* Arithmetic of the fused inverted-residual levels -- the `math` ARGUMENT of hs_patch_ir_fwd / hs_patch_ir_v0_fwd (chosen by
 * the caller per launch: the library keeps no mode, or any other mutable state, of its own):
 *   HS_IR_MATH_F32    v_mfma_f32_16x16x4_f32: bit-exact f32 fma chains (csrc/hs_patch_ir_fused.hip, hs_patch_ir_px.hip).
 *   HS_IR_MATH_SPLIT  f16 matrix cores on split operands wherever that form exists (Op C on patches >= 8 rows x 16 columns,
 *       <= 16 skip and <= 16 previous-level channels: csrc/hs_patch_irc.hip): every f32 operand is scaled by a power of two
 *       and split into two f16 pieces, a product is ah*bh + al*bh + ah*bl accumulated in f32.  f32-class: the measured
 *       error of a dot product is BELOW an f32 fmaf chain's (tools/ubench/f16_probe.hip: 1.3e-7 vs 2.2e-7 of sum|a||b|).
 *       The scales are taken from the data -- the exact maximum of each weight MATRIX of the patch (pw1, pw3; even channel
 *       counts, round 4: the f16 pieces hold 2^15 of range below that maximum, so a row keeps f32-class accuracy as long as
 *       its own maximum is within 2^15 of the matrix's; odd channel counts keep one scale per row) and of every halo
 *       position's input column -- so any overall magnitude is carried; what is not is a dynamic range beyond ~2^18 INSIDE
 *       one reduction (an input 2^20 above its position's other channels meeting a weight 2^-20 of its matrix's maximum).
 *   HS_IR_MATH_AUTO   SPLIT where it is the faster form, F32 elsewhere: SPLIT wherever it applies, except narrow blocks (<= 4 skip
 *       channels) in launches of more than 512 16 x 16 regions, where the exact-f32 kernel measured faster (CamVid HyperSeg-L levels 4-5).
 * The reference runs these layers as fp32 torch convolutions (which cuDNN may run in TF32 there); all modes are held to the
 * same parity tolerance (tests/test_hip_parity.py).  The nn.Module mirror defaults to F32; serving / bench.py opt into AUTO
 * (hyperseg_amd.utils.inference.prepare_for_inference(ir_math='auto')). */
typedef enum { HS_IR_MATH_AUTO = 0, HS_IR_MATH_F32 = 1, HS_IR_MATH_SPLIT = 2 } hs_ir_math;

/* Which kernel hs_patch_ir_fwd would run for a call of this shape (host only: nothing is launched, pointers in `in` may be
 * null): the generic kernel (vector ALU, any channel counts), the exact-f32 matrix-core kernel (the decoder's own shapes) or
 * the f16-split matrix-core kernel (any channel split up to 16 + 16 -> 32 on patches >= 8 x 16).  Negative = error code.
 * For callers that want to know what a configuration gets before they run it, and for the tests that pin the coverage. */
typedef enum { HS_IR_ROUTE_GENERIC = 0, HS_IR_ROUTE_F32_MFMA = 1, HS_IR_ROUTE_SPLIT_MFMA = 2 } hs_ir_route;
int hs_patch_ir_route(const hs_stage_input* in, int32_t fh, int32_t fw, int32_t hidden, int32_t c_out,
                      int32_t residual, int32_t math);

/* Introspection for the tests (host only, no GPU): the matrix-core tile map of the fused inverted-residual kernel for a
 * region edge `reg` (8|16), mode (0 = Op C, 1 = Op D) and patch edge inside the region `pwr`.  out[(t*16+n)*3 + {0,1,2}]
 * = halo coordinates (u, v) and liveness of column n of pw1 tile t (csrc/hs_ir_tiles.h). */
int hs_ir_tile_map(int32_t reg, int32_t mode, int32_t pwr, int32_t* n_tiles, int32_t* n_pixel_tiles,
                   int32_t* out, int32_t capacity);

/* Final logits resize: F.interpolate(p, size, mode='bilinear', align_corners=False)
 * (hyperseg_v1_0.py:250-251).  x (B,C,Hi,Wi) -> y (B,C,Ho,Wo). */
int hs_upsample_bilinear_fwd(const float* x, int32_t batch, int32_t channels, int32_t Hi, int32_t Wi,
                             int32_t Ho, int32_t Wo, float* y, void* stream);
/* ... on bf16 storage (x and y bf16, same taps, f32 arithmetic, one rounding on store): the training path's final logits under
 * torch.autocast(bfloat16) (autograd.UpsampleBilinear). */
int hs_upsample_bilinear_bf16_fwd(const void* x, int32_t batch, int32_t channels, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                  void* y, void* stream);

/* The same resize with the class argmax taken in registers: mask (B, Ho, Wo) uint8 = argmax_c of the upsampled logits,
 * bit-identical to argmax over hs_upsample_bilinear_fwd's output (shared arithmetic), ties -> lowest class; channels
 * <= 256.  Replaces F.interpolate + pred.argmax(1) (hyperseg_v1_0.py:250-251 + test.py:171, test_fps.py:194) for callers
 * that only need masks: 0.5 MB written instead of 39.8 MB at HyperSeg-M 1024x512. */
int hs_upsample_argmax_fwd(const float* x, int32_t batch, int32_t channels, int32_t Hi, int32_t Wi,
                           int32_t Ho, int32_t Wo, uint8_t* mask, void* stream);

/* Backward of hs_patch_conv_fwd (plain input x, no fused prologue / epilogue), fp32 -- SURVEY.md Appendix E.
 * The reference has no backward of its own (autograd over F.pad/unfold/grouped conv2d/fold: meta_patch.py:35-57);
 * these are the adjoints the training path (BASELINE config 5) needs:
 *   hs_patch_conv_bwd_weight: dbank[p, n] = sum over the patch's pixels of dy * padded x  ("per-patch weight-grad kernel")
 *   hs_patch_conv_bwd_input : dx = transposed patch-wise correlation of dy with each output pixel's own patch filters,
 *                             halo gradients folded back through the padding (adjoint of F.pad). */
int hs_patch_conv_bwd_input(const float* dy, const float* bank, int64_t ld, int32_t batch, int32_t c_in,
                            int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad,
                            int32_t pad_mode, int32_t groups, float* dx, void* stream);
int hs_patch_conv_bwd_weight(const float* x, const float* dy, int32_t batch, int32_t c_in, int32_t H, int32_t W,
                             int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad, int32_t pad_mode,
                             int32_t groups, float* dbank, int64_t ld, void* stream);

/* Training-path twins with a storage type: HS_DTYPE_F32, or HS_DTYPE_BF16 = bf16 storage of the ACTIVATIONS and their gradients
 * (x, y, dy, dx read and written as bf16: half the HBM bytes) with fp32 accumulation (BASELINE config 5; the reference has no
 * reduced-precision path: SURVEY 8d).  `bank` and `dbank` are fp32 (const float* / float*) for either dtype: the bank comes out of
 * signal2weights and its gradient goes into that layer's adjoint, both fp32.  Plain (B, C, H, W) tensors, no fused prologue /
 * epilogue -- the training route composes the stage input, BatchNorm and activations with its own differentiable ops
 * (hyperseg_amd/autograd.py).  Same math as hs_patch_conv_fwd / hs_patch_conv_bwd_input / hs_patch_conv_bwd_weight above. */
typedef enum { HS_DTYPE_F32 = 0, HS_DTYPE_BF16 = 1 } hs_dtype;
int hs_patch_conv_plain_fwd(int32_t dtype, const void* x, const void* bank, int64_t ld, int32_t batch, int32_t c_in,
                            int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad,
                            int32_t pad_mode, int32_t groups, void* y, void* stream);
int hs_patch_conv_plain_bwd_in(int32_t dtype, const void* dy, const void* bank, int64_t ld, int32_t batch, int32_t c_in,
                               int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad,
                               int32_t pad_mode, int32_t groups, void* dx, void* stream);
int hs_patch_conv_plain_bwd_w(int32_t dtype, const void* x, const void* dy, int32_t batch, int32_t c_in, int32_t H, int32_t W,
                              int32_t fh, int32_t fw, int32_t c_out, int32_t k, int32_t pad, int32_t pad_mode,
                              int32_t groups, void* dbank, int64_t ld, void* stream);

/* Encoder-side helper ("next" row of SURVEY.md section 8f; opt-in via hyperseg_amd.utils.inference): depthwise k x k
 * convolution (k in {3,5}, stride in {1,2}) with arbitrary top/left zero padding (TF-"SAME"), + per-channel affine
 * (folded BatchNorm) + activation (hs_act) in one launch.  x (B,C,H,W), w (C,1,k,k) -> y (B,C,Ho,Wo).
 * Replaces F.pad + F.conv2d(groups=C) + BatchNorm2d + swish of the reference's MBConvBlock
 * (hyperseg/models/backbones/efficientnet.py:59-66, 101-103). */
int hs_depthwise_conv_fwd(const float* x, int32_t batch, int32_t channels, int32_t H, int32_t W,
                          const float* w, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                          int32_t Ho, int32_t Wo, const float* scale, const float* shift, int32_t act,
                          float* y, float* pool_partial, const float* in_scale, const float* in_shift, void* stream);
/* in_scale / in_shift (optional, both or neither, (C)): the taps become swish(in_scale[c]*x + in_shift[c]) -- the folded
 * BatchNorm + swish of the 1x1 expand convolution that produced x (efficientnet.py:101-103), applied on load so that the
 * raw GEMM output is consumed directly; zero padding stays zero.
 * pool_partial (optional): (B*C, hs_depthwise_pool_blocks(Ho, Wo)) per-workgroup sums of the outputs, the squeeze-excite
 * pooling for free.  hs_se_gate_fwd turns them into the SE gate (pool -> 1x1 reduce + swish -> 1x1 expand -> sigmoid;
 * efficientnet.py:106-111): squeezed (B, c_squeezed) = swish(reduce(pool)), then gate (B, channels) -- one launch when the
 * reduce weights are small (channels <= 768, c_squeezed <= 32: every workgroup re-derives the squeezed vector), two otherwise; if
 * w_proj (c_out, channels) is given it also folds the gate -- and, with out_scale (c_out), the project convolution's folded
 * BatchNorm scale -- into the project weights: w_scaled[b, o, c] = w_proj[o, c] * gate[b, c] * out_scale[o].
 * w_reduce is (c_squeezed, channels); w_expand is passed TRANSPOSED, (c_squeezed, channels): both are read coalesced. */
int hs_depthwise_pool_blocks(int32_t Ho, int32_t Wo);

/* The squeeze-excite gate finished by the LAST workgroups of the pooling launch itself (round 5, csrc/hs_se_tail.h) instead of by
 * hs_se_gate_fwd: hs_depthwise_conv_se_fwd / hs_mbconv_expand_dw_se_fwd are the launches above without pool_partial and with this
 * descriptor; gate (B, channels) and, optionally, squeezed (B, c_squeezed) come back.  Same sums in the same order per channel; the
 * two matrix-vector products are split over <= 32 workgroups and re-assembled in slice order (deterministic).
 * workspace: hs_se_tail_workspace(batch, channels, c_squeezed, nblk, wgs_per_batch) bytes (nblk = hs_depthwise_pool_blocks /
 * hs_mbconv_tiles; wgs_per_batch = channels * nblk for the depthwise launch, hs_mbconv_se_workgroups for the fused one), 8-byte
 * aligned, ZERO before its first use and owned by one launch at a time (two streams need two); the launches keep it consistent.
 * Its word [batch] is an error flag: nonzero after a launch whose bounded waits gave up (the gate is then NaN).
 * hs_se_tail_workspace returns 0 -- and the launches HS_ERR_UNSUPPORTED -- for shapes the tail does not cover
 * (c_squeezed > 96, > 512 partials per channel, more than 32 tails of <= 64 channels, fewer workgroups than tails). */
typedef struct hs_se_tail {
    const float* w_reduce;    /* (c_squeezed, channels)            efficientnet.py:107 _se_reduce */
    const float* b_reduce;    /* (c_squeezed) */
    const float* w_expand_t;  /* (c_squeezed, channels): _se_expand's weight TRANSPOSED */
    const float* b_expand;    /* (channels) */
    int32_t c_squeezed;
    float* gate;              /* out (B, channels) = sigmoid(expand(swish(reduce(mean)))) */
    float* squeezed;          /* out (B, c_squeezed), optional */
    void* workspace;
} hs_se_tail;
int64_t hs_se_tail_workspace(int32_t batch, int32_t channels, int32_t c_squeezed, int32_t nblk, int64_t wgs_per_batch);
int hs_se_tail_tails(int32_t channels, int32_t c_squeezed, int32_t nblk, int64_t wgs_per_batch);   /* workgroups that finish the gate; 0: not covered */
int hs_depthwise_conv_se_fwd(const float* x, int32_t batch, int32_t channels, int32_t H, int32_t W,
                             const float* w, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                             int32_t Ho, int32_t Wo, const float* scale, const float* shift, int32_t act,
                             float* y, const float* in_scale, const float* in_shift, const hs_se_tail* se, void* stream);
int hs_se_gate_fwd(const float* partial, int32_t batch, int32_t channels, int32_t nblk, float inv_hw,
                   const float* w_reduce, const float* b_reduce, int32_t c_squeezed, const float* w_expand,
                   const float* b_expand, float* squeezed, float* gate, const float* w_proj, int32_t c_out,
                   const float* out_scale, float* w_scaled, void* stream);


/* Encoder-side helper: the stem -- dense 3x3 stride-2 convolution of the 3-channel image (zero padding by top/left
 * offsets, TF-"SAME") + folded BatchNorm + swish in one launch.  x (B,3,H,W), w (c_out,3,3,3) -> y (B,c_out,Ho,Wo).
 * Replaces F.pad + conv + BatchNorm2d + swish of efficientnet.py:321-322. */
int hs_stem_conv_fwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W, const float* w, int32_t c_out,
                     int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, const float* scale, const float* shift,
                     float* y, void* stream);

/* Encoder-side helper: the first half of an MBConv block in ONE launch,
 *   y = swish(BN1(depthwise_kxk_stride(zero-pad(swish(BN0(w_expand . x))))))   (+ per-tile sums of y for the SE pool)
 * x (B,c_in,H,W), w_expand (c_mid,c_in), w_dw (c_mid,1,k,k), scale/shift = folded BatchNorms, y (B,c_mid,Ho,Wo),
 * pool_partial (optional) (B*c_mid, hs_mbconv_tiles(k, stride, Ho, Wo)).  k in {3,5}, stride in {1,2}, c_in <= 80.
 * The expanded activation lives in LDS only.  Replaces efficientnet.py:101-106 of the reference's MBConvBlock. */
int hs_mbconv_tiles(int32_t k, int32_t stride, int32_t Ho, int32_t Wo);
/* The stem and the first block's depthwise half in one launch (that block has no expand conv):
 *   y = swish(BN1(depthwise_3x3(zero-pad(swish(BN0(conv3x3/s2(zero-pad(x))))))))   (+ per-tile sums of y for the SE pool)
 * x (B,3,H,W); w_stem28 (c_mid, 28): the stem weight (c_mid,3,3,3) flattened to (c_mid, 27) plus one zero column; (Hs, Ws) = the
 * stem's output size = y's (B,c_mid,Hs,Ws); pool_partial (optional) (B*c_mid, hs_mbconv_tiles(3, 1, Hs, Ws)).  The stem's output
 * map never reaches HBM.  HS_ERR_UNSUPPORTED outside k = 3, c_mid % 16 == 0, Ws % 16 == 0 (the caller then runs
 * hs_stem_conv_fwd + hs_depthwise_conv_fwd).  Replaces efficientnet.py:321-322 and 59-66 / 101-103 of MBConvBlock 0. */
int hs_stem_dw_fwd(const float* x, int32_t batch, int32_t H, int32_t W, const float* w_stem28, int32_t c_mid,
                   const float* scale0, const float* shift0, int32_t stem_pad_t, int32_t stem_pad_l, int32_t Hs, int32_t Ws,
                   const float* w_dw, int32_t k, int32_t pad_t, int32_t pad_l, const float* scale1, const float* shift1,
                   float* y, float* pool_partial, void* stream);
int hs_mbconv_expand_dw_fwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W,
                            const float* w_expand, int32_t c_mid, const float* scale0, const float* shift0,
                            const float* w_dw, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                            int32_t Ho, int32_t Wo, const float* scale1, const float* shift1, float* y,
                            float* pool_partial, void* stream);
int64_t hs_mbconv_se_workgroups(int32_t batch, int32_t c_mid, int32_t k, int32_t stride, int32_t Ho, int32_t Wo);   /* per batch element */
int hs_mbconv_expand_dw_se_fwd(const float* x, int32_t batch, int32_t c_in, int32_t H, int32_t W,
                               const float* w_expand, int32_t c_mid, const float* scale0, const float* shift0,
                               const float* w_dw, int32_t k, int32_t stride, int32_t pad_t, int32_t pad_l,
                               int32_t Ho, int32_t Wo, const float* scale1, const float* shift1, float* y,
                               const hs_se_tail* se, void* stream);

/* Encoder-side helper: 1x1 convolution as an fp32 MFMA GEMM with its surroundings fused,
 *   y[b,o,p] = act(scale[o] * sum_c w[o,c] * (gate[b,c] * x[b,c,p]) + shift[o]) + residual[b,o,p]
 * (gate, scale/shift, residual optional; act = hs_act).  x (B,Cin,P), w (Cout,Cin), y (B,Cout,P), P = H*W.
 * Replaces {SE multiply, 1x1 conv, BatchNorm2d, swish, skip add} of an MBConv block (efficientnet.py:101-103, 110-124). */
int hs_pointwise_conv_fwd(const float* x, int32_t batch, int32_t c_in, int32_t pixels, const float* w, int32_t c_out,
                          const float* gate, const float* scale, const float* shift, int32_t act, const float* residual,
                          float* y, void* stream);

/* y = act(scale[c]*x + shift[c]) + residual over (B, C, P), P % 4 == 0, y may alias x: folded BatchNorm + swish (+ skip
 * add) after a stock 1x1 convolution in one launch. */
int hs_affine_act_fwd(const float* x, int32_t batch, int32_t channels, int32_t pixels, const float* scale,
                      const float* shift, int32_t act, const float* residual, float* y, void* stream);

/* Encoder-side helper (opt-in: prepare_for_inference(split_gemm=True); off by default): a 1x1 convolution as a GEMM on the
 * f16 matrix cores with split operands, f32 storage and accumulation,
 *   y[b,o,p] = act(sum_c w[o,c] * gate[b,c] * x[b,c,p] + shift[o]) + residual[b,o,p]
 * x (B,Cin,P), y (B,Cout,P); gate (B,Cin), shift (Cout), residual (B,Cout,P) optional, residual may be y itself (in-place
 * accumulation onto a skip tensor); act = hs_act.
 * w_frag / w_inv: the STATIC weight split once on the host into two f16 pieces of every row scaled by a power of two to
 * < 2^15, in MFMA-fragment order [ceil(Cout/16)][kp/32][piece][64 lanes][8] (lane = row % 16 + 16 * kgroup holds
 * w[16 R + row % 16][32 S + 8 kgroup + j]), and the inverse row scales padded to a multiple of 16 rows
 * (hyperseg_amd.functional.gemm_split_weights builds both).  kp = hs_gemm_split_kp(Cin): Cin rounded up to the workgroup's
 * K split (HS_ERR_UNSUPPORTED for Cin > 2560; hs_gemm_split_fwd itself takes Cin <= 1280, the deeper K is the 2x2 form's).  Replaces the library GEMM of an MBConv block's expand / project convolution
 * (efficientnet.py:101, 115) and, through `gate`, the SE multiply (:110-111). */
int hs_gemm_split_kp(int32_t c_in);
int hs_gemm_split_fwd(const void* w_frag, const float* w_inv, const float* gate, const float* x, const float* shift,
                      int32_t act, const float* residual, float* y, int32_t batch, int32_t c_out, int32_t c_in, int32_t kp,
                      int32_t pixels, void* stream);
/* Conv2d(c_in, c_out, kernel 2, stride 2, no padding, no bias) + shift + act with the same arithmetic: x (B,c_in,2Ho,2Wo) ->
 * y (B,c_out,Ho,Wo), the window read on load (K = 4 c_in, k = 4 c + 2 dy + dx: the conv weight's own flatten order; no im2col
 * copy).  w_frag / w_inv from the (c_out, 4 c_in) weight, kp = hs_gemm_split_kp(4 c_in); 64 <= c_in <= 640, c_in and Wo even,
 * x 16-byte aligned, else HS_ERR_UNSUPPORTED.  pool_partial (optional): (B, c_out, ceil(Ho Wo / 16)) sums of y over blocks of 16
 * consecutive pixels -- the global average pool that follows, without a launch of its own (hs_pooled_shift_fwd reads it).
 * Replaces F.conv2d + BatchNorm + ReLU of the context head's down blocks (hyperseg_v1_0.py:396-401). */
int hs_gemm_split_conv2x2_fwd(const void* w_frag, const float* w_inv, const float* x, const float* shift, int32_t act, float* y,
                              float* pool_partial, int32_t batch, int32_t c_out, int32_t c_in, int32_t kp, int32_t Ho,
                              int32_t Wo, void* stream);
/* hs_gemm_split_fwd without gate / residual, its (B,c_out,Ho,Wo) result stored nearest-2x upsampled: y (B,c_out,2Ho,2Wo), 8-byte
 * aligned.  The context head's last merge writes straight into the right half of the signal (hyperseg_v1_0.py:409-410: neither
 * the upsampled tensor nor the concatenation is built). */
int hs_gemm_split_up2_fwd(const void* w_frag, const float* w_inv, const float* x, const float* shift, int32_t act, float* y,
                          int32_t batch, int32_t c_out, int32_t c_in, int32_t kp, int32_t Ho, int32_t Wo, void* stream);
/* shift_out[m] = shift[m] + sum_c wb[m][c] * (inv_pixels * sum_j partial[c][j]), partial (channels, nblk) as written by
 * hs_gemm_split_conv2x2_fwd for ONE frame, wb (rows, channels): where the context head concatenates a feature map with its own
 * global average (hyperseg_v1_0.py:404-409) the pooled half contributes a per-row constant to the following 1x1 convolution --
 * this is that constant folded into the BatchNorm shift (mean + mat-vec in one launch).  channels <= 8192. */
int hs_pooled_shift_fwd(const float* partial, int32_t nblk, float inv_pixels, const float* wb, const float* shift,
                        float* shift_out, int32_t rows, int32_t channels, void* stream);

/* Training-path re-layouts and loss reduction (hs_train_aux.hip; dtype = hs_dtype, plain contiguous tensors).
 * Halo tiles: a train-mode v1_0 inverted residual (hyperseg_v1_0.py:328-376) applies each patch's weights to the patch's own
 * reflect-padded (ph+2) x (pw+2) tile; the tiles are laid side by side as one image (B, C, fh (ph+2), fw (pw+2)) so that the three layers
 * are ordinary patch convolutions.  hs_halo_tiles_fwd builds that image from x (B, C, H, W) in one gather (stock ops: F.pad(reflect) ->
 * unfold -> unfold -> permute -> reshape), hs_halo_tiles_bwd is its adjoint (a gather too: per image pixel the <= 16 tile positions that
 * map onto it); hs_tile_interior_fwd drops the halos again, hs_tile_interior_bwd is its adjoint (zeros on the halos). */
int hs_halo_tiles_fwd(int32_t dtype, const void* x, int32_t batch, int32_t channels, int32_t H, int32_t W, int32_t fh, int32_t fw,
                      void* tiled, int32_t patch_major, void* stream);
int hs_halo_tiles_bwd(int32_t dtype, const void* dtiled, int32_t batch, int32_t channels, int32_t H, int32_t W, int32_t fh, int32_t fw,
                      void* dx, int32_t patch_major, void* stream);
/* patch_major (hs_halo_tiles_*, hs_dw_tiles_*): 0 = the tiles side by side as one image (B, C, fh (ph+2), fw (pw+2)); 1 = one tile after the
 * other, (B fh fw, C, ph+2, pw+2) -- every operand of a patch is then one contiguous run and the block's 1x1 layers are patch convolutions
 * with a (1, 1) grid over B fh fw "frames" (same kernels, same bank rows).  hs_tile_interior_* take the image form only. */
int hs_tile_interior_fwd(int32_t dtype, const void* tiled, int32_t batch, int32_t channels, int32_t H, int32_t W, int32_t fh,
                         int32_t fw, void* y, void* stream);
int hs_tile_interior_bwd(int32_t dtype, const void* dy, int32_t batch, int32_t channels, int32_t H, int32_t W, int32_t fh, int32_t fw,
                         void* dtiled, void* stream);
/* The middle layer of that block as the reference states it -- a VALID depthwise 3 x 3 of every halo tile with the patch's own taps
 * (hyperseg_v1_0.py:352-360: F.conv2d on the unfolded, padded patches with padding 0): tile image (B, C, fh (ph+2), fw (pw+2)) -> (B, C, H, W)
 * in one launch, its two adjoints in one launch each.  Replaces hs_patch_conv_plain_* (k = 3, zero padding, on the whole tile image) +
 * hs_tile_interior_*: no ring outputs, no intermediate.  bank / dbank: fp32 (P, ld) with the taps of channel c at columns [9 c, 9 c + 9)
 * (the depthwise range of the block's bank: pass bank + r1); dtype = hs_dtype of the activations.  Even patch widths only
 * (HS_ERR_UNSUPPORTED otherwise: the caller keeps the two-launch route).  autograd.DwTilesValid. */
int hs_dw_tiles_fwd(int32_t dtype, const void* tiled, const float* bank, int64_t ld, int32_t batch, int32_t channels, int32_t H, int32_t W,
                    int32_t fh, int32_t fw, void* y, int32_t patch_major, void* stream);
int hs_dw_tiles_bwd_in(int32_t dtype, const void* dy, const float* bank, int64_t ld, int32_t batch, int32_t channels, int32_t H, int32_t W,
                       int32_t fh, int32_t fw, void* dtiled, int32_t patch_major, void* stream);
int hs_dw_tiles_bwd_w(int32_t dtype, const void* tiled, const void* dy, int32_t batch, int32_t channels, int32_t H, int32_t W, int32_t fh,
                      int32_t fw, float* dbank, int64_t ld, int32_t patch_major, void* stream);
/* The per-image reduction of hyperseg/losses/bootstrapped_ce_loss.py:19-25 over n non-negative f32 losses, with no sort and no host
 * read: if more than k losses exceed thresh, their mean; otherwise the mean of the k largest (the k-th largest found by a three-level
 * radix histogram of the bit patterns; ties at it share the remaining weight).  out5 = {loss, branch, 1/count, t, tie weight}: the
 * state hs_bootstrap_mean_bwd turns into d loss / d values.  workspace: hs_bootstrap_mean_workspace() bytes, scratch.  n > k. */
int64_t hs_bootstrap_mean_workspace(void);
/* The batch form the loss module uses: the batched reduction + the mean of the per-image losses (bootstrapped_ce_loss.py:33, loss / batch)
 * in mean_out (one float), and the adjoint from the ONE upstream gradient of that mean. */
int hs_bootstrap_mean_of_batch_fwd(const float* values, int32_t images, int32_t n, int32_t k, float thresh, void* workspace,
                                   float* out8, float* mean_out, void* stream);
int hs_bootstrap_mean_of_batch_bwd(const float* values, int32_t images, int32_t n, const float* state8, const float* grad_mean,
                                   float* grad_values, void* stream);
/* Adam (torch.optim.Adam / AdamW arithmetic, no amsgrad) for up to 48 fp32 tensors in ONE launch of 1024-element workgroups: the
 * training loop's optimizer step (hyperseg/train.py:185-188) without torch's 65 536-element chunking.  The arrays are HOST arrays of device
 * pointers; `steps`: hs_adam_blocks(numel, n) floats on the device, zero before the first step, owned by this parameter list (one step
 * count per workgroup: graph replay freezes kernel arguments); lr_device (optional) overrides lr.  decoupled: AdamW's weight decay. */
int64_t hs_adam_blocks(const int64_t* numel, int32_t n);
int hs_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                 int32_t n, const float* lr_device, float lr, double beta1, double beta2, float eps, float weight_decay, int32_t decoupled,
                 int32_t maximize, float* steps, void* stream);
int hs_bootstrap_mean_fwd(const float* values, int32_t n, int32_t k, float thresh, void* workspace, float* out5, void* stream);
int hs_bootstrap_mean_bwd(const float* values, int32_t n, const float* state5, const float* grad_out, float* grad_values, void* stream);
/* The same for `images` images at once (one set of launches, grid.y = image): values (images, n), workspace images x
 * hs_bootstrap_mean_workspace() bytes, out8 / state8 (images, 8) floats whose first five are out5 / state5 above, grad_out (images). */
int hs_bootstrap_mean_batched_fwd(const float* values, int32_t images, int32_t n, int32_t k, float thresh, void* workspace, float* out8,
                                  void* stream);
int hs_bootstrap_mean_batched_bwd(const float* values, int32_t images, int32_t n, const float* state8, const float* grad_out,
                                  float* grad_values, void* stream);

/* Adjoint of hs_bank_pack_fwd: patch-major bank (B fh fw, ld) -> channel-major weights (B, hp_total, fh, fw): channels
 * [ch_offset, ch_offset + rows) from the bank's columns [0, rows), exact zeros elsewhere.  Training path (autograd.BankPack). */
int hs_bank_unpack_fwd(const float* bank, int64_t ld, int32_t batch, int32_t hp_total, int32_t fh, int32_t fw, int32_t ch_offset,
                       int32_t rows, float* w, void* stream);

/* Adjoint of hs_upsample_bilinear_fwd (F.interpolate(..., mode='bilinear', align_corners=False), Ho >= Hi, Wo >= Wi): dy (B,C,Ho,Wo) ->
 * dx (B,C,Hi,Wi), a gather over the outputs whose taps touch each input pixel.  dy_batch_stride (floats; 0 = packed C*Ho*Wo): dy may be a
 * channel range of a wider (B, C_total, Ho, Wo) tensor -- the previous level's slice of a stage input's gradient -- read in place.
 * Training path (autograd.UpsampleBilinear, StageMaterialize). */
int hs_upsample_bilinear_bwd(const float* dy, int64_t dy_batch_stride, int32_t batch, int32_t channels, int32_t Hi, int32_t Wi,
                             int32_t Ho, int32_t Wo, float* dx, void* stream);
/* ... with a storage type (hs_dtype; dy and dx alike, dy_batch_stride in ELEMENTS, f32 arithmetic, one rounding on store): the bf16
 * training step keeps the stage inputs and their gradients in bf16 (autograd.StageMaterialize under torch.autocast). */
int hs_upsample_bilinear_typed_bwd(int32_t dtype, const void* dy, int64_t dy_batch_stride, int32_t batch, int32_t channels, int32_t Hi,
                                   int32_t Wi, int32_t Ho, int32_t Wo, void* dx, void* stream);

/* BatchNorm2d in TRAINING mode (torch.nn.functional.batch_norm semantics: batch statistics, biased variance for the normalisation,
 * unbiased for the running estimate, running = (1 - momentum) running + momentum batch) fused with the activation that follows it
 * (act = HS_ACT_NONE | HS_ACT_RELU | HS_ACT_RELU6), x / y / dy / dx (B, C, pixels) in `dtype` storage, parameters and statistics f32.
 * Two launches per direction.  save_mean / save_invstd (C): written by fwd, read by bwd.  gamma / beta / running_* may be NULL.
 * workspace: hs_bn_train_workspace(C) bytes, scratch; num_batches_tracked (optional): the module's int64 step counter, incremented by one.
 * Replaces BatchNorm2d + ReLU6 of hyperseg_v1_0.py:349-376 in train mode. */
int64_t hs_bn_train_workspace(int32_t channels);
/* The statistics pass of hs_bn_act_train_fwd alone: {sum, sum of squares} around the channel's first element, per channel and slice,
 * into `workspace` (hs_bn_train_workspace(C) bytes) -- for consumers that normalise on load instead of reading a normalised copy. */
int hs_bn_train_stats_fwd(int32_t dtype, const void* x, int32_t batch, int32_t channels, int64_t pixels, void* workspace, void* stream);
/* The train-mode inverted residual's BatchNorm1 + ReLU6 + depthwise 3 x 3 (hyperseg_v1_0.py:346-360) without the normalised copy of the
 * halo tiles: hs_dw_tiles_fwd / _bwd_w reading the RAW output of the first 1 x 1 layer and applying act(gamma invstd (x - mean) + beta)
 * on load.  _fwd: `bn_partial` = hs_bn_train_stats_fwd's workspace for the tile tensor viewed as (B', C, pixels) (patch-major: B' =
 * batch fh fw, pixels = (ph + 2)(pw + 2)); writes save_mean / save_invstd (C floats each), updates running_mean / running_var
 * (optional) and num_batches_tracked (optional) exactly as hs_bn_act_train_fwd.  _bwd_w: the tap gradient from the raw tiles and the
 * saved statistics.  The BatchNorm's own adjoint stays hs_bn_act_train_bwd on (raw tiles, hs_dw_tiles_bwd_in's result). */
int hs_dw_tiles_bn_fwd(int32_t dtype, const void* tiled, const float* bn_partial, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, int32_t act, float* save_mean,
                       float* save_invstd, int64_t* num_batches_tracked, const float* bank, int64_t ld, int32_t batch,
                       int32_t channels, int32_t H, int32_t W, int32_t fh, int32_t fw, void* y, int32_t patch_major, void* stream);
int hs_dw_tiles_bn_bwd_w(int32_t dtype, const void* tiled, const void* dy, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, int32_t act, int32_t batch, int32_t channels, int32_t H, int32_t W, int32_t fh,
                         int32_t fw, float* dbank, int64_t ld, int32_t patch_major, void* stream);
/* The same move at the block's second cut (round 5): BatchNorm2 + ReLU6 normalised on load by the LAST 1 x 1 layer (hyperseg_v1_0.py:361-370),
 * x (B, c_in, H, W) RAW, bank rows = this layer's (c_out, c_in) weights per patch, k = 1.  _fwd: bn_partial = hs_bn_train_stats_fwd's
 * workspace for x; save_mean / save_invstd / running_* / num_batches_tracked as hs_bn_act_train_fwd.  _bwd_w: the weight gradient on the raw
 * input + saved statistics.  The input gradient is hs_patch_conv_plain_bwd_in unchanged (it yields the gradient of the normalised map),
 * BatchNorm's adjoint hs_bn_act_train_bwd on (x, that).  c_out <= 32, c_in <= 64, patches >= 64 pixels; else HS_ERR_UNSUPPORTED. */
int hs_patch_conv_bn_fwd(int32_t dtype, const void* x, const float* bn_partial, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, int32_t act, float* save_mean,
                         float* save_invstd, int64_t* num_batches_tracked, const float* bank, int64_t ld, int32_t batch,
                         int32_t c_in, int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t c_out, void* y, void* stream);
int hs_patch_conv_bn_bwd_w(int32_t dtype, const void* x, const void* dy, const float* gamma, const float* beta, const float* save_mean,
                           const float* save_invstd, int32_t act, int32_t batch, int32_t c_in, int32_t H, int32_t W, int32_t fh,
                           int32_t fw, int32_t c_out, float* dbank, int64_t ld, void* stream);
int hs_bn_act_train_fwd(int32_t dtype, const void* x, int32_t batch, int32_t channels, int64_t pixels, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float momentum, float eps, int32_t act,
                        float* save_mean, float* save_invstd, void* workspace, void* y, int64_t* num_batches_tracked, void* stream);
int hs_bn_act_train_bwd(int32_t dtype, const void* x, const void* dy, int32_t batch, int32_t channels, int64_t pixels,
                        const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, float eps, int32_t act,
                        void* workspace, void* dx, float* dgamma, float* dbeta, void* stream);
/* BatchNorm1's adjoint of a train-mode inverted residual WITHOUT its statistics launch (round 6; hyperseg_v1_0.py:352-360 under autograd): the
 * valid depthwise adjoint hs_dw_tiles_bwd_in, given the RAW tiles `tiled` and the statistics hs_dw_tiles_bn_fwd saved, also leaves the adjoint's two
 * per-channel sums over the tile tensor (sum of d, sum of d x_hat; d = dtiled act'(bn(tiled))) as one pair per workgroup: partial holds
 * channels x hs_dw_tiles_bn_bwd_in_partials(batch, H, W, fh, fw) pairs of floats.  hs_bn_act_train_bwd_apply = the second launch of
 * hs_bn_act_train_bwd from such pairs (n_partials per channel, combined in order): dx, dgamma, dbeta.  Same values as the two-launch adjoint up to
 * the association of the sums. */
int64_t hs_dw_tiles_bn_bwd_in_partials(int32_t batch, int32_t H, int32_t W, int32_t fh, int32_t fw);
int hs_dw_tiles_bn_bwd_in(int32_t dtype, const void* dy, const float* bank, int64_t ld, const void* tiled, const float* gamma, const float* beta,
                          const float* save_mean, const float* save_invstd, int32_t act, int32_t batch, int32_t channels, int32_t H, int32_t W,
                          int32_t fh, int32_t fw, void* dtiled, float* partial, int32_t patch_major, void* stream);
int hs_bn_act_train_bwd_apply(int32_t dtype, const void* x, const void* dy, int32_t batch, int32_t channels, int64_t pixels, const float* gamma,
                              const float* beta, const float* save_mean, const float* save_invstd, int32_t act, const float* partial,
                              int64_t n_partials, void* dx, float* dgamma, float* dbeta, void* stream);

/* signal2weights on the TRAINING path, every level of a decoder per launch (hyperseg_v1_0.py:473-484 + the permute / reshape of
 * :334-337, 491, and their autograd): the weights are the Conv2d parameters in their OWN layout (wc, signal_channels / groups) --
 * they change every step, so no transposed or packed copy exists -- and banks / bank gradients are patch-major (P, ld), P = batch*fh*fw.
 *   hs_s2w_train_fwd   bank[p, n] = sum_k w[n, k] * signal[b, signal_index + g(n) * K + k, i, j],  n < rows        (one launch)
 *   hs_s2w_train_bwd   dw[n, k] = sum_p dbank[p, n] * signal[...]  (rows in [rows, wc): exact zeros; dbank NULL = zero gradient),
 *                      dsignal[b, c, i, j] = sum over the layers whose channel range holds c of sum_n dbank[p, n] * w[n, k]
 *                      (channels no layer reads: zeros); `ds` is each layer's private (batch, signal_channels, fh, fw) scratch.
 *                      dw / dsignal may be NULL (not wanted).  workspace (hs_s2w_train_workspace bytes, scratch, optional): the
 *                      reduction of dw over the patches is then cut into slices of 64 patches across workgroups and a fourth launch
 *                      adds the slices in order; without it one workgroup walks all patches of its tile.  No atomics. */
typedef struct hs_s2w_train_layer {
    int32_t signal_index, signal_channels, groups;
    const float* w;            /* (wc, signal_channels / groups) */
    int32_t wc, rows;
    float* bank;               /* forward: (P, ld) out */
    int64_t ld;
    const float* dbank;        /* backward: (P, ld) or NULL */
    float* dw;                 /* backward: (wc, signal_channels / groups) out or NULL */
    float* ds;                 /* backward: scratch, see above */
} hs_s2w_train_layer;
int hs_s2w_train_fwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                     const hs_s2w_train_layer* layers, int32_t n_layers, void* stream);
int64_t hs_s2w_train_workspace(int32_t batch, int32_t fh, int32_t fw, const hs_s2w_train_layer* layers, int32_t n_layers);
int hs_s2w_train_bwd(const float* signal, int32_t batch, int32_t c_signal, int32_t fh, int32_t fw,
                     const hs_s2w_train_layer* layers, int32_t n_layers, float* dsignal, void* workspace, int64_t workspace_bytes,
                     void* stream);

/* Per-pixel cross entropy, F.cross_entropy(logits (N,C,H,W), target (N,H,W) int64, ignore_index, reduction='none') without class
 * weights -- what BootstrappedCrossEntropyLoss.forward computes before its top-k rule (hyperseg/losses/bootstrapped_ce_loss.py:20-23) --
 * and its adjoint, one launch each: loss (N,H,W) (0 at ignored pixels); grad_logits (N,C,H,W) = (softmax - onehot) * grad_loss. */
int hs_cross_entropy_fwd(const float* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                         int64_t ignore_index, float* loss, void* stream);
int hs_cross_entropy_bwd(const float* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                         int64_t ignore_index, const float* grad_loss, float* grad_logits, void* stream);
/* ... with the logits' storage type (hs_dtype; the logits' gradient alike; loss and grad_loss fp32; f32 arithmetic): bf16 logits straight
 * from the decoder under torch.autocast(bfloat16), where the stock op widens them with a cast launch first. */
int hs_cross_entropy_typed_fwd(int32_t dtype, const void* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                               int64_t ignore_index, float* loss, void* stream);
int hs_cross_entropy_typed_bwd(int32_t dtype, const void* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                               int64_t ignore_index, const float* grad_loss, void* grad_logits, void* stream);
/* BootstrappedCrossEntropyLoss.forward / backward as one entry each (hyperseg/losses/bootstrapped_ce_loss.py:15-27: per-pixel cross entropy, then per
 * image the mean of the losses above `thresh` if more than k exceed it, else of the k largest; the batch's mean of those).  Forward:
 * hs_cross_entropy_typed_fwd + hs_bootstrap_mean_of_batch_fwd; loss (N, pixels) fp32 = the per-pixel losses (kept for the adjoint),
 * out8 (N, 8) = the per-image selection state of hs_bootstrap_mean_batched_fwd, mean_out[0] = the loss.  workspace: N x hs_bootstrap_mean_workspace()
 * bytes, any contents.  pixels > k (the reference indexes ranked[k]).  Backward: ONE launch (the pixel weights are formed inside the cross entropy's
 * adjoint: no hs_bootstrap_mean_of_batch_bwd launch, no (N, pixels) gradient tensor), grad_logits (N, C, pixels) in the logits' type from the saved
 * `loss`, `out8` and the one upstream gradient grad_mean[0].  Values identical to the separate entries'. */
int hs_bootstrapped_ce_fwd(int32_t dtype, const void* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                           int64_t ignore_index, int32_t k, float thresh, void* workspace, float* loss, float* out8, float* mean_out, void* stream);
int hs_bootstrapped_ce_bwd(int32_t dtype, const void* logits, const int64_t* target, int32_t batch, int32_t classes, int64_t pixels,
                           int64_t ignore_index, const float* loss, const float* state8, const float* grad_mean, void* grad_logits, void* stream);

/* Materialises a stage input (B, 2*coords + c_skip + c_prev, H, W); test/diagnostic twin of the
 * fused prologue (the product path never calls it). */
int hs_stage_input_fwd(const hs_stage_input* in, float* y, void* stream);
/* ... with storage types (hs_dtype): in->prev points at prev_dtype elements, y at out_dtype elements; in->skip is fp32.  The training
 * path's stage input under bf16 autocast (autograd.StageMaterialize): the previous level is read as it is stored and the result is
 * written in the type the first convolution reads, instead of a cast launch on either side. */
int hs_stage_input_typed_fwd(const hs_stage_input* in, int32_t prev_dtype, int32_t out_dtype, void* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HYPERSEG_HIP_H */

/* libgimhip -- C ABI of the MI355X (gfx950) gim_loftr hot path.
 *
 * The reference (xuelunshen/gim) is 100 % Python and has no FFI of its own; its hot path is the
 * `forward` of `networks/loftr/loftr.py:43-91` built from stock torch ops.  Each entry point below
 * replaces one group of those torch ops (reference file:line cited per function) and is what the
 * Python host shell `gim_amd/loftr/` binds through `ctypes` (see INTEGRATION.md for the stub a
 * reference maintainer would add).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless noted;
 *   - every call is asynchronous on `stream` (pass torch.cuda.current_stream().cuda_stream);
 *   - the library never allocates user-visible memory: outputs / workspaces are caller-allocated;
 *   - return 0 on success, negative GIM_ERR_* otherwise; message via gim_last_error() (thread local);
 *   - `dtype` arguments: GIM_F32 (exact-parity mode, fp32 MFMA), GIM_BF16 (throughput mode, 8 significand bits, fp32 range)
 *     or -- on the gim_loftr entry points (conv, layout / pos-enc / LayerNorm glue, linear attention, coarse matching, fine
 *     gather) -- GIM_F16 (throughput mode, IEEE fp16 operands: 11 significand bits at the same MFMA rate, |x| < 65504; a
 *     quarter of bf16's index flips against the fp32 reference).  A call is of ONE 16-bit kind; the one
 *     mixed form is gim_conv2d_bn_act with dtype GIM_F16 and out_dtype GIM_BF16 (no residual / upsample operand): the bf16
 *     mode's first convolution reads the image as fp16;
 *   - activations are NHWC ("pixel rows"): row m = ((b*H + y)*W + x), `ld*` = row stride in ELEMENTS.
 */
#ifndef GIM_HIP_H
#define GIM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gim_stream_t; /* hipStream_t */

enum { GIM_F32 = 0, GIM_BF16 = 1, GIM_F16 = 2 };
enum { GIM_ACT_NONE = 0, GIM_ACT_RELU = 1, GIM_ACT_LEAKY = 2, GIM_ACT_ELU1 = 3 /* elu(x)+1 */, GIM_ACT_GELU = 4 /* exact erf GELU */ };
enum { GIM_OK = 0, GIM_ERR_INVALID = -1, GIM_ERR_LAUNCH = -2, GIM_ERR_UNSUPPORTED = -3 };

/* 110 (round 5): the fp16 range-guard word is an ARGUMENT of the entry points that use it (`health` of gim_bneck64_fused*,
 * gim_bneck_tail*, gim_conv_args.health; gim_set_range_guard() is gone), and gim_coarse_match's `count` is int32[2 + N] (count[1] = health
 * word, per-pair counts from count[2]) -- a caller bound to version 100 must be rebuilt.
 * 111: struct gim_token_emit grew the kv_part / kv_nchunk / kv_tile0 / kv_len arrays (fused KV state), q_weights (local queries), project_only and pe_*; zero them for
 * the old behaviour;
 * new entries gim_linear_attention_finalize, gim_linear_attention_ws_bytes_chunks.
 * 112 (round 6): gim_coarse_args.precand_per_row appended (zero it for the old behaviour); the library reads no environment variable.
 * 113 (round 6): gim_conv_args.split16 appended (zero it for the old behaviour). */
int gim_version(void);
/* fp16 range guard.  `health` (NULL: no check): a device word into which the fp16 flavour of the kernels that store un-normalised
 * residual streams (gim_bneck64_fused*, gim_bneck_tail*, gim_conv2d_bn_act with a residual operand) OR 4 when a converted value exceeds
 * the IEEE-fp16 range -- ReLUs downstream turn the resulting NaNs into zeros, so the outputs alone do not show it.  A plain argument:
 * no process-wide state, a captured graph keeps the word it was captured with.  gim_amd passes count[1] of gim_coarse_match. */
const char* gim_last_error(void);
/* compile-time facts the host packer needs: K-tile bytes (128) and the N padding granule (64). */
int gim_ktile_bytes(void);
int gim_npad_granule(void);

/* --------------------------------------------------------------------------------------------
 * Layout conversion: [B,C,H,W] fp32 (the reference's boundary layout, loftr.py:59) ->
 * NHWC rows [B*H*W, ld] of `dtype`, channels >= C zero-filled up to `cpad`.
 * `b_off`: first output image index (color0 -> 0, color1 -> bs: replaces torch.cat, loftr.py:60). */
int gim_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int cpad, int ld,
                     int b_off, int dtype, gim_stream_t stream);
/* The same boundary conversion for the split-operand first convolution (16-bit dtypes only): every channel c becomes the pair
 * hi = rn16(x), lo = rn16(x - hi), stored as channels [hi(0..C) | lo(0..C) | hi(0..C) | zeros up to ld]; with weights packed as
 * [w_hi | w_hi | w_lo] (gim_amd/packing.py::pack_conv_split) the 16-bit MFMA evaluates x*w to 2^-22 instead of 2^-11: the input
 * of backbone/resnet.py:306 (conv1 7x7 on the image) is where half of the 16-bit modes' deviation from the fp32 reference arises. */
int gim_nchw_to_nhwc_split(const float* src, void* dst, int B, int C, int H, int W, int ld, int b_off, int dtype,
                           gim_stream_t stream);
/* (ld >= 3 C: the layout above.  2 C <= ld < 3 C: [hi(0..C) | lo(0..C) | zeros] -- one 16-byte piece per RGB pixel, the input of gim_stem7x7.) */

/* --------------------------------------------------------------------------------------------
 * The first convolution of the backbone as its own kernel: conv1 7x7 / stride 2 / pad 3, 3 -> 64, + bn1 (folded) + relu
 * (backbone/resnet.py:306).  x [B,H,W,8] 16-bit pixels (one 16-byte piece each: gim_nchw_to_nhwc with cpad 8 for split = 0,
 * gim_nchw_to_nhwc_split with ld 8 for split = 1), w = the LDS image of the filter bank (gim_stem7x7_weight_bytes(split) bytes,
 * gim_amd/packing.py::pack_stem7x7: split = 1 carries every filter as a hi + lo pair), bias [64] fp32, y [B,Ho,Wo,64] of out_dtype
 * (bf16 / fp16; Ho = (H - 1) / 2 + 1).  dtype = the operands' 16-bit kind.  The implicit-GEMM entry computes the same layer
 * (split: on 16 channels per pixel) and remains the fp32 mode's path. */
int64_t gim_stem7x7_weight_bytes(int split);
int gim_stem7x7(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int split, int dtype, int out_dtype,
                gim_stream_t stream);
int gim_stem7x7_f16(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int split, int dtype, int out_dtype,
                    gim_stream_t stream);
/* inverse, for exposing feature maps in the reference layout (tests / lazy outputs) */
int gim_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int ld, int dtype,
                     gim_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution with fused epilogue (MFMA):
 *     y[m, n] = act( sum_k A[m,k] * w[n,k] + bias[n] + res[m or m % res_mod, n] )
 * Replaces Conv2d+BatchNorm2d(eval)+ReLU/LeakyReLU(+residual add) of
 * `networks/loftr/backbone/resnet.py:109-126,230-233,316-327` and, with a 1x1 "pixel = row" view,
 * every bias-free nn.Linear of `submodules/transformer.py:47-55` (+ elu+1 of attentions.py:31-32).
 * A[m,k] is gathered on the fly: k -> (dy,dx,c) through `ktab` (one int per 16-byte K group:
 * c | dx<<16 | dy<<24, dy=255 marks K padding), pixel (b, ho*stride-pad+dy, wo*stride-pad+dx).
 * Weights `w` are packed [npad][kpad] in `dtype` with BN already folded in (host side). */
typedef struct gim_conv_args {
    const void* x;      /* input rows, dtype */
    const void* w;      /* packed weights [npad][kpad], dtype */
    const int32_t* ktab;/* [(kpad*elemsize/128 + 2) * 8]: two trailing slabs of 0xFF000000 */
    const float* bias;  /* [npad] or NULL */
    const void* res;    /* residual rows or NULL */
    void* y;            /* output rows */
    int64_t x_bytes;    /* size of the x allocation in bytes (buffer-descriptor bound, < 4 GiB) */
    int B, H, W;        /* input images / spatial size */
    int Ho, Wo;         /* output spatial size */
    int stride, pad;
    int ldx, ldy, ldres;/* row strides in elements */
    int N;              /* valid output channels stored (multiple of 4, <= npad) */
    int npad, kpad;
    int act;            /* GIM_ACT_* */
    int act_cols;       /* 0: activation on every column; >0 (multiple of 128): only on columns < act_cols
                           (fused [q|k|v] projection: elu+1 on q,k but not on v) */
    int res_mod;        /* 0: res row = m;  >0: res row = m % res_mod (broadcast over images) */
    int dtype;          /* dtype of x and w */
    int out_dtype;      /* dtype of y */
    int res_dtype;      /* dtype of res */
    int use_lds_dma;    /* 1: buffer_load ... lds staging (default); 0: register staging; 2: 3x3 halo kernel (w / ktab / kpad =
                           the halo packing, res_mod = channels stored per input row); 3: as 1, and the 256 x 256 tile takes the launch
                           whatever its size (its selection is a size heuristic: the parity tests reach it on small shapes this way) */
    const void* ups;    /* NULL, or a half-resolution tensor [B, ups_h, ups_w, ups_ld] (dtype = out_dtype, 16-bit) whose bilinear x2
                           upsampling (align_corners=True) is added to the conv output in fp32, in front of the ONE 16-bit rounding (round 6:
                           as 48 more K of the MFMA per 32-pixel pass; act must be GIM_ACT_NONE): the FPN's
                           `x2_out + F.interpolate(x3_out, scale_factor=2.)` (backbone/resnet.py:321-327) without a second pass
                           over y.  Output rows = (image, Y < 2 ups_h, X < 2 ups_w); needs (2 ups_w) % 32 == 0 and a launch the 256 x 256 tile takes
                           (1x1 conv, bf16, no residual, npad % 256 == 0, >= 4 K slabs, >= 512 tiles): gim_conv_ups_supported() */
    int ups_h, ups_w, ups_ld;
    int32_t* health;    /* fp16 range guard (see the top of this file) or NULL: checked where a residual operand is added (fp16 flavour) */
    int split16;        /* fp32 operands only (ABI 113).  0: products on v_mfma_f32_32x32x2_f32 (exact fp32 products).  1: every fp32 operand value is
                           split in registers into an IEEE-fp16 pair hi + lo and x w is evaluated as hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16
                           with fp32 accumulation (weights scaled by 2^12 for the split, accumulators scaled back): 2^-22 relative per product at
                           3/8 of the matrix-pipe time; needs |x| < 65504 and |w| < 16 (a value beyond becomes inf, the output NaN) */
    int pad_;
} gim_conv_args;
int gim_conv_ups_supported(const gim_conv_args* a);   /* 1 if gim_conv2d_bn_act would take a->ups (set or not) for this launch */
int gim_conv2d_bn_act(const gim_conv_args* a, gim_stream_t stream);

/* y[m,:] += bilinear_upsample_2x(x)[m,:], align_corners=True (resnet.py:321,325: F.interpolate +
 * the `x2_out+x3_out_2x` add).  x: [B,h,w,C] rows (ldx), y: [B,2h,2w,C] rows (ldy), in place. */
int gim_upsample2x_add(const void* x, void* y, int B, int h, int w, int C, int ldx, int ldy,
                       int dtype, gim_stream_t stream);

/* out_f32[m,:] = x[m,:] + pe[m % hw, :]   (loftr.py:74-75 pos_encoding + 'n c h w -> n (h w) c';
 * NHWC rows make the rearrange free).  Also writes a `dtype` copy (GEMM operand) if out_t != NULL. */
int gim_posenc_add(const void* x, const float* pe, float* out_f32, void* out_t, int rows, int hw,
                   int C, int ldx, int ld_f32, int ld_t, int dtype, gim_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * LinearAttention (submodules/attentions.py:20-47), q/k already elu+1'd by the projection epilogue.
 *   step 1  gim_linear_attention_kv:  KV[b,h,:,:] = sum_s K[b,s,h,:]^T (V[b,s,h,:]/S),  Ksum[b,h,:]
 *   step 2  gim_linear_attention_apply: out[b,l,h,:] = (Q KV) / (Q.Ksum + eps) * S
 * k,v: [nb*S, ld] rows, q: [nb*L, ld] rows; `kv_ws` >= gim_linear_attention_ws_bytes(...) bytes.
 * kv_mask [nb*S] / q_mask [nb*L] (uint8, NULL = no mask): padded positions, attentions.py:35-39
 * (K, V rows with mask 0 do not contribute; Q rows with mask 0 give a zero message).
 * Arithmetic of step 1 at the coarse level (D = 32, H = 8): 16-bit operands -- exact products on the 16-bit MFMA, fp32 sums, 1/S applied
 * to the sums; fp32 operands -- fp32 MFMA on K and V/S.  Partial sums of row chunks are combined in a fixed order (run-to-run deterministic).
 * The partials are those of 256-row chunks whatever workgroup shape the launch takes: a sequence's state does not depend on the batch it
 * travels in, bit for bit.
 * gim_linear_attention_finalize (version 111): the last step alone -- state = sum of `nchunk` partial states per sequence and head that somebody
 * else wrote (gim_token_mlp_emit's fused KV state: one partial per 64-row tile) into a workspace of gim_linear_attention_ws_bytes_chunks bytes:
 * [state: nb x H x (D D + D) fp32][partials: nb x H x nchunk x (D D + D) fp32]. */
int64_t gim_linear_attention_ws_bytes(int nb, int S, int H, int D);
int64_t gim_linear_attention_ws_bytes_chunks(int nb, int H, int D, int nchunk);
int gim_linear_attention_finalize(float* kv_ws, int nb, int H, int D, int nchunk, gim_stream_t stream);
int gim_linear_attention_kv(const void* k, const void* v, const uint8_t* kv_mask, float* kv_ws, int nb,
                            int S, int H, int D, int ldk, int ldv, int dtype, gim_stream_t stream);
int gim_linear_attention_apply(const void* q, const uint8_t* q_mask, const float* kv_ws, void* out,
                               int nb, int L, int S, int H, int D, int ldq, int ldo, int dtype,
                               int out_dtype, gim_stream_t stream);

/* Same LinearAttention for SHORT sequences (the fine level's 25-token windows, transformer on [M,25,128],
 * loftr.py:88): one wave per sequence fuses both steps, no workspace.  Requires H == 8, D in {16, 32}. */
int gim_linear_attention_short(const void* q, const void* k, const void* v, const uint8_t* q_mask,
                               const uint8_t* kv_mask, void* out, int nb, int L, int S, int H, int D,
                               int ldq, int ldk, int ldv, int ldo, int dtype, int out_dtype,
                               gim_stream_t stream);

/* LayerNorm (+ optional residual):  v = LN(x[m,:]) * gamma + beta ; if res: v += res[m,:]
 * (transformer.py:52,56,58).  x is `x_dtype` (fp32, or bf16 pre-norm activations in the throughput mode).
 * Writes fp32 (out_f32, may be NULL) and/or `dtype` copy (out_t). */
int gim_layernorm_residual(const void* x, const float* gamma, const float* beta, const float* res,
                           float* out_f32, void* out_t, int rows, int C, int ldx, int ldres,
                           int ld_f32, int ld_t, int x_dtype, int dtype, float eps, gim_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Coarse matching (utils/coarse_matching.py:88-259): dual-softmax + threshold + border + mutual-NN
 * + ordered compaction, fused so that the [N,L,S] confidence matrix is never written.
 * feat0 [N*L, C] / feat1 [N*S, C] fp32 rows (row stride C).  Outputs in torch.where order
 * (ascending b, then i): b_ids/i_ids/j_ids int64, mconf fp32, mkpts0_c/mkpts1_c fp32 [cap,2].
 * `count` (device int32[2 + N], ZERO IT ONCE when allocating): count[0] = M, count[2+b] = matches of pair b, count[1] = health word:
 * bit 0 (rewritten by every call) = a NaN / inf similarity reached the statistics -- the inputs or weights were not finite, or an
 * overflow survived the ReLUs in between; bit 1 (sticky, never cleared here) = gim_fine_fused_dev saw a non-finite fine-level
 * output on this buffer: a host that replays a captured graph on the same buffer learns it with the NEXT call's count read-back;
 * bit 2 (sticky) = the fp16 range guard: a kernel that stores an un-normalised residual stream converted a value beyond 65504
 * that was handed this word as its `health` argument.  The reference has no such word (fp32 has the range);
 * gim_amd/loftr/loftr.py reads it with the match count (coarse_matching.py:193's sync) and re-runs the batch in bf16.
 * scale0/scale1: NULL or fp32 [N,2] per-pair (w,h) scales (coarse_matching.py:237-245). */
typedef struct gim_coarse_args {
    const void* feat0;    /* [N, L, ldf] rows of C features, fp32 or bf16 (feat_dtype) */
    const void* feat1;    /* [N, S, ldf] */
    const float* scale0;
    const float* scale1;
    const uint8_t* mask0; /* NULL or [N, L] padding mask of image0's coarse cells (coarse_matching.py:116-117) */
    const uint8_t* mask1; /* NULL or [N, S]; with masks the border rule is mask_border_with_padding (:29-44) */
    void* ws;             /* >= gim_coarse_match_ws_bytes() */
    int64_t* b_ids;
    int64_t* i_ids;
    int64_t* j_ids;
    float* mconf;
    float* mkpts0_c;
    float* mkpts1_c;
    int32_t* count;
    int N, L, S, C;
    int h0c, w0c, h1c, w1c;
    int cap;              /* capacity of the output arrays (N * min(L,S) always suffices) */
    float temperature;    /* dsmax_temperature (0.1) */
    float thr;            /* 0.2 */
    int border_rm;        /* 2 */
    float scale;          /* hw0_i[0] / hw0_c[0] */
    int feat_dtype;       /* GIM_F32 (0, default) or GIM_BF16: bf16 features run the similarity on the bf16 MFMA -- exact
                           * products, fp32 accumulation, i.e. the fp32 result for bf16-valued inputs up to summation order */
    int ldf;              /* row stride of feat0 / feat1 in elements; 0 = C (contiguous) */
    int precand_per_row;  /* 0 = default (16 pre-candidate slots per row, what gim_coarse_match_ws_bytes sizes); 1..16 = fewer; -1 = none: the
                           * device-side overflow flag then routes every call to the two-pass recompute (tests; replaces an environment hook) */
} gim_coarse_args;
int64_t gim_coarse_match_ws_bytes(int N, int L, int S);
int gim_coarse_match(const gim_coarse_args* a, gim_stream_t stream);
/* materialise conf_matrix [N,L,S] fp32 (data['conf_matrix'], coarse_matching.py:144) on demand;
 * needs the workspace of the preceding gim_coarse_match call (row/column softmax statistics). */
int gim_coarse_conf_matrix(const gim_coarse_args* a, float* conf, gim_stream_t stream);

/* Tail of a ResNet Bottleneck (planes 64) fused with the head of the next block (resnet.py:109-126), one kernel:
 *     x' = relu(bn3(conv3(relu(bn2(conv2_3x3(t1))))) + identity);   t1' = relu(bn1'(conv1'(x')))   (optional)
 * t1: [B,H,W,64] bf16 (the block's conv1 output), res: [B,H,W,256] bf16 identity / downsample branch, x_out: [B,H,W,256],
 * t1_next: [B,H,W,n_next] or NULL (n_next = 64: the next block of the layer; 128: the next layer's first conv1; 0: none).
 * Weights bf16 with eval-BN folded in (biases fp32): w2 [64][576] with K = (ky, kx, c);
 * w3 [256][64] and w1n [n_next][256] with K permuted to the MFMA accumulator layout (gim_amd/packing.py::pack_bneck) -- the three
 * products are chained through registers.  H % 8 == 0, W % 32 == 0. */
int gim_bneck64_fused(const void* t1, const void* res, void* x_out, void* t1_next, const void* w2, const void* w3,
                      const void* w1n, const float* b2, const float* b3, const float* b1n, int B, int H, int W,
                      int n_next, int32_t* health, gim_stream_t stream);
/* First block of layer 1 (resnet.py:120-124: identity = bn(conv1x1(x)), 64 -> 256, stride 1): the downsample convolution runs INSIDE
 * the kernel as extra K of conv3 -- x' = relu([W3 | Wds] [t2 ; x] + b3 + bds) -- so neither its launch nor the 256-channel identity
 * tensor exist.  x_in: [B,H,W,64] the block's input; wds [256][64] bf16 (BN folded, K in channel order); b3ds = b3 + bds; n_next = 64. */
int gim_bneck64_fused_ds(const void* t1, const void* x_in, void* x_out, void* t1_next, const void* w2, const void* w3,
                         const void* wds, const void* w1n, const float* b2, const float* b3ds, const float* b1n, int B, int H, int W,
                         int32_t* health, gim_stream_t stream);
int gim_bneck64_fused_ds_f16(const void* t1, const void* x_in, void* x_out, void* t1_next, const void* w2, const void* w3,
                             const void* wds, const void* w1n, const float* b2, const float* b3ds, const float* b1n, int B, int H, int W,
                             int32_t* health, gim_stream_t stream);
/* the same kernel on IEEE fp16 tensors / weights (GIM_F16 mode) */
int gim_bneck64_fused_f16(const void* t1, const void* res, void* x_out, void* t1_next, const void* w2, const void* w3,
                          const void* w1n, const float* b2, const float* b3, const float* b1n, int B, int H, int W,
                          int n_next, int32_t* health, gim_stream_t stream);

/* The same fusion one layer up (planes 128: layer 2), without the 3x3 -- x' = relu(bn3(conv3_1x1(t2)) + identity), t1' =
 * act(bn1'(conv1'_1x1(x'))) of the NEXT block (resnet.py:117-124, 109-111) -- so that x' [M,512], the widest tensor of the block, is
 * written once and not read back.  t2: [M,128] (the block's conv2 output), res: [M,512], x_out: [M,512], t1_next: [M,n_next], n_next
 * in {128, 256}; M = pixel rows, a multiple of 256 (both convolutions are 1x1: no spatial structure).  w3 [512][128] K in channel
 * order, w1n [8][n_next][64]: per 64-channel chunk of x', K in accumulator order (gim_amd/packing.py::pack_bneck_tail); the 256+ KiB
 * of weights stream through LDS two chunks ahead of the MFMAs.  act_next: GIM_ACT_RELU / GIM_ACT_NONE. */
int gim_bneck_tail128(const void* t2, const void* res, void* x_out, void* t1_next, const void* w3, const void* w1n,
                      const float* b3, const float* b1n, int M, int n_next, int act_next, int32_t* health, gim_stream_t stream);
int gim_bneck_tail128_f16(const void* t2, const void* res, void* x_out, void* t1_next, const void* w3, const void* w1n,
                          const float* b3, const float* b1n, int M, int n_next, int act_next, int32_t* health, gim_stream_t stream);
/* First block of layer 2 with its `downsample` branch INSIDE the kernel (round 5; resnet.py:120-124: identity = bn(conv1x1, stride 2 (x))):
 *     x' = relu([W3 | Wds] [t2 ; x_in(b, 2y, 2x)] + b3 + bds);   t1' = act(bn1'(conv1'(x')))
 * -- neither the downsample launch nor its 512-channel output exist.  t2 [B,Ho,Wo,128] (conv2 output, stride 2), x_in [B,Hin,Win,256] the
 * block's input (Hin >= 2 Ho - 1, Win >= 2 Wo - 1), x_out [B,Ho,Wo,512], t1_next [B,Ho,Wo,128]; B Ho Wo a multiple of 256.
 * w3ds [512][128 + 256] (K: t2's channels, then x_in's, both in channel order, BN folded), w1n [16][128][32] (32-channel chunks of x', K in
 * accumulator order), b3ds = b3 + bds (gim_amd/packing.py::pack_bneck_tail(..., ds=True)). */
int gim_bneck_tail128_ds(const void* t2, const void* x_in, void* x_out, void* t1_next, const void* w3ds, const void* w1n,
                         const float* b3ds, const float* b1n, int B, int Ho, int Wo, int Hin, int Win, int n_next, int act_next,
                         int32_t* health, gim_stream_t stream);
int gim_bneck_tail128_ds_f16(const void* t2, const void* x_in, void* x_out, void* t1_next, const void* w3ds, const void* w1n,
                             const float* b3ds, const float* b1n, int B, int Ho, int Wo, int Hin, int Win, int n_next, int act_next,
                             int32_t* health, gim_stream_t stream);
/* Planes 256 (layer 3): t2 [M,256], res / x_out [M,1024], t1_next [M,256] (n_next = 256); w3 [1024][256], w1n [32][256][32] (chunks of
 * 32 channels).  x_out may be NULL: the last block's output is read by nothing but the fused 1x1 convolution -- the FPN's
 * layer3_outconv (resnet.py:316), act_next = GIM_ACT_NONE, zero bias -- so it is never written. */
int gim_bneck_tail256(const void* t2, const void* res, void* x_out, void* t1_next, const void* w3, const void* w1n,
                      const float* b3, const float* b1n, int M, int n_next, int act_next, int32_t* health, gim_stream_t stream);
int gim_bneck_tail256_f16(const void* t2, const void* res, void* x_out, void* t1_next, const void* w3, const void* w1n,
                          const float* b3, const float* b1n, int M, int n_next, int act_next, int32_t* health, gim_stream_t stream);

/* Token-wise tail of a LoFTREncoderLayer in ONE kernel (bf16 operand mode, d_model 256; transformer.py:52-58):
 *     x += norm2(mlp.2(relu(mlp.0(cat[x, norm1(merge(msg))]))))
 * msg: [R][ldm] bf16 attention output; xb: [R][ldxb] bf16 operand copy of x (read, then overwritten with the new x);
 * x32: [R][ldx32] fp32 residual stream (read-modify-write).  `weights`: gim_token_mlp_weight_bytes() bytes of bf16 in the
 * per-wave fragment order of gim_amd/packing.py::pack_token_mlp; ln_params: [norm1.weight | norm1.bias | norm2.weight |
 * norm2.bias] fp32.  64 rows per workgroup; none of the intermediate row buffers exists in memory.
 * kv != NULL fuses the apply step of the linear attention (attentions.py:44-45) in front: `msg` then holds the elu+1 QUERY rows,
 * kv the [nb][8][32*32 + 32] fp32 state written by gim_linear_attention_kv (nb = R / L sequences of L rows, L % 64 == 0), S the
 * source length, q_mask an optional [R] padding mask. */
int64_t gim_token_mlp_weight_bytes(void);
int gim_token_mlp(const void* msg, void* xb, float* x32, const void* weights, const float* ln_params, const float* kv,
                  const uint8_t* q_mask, int R, int C, int L, int S, int ldm, int ldxb, int ldx32, float ln_eps,
                  gim_stream_t stream);
/* the same kernel with fp16 operand rows / weights (GIM_F16 mode) */
int gim_token_mlp_f16(const void* msg, void* xb, float* x32, const void* weights, const float* ln_params, const float* kv,
                      const uint8_t* q_mask, int R, int C, int L, int S, int ldm, int ldxb, int ldx32, float ln_eps,
                      gim_stream_t stream);

/* gim_token_mlp + projection blocks of the NEW x, computed on the tile while it is still in LDS: the q / k / v projections of
 * the following LoFTREncoderLayer (transformer.py:42-44; in a cross layer also the k / v of the same layer's second call), each
 * a bias-free [256 x 256] Linear with elu(.)+1 on q and k (attentions.py:31-32):   out[b][m, 0:256] = act[b](x_new[m] W_b^T)
 * for the rows m of 64-row tiles whose first row lies in [row_lo[b], row_hi[b]).  weights: nblk x 128 KiB of 16-bit values in the
 * per-wave fragment order of gim_amd/packing.py::pack_token_emit. */
#define GIM_TOKEN_EMIT_MAX 6
typedef struct gim_token_emit {
    int nblk;
    const void* weights;
    void* out[GIM_TOKEN_EMIT_MAX];      /* row 0 of the block's output columns, 16-bit, 16-byte aligned */
    int ld[GIM_TOKEN_EMIT_MAX];         /* row stride in elements (multiple of 8) */
    int act[GIM_TOKEN_EMIT_MAX];        /* GIM_ACT_NONE or GIM_ACT_ELU1 */
    int row_lo[GIM_TOKEN_EMIT_MAX], row_hi[GIM_TOKEN_EMIT_MAX];
    /* Fused KV state (version 111).  kv_part[b] != NULL: blocks b (k, elu + 1) and b + 1 (v; same row range, whole 64-row tiles) are NOT
     * written; each tile stores the partial state K^T V / kv_len, K^T 1 of its 64 rows (attentions.py:38-43) at
     * kv_part[b][sequence][8 heads][kv_nchunk[b]][32 * 32 + 32] fp32, tile index = kv_tile0[b] + (tile's first row - row_lo[b]) / 64 counted
     * over the consuming call's source rows (sequence = index / kv_nchunk, chunk = index % kv_nchunk).  kv_part is the partial area of a
     * gim_linear_attention_kv workspace (gim_linear_attention_part) and gim_linear_attention_finalize turns it into the state.  out[] /
     * ld[] of the two blocks are ignored. */
    float* kv_part[GIM_TOKEN_EMIT_MAX];
    int kv_nchunk[GIM_TOKEN_EMIT_MAX], kv_tile0[GIM_TOKEN_EMIT_MAX];
    float kv_len[GIM_TOKEN_EMIT_MAX];
    /* Local queries (version 111).  q_weights != NULL (needs kv != NULL): `msg` is not read; the elu + 1 query rows of the fused attention
     * apply are projected inside the kernel from xb, q = elu(xb Wq^T) + 1 (transformer.py:42, attentions.py:31).  128 KiB of 16-bit values in
     * the order of ONE block of packing.py::pack_token_emit([Wq]).  nblk may be 0 then. */
    const void* q_weights;
    /* Projection only (version 111).  != 0: nothing of the layer runs -- msg, x32, weights, ln_params, kv may be NULL -- the blocks are
     * projections of the 16-bit rows `xb` as they are (the first layer's k / v pair as partial KV states: replaces its projection GEMM). */
    int project_only;
    /* ... with the positional encoding in front (loftr.py:74-75; pe != NULL, project_only only): the rows are pe_feat[m] (16-bit, stride pe_ld)
     * + pe[m % pe_hw] (fp32 [pe_hw][256]) in gim_posenc_add's arithmetic; the kernel writes them to x32 (fp32) and xb (16-bit) before it projects. */
    const void* pe_feat;
    const float* pe;
    int pe_ld, pe_hw;
} gim_token_emit;
int gim_token_mlp_emit(const void* msg, void* xb, float* x32, const void* weights, const float* ln_params, const float* kv,
                       const uint8_t* q_mask, int R, int C, int L, int S, int ldm, int ldxb, int ldx32, float ln_eps,
                       const gim_token_emit* emit, gim_stream_t stream);
int gim_token_mlp_emit_f16(const void* msg, void* xb, float* x32, const void* weights, const float* ln_params, const float* kv,
                           const uint8_t* q_mask, int R, int C, int L, int S, int ldm, int ldxb, int ldx32, float ln_eps,
                           const gim_token_emit* emit, gim_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Fine level.  gim_fine_gather = F.unfold(k=W,stride,pad=W/2) + [b_ids,i_ids] pick
 * (submodules/fine_preprocess.py:40-47) without materialising the unfold: windows of image0 go to
 * rows [0, M*WW), windows of image1 to rows [M*WW, 2*M*WW).
 * feat_f0: [bs, hf0, wf0, C] rows, feat_f1: [bs, hf1, wf1, C] rows (row stride ldf) in `dtype`. */
int gim_fine_gather(const void* feat_f0, const void* feat_f1, const int64_t* b_ids,
                    const int64_t* i_ids, const int64_t* j_ids, float* out_f32, void* out_t, int M,
                    int hf0, int wf0, int hf1, int wf1, int C, int ldf, int w0c, int w1c, int stride,
                    int W, int ld_f32, int ld_t, int dtype, gim_stream_t stream);
/* FineMatching.forward + get_fine_match (utils/fine_matching.py:43-74): centre-row correlation,
 * softmax over WW, DSNT expectation + std, final coordinates.
 * f0/f1: fp32 [M*WW, C] rows.  scale1: NULL or fp32 [bs,2] (applied iff has_scale0, line 68). */
int gim_fine_match(const float* f0, const float* f1, const float* mkpts1_c, const int64_t* b_ids,
                   const float* scale1, float* expec_f, float* mkpts1_f, int M, int WW, int C, int ld,
                   float scale, int has_scale0, gim_stream_t stream);
/* The whole fine level in ONE kernel (bf16 operand mode, d_model 128, 5x5 windows, layer_names ['self','cross']):
 * window gather (fine_preprocess.py:40-47) + LocalFeatureTransformer (transformer.py:35-58,80-101, LinearAttention
 * attentions.py:20-47) + FineMatching (fine_matching.py:43-74); 2 matches per workgroup, activations never leave the CU.
 * feat_f0/feat_f1: NHWC bf16 fine maps (row stride ldf).  `weights`: gim_fine_fused_weight_bytes() bytes, bf16, per layer
 * [Wq | Wk | Wv | Wmerge | mlp.0 rows 0..127 | mlp.0 rows 128..255 | mlp.2 cols 0..127 | mlp.2 cols 128..255], each block
 * [128 out][K] re-ordered to MFMA fragment order [wave = out/32][k16 step][lane = (k/8 % 2)*32 + out%32][8] (host side:
 * gim_amd/packing.py::pack_fine_fused).  ln_params: per layer [norm1.weight | norm1.bias | norm2.weight | norm2.bias] fp32.
 * dbg_fine0/dbg_fine1: NULL or fp32 [M, 25, 128] dumps of the transformer output (tests). */
int64_t gim_fine_fused_weight_bytes(void);
int gim_fine_fused(const void* feat_f0, const void* feat_f1, const int64_t* b_ids, const int64_t* i_ids,
                   const int64_t* j_ids, const float* mkpts1_c, const float* scale1, const void* weights,
                   const float* ln_params, float* expec_f, float* mkpts1_f, float* dbg_fine0, float* dbg_fine1,
                   int M, int hf0, int wf0, int hf1, int wf1, int C, int ldf, int w0c, int w1c, int stride, int W,
                   float scale, float ln_eps, int has_scale0, gim_stream_t stream);
/* The same launch without the host knowing the match count: it covers M_cap = the capacity of the match lists and processes the first
 * min(M_cap, *count_dev) of them (count_dev = gim_coarse_match's count[0]); expec_f / mkpts1_f hold M_cap rows.  The reference
 * synchronises on the count (torch.where, coarse_matching.py:193) before the fine level; here the read-back overlaps this kernel. */
int gim_fine_fused_dev(const void* feat_f0, const void* feat_f1, const int64_t* b_ids, const int64_t* i_ids,
                       const int64_t* j_ids, const float* mkpts1_c, const float* scale1, const void* weights,
                       const float* ln_params, float* expec_f, float* mkpts1_f, int M_cap, const int* count_dev,
                       int hf0, int wf0, int hf1, int wf1, int C, int ldf, int w0c, int w1c, int stride, int W,
                       float scale, float ln_eps, int has_scale0, gim_stream_t stream);
int gim_fine_fused_dev_f16(const void* feat_f0, const void* feat_f1, const int64_t* b_ids, const int64_t* i_ids,
                           const int64_t* j_ids, const float* mkpts1_c, const float* scale1, const void* weights,
                           const float* ln_params, float* expec_f, float* mkpts1_f, int M_cap, const int* count_dev,
                           int hf0, int wf0, int hf1, int wf1, int C, int ldf, int w0c, int w1c, int stride, int W,
                           float scale, float ln_eps, int has_scale0, gim_stream_t stream);
/* the same kernel on fp16 fine maps / weights (GIM_F16 mode) */
int gim_fine_fused_f16(const void* feat_f0, const void* feat_f1, const int64_t* b_ids, const int64_t* i_ids,
                       const int64_t* j_ids, const float* mkpts1_c, const float* scale1, const void* weights,
                       const float* ln_params, float* expec_f, float* mkpts1_f, float* dbg_fine0, float* dbg_fine1,
                       int M, int hf0, int wf0, int hf1, int wf1, int C, int ldf, int w0c, int w1c, int stride, int W,
                       float scale, float ln_eps, int has_scale0, gim_stream_t stream);


/* ======================================================================================================
 * gim_lightglue path: SuperPoint detector glue + LightGlue matcher.
 * The convolutions and Linear layers of both networks run on gim_conv2d_bn_act above.
 * ====================================================================================================== */

/* nn.MaxPool2d(2, 2) on NHWC rows -- replaces superpoint.py:216,219,222 (`self.pool`). */
int gim_maxpool2x2(const void* x, void* y, int B, int H, int W, int C, int ldx, int ldy, int dtype,
                   gim_stream_t stream);

/* Detector head tail -- replaces superpoint.py:231-236: softmax over the 65 logits of each 8x8 cell
 * (rows [B*h*w][ld], ld >= 65), dustbin dropped, depth-to-space -> scores [B, 8h, 8w] fp32. */
int gim_sp_scores(const void* logits, float* scores, int B, int h, int w, int ld, int dtype,
                  gim_stream_t stream);

/* simple_nms + border removal -- replaces superpoint.py:61-80 (radius, 2 refinement rounds) and :247-258
 * (scores within `border` px of the canvas edge = -1; the reference always uses the canvas size, :207).
 * out [B,H,W] fp32: score at kept maxima, 0 elsewhere, -1 on the border. */
int64_t gim_sp_nms_ws_bytes(int B, int H, int W);
int gim_sp_nms(const float* scores, float* out, void* ws, int B, int H, int W, int radius, int border,
               gim_stream_t stream);

/* Keypoint extraction -- replaces superpoint.py:260-300,308: candidates `score > thr`, top-k by score in
 * descending order (`torch.topk(sorted=True)`; fewer than k candidates: all of them in torch.where order),
 * kpts [B,k,2] = (x, y) as float (without the +0.5 of :346), kscores [B,k], nvalid [B] = min(#candidates, k)
 * (entries >= nvalid are zero; the caller pads them like pad_and_stack).  B <= 64, k <= 4096, thr >= 0. */
int64_t gim_sp_topk_ws_bytes(int B, int H, int W);
int gim_sp_topk(const float* nms_scores, void* ws, float* kpts, float* kscores, int32_t* nvalid, int B,
                int H, int W, int k, float thr, gim_stream_t stream);

/* Descriptor sampling -- replaces superpoint.py:241 (per-pixel F.normalize of the dense map, fused),
 * :120-137 (legacy_sampling: bilinear grid_sample, align_corners=True) and the final F.normalize.
 * dense rows [B*h*w][ld] (raw convDb output), kpts [B,K,2]; out_f32 / out_t rows [B*K][C], C = 256. */
int gim_sp_sample_desc(const void* dense, const float* kpts, float* out_f32, void* out_t, int B, int K,
                       int h, int w, int C, int ld, int ld_f32, int ld_t, int cell, int dtype,
                       gim_stream_t stream);

/* normalize_keypoints + LearnableFourierPositionalEncoding -- replaces lightglue.py:21-33,47-61.
 * kpts [B,K,2], size_wh [B,2] = (w, h), Wr [32,2]; enc [B*K][64] = cos(proj)[32] | sin(proj)[32]
 * (the reference's [2,B,1,K,64] tensor is this table with every entry repeated twice). */
int gim_lg_posenc(const float* kpts, const float* size_wh, const float* Wr, float* enc, int B, int K,
                  gim_stream_t stream);

/* apply_cached_rotary_emb -- replaces lightglue.py:36-44,150-151: in place on columns [0, ncols) of x
 * (ncols % 64 == 0; heads of 64, the same table for every head). */
int gim_lg_rotary(void* x, const float* enc, int rows, int ncols, int ld, int dtype, gim_stream_t stream);

/* V -> V^T per sequence for gim_sdpa: dst[nb][C][Sp], key index contiguous, zero for key >= S (Sp = S
 * rounded up to a multiple of 64). */
int gim_lg_transpose(const void* src, void* dst, int nb, int S, int Sp, int C, int ld, int dtype,
                     gim_stream_t stream);

/* softmax(q k^T / sqrt(D)) v -- replaces Attention.forward (lightglue.py:104-118) and the bidirectional
 * cross attention of CrossBlock.forward (:196-207; the two directions are two calls).
 * q rows [nb*L][ldq], k rows [nb*S][ldk] (head h at columns h*D), vt from gim_lg_transpose, out rows
 * [nb*L][ldo].  Sequence s attends to the keys/values of sequence (s + kv_shift) % nb (self: 0;
 * cross with image0/image1 stacked as [2B]: kv_shift = B).  D = 64.  dtype f32: exact-fp32 MFMA. */
int gim_sdpa(const void* q, const void* k, const void* vt, void* out, int nb, int H, int L, int S, int Sp,
             int D, int ldq, int ldk, int ldo, int kv_shift, int dtype, int out_dtype, gim_stream_t stream);

/* LayerNorm (+ GELU) -- replaces ffn[1], ffn[2] of SelfBlock / CrossBlock (lightglue.py:135-139). */
int gim_layernorm_act(const float* x, const float* gamma, const float* beta, void* out, int rows, int C,
                      int ldx, int ldo, int act, int out_dtype, float eps, gim_stream_t stream);

/* fp32 rows -> dtype rows (residual stream -> GEMM operand copy). */
int gim_cast_rows(const float* src, void* dst, int rows, int C, int ld_src, int ld_dst, int dtype,
                  gim_stream_t stream);

/* MatchAssignment + sigmoid_log_double_softmax + filter_matches, fused -- replaces lightglue.py:226-300. */
typedef struct gim_lg_assign_args {
    const float* desc0;   /* [B*M][ld_desc] final descriptors (matchability input) */
    const float* desc1;   /* [B*N][ld_desc] */
    const float* md0;     /* [B][M][C] final_proj(desc0), unscaled */
    const float* md1;     /* [B][N][C] */
    const float* match_w; /* matchability.weight [C] */
    const float* match_b; /* matchability.bias [1] */
    void* ws;             /* >= gim_lg_assign_ws_bytes() */
    int64_t* matches0;    /* [B][M], -1 = unmatched */
    int64_t* matches1;    /* [B][N] */
    float* mscores0;      /* [B][M] */
    float* mscores1;      /* [B][N] */
    int32_t* pos;         /* [B][M] position of row i in its pair's match list, -1 = unmatched */
    int32_t* count;       /* [B] matches per pair */
    int B, M, N, C;
    int ld_desc;
    float threshold;      /* filter_threshold (0.1) */
} gim_lg_assign_args;
int64_t gim_lg_assign_ws_bytes(int B, int M, int N, int C);
int gim_lg_assign(const gim_lg_assign_args* a, gim_stream_t stream);
/* `pred["log_assignment"]` [B][M+1][N+1] on demand (lightglue.py:226-238). */
int gim_lg_log_assignment(const gim_lg_assign_args* a, float* out, gim_stream_t stream);
/* `matches` / `scores` lists (lightglue.py:497-506) packed over the batch in torch.where order, and the
 * caller-side adapter of trainer/lightning.py:176-183 / demo.py:503-510 (mkpts = keypoints * scale,
 * m_bids); mkpts0 == NULL skips the adapter outputs. */
int gim_lg_emit_matches(const int64_t* matches0, const float* mscores0, const int32_t* pos,
                        const int32_t* count, const float* kpts0, const float* kpts1, const float* scale0,
                        const float* scale1, int64_t* matches, float* scores, float* mkpts0, float* mkpts1,
                        int64_t* m_bids, int B, int M, int N, gim_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Output hand-out.  gim_copy_segments: up to GIM_MAX_COPY_SEGS independent device byte ranges moved (src == NULL: zero
 * filled) in ONE launch -- the private copies of b_ids / i_ids / j_ids / m_bids / mkpts0_c / mkpts1_c / mconf and the
 * all-false gt_mask that `data` receives after coarse matching (coarse_matching.py:236-259: tensor construction only).
 * gim_pack_matches: reporting rows [pair_id, x0, y0, x1, y1, conf] (24 B per match; the packed form of the per-pair rows of
 * trainer/lightning.py:258-270); pair_id = pair_ids[m_bids[m]] (device int64, one per batch element) or, with
 * pair_ids == NULL, pid_base + m_bids[m]. */
enum { GIM_MAX_COPY_SEGS = 12 };
typedef struct gim_copy_segs {
    const void* src[12];
    void* dst[12];
    int64_t bytes[12];
    int n;
} gim_copy_segs;
int gim_copy_segments(const gim_copy_segs* segs, gim_stream_t stream);
int gim_pack_matches(const int64_t* m_bids, const float* mkpts0, const float* mkpts1, const float* mconf,
                     const int64_t* pair_ids, int64_t pid_base, float* out, int M, gim_stream_t stream);

/* ======================================================================================================
 * gim_dkm path (kernels D1-D9).  Convolutions / 1x1 projections / the cosine-kernel and
 * posterior-mean products run on gim_conv2d_bn_act (runtime "weights" = row buffers in [N][K] layout).
 * ====================================================================================================== */

/* torchvision resnet50 `maxpool` (kernel 3, stride 2, padding 1) on NHWC rows -- encoders.py:51. */
int gim_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int ldx, int ldy, int dtype,
                     gim_stream_t stream);
/* F.interpolate(mode='bilinear', align_corners=False) on NHWC rows -- dkm.py:420-425,468-479,518-529,668-701. */
int gim_resize_bilinear(const void* x, void* y, int B, int h, int w, int Ho, int Wo, int C, int ldx, int ldy,
                        int dtype, int out_dtype, gim_stream_t stream);
/* The same resize for the NCHW fp32 input images, written as NHWC rows with cpad channels (images b_off..)
 * -- dkm.py:668-669,700-701. */
int gim_resize_image(const float* x, void* y, int B, int C, int h, int w, int Ho, int Wo, int cpad, int b_off,
                     int out_dtype, gim_stream_t stream);
/* F.grid_sample(bilinear, zeros padding, align_corners=False): feat [B,h,w,C] at grid [B,Ho,Wo,2] (x, y)
 * -- dkm.py:89. */
int gim_grid_sample(const void* feat, const float* grid_xy, void* out, int B, int h, int w, int Ho, int Wo, int C,
                    int ldf, int ldo, int dtype, gim_stream_t stream);
/* disp_emb(flow - query_coords): 1x1 conv 2 -> E on the displacement field -- dkm.py:91-101. wgt [E][2]. */
int gim_dkm_disp_emb(const float* flow, const float* wgt, const float* bias, void* out, int B, int h, int w, int E,
                     int ldo, int out_dtype, gim_stream_t stream);
/* local_correlation(x, y, local_radius=r, flow) -- utils/local_correlation.py:5-40: out[b,y,x,k], k = (2r+1)^2
 * bilinear window taps of f1 around the flow target, dotted with f0, / sqrt(C).  r <= 7. */
int gim_local_corr(const void* f0, const void* f1, const float* flow, void* out, int B, int h, int w, int C, int r,
                   int ld0, int ld1, int ldo, int dtype, int out_dtype, gim_stream_t stream);
/* ConvRefiner.create_block, first half (dw=True): depthwise 5x5 conv (Cout = mult * Cin) + eval BatchNorm + ReLU
 * -- dkm.py:58-73.  wgt [25][cpad], scale / shift [cpad] fp32 (conv bias and BN folded), zero padded. */
int gim_dwconv5x5_bn_relu(const void* x, const float* wgt, const float* scale, const float* shift, void* y, int B,
                          int H, int W, int Cin, int Cout, int cpad, int ldx, int ldy, int dtype, gim_stream_t stream);
/* The whole ConvRefiner block of DKM / RoMa (dkm.py:58-73, create_block: depthwise 5x5 + BatchNorm + ReLU + 1x1 convolution with bias) in ONE launch,
 * 16-bit operands, for the refiners whose channel count fits one channel chunk: cs = 144 stored channels (the scale-2 refiner) or 24 / 32 (scale 1).
 * The depthwise output goes to an LDS tile and is the B operand of the 1x1's MFMAs; the intermediate tensor is never written.  x [B,H,W,ldx],
 * y [B,H,W,ldy] rows of cs stored channels; wgt [25][cs], scale / shift [cs] fp32 (BatchNorm folded, as gim_dwconv5x5_bn_relu); pw_w [NP][KP] 16-bit
 * (rows = output channels padded to NP = 160 / 32, K = input channels padded to KP = 144 / 32, zero padding); pw_b [NP] fp32.  dtype GIM_BF16 / GIM_F16. */
int gim_dwconv5x5_pw(const void* x, const float* wgt, const float* scale, const float* shift, const void* pw_w, const float* pw_b, void* y,
                     int B, int H, int W, int cs, int ldx, int ldy, int dtype, gim_stream_t stream);
/* CosKernel pieces -- dkm.py:135-144: row L2 norms, and K = exp((dot / (nx ny + eps) - 1) / T) in place on the
 * dot-product matrix (diag_add = sigma_noise on the diagonal, dkm.py:352). */
int gim_row_norms(const void* x, float* out, int rows, int C, int ld, int dtype, gim_stream_t stream);
int gim_cos_kernel_finish(float* k, const float* nx, const float* ny, int B, int n, int m, int ld, float T, float eps,
                          float diag_add, gim_stream_t stream);
/* (K_yy + sigma I)^-1 f of GP.forward -- dkm.py:352-359 -- as a blocked fp64 Cholesky solve (the reference inverts
 * with LU in fp32).  K [B][n][ldk] fp32 with sigma on the diagonal, F [B][n][nrhs] fp32;
 * Xt [B][nrhs][npad] fp32 = (K^-1 F)^T zero padded: the [N][K] operand layout of gim_conv2d_bn_act. */
int64_t gim_gp_solve_ws_bytes(int B, int n, int nrhs);
int gim_gp_solve(const float* K, const float* F, float* Xt, void* ws, int B, int n, int ldk, int nrhs, int npad,
                 gim_stream_t stream);
/* The same posterior evaluated ENTIRELY in fp64 from fp32 feature rows (the dense matchers' parity mode): cosine-kernel entries,
 * Cholesky and both products on the fp64 MFMA.  X (queries) / Y (supports): [B][n][ldx] fp32, d features per row; F [n][nrhs] fp32
 * (shared by all B directions); mu [B][n][ld_mu] fp32 = K_xy (K_yy + sigma I)^-1 F with K = exp((cos - 1) / T), cos = x.y / (|x||y| + eps)
 * (dkm.py:135-144, 340-370; roma.py:94-136).  The system's condition number (~2e4) turns the 1e-7 rounding of fp32 kernel entries
 * into ~1e-4 of mu; this entry point removes that term from the engine's side. */
int64_t gim_gp_posterior_f64_ws_bytes(int B, int n, int d, int nrhs);
int gim_gp_posterior_f64(const float* X, const float* Y, const float* F, float* mu, void* ws, int B, int n, int d, int ldx,
                         int nrhs, int ld_mu, float T, float eps, float sigma, gim_stream_t stream);
/* CAB -- dkm.py:160-168: global average pool of NHWC rows into out[b][c_off + c]; out = sigmoid(g) * x2 + x1. */
int gim_global_avgpool(const void* x, float* out, int B, int HW, int C, int ld, int ldo, int c_off, int dtype,
                       gim_stream_t stream);
int gim_cab_scale_add(const float* g, const void* x1, const void* x2, void* out, int B, int HW, int C, int ldg, int ld1,
                      int ld2, int ldo, int dtype, gim_stream_t stream);
/* Decoder.forward flow / certainty update -- dkm.py:505-514, roma.py:318-331.  cert_init bit 0: certainty = delta
 * (instead of +=); bit 1: d rows = [dx, dy, delta certainty] (RoMa, roma.py:579) instead of [delta certainty, dx, dy]. */
int gim_dkm_flow_update(float* flow, float* cert, const void* d, int64_t npix, int ldd, float sx, float sy,
                        int cert_init, int dtype, gim_stream_t stream);
/* get_placeholder_flow -- dkm.py:437-448. */
int gim_dkm_grid_coords(float* flow, int B, int h, int w, gim_stream_t stream);
/* match() tail of one symmetric pair -- dkm.py:693-741: flow / certainty / low-res certainty maps [H,W,.] of the two
 * directions (0 = query -> support) -> warp [H,2W,4], certainty [H,2W]; black masks from gim_dkm_black_mask (:726-729). */
int gim_dkm_match_post(const float* flow0, const float* flow1, const float* cert0, const float* cert1, const float* low0,
                       const float* low1, const uint8_t* black0, const uint8_t* black1, float* warp, float* certainty,
                       int H, int W, gim_stream_t stream);
int gim_dkm_black_mask(const float* im, uint8_t* mask, int h, int w, int Ho, int Wo, gim_stream_t stream);
/* kde(x, std) -- utils/kde.py:17-26: density[i] = sum_j exp(-cdist(x_i, x_j)^2 / (2 std^2)), x [n,4] fp32 (the
 * reference materialises the n x n distance matrix: 1.6 GB at n = 20000). */
int gim_kde(const float* x, float* density, int n, float std, gim_stream_t stream);  /* std < 0: |std| on fp16-rounded x (roma.py:1018-1023) */
/* cls_to_flow_refine + the certainty channel of TransformerDecoder's output -- roma.py:1092-1121, 1013-1014: logits rows
 * [npix][ld] fp32 = ncls (= res^2) anchor classes then the certainty logit -> flow [npix,2], cert [npix]. */
int gim_cls_to_flow(const float* logits, float* flow, float* cert, int npix, int ncls, int ld, gim_stream_t stream);
/* torch.multinomial(w, k, replacement=False) of RegressionMatcher.sample -- dkm.py:603-605,617-619: k distinct indices
 * drawn with probability proportional to w (exponential clocks + radix-select top-k), as an unordered set; reproducible
 * from `seed`.  Needs at least k positive weights. */
/* Caller-side dense adapter -- trainer/lightning.py:141-144 / demo.py:438-443: normalised [n,4] matches -> pixel
 * coordinates in the two images, kpts = size * (x + 1) / 2. */
int gim_dense_to_pixels(const float* matches, float* kpts0, float* kpts1, int n, float w0, float h0, float w1, float h1,
                        gim_stream_t stream);
int64_t gim_weighted_sample_ws_bytes(int n);
int gim_weighted_sample(const float* w, int64_t* out, void* ws, int n, int k, uint32_t seed, gim_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GIM_HIP_H */

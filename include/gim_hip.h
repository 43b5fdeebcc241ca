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
 *   - `dtype` arguments: GIM_F32 (exact-parity mode, fp32 MFMA) or GIM_BF16 (throughput mode);
 *   - activations are NHWC ("pixel rows"): row m = ((b*H + y)*W + x), `ld*` = row stride in ELEMENTS.
 */
#ifndef GIM_HIP_H
#define GIM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gim_stream_t; /* hipStream_t */

enum { GIM_F32 = 0, GIM_BF16 = 1 };
enum { GIM_ACT_NONE = 0, GIM_ACT_RELU = 1, GIM_ACT_LEAKY = 2, GIM_ACT_ELU1 = 3 /* elu(x)+1 */ };
enum { GIM_OK = 0, GIM_ERR_INVALID = -1, GIM_ERR_LAUNCH = -2, GIM_ERR_UNSUPPORTED = -3 };

int gim_version(void);
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
    int use_lds_dma;    /* 1: buffer_load ... lds staging (default); 0: register staging */
} gim_conv_args;
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
 * (K, V rows with mask 0 do not contribute; Q rows with mask 0 give a zero message). */
int64_t gim_linear_attention_ws_bytes(int nb, int S, int H, int D);
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
 * (transformer.py:52,56,58).  Writes fp32 (out_f32, may be NULL) and/or `dtype` copy (out_t). */
int gim_layernorm_residual(const float* x, const float* gamma, const float* beta, const float* res,
                           float* out_f32, void* out_t, int rows, int C, int ldx, int ldres,
                           int ld_f32, int ld_t, int dtype, float eps, gim_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Coarse matching (utils/coarse_matching.py:88-259): dual-softmax + threshold + border + mutual-NN
 * + ordered compaction, fused so that the [N,L,S] confidence matrix is never written.
 * feat0 [N*L, C] / feat1 [N*S, C] fp32 rows (row stride C).  Outputs in torch.where order
 * (ascending b, then i): b_ids/i_ids/j_ids int64, mconf fp32, mkpts0_c/mkpts1_c fp32 [cap,2].
 * `count` (device int32[1 + N]): count[0] = M, count[1+b] = matches of pair b.
 * scale0/scale1: NULL or fp32 [N,2] per-pair (w,h) scales (coarse_matching.py:237-245). */
typedef struct gim_coarse_args {
    const float* feat0;
    const float* feat1;
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
} gim_coarse_args;
int64_t gim_coarse_match_ws_bytes(int N, int L, int S);
int gim_coarse_match(const gim_coarse_args* a, gim_stream_t stream);
/* materialise conf_matrix [N,L,S] fp32 (data['conf_matrix'], coarse_matching.py:144) on demand;
 * needs the workspace of the preceding gim_coarse_match call (row/column softmax statistics). */
int gim_coarse_conf_matrix(const gim_coarse_args* a, float* conf, gim_stream_t stream);

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

#ifdef __cplusplus
}
#endif
#endif /* GIM_HIP_H */

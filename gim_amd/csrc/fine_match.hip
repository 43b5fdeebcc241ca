// Fine level of gim_loftr for gfx950.
//
// gim_fine_gather: networks/loftr/submodules/fine_preprocess.py:40-47 -- the reference unfolds *every*
//   5x5 window of both fine maps ([N, 3200, 4800], 61 MB per image) and then picks the M matched ones;
//   here only the M selected windows are gathered (zero padded at the map border like F.unfold
//   padding=W//2) straight from the NHWC fine map: 25 rows x C channels per match and side.
// gim_fine_match:  networks/loftr/utils/fine_matching.py:43-74 -- centre-row correlation, softmax over
//   the WW window cells, DSNT expectation on the normalised grid, std, final sub-pixel coordinates.
#include "gim_common.h"

namespace {

template <bool BF16>
__global__ void fine_gather_kernel(const void* __restrict__ feat0, const void* __restrict__ feat1,
                                   const int64_t* __restrict__ b_ids, const int64_t* __restrict__ i_ids,
                                   const int64_t* __restrict__ j_ids, float* __restrict__ out_f32,
                                   void* __restrict__ out_t, int M, int hf0, int wf0, int hf1, int wf1, int C4,
                                   int ldf, int w0c, int w1c, int stride, int W, int ld_f32, int ld_t) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int WW = W * W;
    const size_t total = (size_t)2 * M * WW * C4;
    if (idx >= total) return;
    const int cq = (int)(idx % C4);
    const size_t row = idx / C4;  // side*M*WW + m*WW + ww
    const int ww = (int)(row % WW);
    const size_t sm = row / WW;
    const int side = sm >= (size_t)M ? 1 : 0;
    const int m = (int)(sm - (size_t)side * M);
    const int b = (int)b_ids[m];
    const int cell = (int)(side ? j_ids[m] : i_ids[m]);
    const int wc = side ? w1c : w0c;
    const int cy = cell / wc, cx = cell - cy * wc;
    const int y = cy * stride - W / 2 + ww / W, x = cx * stride - W / 2 + ww % W;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int hf = side ? hf1 : hf0, wf = side ? wf1 : wf0;
    if (y >= 0 && y < hf && x >= 0 && x < wf)
        v = ElemIO<BF16>::ld4(side ? feat1 : feat0, (((size_t)b * hf + y) * wf + x) * ldf + cq * 4);
    if (out_f32) *(float4*)(out_f32 + row * ld_f32 + cq * 4) = v;
    if (out_t) ElemIO<BF16>::st4(out_t, row * ld_t + cq * 4, v);
}

// one wave per match; lane r < WW owns window cell r of image1
__global__ void __launch_bounds__(256)
fine_match_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ mkpts1_c,
                  const int64_t* __restrict__ b_ids, const float* __restrict__ scale1, float* __restrict__ expec_f,
                  float* __restrict__ mkpts1_f, int M, int W, int C, int ld, float scale, int has_scale0) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int WW = W * W;
    const float* q = f0 + ((size_t)m * WW + WW / 2) * ld;  // feat_f0[:, WW//2, :]
    float s = -INFINITY;
    if (lane < WW) {
        const float* kr = f1 + ((size_t)m * WW + lane) * ld;
        float acc = 0.f;
        for (int c = 0; c < C; c += 4) {
            const float4 a = *(const float4*)(q + c), b = *(const float4*)(kr + c);
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
            acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
        s = (1.0f / sqrtf((float)C)) * acc;  // softmax_temp * sim_matrix
    }
    const float mx = wave_max(s);
    const float e = lane < WW ? expf(s - mx) : 0.f;
    const float heat = e / wave_sum(e);
    const float step = W > 1 ? 2.0f / (float)(W - 1) : 0.f;
    const float gx = lane < WW ? -1.0f + step * (float)(lane % W) : 0.f;
    const float gy = lane < WW ? -1.0f + step * (float)(lane / W) : 0.f;
    const float cx = wave_sum(gx * heat), cy = wave_sum(gy * heat);
    const float vx = wave_sum(gx * gx * heat) - cx * cx, vy = wave_sum(gy * gy * heat) - cy * cy;
    if (lane == 0) {
        const float sd = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f));
        expec_f[3 * m + 0] = cx; expec_f[3 * m + 1] = cy; expec_f[3 * m + 2] = sd;
        float s1x = scale, s1y = scale;
        if (has_scale0) {  // quirk preserved: keyed on scale0, multiplies scale1 (fine_matching.py:68)
            const int b = (int)b_ids[m];
            s1x = scale * scale1[2 * b + 0];
            s1y = scale * scale1[2 * b + 1];
        }
        const float half = (float)(W / 2);
        mkpts1_f[2 * m + 0] = mkpts1_c[2 * m + 0] + cx * half * s1x;
        mkpts1_f[2 * m + 1] = mkpts1_c[2 * m + 1] + cy * half * s1y;
    }
}

}  // namespace

#if !GIM_HALF_KIND
extern "C" int gim_fine_gather_f16(const void* feat_f0, const void* feat_f1, const int64_t* b_ids, const int64_t* i_ids,
                               const int64_t* j_ids, float* out_f32, void* out_t, int M, int hf0, int wf0, int hf1,
                               int wf1, int C, int ldf, int w0c, int w1c, int stride, int W, int ld_f32, int ld_t,
                               int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_fine_gather)(const void* feat_f0, const void* feat_f1, const int64_t* b_ids, const int64_t* i_ids,
                               const int64_t* j_ids, float* out_f32, void* out_t, int M, int hf0, int wf0, int hf1,
                               int wf1, int C, int ldf, int w0c, int w1c, int stride, int W, int ld_f32, int ld_t,
                               int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_fine_gather_f16(feat_f0, feat_f1, b_ids, i_ids, j_ids, out_f32, out_t, M, hf0, wf0, hf1, wf1, C, ldf, w0c, w1c, stride, W, ld_f32, ld_t, dtype, stream);   // the fp16 objects of this file
#endif
    if (M == 0) return GIM_OK;
    GIM_REQUIRE(feat_f0 && feat_f1 && b_ids && i_ids && j_ids && (out_f32 || out_t), "fine_gather: NULL pointer");
    GIM_REQUIRE(M > 0 && hf0 > 0 && wf0 > 0 && hf1 > 0 && wf1 > 0 && C > 0 && C % 4 == 0 && W > 0 && (W & 1) && stride > 0, "fine_gather: bad sizes");
    GIM_REQUIRE(ldf % 4 == 0 && ld_f32 % 4 == 0 && ld_t % 4 == 0, "fine_gather: ld alignment");
    const size_t total = (size_t)2 * M * W * W * (C / 4);
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GIM_H16)
        hipLaunchKernelGGL(fine_gather_kernel<true>, dim3(grid), dim3(256), 0, s, feat_f0, feat_f1, b_ids, i_ids, j_ids, out_f32, out_t, M, hf0, wf0, hf1, wf1, C / 4, ldf, w0c, w1c, stride, W, ld_f32, ld_t);
    else
        hipLaunchKernelGGL(fine_gather_kernel<false>, dim3(grid), dim3(256), 0, s, feat_f0, feat_f1, b_ids, i_ids, j_ids, out_f32, out_t, M, hf0, wf0, hf1, wf1, C / 4, ldf, w0c, w1c, stride, W, ld_f32, ld_t);
    return gim_check_launch("fine_gather");
}

extern "C" int GIM_FN(gim_fine_match)(const float* f0, const float* f1, const float* mkpts1_c, const int64_t* b_ids,
                              const float* scale1, float* expec_f, float* mkpts1_f, int M, int WW, int C, int ld,
                              float scale, int has_scale0, gim_stream_t stream) {
    if (M == 0) return GIM_OK;
    GIM_REQUIRE(f0 && f1 && mkpts1_c && expec_f && mkpts1_f, "fine_match: NULL pointer");
    int W = 1;
    while (W * W < WW) ++W;
    GIM_REQUIRE(W * W == WW && WW <= 64, "fine_match: WW=%d must be a square <= 64", WW);
    GIM_REQUIRE(C % 4 == 0 && ld % 4 == 0, "fine_match: C/ld alignment");
    GIM_REQUIRE(!has_scale0 || (scale1 && b_ids), "fine_match: scale1/b_ids required when has_scale0");
    hipLaunchKernelGGL(fine_match_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f0, f1,
                       mkpts1_c, b_ids, scale1, expec_f, mkpts1_f, M, W, C, ld, scale, has_scale0);
    return gim_check_launch("fine_match");
}

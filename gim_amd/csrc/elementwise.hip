// Memory-bound glue kernels of the gim_loftr path (gfx950): layout conversion, FPN upsample-add,
// positional encoding add, LayerNorm(+residual).  All are coalesced 8/16-byte-per-lane streams.
#include "gim_common.h"

namespace {

// [B,C,H,W] fp32 -> NHWC rows, channels padded with zeros to cpad.  One thread per pixel.
template <bool BF16>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, void* __restrict__ dst, int B, int C,
                                    int HW, int cpad, int ld, int b_off) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)B * HW) return;
    const size_t b = p / HW, r = p - b * HW;
    const size_t row = ((size_t)(b + b_off) * HW + r) * ld;
    for (int c0 = 0; c0 < cpad; c0 += 4) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (c0 + k < C) ? src[(b * C + c0 + k) * HW + r] : 0.f;
        ElemIO<BF16>::st4(dst, row + c0, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// [B,C,H,W] fp32 -> NHWC rows carrying every channel as a 16-bit hi + lo pair, laid out for the split-operand first convolution
// (packing.pack_conv_split): channels [hi(C) | lo(C) | hi(C) | 0 ...] against weights [w_hi | w_hi | w_lo]: the MFMA sums
// x_hi w_hi + x_lo w_hi + x_hi w_lo = x w up to 2^-22 -- the image and the 7x7 filters rounded to one 16-bit value are half of the
// 16-bit modes' error (profiles/r04_precision_sweep.txt).  One thread per pixel.
__global__ void nchw_to_nhwc_split_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int B, int C, int HW, int ld, int b_off, int third) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)B * HW) return;
    const size_t b = p / HW, r = p - b * HW;
    unsigned short* row = dst + ((size_t)(b + b_off) * HW + r) * ld;
    for (int c = (third ? 3 : 2) * C; c < ld; ++c) row[c] = 0;
    for (int c = 0; c < C; ++c) {
        const float v = src[(b * C + c) * HW + r];
        const unsigned short hi = f32_to_h16(v);
        const unsigned short lo = f32_to_h16(v - h16_to_f32(hi));
        row[c] = hi; row[C + c] = lo;
        if (third) row[2 * C + c] = hi;
    }
}
// the same for C and ld known at compile time (RGB -> 16 stored channels, gray -> 8): the row is built in registers and leaves in
// 16-byte stores (the generic kernel writes 2 bytes at a time: 0.1 ms per batch-8 step)
template <int C, int LD, bool THIRD>
__global__ void nchw_to_nhwc_split_fixed_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int B, int HW, int b_off) {
    static_assert((THIRD ? 3 : 2) * C <= LD && LD % 8 == 0, "split layout");
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)B * HW) return;
    const size_t b = p / HW, r = p - b * HW;
    unsigned short h[LD];
#pragma unroll
    for (int c = 0; c < LD; ++c) h[c] = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float v = src[(b * C + c) * HW + r];
        const unsigned short hi = f32_to_h16(v);
        h[c] = hi; h[C + c] = f32_to_h16(v - h16_to_f32(hi));
        if (THIRD) h[2 * C + c] = hi;
    }
    uint4* row = dst + ((size_t)(b + b_off) * HW + r) * (LD / 8);
#pragma unroll
    for (int q = 0; q < LD / 8; ++q)
        row[q] = make_uint4(h[8 * q] | (unsigned)h[8 * q + 1] << 16, h[8 * q + 2] | (unsigned)h[8 * q + 3] << 16,
                            h[8 * q + 4] | (unsigned)h[8 * q + 5] << 16, h[8 * q + 6] | (unsigned)h[8 * q + 7] << 16);
}

// NHWC rows -> [B,C,H,W] fp32.  LDS-tiled transpose: 64 pixels x 64 channels per block.
template <bool BF16>
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ src, float* __restrict__ dst, int C, int HW, int ld) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 64 x 4
    for (int i = ty; i < 64; i += 4) {
        const int p = p0 + i, c = c0 + tx;
        tile[i][tx] = (p < HW && c < C) ? ElemIO<BF16>::ld(src, ((size_t)b * HW + p) * ld + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, p = p0 + tx;
        if (p < HW && c < C) dst[((size_t)b * C + c) * HW + p] = tile[tx][i];
    }
}

// y[b, Y, X, :] += bilinear(x)[b, Y, X, :], scale 2, align_corners=True.
// Weights exactly as ATen's area_pixel_compute_source_index(align_corners=true): src = dst*(in-1)/(out-1).
// One thread = one 16-byte channel group (8 bf16 / 4 fp32) of one output pixel.
template <bool BF16>
__global__ void upsample2x_add_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int h, int w,
                                      int CG, int ldx, int ldy) {
    constexpr int G = BF16 ? 8 : 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int H2 = 2 * h, W2 = 2 * w;
    const size_t total = (size_t)B * H2 * W2 * CG;
    if (idx >= total) return;
    const int cg = (int)(idx % CG);
    const size_t pix = idx / CG;
    const int X = (int)(pix % W2);
    const int Y = (int)((pix / W2) % H2);
    const int b = (int)(pix / ((size_t)W2 * H2));
    const float sy = H2 > 1 ? (float)(h - 1) / (float)(H2 - 1) : 0.f;
    const float sx = W2 > 1 ? (float)(w - 1) / (float)(W2 - 1) : 0.f;
    const float fy = sy * Y, fx = sx * X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly1 = fy - y0, ly0 = 1.f - ly1, lx1 = fx - x0, lx0 = 1.f - lx1;
    const size_t base = (size_t)b * h * w;
    const size_t o00 = (base + (size_t)y0 * w + x0) * ldx + cg * G, o01 = (base + (size_t)y0 * w + x1) * ldx + cg * G;
    const size_t o10 = (base + (size_t)y1 * w + x0) * ldx + cg * G, o11 = (base + (size_t)y1 * w + x1) * ldx + cg * G;
    const size_t yo = pix * ldy + cg * G;
#pragma unroll
    for (int e = 0; e < G; e += 4) {
        const float4 a = ElemIO<BF16>::ld4(x, o00 + e), bq = ElemIO<BF16>::ld4(x, o01 + e);
        const float4 c = ElemIO<BF16>::ld4(x, o10 + e), d = ElemIO<BF16>::ld4(x, o11 + e);
        float4 o = ElemIO<BF16>::ld4(y, yo + e);
        o.x += ly0 * (lx0 * a.x + lx1 * bq.x) + ly1 * (lx0 * c.x + lx1 * d.x);
        o.y += ly0 * (lx0 * a.y + lx1 * bq.y) + ly1 * (lx0 * c.y + lx1 * d.y);
        o.z += ly0 * (lx0 * a.z + lx1 * bq.z) + ly1 * (lx0 * c.z + lx1 * d.z);
        o.w += ly0 * (lx0 * a.w + lx1 * bq.w) + ly1 * (lx0 * c.w + lx1 * d.w);
        ElemIO<BF16>::st4(y, yo + e, o);
    }
}

template <bool BF16>
__global__ void posenc_add_kernel(const void* __restrict__ x, const float* __restrict__ pe, float* __restrict__ out_f32,
                                  void* __restrict__ out_t, size_t rows, int hw, int C4, int ldx, int ld_f32, int ld_t) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C4) return;
    const int cq = (int)(idx % C4);
    const size_t m = idx / C4;
    float4 v = ElemIO<BF16>::ld4(x, m * ldx + cq * 4);
    const float4 p = *(const float4*)(pe + (m % hw) * (size_t)(C4 * 4) + cq * 4);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    if (out_f32) *(float4*)(out_f32 + m * ld_f32 + cq * 4) = v;
    if (out_t) ElemIO<BF16>::st4(out_t, m * ld_t + cq * 4, v);
}

// One wave per row, C <= 512, C % 4 == 0.
template <bool BF16, bool XBF16>
__global__ void __launch_bounds__(256)
layernorm_kernel(const void* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                 const float* __restrict__ res, float* __restrict__ out_f32, void* __restrict__ out_t, int rows,
                 int C, int ldx, int ldres, int ld_f32, int ld_t, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[2];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane * 4 + k * 256;
        v[k] = c < C ? ElemIO<XBF16>::ld4(x, (size_t)row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane * 4 + k * 256;
        if (c < C) {
            const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float var = wave_sum(q) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane * 4 + k * 256;
        if (c >= C) continue;
        const float4 g = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
        float4 o;
        o.x = (v[k].x - mean) * rstd * g.x + bt.x;
        o.y = (v[k].y - mean) * rstd * g.y + bt.y;
        o.z = (v[k].z - mean) * rstd * g.z + bt.z;
        o.w = (v[k].w - mean) * rstd * g.w + bt.w;
        if (res) {
            const float4 r = *(const float4*)(res + (size_t)row * ldres + c);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (out_f32) *(float4*)(out_f32 + (size_t)row * ld_f32 + c) = o;
        if (out_t) ElemIO<BF16>::st4(out_t, (size_t)row * ld_t + c, o);
    }
}

inline unsigned nblocks(size_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

#if !GIM_HALF_KIND
extern "C" int gim_nchw_to_nhwc_f16(const float* src, void* dst, int B, int C, int H, int W, int cpad, int ld,
                                int b_off, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_nchw_to_nhwc)(const float* src, void* dst, int B, int C, int H, int W, int cpad, int ld,
                                int b_off, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_nchw_to_nhwc_f16(src, dst, B, C, H, W, cpad, ld, b_off, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad args");
    GIM_REQUIRE(cpad % 4 == 0 && cpad >= C && ld >= cpad && ld % 4 == 0, "nchw_to_nhwc: cpad=%d ld=%d", cpad, ld);
    const size_t n = (size_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GIM_H16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<true>, dim3(nblocks(n, 256)), dim3(256), 0, s, src, dst, B, C, H * W, cpad, ld, b_off);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<false>, dim3(nblocks(n, 256)), dim3(256), 0, s, src, dst, B, C, H * W, cpad, ld, b_off);
    return gim_check_launch("nchw_to_nhwc");
}

#if !GIM_HALF_KIND
extern "C" int gim_nchw_to_nhwc_split_f16(const float* src, void* dst, int B, int C, int H, int W, int ld, int b_off, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_nchw_to_nhwc_split)(const float* src, void* dst, int B, int C, int H, int W, int ld, int b_off, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_nchw_to_nhwc_split_f16(src, dst, B, C, H, W, ld, b_off, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc_split: bad args");
    GIM_REQUIRE(dtype == GIM_H16, "nchw_to_nhwc_split: 16-bit output only (dtype %d)", dtype);
    // ld >= 3 C: [hi | lo | hi | 0 ...] (the implicit-GEMM form of the split convolution); 2 C <= ld < 3 C: [hi | lo | 0 ...] (gim_stem7x7)
    GIM_REQUIRE(ld >= 2 * C && ld % 8 == 0, "nchw_to_nhwc_split: ld=%d must hold 2 or 3 x %d channels in 16-byte groups", ld, C);
    const int third = ld >= 3 * C;
    const size_t n = (size_t)B * H * W;
    if (C == 3 && ld == 16)
        hipLaunchKernelGGL((nchw_to_nhwc_split_fixed_kernel<3, 16, true>), dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, src, (uint4*)dst, B, H * W, b_off);
    else if (C == 3 && ld == 8)
        hipLaunchKernelGGL((nchw_to_nhwc_split_fixed_kernel<3, 8, false>), dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, src, (uint4*)dst, B, H * W, b_off);
    else if (C == 1 && ld == 8)
        hipLaunchKernelGGL((nchw_to_nhwc_split_fixed_kernel<1, 8, true>), dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, src, (uint4*)dst, B, H * W, b_off);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_split_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, B, C, H * W, ld, b_off, third);
    return gim_check_launch("nchw_to_nhwc_split");
}

#if !GIM_HALF_KIND
extern "C" int gim_nhwc_to_nchw_f16(const void* src, float* dst, int B, int C, int H, int W, int ld, int dtype,
                                gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_nhwc_to_nchw)(const void* src, float* dst, int B, int C, int H, int W, int ld, int dtype,
                                gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_nhwc_to_nchw_f16(src, dst, B, C, H, W, ld, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "nhwc_to_nchw: bad args");
    const int HW = H * W;
    dim3 grid((HW + 63) / 64, (C + 63) / 64, B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GIM_H16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<true>, grid, dim3(256), 0, s, src, dst, C, HW, ld);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<false>, grid, dim3(256), 0, s, src, dst, C, HW, ld);
    return gim_check_launch("nhwc_to_nchw");
}

#if !GIM_HALF_KIND
extern "C" int gim_upsample2x_add_f16(const void* x, void* y, int B, int h, int w, int C, int ldx, int ldy, int dtype,
                                  gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_upsample2x_add)(const void* x, void* y, int B, int h, int w, int C, int ldx, int ldy, int dtype,
                                  gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_upsample2x_add_f16(x, y, B, h, w, C, ldx, ldy, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && y && B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "upsample2x_add: bad args (C=%d)", C);
    GIM_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "upsample2x_add: ld must be a multiple of 4");
    const int G = dtype == GIM_H16 ? 8 : 4;
    GIM_REQUIRE(C % G == 0, "upsample2x_add: C=%d must be a multiple of %d (16-byte groups)", C, G);
    const size_t n = (size_t)B * 2 * h * 2 * w * (C / G);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GIM_H16)
        hipLaunchKernelGGL(upsample2x_add_kernel<true>, dim3(nblocks(n, 256)), dim3(256), 0, s, x, y, B, h, w, C / G, ldx, ldy);
    else
        hipLaunchKernelGGL(upsample2x_add_kernel<false>, dim3(nblocks(n, 256)), dim3(256), 0, s, x, y, B, h, w, C / G, ldx, ldy);
    return gim_check_launch("upsample2x_add");
}

#if !GIM_HALF_KIND
extern "C" int gim_posenc_add_f16(const void* x, const float* pe, float* out_f32, void* out_t, int rows, int hw, int C,
                              int ldx, int ld_f32, int ld_t, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_posenc_add)(const void* x, const float* pe, float* out_f32, void* out_t, int rows, int hw, int C,
                              int ldx, int ld_f32, int ld_t, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_posenc_add_f16(x, pe, out_f32, out_t, rows, hw, C, ldx, ld_f32, ld_t, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && pe && (out_f32 || out_t) && rows > 0 && hw > 0 && C > 0 && C % 4 == 0, "posenc_add: bad args");
    GIM_REQUIRE(ldx % 4 == 0 && ld_f32 % 4 == 0 && ld_t % 4 == 0, "posenc_add: ld must be a multiple of 4");
    const size_t n = (size_t)rows * (C / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GIM_H16)
        hipLaunchKernelGGL(posenc_add_kernel<true>, dim3(nblocks(n, 256)), dim3(256), 0, s, x, pe, out_f32, out_t, (size_t)rows, hw, C / 4, ldx, ld_f32, ld_t);
    else
        hipLaunchKernelGGL(posenc_add_kernel<false>, dim3(nblocks(n, 256)), dim3(256), 0, s, x, pe, out_f32, out_t, (size_t)rows, hw, C / 4, ldx, ld_f32, ld_t);
    return gim_check_launch("posenc_add");
}

#if !GIM_HALF_KIND
extern "C" int gim_layernorm_residual_f16(const void* x, const float* gamma, const float* beta, const float* res,
                                      float* out_f32, void* out_t, int rows, int C, int ldx, int ldres, int ld_f32,
                                      int ld_t, int x_dtype, int dtype, float eps, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_layernorm_residual)(const void* x, const float* gamma, const float* beta, const float* res,
                                      float* out_f32, void* out_t, int rows, int C, int ldx, int ldres, int ld_f32,
                                      int ld_t, int x_dtype, int dtype, float eps, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (x_dtype == GIM_F16 || dtype == GIM_F16) return gim_layernorm_residual_f16(x, gamma, beta, res, out_f32, out_t, rows, C, ldx, ldres, ld_f32, ld_t, x_dtype, dtype, eps, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && gamma && beta && (out_f32 || out_t) && rows > 0, "layernorm: bad args");
    GIM_REQUIRE(C > 0 && C % 4 == 0 && C <= 512, "layernorm: C=%d unsupported (multiple of 4, <= 512)", C);
    GIM_REQUIRE(ldx % 4 == 0 && ld_f32 % 4 == 0 && ld_t % 4 == 0 && (!res || ldres % 4 == 0), "layernorm: ld alignment");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)((rows + 3) / 4);
    const bool ob = dtype == GIM_H16, xb = x_dtype == GIM_H16;
    if (ob && xb) hipLaunchKernelGGL((layernorm_kernel<true, true>), dim3(g), dim3(256), 0, s, x, gamma, beta, res, out_f32, out_t, rows, C, ldx, ldres, ld_f32, ld_t, eps);
    else if (ob) hipLaunchKernelGGL((layernorm_kernel<true, false>), dim3(g), dim3(256), 0, s, x, gamma, beta, res, out_f32, out_t, rows, C, ldx, ldres, ld_f32, ld_t, eps);
    else if (xb) hipLaunchKernelGGL((layernorm_kernel<false, true>), dim3(g), dim3(256), 0, s, x, gamma, beta, res, out_f32, out_t, rows, C, ldx, ldres, ld_f32, ld_t, eps);
    else hipLaunchKernelGGL((layernorm_kernel<false, false>), dim3(g), dim3(256), 0, s, x, gamma, beta, res, out_f32, out_t, rows, C, ldx, ldres, ld_f32, ld_t, eps);
    return gim_check_launch("layernorm");
}

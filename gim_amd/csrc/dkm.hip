// Glue kernels of the gim_dkm path (networks/dkm/models/dkm.py) for gfx950: everything around the convolutions
// (which run on conv_igemm.hip) and the GP solve (gp_solve.hip).  All tensors are NHWC rows; flow / certainty maps
// are small fp32 NHWC tensors ([b,h,w,2], [b,h,w,1]).  Everything here is HBM/L2-bound gather + elementwise work.
//
//   maxpool3x3s2        encoders.py:51 (torchvision resnet50 `maxpool`: kernel 3, stride 2, padding 1)
//   resize_bilinear     F.interpolate(mode='bilinear', align_corners=False)   dkm.py:420-425,468-479,518-529,668-701
//   grid_sample         F.grid_sample(bilinear, zeros, align_corners=False)   dkm.py:89
//   disp_emb            disp_emb(flow - query_coords), a 1x1 conv on 2 channels dkm.py:91-101
//   local_corr          local_correlation(x, y, r, flow)                      utils/local_correlation.py:5-40
//   dwconv5x5_bn_relu   depthwise 5x5 conv + BatchNorm(eval) + ReLU           dkm.py:58-73 (create_block, dw=True)
//   row_norms / cos_kernel_finish   CosKernel                                 dkm.py:135-144
//   global_avgpool / cab_scale_add  CAB                                       dkm.py:160-168
//   flow_update         dense_flow += ins * displacement / (4 w | 4 h), certainty += delta   dkm.py:505-514
//   match_post          match(): certainty attenuation, sigmoid, out-of-range / black masks, clamp, symmetric layout
//                                                                             dkm.py:693-741
#include <cstdlib>
#include <type_traits>
#include "gim_common.h"

namespace {

inline unsigned nblocks(size_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// ---- 3x3 / stride 2 / pad 1 max pooling, one thread per 16-byte channel group of one output pixel ---------------
template <bool BF16>
__global__ void maxpool3x3s2_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int H, int W, int Ho, int Wo,
                                    int CG, int ldx, int ldy) {
    constexpr int G = BF16 ? 8 : 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * Ho * Wo * CG) return;
    const int cg = (int)(idx % CG);
    const size_t pix = idx / CG;
    const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), b = (int)(pix / ((size_t)Wo * Ho));
#pragma unroll
    for (int e = 0; e < G; e += 4) {
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * yo - 1 + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * xo - 1 + dx;
                if (xx < 0 || xx >= W) continue;
                const float4 v = ElemIO<BF16>::ld4(x, (((size_t)b * H + yy) * W + xx) * ldx + cg * G + e);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        ElemIO<BF16>::st4(y, pix * ldy + cg * G + e, m);
    }
}

// ATen area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false)
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

// bilinear resize, align_corners=False; one thread per output element (channel fastest)
template <bool IN_BF16, bool OUT_BF16>
__global__ void resize_bilinear_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int h, int w, int Ho, int Wo,
                                       int C, int ldx, int ldy) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * Ho * Wo * C) return;
    const int c = (int)(idx % C);
    const size_t pix = idx / C;
    const int X = (int)(pix % Wo), Y = (int)((pix / Wo) % Ho), b = (int)(pix / ((size_t)Wo * Ho));
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index((float)h / (float)Ho, Y, h, y0, y1, ly0, ly1);
    src_index((float)w / (float)Wo, X, w, x0, x1, lx0, lx1);
    const size_t base = (size_t)b * h * w;
    const float v00 = ElemIO<IN_BF16>::ld(x, (base + (size_t)y0 * w + x0) * ldx + c), v01 = ElemIO<IN_BF16>::ld(x, (base + (size_t)y0 * w + x1) * ldx + c);
    const float v10 = ElemIO<IN_BF16>::ld(x, (base + (size_t)y1 * w + x0) * ldx + c), v11 = ElemIO<IN_BF16>::ld(x, (base + (size_t)y1 * w + x1) * ldx + c);
    ElemIO<OUT_BF16>::st(y, pix * ldy + c, ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11));
}

// NCHW fp32 image -> bilinear resize (align_corners=False) -> NHWC rows with zero channel padding
template <bool OUT_BF16>
__global__ void resize_image_kernel(const float* __restrict__ x, void* __restrict__ y, int B, int C, int h, int w, int Ho,
                                    int Wo, int cpad, int b_off) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * Ho * Wo) return;
    const int X = (int)(idx % Wo), Y = (int)((idx / Wo) % Ho), b = (int)(idx / ((size_t)Wo * Ho));
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index((float)h / (float)Ho, Y, h, y0, y1, ly0, ly1);
    src_index((float)w / (float)Wo, X, w, x0, x1, lx0, lx1);
    const size_t orow = (((size_t)(b + b_off) * Ho + Y) * Wo + X) * cpad;
    for (int c = 0; c < cpad; ++c) {
        float v = 0.f;
        if (c < C) {
            const float* p = x + ((size_t)b * C + c) * h * w;
            v = ly0 * (lx0 * p[(size_t)y0 * w + x0] + lx1 * p[(size_t)y0 * w + x1]) +
                ly1 * (lx0 * p[(size_t)y1 * w + x0] + lx1 * p[(size_t)y1 * w + x1]);
        }
        ElemIO<OUT_BF16>::st(y, orow + c, v);
    }
}

// grid_sample: feat [b,h,w,C] at grid [b,ho,wo,2] (x, y in [-1,1]) -> out rows (channel slice of a wider row allowed)
template <bool BF16>
__global__ void grid_sample_kernel(const void* __restrict__ feat, const float* __restrict__ grid, void* __restrict__ out,
                                   int B, int h, int w, int HoWo, int CG, int ldf, int ldo) {
    constexpr int G = BF16 ? 8 : 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * HoWo * CG) return;
    const int cg = (int)(idx % CG);
    const size_t pix = idx / CG;
    const int b = (int)(pix / HoWo);
    const float gx = grid[pix * 2 + 0], gy = grid[pix * 2 + 1];
    // grid_sampler_unnormalize(align_corners=false): ((g + 1) * size - 1) / 2
    const float ix = ((gx + 1.f) * (float)w - 1.f) / 2.f, iy = ((gy + 1.f) * (float)h - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
    const float wt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
    for (int e = 0; e < G; e += 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
            if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
            const float4 v = ElemIO<BF16>::ld4(feat, (((size_t)b * h + yy) * w + xx) * ldf + cg * G + e);
            acc.x += v.x * wt[k]; acc.y += v.y * wt[k]; acc.z += v.z * wt[k]; acc.w += v.w * wt[k];
        }
        ElemIO<BF16>::st4(out, pix * ldo + cg * G + e, acc);
    }
}

// emb[c] = W[c,0] * (fx - qx) + W[c,1] * (fy - qy) + bias[c]; query grid = pixel centres (dkm.py:91-101)
template <bool OUT_BF16>
__global__ void disp_emb_kernel(const float* __restrict__ flow, const float* __restrict__ wgt, const float* __restrict__ bias,
                                void* __restrict__ out, int B, int h, int w, int E, int ldo) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * h * w * E) return;
    const int c = (int)(idx % E);
    const size_t pix = idx / E;
    const int X = (int)(pix % w), Y = (int)((pix / w) % h);
    // torch.linspace(-1 + 1/n, 1 - 1/n, n)[i]: start + i * step for i < n/2, end - (n-1-i) * step otherwise
    auto lin = [](int i, int n) {
        const float start = -1.f + 1.f / (float)n, end = 1.f - 1.f / (float)n;
        const float step = (end - start) / (float)(n - 1);
        return i < n / 2 ? start + (float)i * step : end - (float)(n - 1 - i) * step;
    };
    const float dx = flow[pix * 2 + 0] - (w > 1 ? lin(X, w) : 0.f), dy = flow[pix * 2 + 1] - (h > 1 ? lin(Y, h) : 0.f);
    ElemIO<OUT_BF16>::st(out, pix * ldo + c, wgt[c * 2 + 0] * dx + wgt[c * 2 + 1] * dy + bias[c]);
}

// ---- local correlation: one wave per query pixel ---------------------------------------------------------------
// All (2r+1)^2 window taps are spaced exactly one pixel apart, so they share their bilinear fractions: the wave
// computes the (2r+2)^2 integer-grid dot products D once and every tap is a 4-term combination of them (3.5x fewer
// feature reads than sampling each tap's four corners).  Dot products: lanes split the channels (coalesced row
// reads), 16 positions are reduced together with a halving butterfly (17 shuffles instead of 96).
// Everything that depends only on the query (its flow, the patch origin, row bases, bounds) is wave-uniform: the wave index
// goes through readfirstlane so that the compiler keeps it in SGPRs -- scalar loads for the flow, SGPR base + lane offset
// addressing, bounds as scalar selects.  Out-of-range taps load a clamped (valid) address and are multiplied by 0, so all
// PT loads of a patch row are in flight together (scalar branches around single loads serialise them; per-lane predication,
// the first version, cost ~650 VALU instructions per patch row: 3 ms for the 226 k queries of the 336 x 336 level).
// PT: compile-time patch side (6 / 8 / 16 >= 2r + 2).  V8: 16-byte loads, 8 channels per lane (C % 512 == 0).
template <bool BF16, bool OUT_BF16, bool V8, int PT>
__global__ void __launch_bounds__(256) local_corr_kernel(const void* __restrict__ f0, const void* __restrict__ f1,
                                                         const float* __restrict__ flow, void* __restrict__ out, int B, int h,
                                                         int w, int C, int r, int ld0, int ld1, int ldo) {
    static_assert(!V8 || BF16, "16-byte channel groups are the bf16 layout");
    static_assert(PT == 6 || PT == 8 || PT == 16, "patch side");
    __shared__ float Dall[4][18 * 18];
    constexpr int VEC = V8 ? 8 : 4, ES = BF16 ? 2 : 4, NPOS = PT <= 8 ? 8 : 16;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t q = (size_t)blockIdx.x * 4 + wv;
    if (q >= (size_t)B * h * w) return;
    const int b = (int)(q / ((size_t)h * w));
    float* D = Dall[wv];
    const int P = 2 * r + 2;  // patch side (<= PT)
    const float gx = flow[q * 2 + 0], gy = flow[q * 2 + 1];
    // tap (0,0) sits at flow + (-2r/w, -2r/h) in normalised units = -r pixels
    const float ix = ((gx + 1.f) * (float)w - 1.f) / 2.f - (float)r, iy = ((gy + 1.f) * (float)h - 1.f) / 2.f - (float)r;
    const float fx = floorf(ix), fy = floorf(iy);
    // flows far outside the image (random weights) would overflow the int conversion: every tap is out of range anyway
    const int x0 = __builtin_amdgcn_readfirstlane((int)fminf(fmaxf(fx, -65536.f), 65536.f));
    const int y0 = __builtin_amdgcn_readfirstlane((int)fminf(fmaxf(fy, -65536.f), 65536.f));
    const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
    const int nch = (C + 64 * VEC - 1) / (64 * VEC);  // channel groups per lane, strided by 64 lanes
    const char* q0 = (const char*)f0 + q * (size_t)ld0 * ES;
    float qa[VEC];                                      // the query's channels of this lane (nch == 1: loaded once)
    auto load_vec = [&](const char* base, int c, float* dst) {
        if constexpr (V8) {
            const uint4 u = *(const uint4*)(base + (size_t)c * ES);
            dst[0] = h16_lo(u.x); dst[1] = h16_hi(u.x);
            dst[2] = h16_lo(u.y); dst[3] = h16_hi(u.y);
            dst[4] = h16_lo(u.z); dst[5] = h16_hi(u.z);
            dst[6] = h16_lo(u.w); dst[7] = h16_hi(u.w);
        } else {
            const float4 v = ElemIO<BF16>::ld4(base, c);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    };
    if (nch == 1 && lane * VEC < C) load_vec(q0, lane * VEC, qa);
    for (int py = 0; py < P; ++py) {
        const int yy = y0 + py;
        float part[NPOS];
#pragma unroll
        for (int px = 0; px < NPOS; ++px) part[px] = 0.f;
        if (yy >= 0 && yy < h) {
            const char* rowp = (const char*)f1 + (((size_t)b * h + yy) * w) * (size_t)ld1 * ES;
            for (int k = 0; k < nch; ++k) {
                const int c = (k * 64 + lane) * VEC;
                if (c >= C) break;
                if (nch > 1) load_vec(q0, c, qa);
                float v[PT][VEC];
#pragma unroll
                for (int px = 0; px < PT; ++px) {
                    const int xx = x0 + px;
                    load_vec(rowp + (size_t)min(max(xx, 0), w - 1) * ld1 * ES, c, v[px]);
                }
#pragma unroll
                for (int px = 0; px < PT; ++px) {
                    const int xx = x0 + px;
                    const float m = (px < P && xx >= 0 && xx < w) ? 1.f : 0.f;   // wave-uniform
                    float sum = (qa[0] * v[px][0] + qa[1] * v[px][1]) + (qa[2] * v[px][2] + qa[3] * v[px][3]);
                    if constexpr (V8) sum += (qa[4] * v[px][4] + qa[5] * v[px][5]) + (qa[6] * v[px][6] + qa[7] * v[px][7]);
                    part[px] = fmaf(m, sum, part[px]);
                }
            }
        }
        // halving butterfly over NPOS positions: after the step with mask m the lane keeps the positions whose bit matches
        constexpr int M0 = NPOS == 16 ? 32 : 16;
#pragma unroll
        for (int s = 0; s < (NPOS == 16 ? 4 : 3); ++s) {
            const int m = M0 >> s, half = (NPOS / 2) >> s;  // half = positions kept
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < NPOS / 2; ++i) {
                if (i < half) {
                    const float mine = up ? part[i + half] : part[i];
                    const float theirs = up ? part[i] : part[i + half];
                    part[i] = mine + __shfl_xor(theirs, m, 64);
                }
            }
        }
        float v = part[0];
        if constexpr (NPOS == 8) v += __shfl_xor(v, 32, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        // 16 positions: px = bits 5..2 of the lane (MSB first); 8 positions: bits 4..2 (both 32-lane halves hold the sums)
        const int px = NPOS == 16 ? ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1)
                                  : ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
        if ((lane & 3) == 0 && (NPOS == 16 || lane < 32) && px < P) D[py * 18 + px] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes are visible to its own reads
    const int K = 2 * r + 1;
    const float inv = 1.0f / sqrtf((float)C);
    for (int k = lane; k < K * K; k += 64) {
        const int ky = k / K, kx = k - ky * K;
        const float v = (D[ky * 18 + kx] * (wx0 * wy0) + D[ky * 18 + kx + 1] * (wx1 * wy0)) +
                        (D[(ky + 1) * 18 + kx] * (wx0 * wy1) + D[(ky + 1) * 18 + kx + 1] * (wx1 * wy1));
        ElemIO<OUT_BF16>::st(out, q * ldo + k, v * inv);
    }
}

// depthwise 5x5 (pad 2) + per-channel affine (conv bias and eval BatchNorm folded) + ReLU; Cout = mult * Cin,
// output channel co reads input channel co / mult; wgt [25][cpad] fp32, scale/shift [cpad]
template <bool BF16>
__global__ void dwconv5x5_kernel(const void* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ scale,
                                 const float* __restrict__ shift, void* __restrict__ y, int B, int H, int W, int Cout4,
                                 int mult, int cpad, int ldx, int ldy) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * H * W * Cout4) return;
    const int co = (int)(idx % Cout4) * 4;
    const size_t pix = idx / Cout4;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < 5; ++dy) {
        const int yy = Y + dy - 2;
        if (yy < 0 || yy >= H) continue;
        for (int dx = 0; dx < 5; ++dx) {
            const int xx = X + dx - 2;
            if (xx < 0 || xx >= W) continue;
            const size_t row = (((size_t)b * H + yy) * W + xx) * ldx;
            const float4 wv = *(const float4*)(wgt + (size_t)(dy * 5 + dx) * cpad + co);
            float4 v;
            if (mult == 1) {
                v = ElemIO<BF16>::ld4(x, row + co);
            } else {
                v.x = ElemIO<BF16>::ld(x, row + (co + 0) / mult); v.y = ElemIO<BF16>::ld(x, row + (co + 1) / mult);
                v.z = ElemIO<BF16>::ld(x, row + (co + 2) / mult); v.w = ElemIO<BF16>::ld(x, row + (co + 3) / mult);
            }
            acc.x = fmaf(v.x, wv.x, acc.x); acc.y = fmaf(v.y, wv.y, acc.y);
            acc.z = fmaf(v.z, wv.z, acc.z); acc.w = fmaf(v.w, wv.w, acc.w);
        }
    }
    const float4 sc = *(const float4*)(scale + co), sh = *(const float4*)(shift + co);
    acc.x = fmaxf(acc.x * sc.x + sh.x, 0.f); acc.y = fmaxf(acc.y * sc.y + sh.y, 0.f);
    acc.z = fmaxf(acc.z * sc.z + sh.z, 0.f); acc.w = fmaxf(acc.w * sc.w + sh.w, 0.f);
    ElemIO<BF16>::st4(y, pix * ldy + co, acc);
}

// depthwise 5x5, channel multiplier 1, register-tiled: one thread = 4 consecutive output pixels x one 16-byte channel
// group.  Each of the 5 input rows is read once as 8 pixels (x0-2 .. x0+5) and feeds all 4 outputs: 40 16-byte loads
// per 4 outputs instead of 100 8-byte loads (the first version, 40 % of match() time on the 1152x1536 / 576x768 maps).
// (An LDS-staged 16 x 8 tile variant with the weights in LDS was measured slower: 46.9 vs 44.4 ms per match().)
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <bool BF16, int MULT>
__global__ void __launch_bounds__(256) dwconv5x5_tiled_kernel(const void* __restrict__ x, const float* __restrict__ wgt,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              void* __restrict__ y, int B, int H, int W, int CG, int cpad, int ldx,
                                                              int ldy) {
    // The first register-tiled version was VALU-bound (1500 instructions per thread: 770 scalar FMAs, 450 v_cndmask for
    // the border zeroing of unpacked values, 64-bit address arithmetic per load): accumulate in float2 (v_pk_fma_f32),
    // zero the raw 16-byte group instead of its 8 unpacked values, one row base + 32-bit pixel offsets.
    constexpr int G = BF16 ? 8 : 4, ES = BF16 ? 2 : 4, G2 = G / 2;
    constexpr int LG = G / MULT;                 // input channels feeding this output group
    const int WS = (W + 3) / 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * H * WS * CG) return;
    const int cg = (int)(idx % CG);
    const size_t strip = idx / CG;
    const int xs = (int)(strip % WS) * 4, Y = (int)((strip / WS) % H), b = (int)(strip / ((size_t)WS * H));
    const int co = cg * G;
    f32x2_t acc[4][G2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < G2; ++e) acc[p][e] = (f32x2_t){0.f, 0.f};
    const unsigned pstride = (unsigned)ldx * ES;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const int yy = Y + dy - 2;
        if (yy < 0 || yy >= H) continue;
        const char* rp = (const char*)x + (((size_t)b * H + yy) * W) * pstride + (size_t)(co / MULT) * ES;
        f32x2_t in[8][G2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int xx = xs + q - 2;
            const bool ok = xx >= 0 && xx < W;
            const unsigned off = (unsigned)(ok ? xx : 0) * pstride;
            if constexpr (BF16) {
                if constexpr (MULT == 1) {
                    uint4 u = *(const uint4*)(rp + off);
                    if (!ok) u = make_uint4(0u, 0u, 0u, 0u);
                    in[q][0] = (f32x2_t){h16_lo(u.x), h16_hi(u.x)};
                    in[q][1] = (f32x2_t){h16_lo(u.y), h16_hi(u.y)};
                    in[q][2] = (f32x2_t){h16_lo(u.z), h16_hi(u.z)};
                    in[q][3] = (f32x2_t){h16_lo(u.w), h16_hi(u.w)};
                } else {  // multiplier 2: output channels (2c, 2c+1) read input channel c
                    uint2 u = *(const uint2*)(rp + off);
                    if (!ok) u = make_uint2(0u, 0u);
                    const float c0 = h16_lo(u.x), c1 = h16_hi(u.x);
                    const float c2 = h16_lo(u.y), c3 = h16_hi(u.y);
                    in[q][0] = (f32x2_t){c0, c0}; in[q][1] = (f32x2_t){c1, c1}; in[q][2] = (f32x2_t){c2, c2}; in[q][3] = (f32x2_t){c3, c3};
                }
            } else {
                static_assert(BF16 || MULT == 1, "fp32 tiled kernel: multiplier 1 only");
                float4 v = *(const float4*)(rp + off);
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                in[q][0] = (f32x2_t){v.x, v.y}; in[q][1] = (f32x2_t){v.z, v.w};
            }
        }
        (void)LG;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            f32x2_t wv[G2];
#pragma unroll
            for (int e = 0; e < G2; e += 2) {
                const float4 w4 = *(const float4*)(wgt + (size_t)(dy * 5 + dx) * cpad + co + 2 * e);
                wv[e] = (f32x2_t){w4.x, w4.y}; wv[e + 1] = (f32x2_t){w4.z, w4.w};
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int e = 0; e < G2; ++e) acc[p][e] = __builtin_elementwise_fma(in[p + dx][e], wv[e], acc[p][e]);
        }
    }
    f32x2_t sc[G2], sh[G2];
#pragma unroll
    for (int e = 0; e < G2; e += 2) {
        const float4 a = *(const float4*)(scale + co + 2 * e), c = *(const float4*)(shift + co + 2 * e);
        sc[e] = (f32x2_t){a.x, a.y}; sc[e + 1] = (f32x2_t){a.z, a.w};
        sh[e] = (f32x2_t){c.x, c.y}; sh[e + 1] = (f32x2_t){c.z, c.w};
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (xs + p >= W) break;
        const size_t o = (((size_t)b * H + Y) * W + xs + p) * ldy + co;
#pragma unroll
        for (int e = 0; e < G2; e += 2) {
            const f32x2_t r0 = acc[p][e] * sc[e] + sh[e], r1 = acc[p][e + 1] * sc[e + 1] + sh[e + 1];
            ElemIO<BF16>::st4(y, o + 2 * e, make_float4(fmaxf(r0.x, 0.f), fmaxf(r0.y, 0.f), fmaxf(r1.x, 0.f), fmaxf(r1.y, 0.f)));
        }
    }
}

// depthwise 5x5, channel multiplier 1, second generation.  The 4-pixel kernel above turned out to be bound by the
// vector-memory front end (16 clk per 1 KB wave load): per 4 outputs it issues 40 input loads AND 50 weight loads (25 taps x
// 2 float4; L1 hits, but the same TA slots), and every input row is fetched by 5 workgroups on different CUs.  Here
//   * one thread = 2 rows x 4 pixels x one 16-byte channel group: 6 input rows x 8 pixels feed 8 outputs (6 loads per
//     output pixel instead of 10) and one weight read serves both rows;
//   * the 25 x (channels of the block) weights sit in LDS (two float4 planes, lane stride 16 B: conflict-free
//     ds_read_b128, 4 clk per wave read, on the LDS pipe instead of the TA);
//   * a block covers CGB <= 32 channel groups x 256 / CGB strips; blocks are XCD-remapped so that vertically adjacent
//     row pairs (which share 4 of their 6 input rows) run on the same XCD's L2;
//   * one 16-byte store per output pixel.
// compiler fence for memory operations (IR level) + scheduling barrier (machine level): pins the software pipeline
#define ORDER_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// NF > 0 (16-bit only; round 5): the WHOLE ConvRefiner block -- depthwise 5x5 + BatchNorm + ReLU, then the block's 1x1 convolution with bias
// (dkm.py:58-73, create_block) -- in this launch, for the refiners whose channel count fits one chunk: C = 144 (NKS = 9 k16 steps, NF = 5 output
// fragments of 32) and C = 24 / 32 (NKS = 2, NF = 1).  The depthwise outputs of the block's 8 x (256 / CGB) pixels go to an LDS tile as 16-bit rows
// [px][K] (K padded to 16 NKS by zero groups written by the threads of the padding groups: CGB = 2 NKS), every wave takes PFW 32-pixel fragments of
// it as the B operand of the MFMA (weights = A operand: 16 bytes per lane and (fragment, k16 step) straight from L2, requested one step ahead),
// writes the results back over its own rows and stores 16-byte pieces.  The intermediate tensor -- 255 MB written and read per launch at 144
// channels -- never exists, and the 1x1's MFMAs run under the depthwise arithmetic of the co-resident workgroup (two per CU).  The 569-channel
// refiner does not fit this way (576 output channels of fp32 accumulators beside the depthwise working set: DESIGN.md section 4, round 5).
template <bool BF16, int NF = 0, int NKS = 0>
__global__ void __launch_bounds__(256, 2) dwconv5x5_rows2_kernel(const void* __restrict__ x, const float* __restrict__ wgt,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 void* __restrict__ y, int B, int H, int W, int CG, int CGB, int NCH,
                                                                 int cpad, int ldx, int ldy, unsigned nblk,
                                                                 const unsigned short* __restrict__ pww = nullptr, const float* __restrict__ pwb = nullptr) {
    constexpr bool PW = NF > 0;
    static_assert(!PW || BF16, "fused 1x1: 16-bit operands only");
    constexpr int G = BF16 ? 8 : 4, ES = BF16 ? 2 : 4, G2 = G / 2, NP = G / 4;   // NP float4 planes of weights
    extern __shared__ float4 wl[];                                              // [NP][25][CGB]  (+ PW: the pixel tile behind it)
    const unsigned lb = xcd_remap(blockIdx.x, nblk);
    const int chunk = (int)(lb % (unsigned)NCH);
    const unsigned sblk = lb / (unsigned)NCH;
    const int cg0 = chunk * CGB;
    const int ncg = min(CGB, CG - cg0);
    for (int e = threadIdx.x; e < NP * 25 * ncg; e += 256) {
        const int c = e % ncg, tp = e / ncg;            // tp = plane * 25 + tap
        const int pl = tp / 25, tap = tp - pl * 25;
        wl[(pl * 25 + tap) * CGB + c] = *(const float4*)(wgt + (size_t)tap * cpad + (size_t)(cg0 + c) * G + pl * 4);
    }
    __syncthreads();
    const int SPB = 256 / CGB;
    const int cgl = threadIdx.x % CGB, sl = threadIdx.x / CGB;
    const int WS = (W + 3) / 4, HS = (H + 1) / 2;
    const size_t strip = (size_t)sblk * SPB + sl;
    const bool live = !(sl >= SPB || cgl >= ncg || strip >= (size_t)B * HS * WS);
    if (!PW && !live) return;
    // (PW: a thread without a strip computes a clamped one and stores nothing -- it has to reach the barrier)
    const size_t strip_c = live ? strip : 0;
    const int xs = (int)(strip_c % WS) * 4, Y = (int)((strip_c / WS) % HS) * 2, b = (int)(strip_c / ((size_t)WS * HS));
    const int co = live ? (cg0 + cgl) * G : 0;
    f32x2_t acc[2][4][G2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_)
#pragma unroll
            for (int e = 0; e < G2; ++e) acc[o][p_][e] = (f32x2_t){0.f, 0.f};
    const unsigned pstride = (unsigned)ldx * ES;
    // Software pipeline over the 6 input rows: the 8 raw 16-byte groups of row r + 1 are requested before row r is
    // accumulated (the kernel runs at 2 waves per SIMD; without the prefetch every row exposed a full memory latency).
    // Out-of-range rows / pixels load a clamped address and are zeroed when unpacked: straight-line code.
    typedef typename std::conditional<BF16, uint4, float4>::type Raw;
    Raw raw[8];
    auto issue = [&](int r) {
        const int yy = min(max(Y + r - 2, 0), H - 1);
        const char* rp = (const char*)x + (((size_t)b * H + yy) * W) * pstride + (size_t)co * ES;
#pragma unroll
        for (int q = 0; q < 8; ++q) raw[q] = *(const Raw*)(rp + (unsigned)min(max(xs + q - 2, 0), W - 1) * pstride);
    };
    issue(0);
    // a real loop (not unrolled): each iteration is its own scheduling region, so the prefetch stays one row deep (fully
    // unrolled, the compiler hoisted all 48 loads and all 50 weight reads to the top and spilled 2 KB per thread)
#pragma unroll 1
    for (int r = 0; r < 6; ++r) {
        const int yy = Y + r - 2;
        const bool rok = yy >= 0 && yy < H;
        f32x2_t in[8][G2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int xx = xs + q - 2;
            const bool ok = rok && xx >= 0 && xx < W;
            if constexpr (BF16) {
                uint4 u = raw[q];
                if (!ok) u = make_uint4(0u, 0u, 0u, 0u);
                in[q][0] = (f32x2_t){h16_lo(u.x), h16_hi(u.x)};
                in[q][1] = (f32x2_t){h16_lo(u.y), h16_hi(u.y)};
                in[q][2] = (f32x2_t){h16_lo(u.z), h16_hi(u.z)};
                in[q][3] = (f32x2_t){h16_lo(u.w), h16_hi(u.w)};
            } else {
                float4 v = raw[q];
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                in[q][0] = (f32x2_t){v.x, v.y}; in[q][1] = (f32x2_t){v.z, v.w};
            }
        }
        ORDER_FENCE();
        if (r < 5) issue(r + 1);
        ORDER_FENCE();
        // output row Y + o takes this input row with dy = r - o
        if (r <= 4) {
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                f32x2_t wv[G2];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    const float4 w4 = wl[(pl * 25 + r * 5 + dx) * CGB + cgl];
                    wv[2 * pl] = (f32x2_t){w4.x, w4.y}; wv[2 * pl + 1] = (f32x2_t){w4.z, w4.w};
                }
#pragma unroll
                for (int p_ = 0; p_ < 4; ++p_)
#pragma unroll
                    for (int e = 0; e < G2; ++e) acc[0][p_][e] = __builtin_elementwise_fma(in[p_ + dx][e], wv[e], acc[0][p_][e]);
            }
        }
        if (r >= 1) {
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                f32x2_t wv[G2];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    const float4 w4 = wl[(pl * 25 + (r - 1) * 5 + dx) * CGB + cgl];
                    wv[2 * pl] = (f32x2_t){w4.x, w4.y}; wv[2 * pl + 1] = (f32x2_t){w4.z, w4.w};
                }
#pragma unroll
                for (int p_ = 0; p_ < 4; ++p_)
#pragma unroll
                    for (int e = 0; e < G2; ++e) acc[1][p_][e] = __builtin_elementwise_fma(in[p_ + dx][e], wv[e], acc[1][p_][e]);
            }
        }
    }
    f32x2_t sc[G2], sh[G2];
#pragma unroll
    for (int e = 0; e < G2; e += 2) {
        const float4 a = *(const float4*)(scale + co + 2 * e), c = *(const float4*)(shift + co + 2 * e);
        sc[e] = (f32x2_t){a.x, a.y}; sc[e + 1] = (f32x2_t){a.z, a.w};
        sh[e] = (f32x2_t){c.x, c.y}; sh[e + 1] = (f32x2_t){c.z, c.w};
    }
    if constexpr (PW) {
        constexpr int KP = NKS * 16, NPC = NF * 32;
        constexpr int ROW = (KP > NPC ? KP : NPC) * 2 + 16;     // bytes per pixel row: an ODD number of 16-byte slots (16 consecutive rows hit 16 different slots)
        constexpr int PXMAX = 8 * (256 / (2 * NKS));             // pixels per workgroup (CGB = 2 NKS channel groups incl. the zero padding groups)
        constexpr int PFW = (PXMAX + 127) / 128;                 // 32-pixel fragments per wave
        char* T = (char*)(wl + NP * 25 * CGB);                   // [4 PFW x 32 px][ROW]
        const int tid = threadIdx.x;
        const int lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lh = lane >> 5;
        // A operand of the 1x1 (row m = output channel 32 f + l31, k16 step ks: input channels 16 ks + 8 lh .. + 7): 16 bytes per lane and
        // (fragment, step) straight from L2, a ring of RING steps requested ahead -- the first RING steps HERE, in front of the tile writes and the
        // barrier (one step of lookahead left ~9 dependent L2 round trips per workgroup exposed: the fused block measured 0.25 ms SLOWER per match())
        constexpr int RING = NKS < 4 ? NKS : 4;
        bf16x8_t wa[RING][NF];
#pragma unroll
        for (int k = 0; k < RING; ++k)
#pragma unroll
            for (int f = 0; f < NF; ++f) wa[k][f] = *(const bf16x8_t*)(pww + (size_t)(32 * f + l31) * KP + k * 16 + lh * 8);
        if (sl < SPB) {
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int p_ = 0; p_ < 4; ++p_) {
                    float rr[8];
#pragma unroll
                    for (int e = 0; e < G2; ++e) {
                        const f32x2_t v = acc[o][p_][e] * sc[e] + sh[e];
                        rr[2 * e] = fmaxf(v.x, 0.f); rr[2 * e + 1] = fmaxf(v.y, 0.f);
                    }
                    const int pi = sl * 8 + o * 4 + p_;
                    // (a channel group beyond the stored width must hold exact zeros: it is K of the MFMA)
                    *(uint4*)(T + pi * ROW + cgl * 16) = cgl < CG ?
                        make_uint4(cvt_pk_h16(rr[0], rr[1]), cvt_pk_h16(rr[2], rr[3]), cvt_pk_h16(rr[4], rr[5]), cvt_pk_h16(rr[6], rr[7])) :
                        make_uint4(0u, 0u, 0u, 0u);
                }
        }
        __syncthreads();
        f32x16_t c[PFW][NF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 bb = *(const float4*)(pwb + 32 * f + 8 * rg + 4 * lh);
#pragma unroll
                for (int j = 0; j < PFW; ++j) { c[j][f][rg * 4] = bb.x; c[j][f][rg * 4 + 1] = bb.y; c[j][f][rg * 4 + 2] = bb.z; c[j][f][rg * 4 + 3] = bb.w; }
            }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int j = 0; j < PFW; ++j) {
                const int pi = (wv * PFW + j) * 32 + l31;
                const bf16x8_t bv = *(const bf16x8_t*)(T + pi * ROW + (2 * ks + lh) * 16);
#pragma unroll
                for (int f = 0; f < NF; ++f) c[j][f] = mfma_h16_32x32x16(wa[ks % RING][f], bv, c[j][f]);
            }
            if (ks + RING < NKS) {
#pragma unroll
                for (int f = 0; f < NF; ++f) wa[ks % RING][f] = *(const bf16x8_t*)(pww + (size_t)(32 * f + l31) * KP + (ks + RING) * 16 + lh * 8);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's rows are consumed (a wave's LDS accesses execute in order)
#pragma unroll
        for (int j = 0; j < PFW; ++j) {
            const int pi = (wv * PFW + j) * 32 + l31;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)   // accumulator quad = output channels 32 f + 8 rg + 4 lh .. + 3 of pixel pi
                    *(uint2*)(T + pi * ROW + (32 * f + 8 * rg + 4 * lh) * 2) =
                        make_uint2(cvt_pk_h16(c[j][f][rg * 4], c[j][f][rg * 4 + 1]), cvt_pk_h16(c[j][f][rg * 4 + 2], c[j][f][rg * 4 + 3]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // this wave's PFW x 32 pixel rows out: CG 16-byte pieces each
        const int npc = PFW * 32 * CG;
        for (int idx = lane; idx < npc; idx += 64) {
            const int ql = idx / CG, sq = idx - ql * CG;
            const int q = wv * PFW * 32 + ql;
            if (q >= SPB * 8) continue;
            const uint4 v = *(const uint4*)(T + q * ROW + sq * 16);
            const size_t st = (size_t)sblk * SPB + (q >> 3);
            if (st < (size_t)B * HS * WS) {
                const int qx = (int)(st % WS) * 4 + (q & 3), qy = (int)((st / WS) % HS) * 2 + ((q >> 2) & 1), qb = (int)(st / ((size_t)WS * HS));
                if (qx < W && qy < H) *(uint4*)((unsigned short*)y + (((size_t)qb * H + qy) * W + qx) * ldy + sq * 8) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        if (Y + o >= H) break;
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) {
            if (xs + p_ >= W) break;
            const size_t oo = (((size_t)b * H + Y + o) * W + xs + p_) * ldy + co;
            float rr[G];
#pragma unroll
            for (int e = 0; e < G2; ++e) {
                const f32x2_t v = acc[o][p_][e] * sc[e] + sh[e];
                rr[2 * e] = fmaxf(v.x, 0.f); rr[2 * e + 1] = fmaxf(v.y, 0.f);
            }
            if constexpr (BF16) {
                *(uint4*)((unsigned short*)y + oo) = make_uint4(cvt_pk_h16(rr[0], rr[1]), cvt_pk_h16(rr[2], rr[3]),
                                                                cvt_pk_h16(rr[4], rr[5]), cvt_pk_h16(rr[6], rr[7]));
            } else {
                *(float4*)((float*)y + oo) = make_float4(rr[0], rr[1], rr[2], rr[3]);
            }
        }
    }
}



#undef ORDER_FENCE

// one wave per row: L2 norm of x[r, 0:C]
template <bool BF16>
__global__ void __launch_bounds__(256) row_norms_kernel(const void* __restrict__ x, float* __restrict__ out, int rows, int C, int ld) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = ElemIO<BF16>::ld4(x, (size_t)r * ld + c);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = sqrtf(s);
}

// K[b,i,j] = exp((dot / (nx_i * ny_j + eps) - 1) / T), in place on the dot-product matrix rows [b*n, ld]
__global__ void cos_kernel_finish_kernel(float* __restrict__ k, const float* __restrict__ nx, const float* __restrict__ ny,
                                         int B, int n, int m, int ld, float T, float eps, float diag_add) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * n * m) return;
    const int j = (int)(idx % m);
    const size_t row = idx / m;
    const int b = (int)(row / n), i = (int)(row - (size_t)b * n);
    const float c = k[row * ld + j] / (nx[row] * ny[(size_t)b * m + j] + eps);
    float v = expf((c - 1.0f) / T);
    if (i == j) v += diag_add;
    k[row * ld + j] = v;
}

// mean over the spatial positions of NHWC rows: out[b, c_off + c] (fp32).  Workgroup = (image, 64-channel chunk):
// 4 pixel phases x 64 channels, coalesced along the channels, LDS reduction over the phases.
template <bool BF16>
__global__ void __launch_bounds__(256) global_avgpool_kernel(const void* __restrict__ x, float* __restrict__ out, int B, int HW, int C,
                                                             int ld, int ldo, int c_off) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int p = ph; p < HW; p += 4) s += ElemIO<BF16>::ld(x, ((size_t)b * HW + p) * ld + c);
    red[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0 && c < C) out[(size_t)b * ldo + c_off + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) / (float)HW;
}

// CAB: out = sigmoid(g[b, c]) * x2 + x1     (dkm.py:165-168)
template <bool BF16>
__global__ void cab_scale_add_kernel(const float* __restrict__ g, const void* __restrict__ x1, const void* __restrict__ x2,
                                     void* __restrict__ out, int B, int HW, int C4, int ldg, int ld1, int ld2, int ldo) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * HW * C4) return;
    const int c = (int)(idx % C4) * 4;
    const size_t pix = idx / C4;
    const int b = (int)(pix / HW);
    const float4 gg = *(const float4*)(g + (size_t)b * ldg + c);
    const float4 a = x1 ? ElemIO<BF16>::ld4(x1, pix * ld1 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v = ElemIO<BF16>::ld4(x2, pix * ld2 + c);
    auto sg = [](float t) { return 1.0f / (1.0f + expf(-t)); };
    ElemIO<BF16>::st4(out, pix * ldo + c, make_float4(sg(gg.x) * v.x + a.x, sg(gg.y) * v.y + a.y, sg(gg.z) * v.z + a.z, sg(gg.w) * v.w + a.w));
}

// flow[.,0] += ins * d[.,1] / (4 w_full), flow[.,1] += ins * d[.,2] / (4 h_full), cert += d[.,0]
// d rows: [certainty, dx, dy] (ConvRefiner out_conv: d[:, :-2] certainty, d[:, -2:] displacement)
template <bool BF16>
__global__ void flow_update_kernel(float* __restrict__ flow, float* __restrict__ cert, const void* __restrict__ d, size_t npix,
                                   int ldd, float sx, float sy, int flags) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const bool roma = flags & 2;     // RoMa's refiners emit (dx, dy, certainty), DKM's (certainty, dx, dy)
    const float dc = ElemIO<BF16>::ld(d, p * ldd + (roma ? 2 : 0));
    const float dx = ElemIO<BF16>::ld(d, p * ldd + (roma ? 0 : 1)), dy = ElemIO<BF16>::ld(d, p * ldd + (roma ? 1 : 2));
    flow[p * 2 + 0] = flow[p * 2 + 0] + dx * sx;
    flow[p * 2 + 1] = flow[p * 2 + 1] + dy * sy;
    cert[p] = ((flags & 1) ? 0.f : cert[p]) + dc;
}

// pixel-centre grid (dkm.py:437-448): flow[b,y,x] = (lin(x, w), lin(y, h))
__global__ void grid_coords_kernel(float* __restrict__ flow, int B, int h, int w) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)B * h * w) return;
    const int X = (int)(p % w), Y = (int)((p / w) % h);
    auto lin = [](int i, int n) {
        const float start = -1.f + 1.f / (float)n, end = 1.f - 1.f / (float)n;
        const float step = (end - start) / (float)(n - 1);
        return i < n / 2 ? start + (float)i * step : end - (float)(n - 1 - i) * step;
    };
    flow[p * 2 + 0] = w > 1 ? lin(X, w) : 0.f;
    flow[p * 2 + 1] = h > 1 ? lin(Y, h) : 0.f;
}

// match() tail (dkm.py:693-741), symmetric, one pair: direction 0 = query -> support, direction 1 = support -> query.
// warp [H, 2W, 4], certainty [H, 2W]; black0/black1: uint8 masks already resized (nearest) to [H, W]
__global__ void match_post_kernel(const float* __restrict__ flow0, const float* __restrict__ flow1, const float* __restrict__ cert0,
                                  const float* __restrict__ cert1, const float* __restrict__ low0, const float* __restrict__ low1,
                                  const uint8_t* __restrict__ black0, const uint8_t* __restrict__ black1, float* __restrict__ warp,
                                  float* __restrict__ certainty, int H, int W) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)2 * H * W) return;
    const int dir = (int)(p / ((size_t)H * W));
    const size_t pp = p - (size_t)dir * H * W;
    const int X = (int)(pp % W), Y = (int)(pp / W);
    auto lin = [](int i, int n) {
        const float start = -1.f + 1.f / (float)n, end = 1.f - 1.f / (float)n;
        const float step = (end - start) / (float)(n - 1);
        return i < n / 2 ? start + (float)i * step : end - (float)(n - 1 - i) * step;
    };
    const float qx = lin(X, W), qy = lin(Y, H);
    const float* flow = dir == 0 ? flow0 : flow1;
    float fx = flow[pp * 2 + 0], fy = flow[pp * 2 + 1];
    float l = (dir == 0 ? low0 : low1)[pp];
    l = 0.5f * l * (l < 0.f ? 1.f : 0.f);
    float c = 1.0f / (1.0f + expf(-((dir == 0 ? cert0 : cert1)[pp] - l)));
    if (fabsf(fx) > 1.f || fabsf(fy) > 1.f) c = 0.f;
    if ((dir == 0 ? black0 : black1)[pp]) c = 0.f;
    fx = fminf(fmaxf(fx, -1.f), 1.f);
    fy = fminf(fmaxf(fy, -1.f), 1.f);
    const size_t o = (size_t)Y * 2 * W + (size_t)dir * W + X;
    if (dir == 0) { warp[o * 4 + 0] = qx; warp[o * 4 + 1] = qy; warp[o * 4 + 2] = fx; warp[o * 4 + 3] = fy; }
    else { warp[o * 4 + 0] = fx; warp[o * 4 + 1] = fy; warp[o * 4 + 2] = qx; warp[o * 4 + 3] = qy; }
    certainty[o] = c;
}

// black-pixel mask of an NCHW fp32 image resized with F.interpolate(mode='nearest') (dkm.py:726-729)
__global__ void black_mask_kernel(const float* __restrict__ im, uint8_t* __restrict__ mask, int h, int w, int Ho, int Wo) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)Ho * Wo) return;
    const int X = (int)(p % Wo), Y = (int)(p / Wo);
    // nearest: src = floor(dst * in / out) with the float scale ATen uses
    const int sy = min((int)floorf((float)Y * ((float)h / (float)Ho)), h - 1), sx = min((int)floorf((float)X * ((float)w / (float)Wo)), w - 1);
    const size_t o = (size_t)sy * w + sx, hw = (size_t)h * w;
    mask[p] = (im[o] < 0.03125f) && (im[hw + o] < 0.03125f) && (im[2 * hw + o] < 0.03125f);
}

// kde (utils/kde.py:17-26): density[i] = sum_j exp(-|x_i - x_j|^2 / (2 std^2)), x [n,4]; j tiles staged in LDS
__device__ __forceinline__ float4 round_half4(float4 v) {
    return make_float4((float)(_Float16)v.x, (float)(_Float16)v.y, (float)(_Float16)v.z, (float)(_Float16)v.w);
}

// One block = 64 query points; its 4 waves split the j range (chunks of 64 points, chunk c -> wave c % 4), each staging its
// chunk in its own LDS slice (wave-local: no block barrier in the loop), partial sums combined in a fixed order.  The first
// version ran 256 queries per block with the whole block walking all j: 79 blocks for 20 000 samples -- a quarter of the CUs,
// one wave per SIMD, 1.6 ms per call; exp via v_exp_f32 (relative error ~1e-6 on a sum of 20 000 terms).
__global__ void __launch_bounds__(256) kde_kernel(const float* __restrict__ x, float* __restrict__ density, int n, float inv2s2, int half) {
    __shared__ float4 tile[4][64];
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float4 xi = i < n ? *(const float4*)(x + (size_t)i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (half) xi = round_half4(xi);
    const float c = -inv2s2 * 1.4426950408889634f;      // exp(-d2 * inv2s2) = exp2(d2 * c)
    float acc = 0.f;
    for (int j0 = wv * 64; j0 < n; j0 += 256) {
        const int j = j0 + lane;
        float4 xj = j < n ? *(const float4*)(x + (size_t)j * 4) : make_float4(1e18f, 1e18f, 1e18f, 1e18f);
        if (half && j < n) xj = round_half4(xj);
        tile[wv][lane] = xj;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local LDS hand-off
        const int m = min(64, n - j0);
#pragma unroll 8
        for (int k = 0; k < m; ++k) {
            const float4 v = tile[wv][k];
            const float dx = xi.x - v.x, dy = xi.y - v.y, dz = xi.z - v.z, dw = xi.w - v.w;
            acc += __builtin_amdgcn_exp2f(((dx * dx + dy * dy) + (dz * dz + dw * dw)) * c);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all reads done before the slice is overwritten
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && i < n) density[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

}  // namespace

#define DISPATCH_BF(KERN, bf, grid, ...)                                                                  \
    do {                                                                                                  \
        if (bf) hipLaunchKernelGGL(KERN<true>, grid, dim3(256), 0, s, __VA_ARGS__);                     \
        else hipLaunchKernelGGL(KERN<false>, grid, dim3(256), 0, s, __VA_ARGS__);                       \
    } while (0)

#if !GIM_HALF_KIND
extern "C" int gim_maxpool3x3s2_f16(const void* x, void* y, int B, int H, int W, int C, int ldx, int ldy, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_maxpool3x3s2)(const void* x, void* y, int B, int H, int W, int C, int ldx, int ldy, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_maxpool3x3s2_f16(x, y, B, H, W, C, ldx, ldy, dtype, stream);   // the fp16 objects of this file
#endif
    const int G = dtype == GIM_H16 ? 8 : 4;
    GIM_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % G == 0 && ldx % G == 0 && ldy % G == 0, "maxpool3x3s2: bad args");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks((size_t)B * Ho * Wo * (C / G), 256));
    DISPATCH_BF(maxpool3x3s2_kernel, dtype == GIM_H16, grid, x, y, B, H, W, Ho, Wo, C / G, ldx, ldy);
    return gim_check_launch("maxpool3x3s2");
}

#if !GIM_HALF_KIND
extern "C" int gim_resize_bilinear_f16(const void* x, void* y, int B, int h, int w, int Ho, int Wo, int C, int ldx, int ldy,
                                   int dtype, int out_dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_resize_bilinear)(const void* x, void* y, int B, int h, int w, int Ho, int Wo, int C, int ldx, int ldy,
                                   int dtype, int out_dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16 || out_dtype == GIM_F16) return gim_resize_bilinear_f16(x, y, B, h, w, Ho, Wo, C, ldx, ldy, dtype, out_dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && y && B > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0 && C > 0 && ldx >= C && ldy >= C, "resize_bilinear: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks((size_t)B * Ho * Wo * C, 256));
    const bool ib = dtype == GIM_H16, ob = out_dtype == GIM_H16;
    if (ib && ob) hipLaunchKernelGGL((resize_bilinear_kernel<true, true>), grid, dim3(256), 0, s, x, y, B, h, w, Ho, Wo, C, ldx, ldy);
    else if (ib) hipLaunchKernelGGL((resize_bilinear_kernel<true, false>), grid, dim3(256), 0, s, x, y, B, h, w, Ho, Wo, C, ldx, ldy);
    else if (ob) hipLaunchKernelGGL((resize_bilinear_kernel<false, true>), grid, dim3(256), 0, s, x, y, B, h, w, Ho, Wo, C, ldx, ldy);
    else hipLaunchKernelGGL((resize_bilinear_kernel<false, false>), grid, dim3(256), 0, s, x, y, B, h, w, Ho, Wo, C, ldx, ldy);
    return gim_check_launch("resize_bilinear");
}

#if !GIM_HALF_KIND
extern "C" int gim_resize_image_f16(const float* x, void* y, int B, int C, int h, int w, int Ho, int Wo, int cpad, int b_off,
                                int out_dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_resize_image)(const float* x, void* y, int B, int C, int h, int w, int Ho, int Wo, int cpad, int b_off,
                                int out_dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (out_dtype == GIM_F16) return gim_resize_image_f16(x, y, B, C, h, w, Ho, Wo, cpad, b_off, out_dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && y && B > 0 && C > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0 && cpad >= C, "resize_image: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks((size_t)B * Ho * Wo, 256));
    DISPATCH_BF(resize_image_kernel, out_dtype == GIM_H16, grid, x, y, B, C, h, w, Ho, Wo, cpad, b_off);
    return gim_check_launch("resize_image");
}

#if !GIM_HALF_KIND
extern "C" int gim_grid_sample_f16(const void* feat, const float* grid_xy, void* out, int B, int h, int w, int Ho, int Wo, int C,
                               int ldf, int ldo, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_grid_sample)(const void* feat, const float* grid_xy, void* out, int B, int h, int w, int Ho, int Wo, int C,
                               int ldf, int ldo, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_grid_sample_f16(feat, grid_xy, out, B, h, w, Ho, Wo, C, ldf, ldo, dtype, stream);   // the fp16 objects of this file
#endif
    const int G = dtype == GIM_H16 ? 8 : 4;
    GIM_REQUIRE(feat && grid_xy && out && B > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0 && C > 0, "grid_sample: bad args");
    GIM_REQUIRE(C % G == 0 && ldf % G == 0 && ldo % G == 0, "grid_sample: C / strides must keep 16-byte groups (C=%d)", C);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks((size_t)B * Ho * Wo * (C / G), 256));
    DISPATCH_BF(grid_sample_kernel, dtype == GIM_H16, grid, feat, grid_xy, out, B, h, w, Ho * Wo, C / G, ldf, ldo);
    return gim_check_launch("grid_sample");
}

#if !GIM_HALF_KIND
extern "C" int gim_dkm_disp_emb_f16(const float* flow, const float* wgt, const float* bias, void* out, int B, int h, int w, int E,
                                int ldo, int out_dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_dkm_disp_emb)(const float* flow, const float* wgt, const float* bias, void* out, int B, int h, int w, int E,
                                int ldo, int out_dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (out_dtype == GIM_F16) return gim_dkm_disp_emb_f16(flow, wgt, bias, out, B, h, w, E, ldo, out_dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(flow && wgt && bias && out && B > 0 && h > 0 && w > 0 && E > 0 && ldo >= E, "dkm_disp_emb: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks((size_t)B * h * w * E, 256));
    DISPATCH_BF(disp_emb_kernel, out_dtype == GIM_H16, grid, flow, wgt, bias, out, B, h, w, E, ldo);
    return gim_check_launch("dkm_disp_emb");
}

#if !GIM_HALF_KIND
extern "C" int gim_local_corr_f16(const void* f0, const void* f1, const float* flow, void* out, int B, int h, int w, int C, int r,
                              int ld0, int ld1, int ldo, int dtype, int out_dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_local_corr)(const void* f0, const void* f1, const float* flow, void* out, int B, int h, int w, int C, int r,
                              int ld0, int ld1, int ldo, int dtype, int out_dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16 || out_dtype == GIM_F16) return gim_local_corr_f16(f0, f1, flow, out, B, h, w, C, r, ld0, ld1, ldo, dtype, out_dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(f0 && f1 && flow && out && B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "local_corr: bad args");
    GIM_REQUIRE(r >= 1 && r <= 7, "local_corr: radius %d unsupported (1..7)", r);
    GIM_REQUIRE(ld0 % 4 == 0 && ld1 % 4 == 0 && ldo >= (2 * r + 1) * (2 * r + 1), "local_corr: strides");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks((size_t)B * h * w, 4));
    const bool ib = dtype == GIM_H16, ob = out_dtype == GIM_H16;
    const bool v8 = ib && C % 512 == 0 && ld0 % 8 == 0 && ld1 % 8 == 0;
    const int P = 2 * r + 2;
#define LC_LAUNCH(I, O, V, PT) hipLaunchKernelGGL((local_corr_kernel<I, O, V, PT>), grid, dim3(256), 0, s, f0, f1, flow, out, B, h, w, C, r, ld0, ld1, ldo)
#define LC_PT(I, O, V) do { if (P <= 6) LC_LAUNCH(I, O, V, 6); else if (P <= 8) LC_LAUNCH(I, O, V, 8); else LC_LAUNCH(I, O, V, 16); } while (0)
    if (v8 && ob) LC_PT(true, true, true);
    else if (ib && ob) LC_PT(true, true, false);
    else if (!ib && !ob) LC_PT(false, false, false);
    else if (v8) LC_LAUNCH(true, false, true, 16);
    else if (ib) LC_LAUNCH(true, false, false, 16);
    else LC_LAUNCH(false, true, false, 16);
#undef LC_PT
#undef LC_LAUNCH
    return gim_check_launch("local_corr");
}

#if !GIM_HALF_KIND
extern "C" int gim_dwconv5x5_bn_relu_f16(const void* x, const float* wgt, const float* scale, const float* shift, void* y, int B,
                                     int H, int W, int Cin, int Cout, int cpad, int ldx, int ldy, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_dwconv5x5_bn_relu)(const void* x, const float* wgt, const float* scale, const float* shift, void* y, int B,
                                     int H, int W, int Cin, int Cout, int cpad, int ldx, int ldy, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_dwconv5x5_bn_relu_f16(x, wgt, scale, shift, y, B, H, W, Cin, Cout, cpad, ldx, ldy, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && wgt && scale && shift && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout % Cin == 0, "dwconv5x5: bad args");
    GIM_REQUIRE(cpad % 4 == 0 && cpad >= Cout && ldx % 4 == 0 && ldy % 4 == 0 && ldy >= cpad, "dwconv5x5: cpad / strides");
    GIM_REQUIRE((int64_t)ldx * (Cout / Cin) >= cpad, "dwconv5x5: input rows too narrow for the padded channel range");
    hipStream_t s = (hipStream_t)stream;
    const int G = dtype == GIM_H16 ? 8 : 4;
    const int mult = Cout / Cin;
    if ((mult == 1 || (mult == 2 && dtype == GIM_H16)) && cpad % G == 0 && ldx % 4 == 0 && ldy % G == 0 && (mult == 2 || ldx % G == 0)) {
        const dim3 gt(nblocks((size_t)B * H * ((W + 3) / 4) * (cpad / G), 256));
        if (mult == 1) {
            const int CG = cpad / G;
            const int NCH = (CG + 31) / 32, CGB = (CG + NCH - 1) / NCH, SPB = 256 / CGB;
            const size_t strips = (size_t)B * ((H + 1) / 2) * ((W + 3) / 4);
            const size_t nblk = (strips + SPB - 1) / SPB * NCH;
            GIM_REQUIRE(nblk < 0x7fffffffull, "dwconv5x5: grid too large");
            const size_t shm = (size_t)(G / 4) * 25 * CGB * 16;
            if (dtype == GIM_H16) hipLaunchKernelGGL((dwconv5x5_rows2_kernel<true>), dim3((unsigned)nblk), dim3(256), shm, s, x, wgt, scale, shift, y, B, H, W, CG, CGB, NCH, cpad, ldx, ldy, (unsigned)nblk);
            else hipLaunchKernelGGL((dwconv5x5_rows2_kernel<false>), dim3((unsigned)nblk), dim3(256), shm, s, x, wgt, scale, shift, y, B, H, W, CG, CGB, NCH, cpad, ldx, ldy, (unsigned)nblk);
            return gim_check_launch("dwconv5x5_rows2");
        }
        hipLaunchKernelGGL((dwconv5x5_tiled_kernel<true, 2>), gt, dim3(256), 0, s, x, wgt, scale, shift, y, B, H, W, cpad / G, cpad, ldx, ldy);   // channel multiplier 2
        return gim_check_launch("dwconv5x5_tiled");
    }
    const dim3 grid(nblocks((size_t)B * H * W * (cpad / 4), 256));
    DISPATCH_BF(dwconv5x5_kernel, dtype == GIM_H16, grid, x, wgt, scale, shift, y, B, H, W, cpad / 4, Cout / Cin, cpad, ldx, ldy);
    return gim_check_launch("dwconv5x5");
}

// The ConvRefiner block in one launch (see dwconv5x5_rows2_kernel, NF > 0): x [B,H,W,ldx], y [B,H,W,ldy] 16-bit rows of cs stored channels (cs = 144, or 24 /
// 32); wgt [25][cs], scale / shift [cs] fp32 as for gim_dwconv5x5_bn_relu; pw_w [NP][KP] 16-bit (the 1x1 weights, rows = output channels padded to NP = 160 / 32,
// K = input channels padded to KP = 144 / 32, zero padding), pw_b [NP] fp32.
#if !GIM_HALF_KIND
extern "C" int gim_dwconv5x5_pw_f16(const void* x, const float* wgt, const float* scale, const float* shift, const void* pw_w, const float* pw_b, void* y,
                                    int B, int H, int W, int cs, int ldx, int ldy, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_dwconv5x5_pw)(const void* x, const float* wgt, const float* scale, const float* shift, const void* pw_w, const float* pw_b, void* y,
                                       int B, int H, int W, int cs, int ldx, int ldy, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_dwconv5x5_pw_f16(x, wgt, scale, shift, pw_w, pw_b, y, B, H, W, cs, ldx, ldy, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && wgt && scale && shift && pw_w && pw_b && y && B > 0 && H > 0 && W > 0, "dwconv5x5_pw: bad args");
    GIM_REQUIRE(dtype == GIM_H16, "dwconv5x5_pw: 16-bit operands only");
    GIM_REQUIRE((cs == 24 || cs == 32 || cs == 144) && ldx % 8 == 0 && ldx >= cs && ldy % 8 == 0 && ldy >= cs,
                "dwconv5x5_pw: stored channels %d (24, 32 or 144), row strides (ldx %d, ldy %d)", cs, ldx, ldy);
    const int CG = cs / 8, nks = cs == 144 ? 9 : 2, nf = cs == 144 ? 5 : 1, CGB = 2 * nks, SPB = 256 / CGB;
    const int kp = nks * 16, npc = nf * 32, row = (kp > npc ? kp : npc) * 2 + 16, pfw = (8 * SPB + 127) / 128;
    const size_t strips = (size_t)B * ((H + 1) / 2) * ((W + 3) / 4);
    const size_t nblk = (strips + SPB - 1) / SPB;
    GIM_REQUIRE(nblk < 0x7fffffffull, "dwconv5x5_pw: grid too large");
    const size_t shm = (size_t)2 * 25 * CGB * 16 + (size_t)pfw * 128 * row;
    hipStream_t s = (hipStream_t)stream;
    if (cs == 144) {
        static GimPerDevice attr;
        if (attr.needed()) {
            if (hipFuncSetAttribute((const void*)dwconv5x5_rows2_kernel<true, 5, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) {
                gim_set_error("dwconv5x5_pw: hipFuncSetAttribute(%d B LDS)", (int)shm); return GIM_ERR_LAUNCH;
            }
            attr.done();
        }
        hipLaunchKernelGGL((dwconv5x5_rows2_kernel<true, 5, 9>), dim3((unsigned)nblk), dim3(256), shm, s, x, wgt, scale, shift, y, B, H, W, CG, CGB, 1, cs, ldx, ldy,
                           (unsigned)nblk, (const unsigned short*)pw_w, pw_b);
    } else {
        hipLaunchKernelGGL((dwconv5x5_rows2_kernel<true, 1, 2>), dim3((unsigned)nblk), dim3(256), shm, s, x, wgt, scale, shift, y, B, H, W, CG, CGB, 1, cs, ldx, ldy,
                           (unsigned)nblk, (const unsigned short*)pw_w, pw_b);
    }
    return gim_check_launch("dwconv5x5_pw");
}

#if !GIM_HALF_KIND
extern "C" int gim_row_norms_f16(const void* x, float* out, int rows, int C, int ld, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_row_norms)(const void* x, float* out, int rows, int C, int ld, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_row_norms_f16(x, out, rows, C, ld, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && out && rows > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0, "row_norms: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((rows + 3) / 4);
    DISPATCH_BF(row_norms_kernel, dtype == GIM_H16, grid, x, out, rows, C, ld);
    return gim_check_launch("row_norms");
}

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_cos_kernel_finish(float* k, const float* nx, const float* ny, int B, int n, int m, int ld, float T, float eps,
                                     float diag_add, gim_stream_t stream) {
    GIM_REQUIRE(k && nx && ny && B > 0 && n > 0 && m > 0 && ld >= m, "cos_kernel_finish: bad args");
    hipLaunchKernelGGL(cos_kernel_finish_kernel, dim3(nblocks((size_t)B * n * m, 256)), dim3(256), 0, (hipStream_t)stream, k, nx, ny, B, n, m, ld, T, eps, diag_add);
    return gim_check_launch("cos_kernel_finish");
}
#endif

#if !GIM_HALF_KIND
extern "C" int gim_global_avgpool_f16(const void* x, float* out, int B, int HW, int C, int ld, int ldo, int c_off, int dtype,
                                  gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_global_avgpool)(const void* x, float* out, int B, int HW, int C, int ld, int ldo, int c_off, int dtype,
                                  gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_global_avgpool_f16(x, out, B, HW, C, ld, ldo, c_off, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && out && B > 0 && HW > 0 && C > 0, "global_avgpool: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((C + 63) / 64, B);
    DISPATCH_BF(global_avgpool_kernel, dtype == GIM_H16, grid, x, out, B, HW, C, ld, ldo, c_off);
    return gim_check_launch("global_avgpool");
}

#if !GIM_HALF_KIND
extern "C" int gim_cab_scale_add_f16(const float* g, const void* x1, const void* x2, void* out, int B, int HW, int C, int ldg, int ld1,
                                 int ld2, int ldo, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_cab_scale_add)(const float* g, const void* x1, const void* x2, void* out, int B, int HW, int C, int ldg, int ld1,
                                 int ld2, int ldo, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_cab_scale_add_f16(g, x1, x2, out, B, HW, C, ldg, ld1, ld2, ldo, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(g && x2 && out && B > 0 && HW > 0 && C > 0 && C % 4 == 0 && ldg % 4 == 0 && ld2 % 4 == 0 && ldo % 4 == 0 && (!x1 || ld1 % 4 == 0), "cab_scale_add: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks((size_t)B * HW * (C / 4), 256));
    DISPATCH_BF(cab_scale_add_kernel, dtype == GIM_H16, grid, g, x1, x2, out, B, HW, C / 4, ldg, ld1, ld2, ldo);
    return gim_check_launch("cab_scale_add");
}

#if !GIM_HALF_KIND
extern "C" int gim_dkm_flow_update_f16(float* flow, float* cert, const void* d, int64_t npix, int ldd, float sx, float sy, int cert_init,
                                   int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_dkm_flow_update)(float* flow, float* cert, const void* d, int64_t npix, int ldd, float sx, float sy, int cert_init,
                                   int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_dkm_flow_update_f16(flow, cert, d, npix, ldd, sx, sy, cert_init, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(flow && cert && d && npix > 0 && ldd >= 3, "dkm_flow_update: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks((size_t)npix, 256));
    DISPATCH_BF(flow_update_kernel, dtype == GIM_H16, grid, flow, cert, d, (size_t)npix, ldd, sx, sy, cert_init);  // cert_init: flag bits
    return gim_check_launch("dkm_flow_update");
}

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_dkm_grid_coords(float* flow, int B, int h, int w, gim_stream_t stream) {
    GIM_REQUIRE(flow && B > 0 && h > 0 && w > 0, "dkm_grid_coords: bad args");
    hipLaunchKernelGGL(grid_coords_kernel, dim3(nblocks((size_t)B * h * w, 256)), dim3(256), 0, (hipStream_t)stream, flow, B, h, w);
    return gim_check_launch("dkm_grid_coords");
}
#endif

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_dkm_match_post(const float* flow0, const float* flow1, const float* cert0, const float* cert1, const float* low0,
                                  const float* low1, const uint8_t* black0, const uint8_t* black1, float* warp, float* certainty,
                                  int H, int W, gim_stream_t stream) {
    GIM_REQUIRE(flow0 && flow1 && cert0 && cert1 && low0 && low1 && black0 && black1 && warp && certainty && H > 1 && W > 1, "dkm_match_post: bad args");
    hipLaunchKernelGGL(match_post_kernel, dim3(nblocks((size_t)2 * H * W, 256)), dim3(256), 0, (hipStream_t)stream, flow0, flow1, cert0, cert1,
                       low0, low1, black0, black1, warp, certainty, H, W);
    return gim_check_launch("dkm_match_post");
}
#endif

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_dkm_black_mask(const float* im, uint8_t* mask, int h, int w, int Ho, int Wo, gim_stream_t stream) {
    GIM_REQUIRE(im && mask && h > 0 && w > 0 && Ho > 0 && Wo > 0, "dkm_black_mask: bad args");
    hipLaunchKernelGGL(black_mask_kernel, dim3(nblocks((size_t)Ho * Wo, 256)), dim3(256), 0, (hipStream_t)stream, im, mask, h, w, Ho, Wo);
    return gim_check_launch("dkm_black_mask");
}
#endif

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_kde(const float* x, float* density, int n, float std, gim_stream_t stream) {
    GIM_REQUIRE(x && density && n > 0 && std != 0.f, "kde: bad args");
    // std < 0: coordinates rounded to fp16 first (RoMa evaluates its KDE on x.half(), roma.py:1018-1023)
    hipLaunchKernelGGL(kde_kernel, dim3((n + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, density, n, 1.0f / (2.0f * std * std), std < 0.f ? 1 : 0);
    return gim_check_launch("kde");
}
#endif

namespace {
// normalised (x0, y0, x1, y1) in [-1, 1] -> pixel coordinates of the two images (trainer/lightning.py:141-144, demo.py:438-443)
__global__ void to_pixels_kernel(const float* __restrict__ m, float* __restrict__ k0, float* __restrict__ k1, int n, float w0, float h0,
                                 float w1, float h1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = *(const float4*)(m + (size_t)i * 4);
    k0[i * 2 + 0] = w0 * (v.x + 1.f) / 2.f; k0[i * 2 + 1] = h0 * (v.y + 1.f) / 2.f;
    k1[i * 2 + 0] = w1 * (v.z + 1.f) / 2.f; k1[i * 2 + 1] = h1 * (v.w + 1.f) / 2.f;
}
}  // namespace

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_dense_to_pixels(const float* matches, float* kpts0, float* kpts1, int n, float w0, float h0, float w1, float h1,
                                   gim_stream_t stream) {
    GIM_REQUIRE(matches && kpts0 && kpts1 && n >= 0, "dense_to_pixels: bad args");
    if (n == 0) return GIM_OK;
    hipLaunchKernelGGL(to_pixels_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, matches, kpts0, kpts1, n, w0, h0, w1, h1);
    return gim_check_launch("dense_to_pixels");
}
#endif

namespace {
// cls_to_flow_refine (roma.py:1092-1121) + the certainty channel: one wave per pixel over C = res^2 class logits.
// softmax, arg-max (first index on ties), the arg-max and its 4-neighbourhood (indices clamped to [0, C-1], duplicates
// counted twice like torch.gather does), anchor-weighted mean / total probability.
__global__ void __launch_bounds__(256) cls_to_flow_kernel(const float* __restrict__ logits, float* __restrict__ flow, float* __restrict__ cert,
                                                          int npix, int C, int res, int ld) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npix) return;
    const float* row = logits + (size_t)p * ld;
    float mx = -INFINITY;
    int arg = 0;
    for (int c = lane; c < C; c += 64) {
        const float v = row[c];
        if (v > mx) { mx = v; arg = c; }
    }
    // wave arg-max, smallest index on ties
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    float z = 0.f;
    for (int c = lane; c < C; c += 64) z += expf(row[c] - mx);
    z = wave_sum(z);
    if (lane == 0) {
        auto lin = [res](int i) {
            const float start = -1.f + 1.f / (float)res, end = 1.f - 1.f / (float)res;
            const float step = (end - start) / (float)(res - 1);
            return i < res / 2 ? start + (float)i * step : end - (float)(res - 1 - i) * step;
        };
        const int idx[5] = {arg - 1, arg, arg + 1, arg - res, arg + res};
        float fx = 0.f, fy = 0.f, tot = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int c = min(max(idx[k], 0), C - 1);
            const float pr = expf(row[c] - mx) / z;
            fx += pr * lin(c % res);
            fy += pr * lin(c / res);
            tot += pr;
        }
        flow[(size_t)p * 2 + 0] = fx / tot;
        flow[(size_t)p * 2 + 1] = fy / tot;
        cert[p] = row[C];
    }
}
}  // namespace

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_cls_to_flow(const float* logits, float* flow, float* cert, int npix, int ncls, int ld, gim_stream_t stream) {
    GIM_REQUIRE(logits && flow && cert && npix > 0 && ncls > 0 && ld > ncls, "cls_to_flow: bad args");
    const int res = (int)(sqrtf((float)ncls) + 0.5f);
    GIM_REQUIRE(res * res == ncls, "cls_to_flow: %d classes are not a square grid", ncls);
    hipLaunchKernelGGL(cls_to_flow_kernel, dim3((npix + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, flow, cert, npix, ncls, res, ld);
    return gim_check_launch("cls_to_flow");
}
#endif

// SuperPoint glue kernels of the gim_lightglue path (gfx950): everything between the conv stacks and the
// matcher (networks/lightglue/superpoint.py:206-354).  All of it is HBM/L2-bound integer + fp32 work on
// maps of a few MB, written as coalesced streams; the convolutions themselves run on conv_igemm.hip.
//
//   maxpool2x2        superpoint.py:216,219,222   nn.MaxPool2d(2, 2) on NHWC rows
//   sp_scores         superpoint.py:231-236       softmax over the 65 detector logits, dustbin dropped,
//                                                 8x8 depth-to-space -> [B, 8h, 8w] fp32
//   sp_nms            superpoint.py:61-80,247-258 simple_nms (max-pool NMS, 2 suppression rounds) + borders = -1
//   sp_topk           superpoint.py:260-300       candidates > thr, top-k by score (sorted), (x, y) keypoints
//   sp_sample_desc    superpoint.py:120-137,235-241,337-341   per-pixel L2 normalisation of the dense
//                                                 descriptors, legacy bilinear sampling (align_corners),
//                                                 L2 normalisation of the sample
#include "gim_common.h"

namespace {

inline unsigned nblocks(size_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// ---- 2x2 / stride-2 max pooling, NHWC, one thread per 16-byte channel group of one output pixel -------
template <bool BF16>
__global__ void maxpool2x2_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int H, int W, int CG,
                                  int ldx, int ldy) {
    constexpr int G = BF16 ? 8 : 4;
    const int Ho = H / 2, Wo = W / 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * Ho * Wo * CG) return;
    const int cg = (int)(idx % CG);
    const size_t pix = idx / CG;
    const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), b = (int)(pix / ((size_t)Wo * Ho));
    const size_t r00 = ((size_t)b * H + 2 * yo) * W + 2 * xo;
    const size_t o[4] = {r00 * ldx, (r00 + 1) * ldx, (r00 + W) * ldx, (r00 + W + 1) * ldx};
#pragma unroll
    for (int e = 0; e < G; e += 4) {
        float4 m = ElemIO<BF16>::ld4(x, o[0] + cg * G + e);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float4 v = ElemIO<BF16>::ld4(x, o[k] + cg * G + e);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        ElemIO<BF16>::st4(y, pix * ldy + cg * G + e, m);
    }
}

// ---- detector head: one wave per 8x8 cell -----------------------------------------------------------------
template <bool BF16>
__global__ void __launch_bounds__(256) sp_scores_kernel(const void* __restrict__ logits, float* __restrict__ scores,
                                                        int cells, int h, int w, int ld) {
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= cells) return;
    const float v = ElemIO<BF16>::ld(logits, (size_t)cell * ld + lane);
    const float dust = ElemIO<BF16>::ld(logits, (size_t)cell * ld + 64);
    const float m = fmaxf(wave_max(v), dust);
    const float e = expf(v - m);
    const float z = wave_sum(e) + expf(dust - m);
    const int x = cell % w, y = (cell / w) % h, b = cell / (w * h);
    scores[((size_t)b * (8 * h) + 8 * y + (lane >> 3)) * (size_t)(8 * w) + 8 * x + (lane & 7)] = e / z;
}

// ---- simple_nms ---------------------------------------------------------------------------------------------
// F.max_pool2d pads with -inf, i.e. the window is clipped to the image.  Every pass is a (2r+1)^2 window max
// over a map; done separably per 64x16 output tile: halo tile -> LDS, row max into LDS, column max from LDS
// (2(2r+1) LDS reads per pixel instead of (2r+1)^2 global reads).
constexpr int NT_W = 64, NT_H = 16, NR_MAX = 8;
constexpr int NT_LW = NT_W + 2 * NR_MAX, NT_LH = NT_H + 2 * NR_MAX;

// window max of v around every pixel of the tile; returns it for this thread's pixels via `out[4]`
template <typename F>
__device__ __forceinline__ void tile_window_max(F load, int H, int W, int r, int x0, int y0, float* t0, float* t1, float out[4]) {
    const int lw = NT_W + 2 * r, lh = NT_H + 2 * r;
    for (int i = threadIdx.x; i < lw * lh; i += 256) {
        const int ly = i / lw, lx = i - ly * lw;
        const int y = y0 + ly - r, x = x0 + lx - r;
        t0[ly * NT_LW + lx] = (y >= 0 && y < H && x >= 0 && x < W) ? load(y, x) : -INFINITY;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NT_W * lh; i += 256) {   // horizontal max
        const int ly = i / NT_W, lx = i - ly * NT_W;
        float m = -INFINITY;
        for (int d = 0; d <= 2 * r; ++d) m = fmaxf(m, t0[ly * NT_LW + lx + d]);
        t1[ly * NT_W + lx] = m;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                           // vertical max: rows ty + 4k
        const int ly = ty + 4 * k;
        float m = -INFINITY;
        for (int d = 0; d <= 2 * r; ++d) m = fmaxf(m, t1[(ly + d) * NT_W + tx]);
        out[k] = m;
    }
    __syncthreads();
}

// keep = scores == max_pool(scores)
__global__ void __launch_bounds__(256) nms_init_kernel(const float* __restrict__ s, uint8_t* __restrict__ keep, int H, int W, int r) {
    __shared__ float t0[NT_LH * NT_LW], t1[NT_LH * NT_W];
    const int x0 = blockIdx.x * NT_W, y0 = blockIdx.y * NT_H, b = blockIdx.z;
    const float* sb = s + (size_t)b * H * W;
    float wm[4];
    tile_window_max([&](int y, int x) { return sb[(size_t)y * W + x]; }, H, W, r, x0, y0, t0, t1, wm);
    const int x = x0 + (threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + (threadIdx.x >> 6) + 4 * k;
        if (x < W && y < H) keep[((size_t)b * H + y) * W + x] = sb[(size_t)y * W + x] == wm[k];
    }
}
// supp = max_pool(keep) > 0 ; s2 = supp ? 0 : s
__global__ void __launch_bounds__(256) nms_supp_kernel(const float* __restrict__ s, const uint8_t* __restrict__ keep,
                                                       uint8_t* __restrict__ supp, float* __restrict__ s2, int H, int W, int r) {
    __shared__ float t0[NT_LH * NT_LW], t1[NT_LH * NT_W];
    const int x0 = blockIdx.x * NT_W, y0 = blockIdx.y * NT_H, b = blockIdx.z;
    const uint8_t* kb = keep + (size_t)b * H * W;
    float wm[4];
    tile_window_max([&](int y, int x) { return kb[(size_t)y * W + x] ? 1.f : 0.f; }, H, W, r, x0, y0, t0, t1, wm);
    const int x = x0 + (threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + (threadIdx.x >> 6) + 4 * k;
        if (x < W && y < H) {
            const size_t o = ((size_t)b * H + y) * W + x;
            const bool sp = wm[k] > 0.f;
            supp[o] = sp;
            s2[o] = sp ? 0.f : s[o];
        }
    }
}
// keep |= (s2 == max_pool(s2)) & ~supp ; on the last round also writes the masked score map
__global__ void __launch_bounds__(256) nms_update_kernel(const float* __restrict__ s, const float* __restrict__ s2,
                                                         const uint8_t* __restrict__ supp, uint8_t* __restrict__ keep,
                                                         float* __restrict__ out, int H, int W, int r, int border, int last) {
    __shared__ float t0[NT_LH * NT_LW], t1[NT_LH * NT_W];
    const int x0 = blockIdx.x * NT_W, y0 = blockIdx.y * NT_H, b = blockIdx.z;
    const float* s2b = s2 + (size_t)b * H * W;
    float wm[4];
    tile_window_max([&](int y, int x) { return s2b[(size_t)y * W + x]; }, H, W, r, x0, y0, t0, t1, wm);
    const int x = x0 + (threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + (threadIdx.x >> 6) + 4 * k;
        if (x >= W || y >= H) continue;
        const size_t o = ((size_t)b * H + y) * W + x;
        const bool nm = (s2[o] == wm[k]) && !supp[o];
        const bool kp = keep[o] || nm;   // keep is only read at the thread's own pixel: in-place update is safe
        keep[o] = kp;
        if (last) {
            const bool edge = border > 0 && (x < border || y < border || x >= W - border || y >= H - border);
            out[o] = edge ? -1.f : (kp ? s[o] : 0.f);
        }
    }
}

// ---- candidates + top-k ---------------------------------------------------------------------------------------
// candidate (unordered) list per image: 64-bit keys, unique, "larger is better":
//   normal      : score bits << 32 | ~index     (score > thr >= 0 => bit pattern orders like the value; ties -> lower index)
//   few (<= k)  : re-keyed as ~index << 32 so that the descending sort restores torch.where order
constexpr int CAND_PPB = 4096;  // pixels per workgroup of sp_candidates_kernel
__global__ void __launch_bounds__(256) sp_candidates_kernel(const float* __restrict__ s, unsigned long long* __restrict__ cand,
                                                            int* __restrict__ count, int HW, float thr) {
    // candidates of 4096 pixels are collected in LDS (LDS atomics), then the workgroup reserves its range of the
    // image's list with ONE global atomic (a returning global atomic per candidate, or even per 256 pixels,
    // serialises on the 16 per-image counters: 445 / 141 us for 16 maps of 640x480)
    __shared__ unsigned long long list[CAND_PPB];
    __shared__ int s_n, s_base;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int p0 = blockIdx.x * CAND_PPB;
#pragma unroll 4
    for (int k = 0; k < CAND_PPB / 256; ++k) {
        const int p = p0 + k * 256 + threadIdx.x;
        const float v = p < HW ? s[(size_t)b * HW + p] : -1.f;
        if (v > thr) list[atomicAdd(&s_n, 1)] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(~(unsigned)p);
    }
    __syncthreads();
    const int n = s_n;
    if (threadIdx.x == 0) s_base = n ? atomicAdd(count + b, n) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) cand[(size_t)b * HW + s_base + i] = list[i];
}

constexpr int TOPK_MAX = 4096;
// One 1024-thread workgroup per image: 8-pass radix select of the k-th largest key, gather, bitonic sort.
__global__ void __launch_bounds__(1024) sp_topk_kernel(const unsigned long long* __restrict__ cand, const int* __restrict__ count,
                                                       float* __restrict__ kpts, float* __restrict__ kscores,
                                                       int* __restrict__ nvalid, int HW, int W, int k, int kp2) {
    __shared__ unsigned long long keys[TOPK_MAX];
    __shared__ int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_need, s_n;
    const int b = blockIdx.x, t = threadIdx.x;
    const unsigned long long* c = cand + (size_t)b * HW;
    const int n = count[b];
    const bool few = n <= k;
    const int kk = few ? n : k;
    unsigned long long kth = 0;  // keep keys >= kth
    if (!few) {
        if (t == 0) { s_prefix = 0; s_need = k; }
        __syncthreads();
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            if (t < 256) hist[t] = 0;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            const unsigned long long pmask = pass == 0 ? 0ull : ~0ull << (shift + 8);
            for (int i = t; i < n; i += 1024) {
                const unsigned long long v = c[i];
                if ((v & pmask) == prefix) atomicAdd(&hist[(int)((v >> shift) & 255)], 1);
            }
            __syncthreads();
            if (t == 0) {
                int need = s_need, bin = 255;
                for (; bin > 0; --bin) {
                    if (hist[bin] >= need) break;
                    need -= hist[bin];
                }
                s_need = need;
                s_prefix = prefix | ((unsigned long long)bin << shift);
            }
            __syncthreads();
        }
        kth = s_prefix;  // keys are unique: exactly k keys are >= kth
    }
    if (t == 0) s_n = 0;
    for (int i = t; i < kp2; i += 1024) keys[i] = 0ull;  // padding sorts last
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const unsigned long long v = c[i];
        if (v >= kth) {
            const int pos = atomicAdd(&s_n, 1);
            // few: order by index ascending (torch.where order, superpoint.py:84-85 returns early, unsorted)
            keys[pos] = few ? (((v & 0xffffffffull) << 32) | (v >> 32)) : v;
        }
    }
    __syncthreads();
    // bitonic sort, descending, kp2 = power of two >= kk
    for (int size = 2; size <= kp2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < kp2 / 2; i += 1024) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = keys[lo], bq = keys[hi];
                if ((a < bq) == desc) { keys[lo] = bq; keys[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = t; i < k; i += 1024) {
        float x = 0.f, y = 0.f, sc = 0.f;
        if (i < kk) {
            const unsigned long long v = keys[i];
            const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
            const unsigned p = few ? ~hi : ~lo;
            sc = __uint_as_float(few ? lo : hi);
            x = (float)(p % (unsigned)W);
            y = (float)(p / (unsigned)W);
        }
        kpts[((size_t)b * k + i) * 2 + 0] = x;
        kpts[((size_t)b * k + i) * 2 + 1] = y;
        kscores[(size_t)b * k + i] = sc;
    }
    if (t == 0) nvalid[b] = kk;
}

// ---- descriptor sampling: one wave per keypoint, C = 256 (4 channels per lane) --------------------------------
template <bool BF16>
__global__ void __launch_bounds__(256) sp_sample_desc_kernel(const void* __restrict__ dense, const float* __restrict__ kpts,
                                                             float* __restrict__ out_f32, void* __restrict__ out_t, int B,
                                                             int K, int h, int w, int ld, int ld_f32, int ld_t, float s) {
    const int lane = threadIdx.x & 63;
    const int kp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (kp >= B * K) return;
    const int b = kp / K;
    // superpoint.py:123-131: (kp - s/2 + 0.5) / (w*s - s/2 - 0.5), *2 - 1; grid_sample(align_corners=True)
    // unnormalises with ((g + 1) / 2) * (size - 1)
    float gx = kpts[(size_t)kp * 2 + 0] - s / 2 + 0.5f;
    float gy = kpts[(size_t)kp * 2 + 1] - s / 2 + 0.5f;
    gx = gx / ((float)w * s - s / 2 - 0.5f);
    gy = gy / ((float)h * s - s / 2 - 0.5f);
    gx = gx * 2.f - 1.f;
    gy = gy * 2.f - 1.f;
    const float ix = ((gx + 1.f) / 2.f) * (float)(w - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
    const float wt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};  // nw, ne, sw, se
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
        if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;  // zero padding (wave-uniform branch)
        const float4 v = ElemIO<BF16>::ld4(dense, (((size_t)b * h + yy) * w + xx) * ld + lane * 4);
        // F.normalize(dense, p=2, dim=1): v / max(||v||, 1e-12)   (superpoint.py:241)
        const float nrm = fmaxf(sqrtf(wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w))), 1e-12f);
        acc.x += (v.x / nrm) * wt[k]; acc.y += (v.y / nrm) * wt[k];
        acc.z += (v.z / nrm) * wt[k]; acc.w += (v.w / nrm) * wt[k];
    }
    const float n2 = fmaxf(sqrtf(wave_sum((acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w))), 1e-12f);
    acc.x /= n2; acc.y /= n2; acc.z /= n2; acc.w /= n2;
    if (out_f32) *(float4*)(out_f32 + (size_t)kp * ld_f32 + lane * 4) = acc;
    if (out_t) ElemIO<BF16>::st4(out_t, (size_t)kp * ld_t + lane * 4, acc);
}

}  // namespace

#if !GIM_HALF_KIND
extern "C" int gim_maxpool2x2_f16(const void* x, void* y, int B, int H, int W, int C, int ldx, int ldy, int dtype,
                              gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_maxpool2x2)(const void* x, void* y, int B, int H, int W, int C, int ldx, int ldy, int dtype,
                              gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_maxpool2x2_f16(x, y, B, H, W, C, ldx, ldy, dtype, stream);   // the fp16 objects of this file
#endif
    const int G = dtype == GIM_H16 ? 8 : 4;
    GIM_REQUIRE(x && y && B > 0 && H > 1 && W > 1 && C > 0 && C % G == 0, "maxpool2x2: bad args (C=%d)", C);
    GIM_REQUIRE(ldx % G == 0 && ldy % G == 0, "maxpool2x2: row strides must keep 16-byte groups aligned");
    const size_t n = (size_t)B * (H / 2) * (W / 2) * (C / G);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GIM_H16) hipLaunchKernelGGL(maxpool2x2_kernel<true>, dim3(nblocks(n, 256)), dim3(256), 0, s, x, y, B, H, W, C / G, ldx, ldy);
    else hipLaunchKernelGGL(maxpool2x2_kernel<false>, dim3(nblocks(n, 256)), dim3(256), 0, s, x, y, B, H, W, C / G, ldx, ldy);
    return gim_check_launch("maxpool2x2");
}

#if !GIM_HALF_KIND
extern "C" int gim_sp_scores_f16(const void* logits, float* scores, int B, int h, int w, int ld, int dtype,
                             gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_sp_scores)(const void* logits, float* scores, int B, int h, int w, int ld, int dtype,
                             gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_sp_scores_f16(logits, scores, B, h, w, ld, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(logits && scores && B > 0 && h > 0 && w > 0 && ld >= 65, "sp_scores: bad args");
    const int cells = B * h * w;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GIM_H16) hipLaunchKernelGGL(sp_scores_kernel<true>, dim3((cells + 3) / 4), dim3(256), 0, s, logits, scores, cells, h, w, ld);
    else hipLaunchKernelGGL(sp_scores_kernel<false>, dim3((cells + 3) / 4), dim3(256), 0, s, logits, scores, cells, h, w, ld);
    return gim_check_launch("sp_scores");
}

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int64_t gim_sp_nms_ws_bytes(int B, int H, int W) {
    const size_t n = (size_t)B * H * W;
    return (int64_t)(((n + 255) & ~(size_t)255) * 2 + n * 4);  // keep, supp (u8), s2 (f32)
}
#endif

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_sp_nms(const float* scores, float* out, void* ws, int B, int H, int W, int radius, int border,
                          gim_stream_t stream) {
    GIM_REQUIRE(scores && out && ws && B > 0 && H > 0 && W > 0 && radius >= 0 && border >= 0, "sp_nms: bad args");
    GIM_REQUIRE(radius <= NR_MAX, "sp_nms: radius %d > %d", radius, NR_MAX);
    const size_t n = (size_t)B * H * W, na = (n + 255) & ~(size_t)255;
    uint8_t* keep = (uint8_t*)ws;
    uint8_t* supp = keep + na;
    float* s2 = (float*)(supp + na);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((W + NT_W - 1) / NT_W, (H + NT_H - 1) / NT_H, B), blk(256);
    hipLaunchKernelGGL(nms_init_kernel, grid, blk, 0, s, scores, keep, H, W, radius);
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(nms_supp_kernel, grid, blk, 0, s, scores, keep, supp, s2, H, W, radius);
        hipLaunchKernelGGL(nms_update_kernel, grid, blk, 0, s, scores, s2, supp, keep, out, H, W, radius, border, it == 1);
    }
    return gim_check_launch("sp_nms");
}
#endif

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int64_t gim_sp_topk_ws_bytes(int B, int H, int W) { return (int64_t)B * H * W * 8 + 256; }
#endif

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_sp_topk(const float* nms_scores, void* ws, float* kpts, float* kscores, int32_t* nvalid, int B,
                           int H, int W, int k, float thr, gim_stream_t stream) {
    GIM_REQUIRE(nms_scores && ws && kpts && kscores && nvalid && B > 0 && H > 0 && W > 0, "sp_topk: bad args");
    GIM_REQUIRE(k > 0 && k <= TOPK_MAX, "sp_topk: k=%d unsupported (1..%d)", k, TOPK_MAX);
    GIM_REQUIRE(thr >= 0.f, "sp_topk: threshold must be >= 0 (keys order positive floats by bit pattern)");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    int* count = (int*)ws;
    unsigned long long* cand = (unsigned long long*)((char*)ws + 256);
    GIM_REQUIRE(B * 4 <= 256, "sp_topk: at most 64 images per call");
    if (hipMemsetAsync(count, 0, 256, s) != hipSuccess) return gim_check_launch("sp_topk memset");
    hipLaunchKernelGGL(sp_candidates_kernel, dim3(nblocks(HW, CAND_PPB), B), dim3(256), 0, s, nms_scores, cand, count, HW, thr);
    int kp2 = 1;
    while (kp2 < k) kp2 <<= 1;
    hipLaunchKernelGGL(sp_topk_kernel, dim3(B), dim3(1024), 0, s, cand, count, kpts, kscores, nvalid, HW, W, k, kp2);
    return gim_check_launch("sp_topk");
}
#endif

#if !GIM_HALF_KIND
extern "C" int gim_sp_sample_desc_f16(const void* dense, const float* kpts, float* out_f32, void* out_t, int B, int K,
                                  int h, int w, int C, int ld, int ld_f32, int ld_t, int cell, int dtype,
                                  gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_sp_sample_desc)(const void* dense, const float* kpts, float* out_f32, void* out_t, int B, int K,
                                  int h, int w, int C, int ld, int ld_f32, int ld_t, int cell, int dtype,
                                  gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_sp_sample_desc_f16(dense, kpts, out_f32, out_t, B, K, h, w, C, ld, ld_f32, ld_t, cell, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(dense && kpts && (out_f32 || out_t) && B > 0 && K > 0 && h > 1 && w > 1, "sp_sample_desc: bad args");
    GIM_REQUIRE(C == 256 && ld % 4 == 0 && ld_f32 % 4 == 0 && ld_t % 4 == 0, "sp_sample_desc: C must be 256, strides % 4");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)((B * K + 3) / 4);
    if (dtype == GIM_H16) hipLaunchKernelGGL(sp_sample_desc_kernel<true>, dim3(g), dim3(256), 0, s, dense, kpts, out_f32, out_t, B, K, h, w, ld, ld_f32, ld_t, (float)cell);
    else hipLaunchKernelGGL(sp_sample_desc_kernel<false>, dim3(g), dim3(256), 0, s, dense, kpts, out_f32, out_t, B, K, h, w, ld, ld_f32, ld_t, (float)cell);
    return gim_check_launch("sp_sample_desc");
}

// LightGlue transformer kernels for gfx950 (networks/lightglue/models/matchers/lightglue.py).
//
//   lg_posenc      lightglue.py:21-33,47-61   keypoint normalisation + learnable Fourier encoding -> cos/sin table
//   lg_rotary      lightglue.py:36-44,150-151 rotary embedding applied in place to the q|k columns
//   lg_transpose   V -> V^T per sequence (key index contiguous), zero-padded to the 64-key tile
//   sdpa           lightglue.py:106-118,196-207  softmax(q k^T / sqrt(d)) v, flash-style: the [L,S] score matrix
//                  never leaves the CU.  MFMA 32x32x16 bf16 (throughput mode) or 32x32x2 f32 (parity mode).
//   layernorm_act  lightglue.py:135-139       LayerNorm(+GELU) of the FFN hidden layer
//   cast_rows      fp32 residual stream -> compute-dtype copy for the next GEMM
//
// sdpa tile scheme (one workgroup = 4 waves = 128 queries of one head, keys in tiles of 64):
//   * scores are computed TRANSPOSED, S^T = K Q^T: the MFMA C layout then gives every lane ONE query column
//     (lane & 31) and 16 key rows per 32-key block, so the row max / row sum of the softmax are in-lane
//     reductions plus one exchange with lane ^ 32 (instead of 32-lane butterflies per row).
//   * P^T stays in registers: C-layout registers [8s .. 8s+7] of a 32-key block are exactly a B operand of the
//     second MFMA (O^T = V^T P^T) if the A operand (V^T) is read with the same key permutation -- two 8-byte
//     LDS reads per fragment from a V^T tile stored [d][key] (hence lg_transpose).
//   * LDS tiles are XOR-swizzled so that both the 16-byte K reads and the 8-byte V^T reads are conflict-free.
#include "gim_common.h"
#include <math.h>

namespace {

inline unsigned nblocks(size_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// enc[row][0..31] = cos(proj), enc[row][32..63] = sin(proj), proj = Wr . normalised keypoint
__global__ void lg_posenc_kernel(const float* __restrict__ kpts, const float* __restrict__ size_wh,
                                 const float* __restrict__ Wr, float* __restrict__ enc, int rows, int K, int F) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * F) return;
    const int f = idx % F, r = idx / F, b = r / K;
    const float w = size_wh[b * 2 + 0], h = size_wh[b * 2 + 1];
    const float scale = fmaxf(w, h) / 2.f;
    const float x = (kpts[(size_t)r * 2 + 0] - w / 2.f) / scale;
    const float y = (kpts[(size_t)r * 2 + 1] - h / 2.f) / scale;
    const float p = x * Wr[f * 2 + 0] + y * Wr[f * 2 + 1];
    enc[(size_t)r * 2 * F + f] = cosf(p);
    enc[(size_t)r * 2 * F + F + f] = sinf(p);
}

// x[row][c..c+3] (two rotary pairs) for c < ncols; head dim 64 = 32 pairs, the same table for every head:
//   out[2f] = t[2f] cos_f - t[2f+1] sin_f ; out[2f+1] = t[2f+1] cos_f + t[2f] sin_f
template <bool BF16>
__global__ void lg_rotary_kernel(void* __restrict__ x, const float* __restrict__ enc, size_t rows, int ncols4, int ld) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * ncols4) return;
    const int c = (int)(idx % ncols4) * 4;
    const size_t r = idx / ncols4;
    const int f = (c & 63) >> 1;
    const float2 cs = *(const float2*)(enc + r * 64 + f), sn = *(const float2*)(enc + r * 64 + 32 + f);
    float4 t = ElemIO<BF16>::ld4(x, r * ld + c);
    float4 o;
    o.x = t.x * cs.x + (-t.y) * sn.x;
    o.y = t.y * cs.x + t.x * sn.x;
    o.z = t.z * cs.y + (-t.w) * sn.y;
    o.w = t.w * cs.y + t.z * sn.y;
    ElemIO<BF16>::st4(x, r * ld + c, o);
}

// src rows [nb*S][ld] (columns [0, C)) -> dst [nb][C][Sp], dst[s][c][key] = src[s*S + key][c], zeros for key >= S
template <bool BF16>
__global__ void lg_transpose_kernel(const void* __restrict__ src, void* __restrict__ dst, int S, int Sp, int C, int ld) {
    __shared__ float tile[64][65];
    const int s = blockIdx.z, k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int key = k0 + i, c = c0 + tx;
        tile[i][tx] = (key < S && c < C) ? ElemIO<BF16>::ld(src, ((size_t)s * S + key) * ld + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, key = k0 + tx;
        if (c < C && key < Sp) ElemIO<BF16>::st(dst, ((size_t)s * C + c) * Sp + key, tile[tx][i]);
    }
}

template <bool BF16>
__global__ void cast_rows_kernel(const float* __restrict__ src, void* __restrict__ dst, size_t rows, int C4, int lds_, int ldd) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C4) return;
    const int c = (int)(idx % C4) * 4;
    const size_t r = idx / C4;
    ElemIO<BF16>::st4(dst, r * ldd + c, *(const float4*)(src + r * lds_ + c));
}

// LayerNorm over C <= 1024 (one wave per row, fp32 input) optionally followed by exact GELU (F.gelu: 0.5 x (1 + erf(x / sqrt 2)))
template <bool BF16, bool GELU>
__global__ void __launch_bounds__(256)
layernorm_act_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                     void* __restrict__ out, int rows, int C, int ldx, int ldo, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + k * 256;
        v[k] = c < C ? *(const float4*)(x + (size_t)row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + k * 256;
        if (c < C) {
            const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + k * 256;
        if (c >= C) continue;
        const float4 g = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
        float o[4] = {(v[k].x - mean) * rstd * g.x + bt.x, (v[k].y - mean) * rstd * g.y + bt.y,
                      (v[k].z - mean) * rstd * g.z + bt.z, (v[k].w - mean) * rstd * g.w + bt.w};
        if (GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = 0.5f * o[e] * (1.f + erff(o[e] * 0.70710678118654752440f));
        }
        ElemIO<BF16>::st4(out, (size_t)row * ldo + c, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// ------------------------------------------------------------------------------------------------- sdpa
struct SdpaArgs {
    const void* q;   // rows [nb*L][ldq], head h at columns h*64
    const void* k;   // rows [nb*S][ldk]
    const void* vt;  // [nb][H*64][Sp]
    void* out;       // rows [nb*L][ldo]
    int nb, H, L, S, Sp, ldq, ldk, ldo, kv_shift;
    float scale_log2e;  // d^-0.5 * log2(e)
};

// raw v_exp_f32 (exp2f() wraps it in a 6-instruction denormal-range fix-up; arguments here are <= 0 and a
// flushed denormal probability is exactly what the softmax wants)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// D = head dimension (64: LightGlue, DINOv2 ViT-L; 128: RoMa's match decoder)
template <bool BF16, bool OUT_BF16, int D>
__global__ void __launch_bounds__(256, 2) sdpa_kernel(const SdpaArgs a) {
    constexpr int ES = BF16 ? 2 : 4;
    constexpr int KROW = D * ES;                   // bytes of one K tile row (one key, D channels)
    constexpr int VROW = 64 * ES;                  // bytes of one V^T tile row (one channel, 64 keys)
    constexpr int KSLOT = KROW / 16, VSLOT = VROW / 16;
    constexpr int KBYTES = 64 * KROW, VBYTES = D * VROW;
    constexpr int NPK = 64 * KSLOT / 256, NPV = D * VSLOT / 256;   // 16-byte chunks per thread and tile
    constexpr int NDB = D / 32;                    // 32-channel blocks of the output
    extern __shared__ __attribute__((aligned(16))) char sdpa_smem[];
    // K / V^T tiles, double-buffered: tile t lives in buffer t & 1
    char* const sK0 = sdpa_smem;
    char* const sV0 = sdpa_smem + 2 * KBYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, lh = lane >> 5;
    const int seq = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128 + wave * 32;
    const int kvseq = (seq + a.kv_shift) % a.nb;
    // 16-byte-slot swizzles of the tiles (conflict-free fragment reads): 128-byte rows pair up over the bank cycle
    auto kswz = [](int row) { return KROW == 128 ? ((row >> 1) & 7) : (row & 15); };
    // ---- Q fragments (B operand of S^T = K Q^T), held for the whole kernel --------------------------------
    const int qrow = min(q0 + l31, a.L - 1);
    const char* qp = (const char*)a.q + ((size_t)(seq * (size_t)a.L + qrow) * a.ldq + h * D) * ES;
    bf16x8_t qb[BF16 ? D / 16 : 1];
    f32x4_t qf[BF16 ? 1 : D / 8];
    if constexpr (BF16) {
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) qb[ks] = *(const bf16x8_t*)(qp + (ks * 16 + lh * 8) * 2);
    } else {
        // f32 MFMA (k = 2 per step): step st of half lh uses d = lh*(D/2) + st (any bijection works as long as A matches)
#pragma unroll
        for (int i = 0; i < D / 8; ++i) qf[i] = *(const f32x4_t*)(qp + (lh * (D / 2) + i * 4) * 4);
    }
    f32x16_t accO[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) accO[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- staging: thread t moves 16-byte chunks c = t + 256 p of the K tile (row = c / KSLOT) and of the V^T tile ----
    const char* kbase = (const char*)a.k + ((size_t)kvseq * a.S * a.ldk + h * D) * ES;
    const char* vbase = (const char*)a.vt + ((size_t)(kvseq * (size_t)a.H + h) * D * a.Sp) * ES;
    uint4 rk[NPK], rv[NPV];
    auto fetch = [&](int key0) {
#pragma unroll
        for (int p = 0; p < NPK; ++p) {
            const int c = t + 256 * p, row = c / KSLOT, slot = c % KSLOT, key = key0 + row;
            rk[p] = key < a.S ? *(const uint4*)(kbase + (size_t)key * a.ldk * ES + slot * 16) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < NPV; ++p) {
            const int c = t + 256 * p, row = c / VSLOT, slot = c % VSLOT;
            rv[p] = *(const uint4*)(vbase + ((size_t)row * a.Sp + key0) * ES + slot * 16);
        }
    };
    auto stash = [&](int buf) {
        char* sK = sK0 + buf * KBYTES;
        char* sV = sV0 + buf * VBYTES;
#pragma unroll
        for (int p = 0; p < NPK; ++p) {
            const int c = t + 256 * p, row = c / KSLOT, slot = c % KSLOT;
            *(uint4*)(sK + row * KROW + ((slot ^ kswz(row)) << 4)) = rk[p];
        }
#pragma unroll
        for (int p = 0; p < NPV; ++p) {
            const int c = t + 256 * p, row = c / VSLOT, slot = c % VSLOT;
            if constexpr (BF16) {  // V^T: 8-byte chunks 2*slot, 2*slot+1, swizzled by (d >> 1) & 15
                const int sw = (row >> 1) & 15;
                *(uint2*)(sV + row * VROW + (((2 * slot) ^ sw) << 3)) = make_uint2(rv[p].x, rv[p].y);
                *(uint2*)(sV + row * VROW + (((2 * slot + 1) ^ sw) << 3)) = make_uint2(rv[p].z, rv[p].w);
            } else {
                *(uint4*)(sV + row * VROW + ((slot ^ (row & 15)) << 4)) = rv[p];
            }
        }
    };
    // S^T = K Q^T of the tile in buffer `buf`: accS[kb][r] = key kb*32 + (r/4)*8 + lh*4 + r%4, query l31
    auto scores = [&](int buf, f32x16_t (&accS)[2]) {
        const char* sK = sK0 + buf * KBYTES;
        const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            accS[kb] = zero16;  // folded into the first MFMA's inline-constant C operand
            const int krow = kb * 32 + l31;
            const char* rp = sK + krow * KROW;
            const int sw = kswz(krow);
            if constexpr (BF16) {
#pragma unroll
                for (int ks = 0; ks < D / 16; ++ks) {
                    const bf16x8_t fa = *(const bf16x8_t*)(rp + (((2 * ks + lh) ^ sw) << 4));
                    accS[kb] = mfma_h16_32x32x16(fa, qb[ks], accS[kb]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < D / 8; ++i) {
                    const f32x4_t fa = *(const f32x4_t*)(rp + (((lh * (D / 8) + i) ^ sw) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        accS[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], qf[i][e], accS[kb], 0, 0, 0);
                }
            }
        }
    };
    // online softmax of tile kt (scores in accS) and O^T += V^T P^T with the V^T tile in buffer `buf`
    auto softmax_pv = [&](int buf, int key0, f32x16_t (&accS)[2]) {
        const char* sV = sV0 + buf * VBYTES;
        if (key0 + 64 > a.S) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kb * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
                    if (key >= a.S) accS[kb][r] = -INFINITY;
                }
        }
        float mx = accS[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, accS[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float mb = m_new * a.scale_log2e;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(accS[kb][r] * a.scale_log2e - mb);
                accS[kb][r] = p;
                psum += p;
            }
        // rescale only when some query of the wave saw a new maximum (rare after the first tiles; wave-uniform
        // branch).  Per-half partial sums share alpha; the halves are added at the end.
        if (__builtin_amdgcn_ballot_w64(m_new != m_run)) {
            const float alpha = fast_exp2((m_run - m_new) * a.scale_log2e);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) accO[d][r] *= alpha;
        }
        l_run += psum;
        m_run = m_new;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if constexpr (BF16) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    union { unsigned u[4]; bf16x8_t v; } pb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pb.u[e] = cvt_pk_h16(accS[kb][8 * s + 2 * e], accS[kb][8 * s + 2 * e + 1]);
                    const int c = kb * 8 + 4 * s + lh;  // 8-byte chunk of keys kb*32 + 16 s + lh*4 .. +3 ; c + 2 = +8 keys
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        const int drow = d * 32 + l31;
                        const int sw = (drow >> 1) & 15;
                        union { uint2 h2[2]; bf16x8_t v; } va;
                        va.h2[0] = *(const uint2*)(sV + drow * VROW + ((c ^ sw) << 3));
                        va.h2[1] = *(const uint2*)(sV + drow * VROW + (((c + 2) ^ sw) << 3));
                        accO[d] = mfma_h16_32x32x16(va.v, pb.v, accO[d]);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 = keys kb*32 + g*8 + lh*4 + 0..3
                    const int slot = kb * 8 + g * 2 + lh;
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        const int drow = d * 32 + l31;
                        const f32x4_t va = *(const f32x4_t*)(sV + drow * VROW + ((slot ^ (drow & 15)) << 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            accO[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e], accS[kb][4 * g + e], accO[d], 0, 0, 0);
                    }
                }
            }
        }
    };
    // Double-buffered tiles, ONE barrier per tile: while tile kt (buffer kt & 1) is consumed, tile kt+1 -- fetched
    // into registers one step earlier -- is written to the other buffer, which every wave left at the previous
    // barrier.  (Issuing the next tile's score MFMAs ahead of this tile's softmax inside one wave was measured
    // slower, 69 vs 64 us: in-order issue stalls the VALU behind the dependent MFMA chain, and the second wave
    // on the SIMD already fills those gaps.)
    const int ntiles = (a.S + 63) / 64;
    f32x16_t accS[2];
    fetch(0);
    stash(0);
    if (ntiles > 1) fetch(64);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int b = kt & 1;
        scores(b, accS);
        softmax_pv(b, kt * 64, accS);
        if (kt + 1 < ntiles) stash(b ^ 1);
        if (kt + 2 < ntiles) fetch((kt + 2) * 64);
        __syncthreads();
    }
    // ---- normalise and store: lane holds query l31, d = db*32 + (r/4)*8 + lh*4 + r%4 ---------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    const int q = q0 + l31;
    if (q < a.L) {
        const size_t orow = (size_t)(seq * (size_t)a.L + q) * a.ldo + h * D;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 o = make_float4(accO[d][rg * 4 + 0] * inv, accO[d][rg * 4 + 1] * inv,
                                             accO[d][rg * 4 + 2] * inv, accO[d][rg * 4 + 3] * inv);
                ElemIO<OUT_BF16>::st4(a.out, orow + d * 32 + rg * 8 + lh * 4, o);
            }
    }
}

template <bool BF16, bool OUT_BF16, int D>
int launch_sdpa(const SdpaArgs& a, hipStream_t s) {
    constexpr int ES = BF16 ? 2 : 4;
    constexpr int smem = 2 * (64 * D * ES + D * 64 * ES);
    auto kern = sdpa_kernel<BF16, OUT_BF16, D>;
    static GimPerDevice attr;
    if (attr.needed()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            gim_set_error("sdpa: hipFuncSetAttribute(%d B LDS) failed", smem);
            return GIM_ERR_LAUNCH;
        }
        attr.done();
    }
    const dim3 grid((a.L + 127) / 128, a.H, a.nb), blk(256);
    hipLaunchKernelGGL(kern, grid, blk, smem, s, a);
    return gim_check_launch("sdpa");
}

}  // namespace

#if !GIM_HALF_KIND   // no 16-bit operand: one copy, in the bf16 objects
extern "C" int gim_lg_posenc(const float* kpts, const float* size_wh, const float* Wr, float* enc, int B, int K,
                             gim_stream_t stream) {
    GIM_REQUIRE(kpts && size_wh && Wr && enc && B > 0 && K > 0, "lg_posenc: bad args");
    const int n = B * K * 32;
    hipLaunchKernelGGL(lg_posenc_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, kpts, size_wh, Wr, enc, B * K, K, 32);
    return gim_check_launch("lg_posenc");
}
#endif

#if !GIM_HALF_KIND
extern "C" int gim_lg_rotary_f16(void* x, const float* enc, int rows, int ncols, int ld, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_lg_rotary)(void* x, const float* enc, int rows, int ncols, int ld, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_lg_rotary_f16(x, enc, rows, ncols, ld, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && enc && rows > 0 && ncols > 0 && ncols % 64 == 0 && ld >= ncols && ld % 4 == 0, "lg_rotary: bad args");
    const size_t n = (size_t)rows * (ncols / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GIM_H16) hipLaunchKernelGGL(lg_rotary_kernel<true>, dim3(nblocks(n, 256)), dim3(256), 0, s, x, enc, (size_t)rows, ncols / 4, ld);
    else hipLaunchKernelGGL(lg_rotary_kernel<false>, dim3(nblocks(n, 256)), dim3(256), 0, s, x, enc, (size_t)rows, ncols / 4, ld);
    return gim_check_launch("lg_rotary");
}

#if !GIM_HALF_KIND
extern "C" int gim_lg_transpose_f16(const void* src, void* dst, int nb, int S, int Sp, int C, int ld, int dtype,
                                gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_lg_transpose)(const void* src, void* dst, int nb, int S, int Sp, int C, int ld, int dtype,
                                gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_lg_transpose_f16(src, dst, nb, S, Sp, C, ld, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(src && dst && nb > 0 && S > 0 && Sp >= S && Sp % 64 == 0 && C > 0 && ld >= C, "lg_transpose: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(Sp / 64, (C + 63) / 64, nb);
    if (dtype == GIM_H16) hipLaunchKernelGGL(lg_transpose_kernel<true>, grid, dim3(256), 0, s, src, dst, S, Sp, C, ld);
    else hipLaunchKernelGGL(lg_transpose_kernel<false>, grid, dim3(256), 0, s, src, dst, S, Sp, C, ld);
    return gim_check_launch("lg_transpose");
}

#if !GIM_HALF_KIND
extern "C" int gim_cast_rows_f16(const float* src, void* dst, int rows, int C, int ld_src, int ld_dst, int dtype,
                             gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_cast_rows)(const float* src, void* dst, int rows, int C, int ld_src, int ld_dst, int dtype,
                             gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_cast_rows_f16(src, dst, rows, C, ld_src, ld_dst, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(src && dst && rows > 0 && C > 0 && C % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0, "cast_rows: bad args");
    const size_t n = (size_t)rows * (C / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GIM_H16) hipLaunchKernelGGL(cast_rows_kernel<true>, dim3(nblocks(n, 256)), dim3(256), 0, s, src, dst, (size_t)rows, C / 4, ld_src, ld_dst);
    else hipLaunchKernelGGL(cast_rows_kernel<false>, dim3(nblocks(n, 256)), dim3(256), 0, s, src, dst, (size_t)rows, C / 4, ld_src, ld_dst);
    return gim_check_launch("cast_rows");
}

#if !GIM_HALF_KIND
extern "C" int gim_layernorm_act_f16(const float* x, const float* gamma, const float* beta, void* out, int rows, int C,
                                 int ldx, int ldo, int act, int out_dtype, float eps, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_layernorm_act)(const float* x, const float* gamma, const float* beta, void* out, int rows, int C,
                                 int ldx, int ldo, int act, int out_dtype, float eps, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (out_dtype == GIM_F16) return gim_layernorm_act_f16(x, gamma, beta, out, rows, C, ldx, ldo, act, out_dtype, eps, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(x && gamma && beta && out && rows > 0, "layernorm_act: bad args");
    GIM_REQUIRE(C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ldo % 4 == 0, "layernorm_act: C=%d (multiple of 4, <= 1024)", C);
    GIM_REQUIRE(act == GIM_ACT_NONE || act == GIM_ACT_GELU, "layernorm_act: act must be NONE or GELU");
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((rows + 3) / 4), b(256);
    const bool bf = out_dtype == GIM_H16, ge = act == GIM_ACT_GELU;
    if (bf && ge) hipLaunchKernelGGL((layernorm_act_kernel<true, true>), g, b, 0, s, x, gamma, beta, out, rows, C, ldx, ldo, eps);
    else if (bf) hipLaunchKernelGGL((layernorm_act_kernel<true, false>), g, b, 0, s, x, gamma, beta, out, rows, C, ldx, ldo, eps);
    else if (ge) hipLaunchKernelGGL((layernorm_act_kernel<false, true>), g, b, 0, s, x, gamma, beta, out, rows, C, ldx, ldo, eps);
    else hipLaunchKernelGGL((layernorm_act_kernel<false, false>), g, b, 0, s, x, gamma, beta, out, rows, C, ldx, ldo, eps);
    return gim_check_launch("layernorm_act");
}

#if !GIM_HALF_KIND
extern "C" int gim_sdpa_f16(const void* q, const void* k, const void* vt, void* out, int nb, int H, int L, int S, int Sp,
                        int D, int ldq, int ldk, int ldo, int kv_shift, int dtype, int out_dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_sdpa)(const void* q, const void* k, const void* vt, void* out, int nb, int H, int L, int S, int Sp,
                        int D, int ldq, int ldk, int ldo, int kv_shift, int dtype, int out_dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16 || out_dtype == GIM_F16) return gim_sdpa_f16(q, k, vt, out, nb, H, L, S, Sp, D, ldq, ldk, ldo, kv_shift, dtype, out_dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(q && k && vt && out && nb > 0 && H > 0 && L > 0 && S > 0, "sdpa: bad args");
    GIM_REQUIRE(D == 64 || D == 128, "sdpa: head dim %d unsupported (64, 128)", D);
    GIM_REQUIRE(Sp >= S && Sp % 64 == 0, "sdpa: Sp=%d must be S rounded up to a multiple of 64", Sp);
    const int g = dtype == GIM_H16 ? 8 : 4;
    GIM_REQUIRE(ldq % g == 0 && ldk % g == 0 && ldo % 4 == 0, "sdpa: row strides must keep 16-byte alignment");
    GIM_REQUIRE(kv_shift >= 0 && kv_shift < nb, "sdpa: kv_shift");
    SdpaArgs a;
    a.q = q; a.k = k; a.vt = vt; a.out = out;
    a.nb = nb; a.H = H; a.L = L; a.S = S; a.Sp = Sp; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.kv_shift = kv_shift;
    a.scale_log2e = (1.0f / sqrtf((float)D)) * 1.44269504088896340736f;
    hipStream_t s = (hipStream_t)stream;
    const bool bf = dtype == GIM_H16, obf = out_dtype == GIM_H16;
    if (D == 64) {
        if (bf && obf) return launch_sdpa<true, true, 64>(a, s);
        if (bf) return launch_sdpa<true, false, 64>(a, s);
        if (obf) return launch_sdpa<false, true, 64>(a, s);
        return launch_sdpa<false, false, 64>(a, s);
    }
    if (bf && obf) return launch_sdpa<true, true, 128>(a, s);
    if (bf) return launch_sdpa<true, false, 128>(a, s);
    if (obf) return launch_sdpa<false, true, 128>(a, s);
    return launch_sdpa<false, false, 128>(a, s);
}

// LightGlue match assignment for gfx950, fused (networks/lightglue/models/matchers/lightglue.py:226-300):
//
//   sim[b,i,j]    = (md0[b,i] / d^(1/4)) . (md1[b,j] / d^(1/4))                    MatchAssignment.forward :261-264
//   scores[b,i,j] = log_softmax_j(sim) + log_softmax_i(sim) + logsigmoid(z0_i) + logsigmoid(z1_j)   :226-238
//   m0[i] = argmax_j scores, m1[j] = argmax_i scores, mutual check, exp(max) > th    filter_matches :284-300
//
// without writing the [B, M+1, N+1] matrix (16.8 MB per pair at 2048 keypoints) unless a caller asks for
// `log_assignment` (gim_lg_log_assignment).  Three passes over 128x128 similarity tiles on the fp32 MFMA
// (exact fp32 products: the argmax indices must equal the fp32 CPU reference's):
//   lga_prep     matchability logits z = w . desc + b, logsigmoid(z), logsigmoid(-z)   (one wave per keypoint)
//   lga_stats    per-tile row / column (max, sum exp) partials        lga_combine -> (max, log sum)
//   lga_best     tile recomputed, scores formed in the reference's operation order, row / column maxima
//                folded into one 64-bit atomicMax per row (column) and tile: ordered(score) << 32 | ~index
//                (ties -> lowest index, like torch.max)
//   lga_filter   one workgroup per pair: mutual / threshold logic, matches0/1, matching_scores0/1 and the
//                compacted `matches` / `scores` lists in torch.where order (+ the caller-side adapter of
//                trainer/lightning.py:176-183: matched keypoints scaled to original-image pixels)
#include "igemm_mainloop.h"

namespace {

constexpr int BM = 128, BN = 128, WM = 2, WN = 2;
constexpr int TLD = BN + 4;
constexpr int TILE_SMEM = BM * TLD * 4 + 8 * 128 * 4 + 64;
static_assert(TILE_SMEM >= gim::mainloop_smem_bytes<BM, BN>(), "stage buffers must fit in the tile allocation");

struct LgaWs {
    float2* rowpart;  // [B][ntN][M]
    float2* colpart;  // [B][ntM][N]
    float2* rowstat;  // [B][M] (max, log sum exp(x - max))
    float2* colstat;  // [B][N]
    float* ls0;       // [B][M] logsigmoid(z0)
    float* ls1;       // [B][N]
    float* lsn0;      // [B][M] logsigmoid(-z0)
    float* lsn1;      // [B][N]
    unsigned long long* best0;  // [B][M]
    unsigned long long* best1;  // [B][N]
    int* count;       // [B] matches per pair
    int* ktab;
    int ntM, ntN;
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

size_t carve(LgaWs& w, char* base, int B, int M, int N, int C) {
    w.ntM = (M + BM - 1) / BM;
    w.ntN = (N + BN - 1) / BN;
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += al(bytes); return p; };
    w.rowpart = (float2*)take((size_t)B * w.ntN * M * 8);
    w.colpart = (float2*)take((size_t)B * w.ntM * N * 8);
    w.rowstat = (float2*)take((size_t)B * M * 8);
    w.colstat = (float2*)take((size_t)B * N * 8);
    w.ls0 = (float*)take((size_t)B * M * 4);
    w.ls1 = (float*)take((size_t)B * N * 4);
    w.lsn0 = (float*)take((size_t)B * M * 4);
    w.lsn1 = (float*)take((size_t)B * N * 4);
    w.best0 = (unsigned long long*)take((size_t)B * M * 8);
    w.best1 = (unsigned long long*)take((size_t)B * N * 8);
    w.count = (int*)take((size_t)B * 4);
    w.ktab = (int*)take((size_t)(C / 32 + 2) * 8 * 4);
    return o;
}

struct LgaGeom {
    const float* md0;  // [B][M][C] final_proj outputs (unscaled)
    const float* md1;  // [B][N][C]
    int B, M, N, C;
    float inv_sqrt_d;  // 1 / (d^(1/4))^2
};

__device__ __forceinline__ float logsigmoid(float x) {  // ATen: min(x, 0) - log1p(exp(-|x|))
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// monotone map float -> uint (handles negative scores)
__device__ __forceinline__ unsigned ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void lga_ktab_kernel(int* ktab, int C) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int ng = (C / 32 + 2) * 8;
    if (g < ng) ktab[g] = (g * 4 < C) ? g * 4 : (int)0xFF000000;
}

// one wave per keypoint: z = desc . w + b  (C = 256)
__global__ void __launch_bounds__(256) lga_prep_kernel(const float* __restrict__ desc, const float* __restrict__ wgt,
                                                       const float* __restrict__ bias, float* __restrict__ ls,
                                                       float* __restrict__ lsn, unsigned long long* __restrict__ best,
                                                       int rows, int ld) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float4 v = *(const float4*)(desc + (size_t)r * ld + lane * 4);
    const float4 w = *(const float4*)(wgt + lane * 4);
    const float z = wave_sum((v.x * w.x + v.y * w.y) + (v.z * w.z + v.w * w.w)) + bias[0];
    if (lane == 0) {
        ls[r] = logsigmoid(z);
        lsn[r] = logsigmoid(-z);
        best[r] = 0ull;
    }
}

__device__ __forceinline__ void sim_tile_to_lds(const LgaGeom& g, const int* ktab, int b, int m0, int n0, char* smem) {
    gim::MainloopArgs ml;
    ml.x = g.md0 + (size_t)b * g.M * g.C;
    ml.w = g.md1 + (size_t)b * g.N * g.C;
    ml.ktab = ktab;
    ml.x_bytes = (unsigned)((size_t)g.M * g.C * 4);
    ml.w_bytes = (unsigned)((size_t)g.N * g.C * 4);
    ml.H = 1; ml.W = g.M; ml.Ho = 1; ml.Wo = g.M; ml.stride = 1; ml.pad = 0; ml.ldx = g.C;
    ml.kpad = g.C; ml.ldw = g.C; ml.M = g.M;
    f32x16_t acc[2][2];
    gim::igemm_mainloop<BM, BN, WM, WN, false, true>(ml, smem, m0, n0, acc);
    float* St = (float*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5, wm = wave / WN, wn = wave - wm * WN;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i_loc = wm * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int j_loc = wn * 64 + i * 32 + rg * 8 + lh * 4;
                *(float4*)(St + i_loc * TLD + j_loc) =
                    make_float4(acc[i][j][rg * 4 + 0] * g.inv_sqrt_d, acc[i][j][rg * 4 + 1] * g.inv_sqrt_d,
                                acc[i][j][rg * 4 + 2] * g.inv_sqrt_d, acc[i][j][rg * 4 + 3] * g.inv_sqrt_d);
            }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) lga_stats_kernel(const LgaGeom g, const LgaWs w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    const int mt = blockIdx.x / w.ntN, nt = blockIdx.x - mt * w.ntN;
    const int m0 = mt * BM, n0 = nt * BN;
    sim_tile_to_lds(g, w.ktab, b, m0, n0, smem);
    const float* St = (const float*)smem;
    float* red = (float*)(smem + BM * TLD * 4);
    const int t = threadIdx.x, idx = t & 127, half = t >> 7;
    {   // rows
        float mx = -INFINITY;
        for (int jj = 0; jj < 64; ++jj) {
            const int j = half * 64 + jj;
            if (n0 + j < g.N) mx = fmaxf(mx, St[idx * TLD + j]);
        }
        red[half * 128 + idx] = mx;
        __syncthreads();
        const float m = fmaxf(red[idx], red[128 + idx]);
        float z = 0.f;
        for (int jj = 0; jj < 64; ++jj) {
            const int j = half * 64 + jj;
            if (n0 + j < g.N) z += __expf(St[idx * TLD + j] - m);
        }
        red[256 + half * 128 + idx] = z;
        __syncthreads();
        if (half == 0 && m0 + idx < g.M)
            w.rowpart[((size_t)b * w.ntN + nt) * g.M + m0 + idx] = make_float2(m, red[256 + idx] + red[384 + idx]);
        __syncthreads();
    }
    {   // columns
        float mx = -INFINITY;
        for (int r = half * 64; r < half * 64 + 64; ++r)
            if (m0 + r < g.M) mx = fmaxf(mx, St[r * TLD + idx]);
        red[half * 128 + idx] = mx;
        __syncthreads();
        const float m = fmaxf(red[idx], red[128 + idx]);
        float z = 0.f;
        for (int r = half * 64; r < half * 64 + 64; ++r)
            if (m0 + r < g.M) z += __expf(St[r * TLD + idx] - m);
        red[256 + half * 128 + idx] = z;
        __syncthreads();
        if (half == 0 && n0 + idx < g.N)
            w.colpart[((size_t)b * w.ntM + mt) * g.N + n0 + idx] = make_float2(m, red[256 + idx] + red[384 + idx]);
    }
}

// stat = (max over tiles, log(sum_t z_t exp(m_t - max)))
__global__ void lga_combine_kernel(const float2* __restrict__ part, float2* __restrict__ stat, int B, int len, int ntile) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * len) return;
    const size_t n = idx / len, x = idx - n * len;
    float m = -INFINITY;
    for (int t = 0; t < ntile; ++t) m = fmaxf(m, part[(n * ntile + t) * len + x].x);
    float z = 0.f;
    for (int t = 0; t < ntile; ++t) {
        const float2 p = part[(n * ntile + t) * len + x];
        z += p.y * expf(p.x - m);
    }
    stat[idx] = make_float2(m, logf(z));
}

// MODE 0: row / column best.  MODE 1: write the core of log_assignment [B][M+1][N+1].
template <int MODE>
__global__ void __launch_bounds__(256) lga_best_kernel(const LgaGeom g, const LgaWs w, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    const int mt = blockIdx.x / w.ntN, nt = blockIdx.x - mt * w.ntN;
    const int m0 = mt * BM, n0 = nt * BN;
    sim_tile_to_lds(g, w.ktab, b, m0, n0, smem);
    float* St = (float*)smem;
    float* sc = (float*)(smem + BM * TLD * 4);
    float* rmax = sc;          // [128] row max of the softmax statistics
    float* rlz = sc + 128;     // [128] row log-sum
    float* rls = sc + 256;     // [128] logsigmoid(z0)
    float* cmax = sc + 384;
    float* clz = sc + 512;
    float* cls = sc + 640;
    unsigned long long* red = (unsigned long long*)(sc + 768);  // [128] second-half partial best
    const int t = threadIdx.x, idx = t & 127, half = t >> 7;
    if (half == 0) {
        const bool ok = m0 + idx < g.M;
        const float2 s = ok ? w.rowstat[(size_t)b * g.M + m0 + idx] : make_float2(0.f, 0.f);
        rmax[idx] = s.x; rlz[idx] = s.y;
        rls[idx] = ok ? w.ls0[(size_t)b * g.M + m0 + idx] : 0.f;
    } else {
        const bool ok = n0 + idx < g.N;
        const float2 s = ok ? w.colstat[(size_t)b * g.N + n0 + idx] : make_float2(0.f, 0.f);
        cmax[idx] = s.x; clz[idx] = s.y;
        cls[idx] = ok ? w.ls1[(size_t)b * g.N + n0 + idx] : 0.f;
    }
    __syncthreads();
    // scores in place, in the reference's order: (log_softmax_j + log_softmax_i) + (ls0 + ls1)
    for (int e = t; e < BM * BN / 4; e += 256) {
        const int i = e >> 5, j = (e & 31) * 4;
        float4 v = *(float4*)(St + i * TLD + j);
        float* pv = (float*)&v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float s = pv[q];
            const float s0 = (s - rmax[i]) - rlz[i];
            const float s1 = (s - cmax[j + q]) - clz[j + q];
            pv[q] = (s0 + s1) + (rls[i] + cls[j + q]);
        }
        *(float4*)(St + i * TLD + j) = v;
    }
    __syncthreads();
    if (MODE == 1) {
        for (int e = t; e < BM * BN; e += 256) {
            const int i = e >> 7, j = e & 127;
            if (m0 + i < g.M && n0 + j < g.N)
                out[((size_t)b * (g.M + 1) + m0 + i) * (size_t)(g.N + 1) + n0 + j] = St[i * TLD + j];
        }
        return;
    }
    {   // row best: thread (row idx, column half); ties -> lowest j
        unsigned long long best = 0ull;
        if (m0 + idx < g.M) {
            for (int jj = 0; jj < 64; ++jj) {
                const int j = half * 64 + jj;
                if (n0 + j >= g.N) break;
                const unsigned long long key = ((unsigned long long)ord(St[idx * TLD + j]) << 32) | (unsigned)(~(unsigned)(n0 + j));
                best = key > best ? key : best;
            }
        }
        if (half == 1) red[idx] = best;
        __syncthreads();
        if (half == 0 && m0 + idx < g.M) {
            const unsigned long long o = red[idx];
            best = o > best ? o : best;
            atomicMax(&w.best0[(size_t)b * g.M + m0 + idx], best);
        }
        __syncthreads();
    }
    {   // column best: thread (column idx, row half); ties -> lowest i
        unsigned long long best = 0ull;
        if (n0 + idx < g.N) {
            for (int r = half * 64; r < half * 64 + 64; ++r) {
                if (m0 + r >= g.M) break;
                const unsigned long long key = ((unsigned long long)ord(St[r * TLD + idx]) << 32) | (unsigned)(~(unsigned)(m0 + r));
                best = key > best ? key : best;
            }
        }
        if (half == 1) red[idx] = best;
        __syncthreads();
        if (half == 0 && n0 + idx < g.N) {
            const unsigned long long o = red[idx];
            best = o > best ? o : best;
            atomicMax(&w.best1[(size_t)b * g.N + n0 + idx], best);
        }
    }
}

// dustbin row / column of log_assignment (lightglue.py:236-237), corner = 0
__global__ void lga_border_kernel(const LgaWs w, float* __restrict__ out, int B, int M, int N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (idx < M) out[((size_t)b * (M + 1) + idx) * (size_t)(N + 1) + N] = w.lsn0[(size_t)b * M + idx];
    if (idx < N) out[((size_t)b * (M + 1) + M) * (size_t)(N + 1) + idx] = w.lsn1[(size_t)b * N + idx];
    if (idx == 0) out[((size_t)b * (M + 1) + M) * (size_t)(N + 1) + N] = 0.f;
}

struct LgaOut {
    int64_t* matches0;   // [B][M]
    int64_t* matches1;   // [B][N]
    float* mscores0;     // [B][M]
    float* mscores1;     // [B][N]
    int32_t* pos;        // [B][M] scratch: position of row i in its pair's match list
};

// one workgroup per pair
__global__ void __launch_bounds__(1024) lga_filter_kernel(const LgaWs w, const LgaOut o, int M, int N, float th) {
    __shared__ int warp_tot[16];
    __shared__ int s_base;
    const int b = blockIdx.x, t = threadIdx.x;
    const unsigned long long* b0 = w.best0 + (size_t)b * M;
    const unsigned long long* b1 = w.best1 + (size_t)b * N;
    // rows
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < M; i0 += 1024) {
        const int i = i0 + t;
        bool valid = false;
        if (i < M) {
            const unsigned long long k = b0[i];
            int j = (int)(~(unsigned)k);
            if ((unsigned)j >= (unsigned)N) j = 0;  // only reachable with NaN scores
            const int back = (int)(~(unsigned)b1[j]);
            const bool mutual = back == i;
            const float ms = mutual ? expf(unord((unsigned)(k >> 32))) : 0.f;
            valid = mutual && ms > th;
            o.matches0[(size_t)b * M + i] = valid ? (int64_t)j : -1;
            o.mscores0[(size_t)b * M + i] = ms;
        }
        // ordered compaction (torch.where order): wave ballot + block scan
        const unsigned long long bal = __ballot(valid);
        const int lane = t & 63, wv = t >> 6;
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) warp_tot[wv] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int q = 0; q < wv; ++q) off += warp_tot[q];
        if (i < M) o.pos[(size_t)b * M + i] = valid ? off + pre : -1;
        __syncthreads();
        if (t == 0) {
            int tot = 0;
            for (int q = 0; q < 16; ++q) tot += warp_tot[q];
            s_base += tot;
        }
        __syncthreads();
    }
    if (t == 0) w.count[b] = s_base;
    // columns: mscores1 = mutual1 ? mscores0[m1] : 0 ; valid1 = mutual1 & valid0[m1]
    for (int j = t; j < N; j += 1024) {
        const unsigned long long k = b1[j];
        int i = (int)(~(unsigned)k);
        if ((unsigned)i >= (unsigned)M) i = 0;
        const unsigned long long k0 = b0[i];
        const bool mutual = (int)(~(unsigned)k0) == j;
        const float ms0 = mutual ? expf(unord((unsigned)(k0 >> 32))) : 0.f;  // mutual1 => mutual0 for row i
        const bool valid = mutual && ms0 > th;
        o.matches1[(size_t)b * N + j] = valid ? (int64_t)i : -1;
        o.mscores1[(size_t)b * N + j] = ms0;
    }
}

struct LgaEmit {
    const int64_t* matches0;
    const float* mscores0;
    const int32_t* pos;
    const int* count;
    const float* kpts0;   // [B][M][2] or NULL
    const float* kpts1;   // [B][N][2]
    const float* scale0;  // [B][2] or NULL
    const float* scale1;
    int64_t* matches;     // [total][2]
    float* scores;        // [total]
    float* mkpts0;        // [total][2] or NULL
    float* mkpts1;
    int64_t* m_bids;      // [total] or NULL
};

__global__ void lga_emit_kernel(const LgaEmit e, int B, int M, int N) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int p = e.pos[(size_t)b * M + i];
    if (p < 0) return;
    int off = 0;
    for (int q = 0; q < b; ++q) off += e.count[q];
    const size_t d = (size_t)off + p;
    const int64_t j = e.matches0[(size_t)b * M + i];
    e.matches[d * 2 + 0] = i;
    e.matches[d * 2 + 1] = j;
    e.scores[d] = e.mscores0[(size_t)b * M + i];
    if (e.mkpts0) {
        const float sx0 = e.scale0 ? e.scale0[b * 2 + 0] : 1.f, sy0 = e.scale0 ? e.scale0[b * 2 + 1] : 1.f;
        const float sx1 = e.scale1 ? e.scale1[b * 2 + 0] : 1.f, sy1 = e.scale1 ? e.scale1[b * 2 + 1] : 1.f;
        e.mkpts0[d * 2 + 0] = e.kpts0[((size_t)b * M + i) * 2 + 0] * sx0;
        e.mkpts0[d * 2 + 1] = e.kpts0[((size_t)b * M + i) * 2 + 1] * sy0;
        e.mkpts1[d * 2 + 0] = e.kpts1[((size_t)b * N + j) * 2 + 0] * sx1;
        e.mkpts1[d * 2 + 1] = e.kpts1[((size_t)b * N + j) * 2 + 1] * sy1;
        e.m_bids[d] = b;
    }
}

int check_args(const gim_lg_assign_args* a) {
    GIM_REQUIRE(a && a->desc0 && a->desc1 && a->md0 && a->md1 && a->match_w && a->match_b && a->ws, "lg_assign: null pointer");
    GIM_REQUIRE(a->B > 0 && a->M > 0 && a->N > 0 && a->B <= 65535, "lg_assign: bad sizes");
    GIM_REQUIRE(a->C == 256, "lg_assign: descriptor dim %d unsupported (256)", a->C);
    GIM_REQUIRE((size_t)a->M * a->C * 4 < (1ull << 31) && (size_t)a->N * a->C * 4 < (1ull << 31), "lg_assign: too many keypoints");
    return GIM_OK;
}

}  // namespace

extern "C" int64_t gim_lg_assign_ws_bytes(int B, int M, int N, int C) {
    LgaWs w;
    return (int64_t)carve(w, nullptr, B, M, N, C);
}

static int lga_run_stats(const gim_lg_assign_args* a, LgaWs& w, LgaGeom& g, hipStream_t s) {
    carve(w, (char*)a->ws, a->B, a->M, a->N, a->C);
    g.md0 = a->md0; g.md1 = a->md1; g.B = a->B; g.M = a->M; g.N = a->N; g.C = a->C;
    g.inv_sqrt_d = 1.0f / sqrtf((float)a->C);
    static GimPerDevice attr_set;
    if (attr_set.needed()) {
        const void* kerns[3] = {(const void*)lga_stats_kernel, (const void*)lga_best_kernel<0>, (const void*)lga_best_kernel<1>};
        for (const void* k : kerns) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_SMEM);
            if (e != hipSuccess) { gim_set_error("lg_assign: hipFuncSetAttribute: %s", hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
        }
        attr_set.done();
    }
    hipLaunchKernelGGL(lga_ktab_kernel, dim3(1), dim3(256), 0, s, w.ktab, a->C);
    hipLaunchKernelGGL(lga_prep_kernel, dim3((a->B * a->M + 3) / 4), dim3(256), 0, s, a->desc0, a->match_w, a->match_b, w.ls0, w.lsn0, w.best0, a->B * a->M, a->ld_desc);
    hipLaunchKernelGGL(lga_prep_kernel, dim3((a->B * a->N + 3) / 4), dim3(256), 0, s, a->desc1, a->match_w, a->match_b, w.ls1, w.lsn1, w.best1, a->B * a->N, a->ld_desc);
    const dim3 grid(w.ntM * w.ntN, a->B);
    hipLaunchKernelGGL(lga_stats_kernel, grid, dim3(256), TILE_SMEM, s, g, w);
    hipLaunchKernelGGL(lga_combine_kernel, dim3((unsigned)(((size_t)a->B * a->M + 255) / 256)), dim3(256), 0, s, w.rowpart, w.rowstat, a->B, a->M, w.ntN);
    hipLaunchKernelGGL(lga_combine_kernel, dim3((unsigned)(((size_t)a->B * a->N + 255) / 256)), dim3(256), 0, s, w.colpart, w.colstat, a->B, a->N, w.ntM);
    return GIM_OK;
}

extern "C" int gim_lg_assign(const gim_lg_assign_args* a, gim_stream_t stream) {
    if (int rc = check_args(a)) return rc;
    GIM_REQUIRE(a->matches0 && a->matches1 && a->mscores0 && a->mscores1 && a->pos && a->count, "lg_assign: null output");
    hipStream_t s = (hipStream_t)stream;
    LgaWs w;
    LgaGeom g;
    lga_run_stats(a, w, g, s);
    const dim3 grid(w.ntM * w.ntN, a->B);
    hipLaunchKernelGGL(lga_best_kernel<0>, grid, dim3(256), TILE_SMEM, s, g, w, (float*)nullptr);
    LgaOut o;
    o.matches0 = a->matches0; o.matches1 = a->matches1; o.mscores0 = a->mscores0; o.mscores1 = a->mscores1; o.pos = a->pos;
    hipLaunchKernelGGL(lga_filter_kernel, dim3(a->B), dim3(1024), 0, s, w, o, a->M, a->N, a->threshold);
    if (hipMemcpyAsync(a->count, w.count, (size_t)a->B * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return gim_check_launch("lg_assign count copy");
    return gim_check_launch("lg_assign");
}

extern "C" int gim_lg_emit_matches(const int64_t* matches0, const float* mscores0, const int32_t* pos, const int32_t* count,
                                   const float* kpts0, const float* kpts1, const float* scale0, const float* scale1,
                                   int64_t* matches, float* scores, float* mkpts0, float* mkpts1, int64_t* m_bids, int B,
                                   int M, int N, gim_stream_t stream) {
    GIM_REQUIRE(matches0 && mscores0 && pos && count && matches && scores && B > 0 && M > 0 && N > 0, "lg_emit_matches: bad args");
    GIM_REQUIRE(!mkpts0 || (mkpts1 && m_bids && kpts0 && kpts1), "lg_emit_matches: adapter outputs need keypoints");
    LgaEmit e;
    e.matches0 = matches0; e.mscores0 = mscores0; e.pos = pos; e.count = count; e.kpts0 = kpts0; e.kpts1 = kpts1;
    e.scale0 = scale0; e.scale1 = scale1; e.matches = matches; e.scores = scores; e.mkpts0 = mkpts0; e.mkpts1 = mkpts1;
    e.m_bids = m_bids;
    hipLaunchKernelGGL(lga_emit_kernel, dim3((M + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, e, B, M, N);
    return gim_check_launch("lg_emit_matches");
}

extern "C" int gim_lg_log_assignment(const gim_lg_assign_args* a, float* out, gim_stream_t stream) {
    if (int rc = check_args(a)) return rc;
    GIM_REQUIRE(out, "lg_log_assignment: null output");
    hipStream_t s = (hipStream_t)stream;
    LgaWs w;
    LgaGeom g;
    lga_run_stats(a, w, g, s);
    const dim3 grid(w.ntM * w.ntN, a->B);
    hipLaunchKernelGGL(lga_best_kernel<1>, grid, dim3(256), TILE_SMEM, s, g, w, out);
    const int mx = a->M > a->N ? a->M : a->N;
    hipLaunchKernelGGL(lga_border_kernel, dim3((mx + 255) / 256, a->B), dim3(256), 0, s, w, out, a->B, a->M, a->N);
    return gim_check_launch("lg_log_assignment");
}

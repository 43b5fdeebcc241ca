// Fused coarse matching for gfx950: dual-softmax + threshold + border + mutual-NN + ordered compaction
// (networks/loftr/utils/coarse_matching.py:88-259) without ever writing the [N,L,S] matrix to HBM.
//
//   sim[n,i,j]  = (f0[n,i]/sqrt(C)) . (f1[n,j]/sqrt(C)) / T                       (coarse_matching.py:111,115)
//   conf[n,i,j] = softmax_i(sim)[i,j] * softmax_j(sim)[i,j]                       (coarse_matching.py:118)
//   match (i,j) <=> conf > thr, (i,j) off the border, conf == max_j conf[i,:], conf == max_i conf[:,j]
//                                                                                 (coarse_matching.py:174-190)
//   outputs ordered like torch.where(mask.max(dim=2)): ascending (n, i)           (coarse_matching.py:192-195)
//
// Pass A  (cm_stats):  128x128 similarity tiles on the fp32 MFMA (exact fp32: the parity bar asks for
//          bit-exact indices against a CPU fp32 oracle, which bf16 operands cannot give -- SURVEY 7),
//          tile staged in LDS, per-tile row/column (max, sum-exp) partials.
// combine: partials -> per-row / per-column softmax statistics.
//          Pass A also emits *pre-candidates*: the softmax over a tile's 128 columns (rows) bounds the
//          true row (column) softmax from above, so only elements whose tile-local row factor, column
//          factor and product all exceed thr can end up with conf > thr (<= 4 per row per tile); they are
//          stored as (i, j, sim).  Exact, and it removes the second similarity pass.
// Pass B  (cm_precand): every pre-candidate is evaluated with the final statistics; conf > thr makes
//          it a candidate (<= 4 per row, since sum_j softmax_j <= 1): atomicMax of conf into
//          rowmax/colmax (positive floats order like their bit patterns).  Restricting the maxima to
//          candidates is exact: if conf[i,j] > thr and some conf[i,j'] >= conf[i,j], then (i,j') is a
//          candidate too.  If the pre-candidate buffer overflows (cannot happen for thr >= 0.2 with the
//          default capacity unless more than 16 entries per row survive on average) the device-side flag
//          makes cm_cand recompute the tiles instead (the original two-pass scheme); it exits at once
//          otherwise.
// select:  candidate survives iff it equals both maxima and is off the border; ties -> smallest j
//          (mask.max(dim=2) returns the first True).  compact: block scan per pair, ascending i.
#include "igemm_mainloop.h"
#include <limits.h>
#include <stdlib.h>

namespace {

using gim::KTB;
constexpr int BM = 128, BN = 128, WM = 2, WN = 2;
constexpr int TLD = BN + 4;  // LDS similarity tile row stride in floats
constexpr int PLIST = 512;  // per-tile pre-candidate list in LDS (4 per row can pass the tile-local test)
constexpr int TILE_SMEM = BM * TLD * 4 + (4 * 128 + 4 * 128) * 4 + PLIST * 12 + 16;  // tile + scratch + list
static_assert(TILE_SMEM >= gim::mainloop_smem_bytes<BM, BN>(), "stage buffers must fit in the tile allocation");

struct Cand { int i, j; float p; int pad; };
struct PreCand { int i, j; float s; };

struct CmWs {  // device pointers carved out of the caller's workspace
    float2* rowpart;   // [N][ntS][L]
    float2* colpart;   // [N][ntL][S]
    float2* rowstat;   // [N][L]  (max, sum)
    float2* colstat;   // [N][S]
    unsigned* rowmaxP; // [N][L]
    unsigned* colmaxP; // [N][S]
    int* jsel;         // [N][L]
    float* psel;       // [N][L]
    int* lpos;         // [N][L]
    int* ncand;        // [N]
    Cand* cand;        // [N][capc]
    int* ext;          // [N][4] valid extents of the padding masks
    int* npre;         // [N] pre-candidate counters, npre[N] = overflow flag, npre[N + 1] = "wide logit range" flag of the panel kernel
    PreCand* pre;      // [N][capp]
    int* ktab;         // dense K table for the mainloop
    int ntL, ntS, capc, capp;
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

size_t carve(CmWs& w, char* base, int N, int L, int S, int C) {
    w.ntL = (L + BM - 1) / BM;
    w.ntS = (S + BN - 1) / BN;
    w.capc = 20 * L + 64;  // < 1/thr entries of a row can exceed thr (sum_j softmax_j <= 1); validate() enforces thr >= 0.05
    // pre-candidate capacity; GIM_CM_PRECAND_PER_ROW (default 16) exists so that tests can force the
    // overflow -> recompute fallback
    static const int per_row = [] { const char* e = getenv("GIM_CM_PRECAND_PER_ROW"); int v = e ? atoi(e) : 16; return v < 0 ? 0 : v; }();
    w.capp = per_row * L + 1024;
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += al(bytes); return p; };
    w.rowpart = (float2*)take((size_t)N * w.ntS * L * 8);
    w.colpart = (float2*)take((size_t)N * w.ntL * S * 8);
    w.rowstat = (float2*)take((size_t)N * L * 8);
    w.colstat = (float2*)take((size_t)N * S * 8);
    w.rowmaxP = (unsigned*)take((size_t)N * L * 4);
    w.colmaxP = (unsigned*)take((size_t)N * S * 4);
    w.jsel = (int*)take((size_t)N * L * 4);
    w.psel = (float*)take((size_t)N * L * 4);
    w.lpos = (int*)take((size_t)N * L * 4);
    w.ncand = (int*)take((size_t)N * 4);
    w.cand = (Cand*)take((size_t)N * w.capc * sizeof(Cand));
    w.ext = (int*)take((size_t)N * 16);
    w.npre = (int*)take((size_t)(N + 2) * 4);
    w.pre = (PreCand*)take((size_t)N * w.capp * sizeof(PreCand));
    w.ktab = (int*)take((size_t)(C / 32 + 2) * 8 * 4);
    return o;
}

struct CmGeom {
    const void* feat0;     // [N][L][ldf] rows, fp32 or bf16 (bf16: the token buffers the coarse transformer already holds)
    const void* feat1;
    int bf16, ldf;
    const uint8_t* mask0;  // [N][L] or NULL
    const uint8_t* mask1;  // [N][S] or NULL
    int N, L, S, C;
    float inv_c, temperature, thr;
    float inv_ct;  // 1 / (C * temperature)
};

// similarity tile -> LDS (St[i][j], i = feat0 row, j = feat1 row), scaled like the reference
// BF16: bf16 x bf16 products are exact in fp32 and accumulate in fp32, so on bf16-valued features the bf16 MFMA (8x the fp32
// matrix rate) computes the same similarity as the fp32 path up to summation order.
template <int MODE, bool BF16>
__device__ __forceinline__ void sim_tile_to_lds(const CmGeom& g, const int* ktab, int n, int m0, int n0, char* smem) {
    constexpr int ES = BF16 ? 2 : 4;
    gim::MainloopArgs ml;
    ml.x = (const char*)g.feat0 + (size_t)n * g.L * g.ldf * ES;
    ml.w = (const char*)g.feat1 + (size_t)n * g.S * g.ldf * ES;
    ml.ktab = ktab;
    ml.x_bytes = (unsigned)(((size_t)(g.L - 1) * g.ldf + g.C) * ES);
    ml.w_bytes = (unsigned)(((size_t)(g.S - 1) * g.ldf + g.C) * ES);
    ml.H = 1; ml.W = g.L; ml.Ho = 1; ml.Wo = g.L; ml.stride = 1; ml.pad = 0; ml.ldx = g.ldf;
    ml.kpad = g.C; ml.ldw = g.ldf; ml.M = g.L;
    f32x16_t acc[2][2];
    gim::igemm_mainloop<BM, BN, WM, WN, BF16, true>(ml, smem, m0, n0, acc);
    // mainloop ends with a barrier: stage buffers are free, reuse them as the similarity tile
    float* St = (float*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5, wm = wave / WN, wn = wave - wm * WN;
    const bool masked = g.mask0 != nullptr;
    const float NEG_INF = -1e9f;  // INF = 1e9 (coarse_matching.py:6): sim.masked_fill_(~(mask0 x mask1), -INF)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i_loc = wm * 64 + j * 32 + l31;
        const int gi = m0 + i_loc;
        const bool vi = !masked || (gi < g.L && g.mask0[(size_t)n * g.L + gi]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int j_loc = wn * 64 + i * 32 + rg * 8 + lh * 4;
                float4 v;
                // (f0/sqrt C).(f1/sqrt C)/T: 1/C is a power of two, so one multiply by 1/(C T) differs from the
                // reference's divide by T by at most 1 ulp (an IEEE division costs ~12 VALU instructions per value)
                v.x = acc[i][j][rg * 4 + 0] * g.inv_ct;
                v.y = acc[i][j][rg * 4 + 1] * g.inv_ct;
                v.z = acc[i][j][rg * 4 + 2] * g.inv_ct;
                v.w = acc[i][j][rg * 4 + 3] * g.inv_ct;
                if (masked) {
                    const int gj = n0 + j_loc;
                    const uint8_t* m1 = g.mask1 + (size_t)n * g.S;
                    if (!vi || gj + 0 >= g.S || !m1[gj + 0]) v.x = NEG_INF;
                    if (!vi || gj + 1 >= g.S || !m1[gj + 1]) v.y = NEG_INF;
                    if (!vi || gj + 2 >= g.S || !m1[gj + 2]) v.z = NEG_INF;
                    if (!vi || gj + 3 >= g.S || !m1[gj + 3]) v.w = NEG_INF;
                }
                *(float4*)(St + i_loc * TLD + j_loc) = v;
            }
    }
    __syncthreads();
}

// exp for the softmax *statistics* (sums of up to S terms): v_exp_f32 on x*log2(e).  Relative error
// <= ~|x| * 6e-8 -- 1e-6 for every term that contributes more than e^-15 of a sum -- while each of the few
// final confidences is evaluated with the accurate expf (cm_precand / cm_cand).
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
// `gated`: launched behind the row-panel kernel as its fallback -- runs only if that kernel found a logit range too wide for its
// shared exponentials (npre[N + 1]); it then redoes all partials (and adds its own pre-candidates to the list, which may hold
// duplicates afterwards: every entry is an exact (i, j, similarity) triple that cm_precand evaluates with the final statistics).
template <bool BF16>
__global__ void __launch_bounds__(256) cm_stats_kernel(const CmGeom g, const CmWs w, const int gated) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gated && !w.npre[g.N + 1]) return;
    const int n = blockIdx.y;
    const int mt = blockIdx.x / w.ntS, nt = blockIdx.x - mt * w.ntS;
    const int m0 = mt * BM, n0 = nt * BN;
    sim_tile_to_lds<0, BF16>(g, w.ktab, n, m0, n0, smem);
    const float* St = (const float*)smem;
    float* red = (float*)(smem + BM * TLD * 4);  // [4][128] reduction scratch
    float* rowm = red + 512;                     // tile-local row max / sum, column max / sum
    float* rowz = red + 640;
    float* colm = red + 768;
    float* colz = red + 896;
    const int t = threadIdx.x, idx = t & 127, half = t >> 7;
    const float NEG = -INFINITY;
    // Interior tiles (all but the last tile row / column) take fixed-trip-count paths: the 64 values of a thread's half row
    // (half column) are fetched into registers with all LDS reads in flight at once and both passes (max, sum-exp) run on
    // the registers.  With run-time loop bounds the compiler leaves read -> wait -> use loops, one LDS latency per
    // element: the statistics phase then took 3.5x as long as the bf16 MFMA main loop (460 vs 130 us per call).
    const bool interior = (m0 + BM <= g.L) && (n0 + BN <= g.S);  // block-uniform
    float rv[64];  // interior: this thread's half row, kept for the pre-candidate scan
    // ---- rows: thread (row idx, column half).  nv = valid columns of this half (bounds test hoisted) ----
    {
        const int nv = min(64, max(0, g.S - n0 - half * 64)), nv4 = nv & ~3;
        const float* rp = St + idx * TLD + half * 64;
        float mx = NEG;
        if (interior) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float4 v = *(const float4*)(rp + 4 * q);
                rv[4 * q + 0] = v.x; rv[4 * q + 1] = v.y; rv[4 * q + 2] = v.z; rv[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) mx = fmaxf(fmaxf(mx, fmaxf(rv[4 * q], rv[4 * q + 1])), fmaxf(rv[4 * q + 2], rv[4 * q + 3]));
        } else {
            for (int jj = 0; jj < nv4; jj += 4) {
                const float4 v = *(const float4*)(rp + jj);
                mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            }
            for (int jj = nv4; jj < nv; ++jj) mx = fmaxf(mx, rp[jj]);
        }
        red[half * 128 + idx] = mx;
        __syncthreads();
        const float m = fmaxf(red[idx], red[128 + idx]);
        float z = 0.f;
        if (interior) {  // same summation order as the generic loop
#pragma unroll
            for (int q = 0; q < 16; ++q)  // (v - m is exact for the values that matter; folding it into the exp2 scaling is not)
                z += (fast_exp(rv[4 * q] - m) + fast_exp(rv[4 * q + 1] - m)) + (fast_exp(rv[4 * q + 2] - m) + fast_exp(rv[4 * q + 3] - m));
        } else {
            for (int jj = 0; jj < nv4; jj += 4) {
                const float4 v = *(const float4*)(rp + jj);
                z += (fast_exp(v.x - m) + fast_exp(v.y - m)) + (fast_exp(v.z - m) + fast_exp(v.w - m));
            }
            for (int jj = nv4; jj < nv; ++jj) z += fast_exp(rp[jj] - m);
        }
        red[256 + half * 128 + idx] = z;
        __syncthreads();
        if (half == 0) {
            const float zz = red[256 + idx] + red[384 + idx];
            rowm[idx] = m; rowz[idx] = zz;
            if (m0 + idx < g.L) w.rowpart[((size_t)n * w.ntS + nt) * g.L + m0 + idx] = make_float2(m, zz);
        }
        __syncthreads();
    }
    // ---- columns: thread (column idx, row half) ----
    {
        const int r0 = half * 64, r1 = r0 + min(64, max(0, g.L - m0 - r0));
        float mx = NEG;
        float cv[64];
        if (interior) {
#pragma unroll
            for (int r = 0; r < 64; ++r) cv[r] = St[(r0 + r) * TLD + idx];
#pragma unroll
            for (int r = 0; r < 64; ++r) mx = fmaxf(mx, cv[r]);
        } else {
            for (int r = r0; r < r1; ++r) mx = fmaxf(mx, St[r * TLD + idx]);
        }
        red[half * 128 + idx] = mx;
        __syncthreads();
        const float m = fmaxf(red[idx], red[128 + idx]);
        float z = 0.f;
        if (interior) {
#pragma unroll
            for (int r = 0; r < 64; ++r) z += fast_exp(cv[r] - m);
        } else {
            for (int r = r0; r < r1; ++r) z += fast_exp(St[r * TLD + idx] - m);
        }
        red[256 + half * 128 + idx] = z;
        __syncthreads();
        if (half == 0) {
            const float zz = red[256 + idx] + red[384 + idx];
            colm[idx] = m; colz[idx] = zz;
            if (n0 + idx < g.S) w.colpart[((size_t)n * w.ntL + mt) * g.S + n0 + idx] = make_float2(m, zz);
        }
        __syncthreads();
    }
    // ---- pre-candidates.  softmax over the tile's columns (rows) >= the true row (column) softmax, so
    // conf > thr needs: tile-local row factor > thr, column factor > thr (cheap tests  s > m + log(thr z))
    // and their product > thr (evaluated only for the few survivors).  Survivors are collected in an LDS
    // list (LDS atomics) and the tile reserves its global range with ONE atomic: a returning global atomic
    // per survivor stalled the waves for microseconds each (measured 2.8x on wide-spread similarities).
    {
        PreCand* plist = (PreCand*)(smem + BM * TLD * 4 + 4096);
        int* pcnt = (int*)(plist + PLIST);  // [0] = local count, [1] = global base
        const float thr_pre = g.thr * (1.0f - 1e-4f);  // slack: rounding must never drop a true candidate
        if (half == 0) red[idx] = rowm[idx] + logf(thr_pre * rowz[idx]);
        else red[128 + idx] = colm[idx] + logf(thr_pre * colz[idx]);
        if (t == 0) pcnt[0] = 0;
        __syncthreads();
        const int i = m0 + idx;
        if (i < g.L) {
            const float trow = red[idx], rm = rowm[idx], rz = rowz[idx];
            const int nv = min(64, max(0, g.S - n0 - half * 64));
            auto scan4 = [&](const int jj, const float4 s4, const float4 t4) __attribute__((always_inline)) {
                const float sv4[4] = {s4.x, s4.y, s4.z, s4.w}, tc4[4] = {t4.x, t4.y, t4.z, t4.w};
                if (!((s4.x > trow && s4.x > t4.x) || (s4.y > trow && s4.y > t4.y) || (s4.z > trow && s4.z > t4.z) ||
                      (s4.w > trow && s4.w > t4.w)))
                    return;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int jl = half * 64 + jj + e, j = n0 + jl;
                    const float sv = sv4[e];
                    if (jj + e < nv && sv > trow && sv > tc4[e]) {
                        const float pr = expf(sv - rm) / rz;
                        const float pc = expf(sv - colm[jl]) / colz[jl];
                        if (pr * pc > thr_pre) {
                            const int k = atomicAdd(&pcnt[0], 1);
                            if (k < PLIST) plist[k] = PreCand{i, j, sv};
                        }
                    }
                }
            };
            if (interior) {  // cheap test unrolled on the registers, the rare path once per surviving quad
                unsigned hit = 0u;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 t4 = *(const float4*)(red + 128 + half * 64 + 4 * q);
                    const bool h = (rv[4 * q] > trow && rv[4 * q] > t4.x) || (rv[4 * q + 1] > trow && rv[4 * q + 1] > t4.y) ||
                                   (rv[4 * q + 2] > trow && rv[4 * q + 2] > t4.z) || (rv[4 * q + 3] > trow && rv[4 * q + 3] > t4.w);
                    hit |= h ? 1u << q : 0u;
                }
                while (hit) {
                    const int jj = 4 * (__ffs(hit) - 1);
                    hit &= hit - 1;
                    scan4(jj, *(const float4*)(St + idx * TLD + half * 64 + jj), *(const float4*)(red + 128 + half * 64 + jj));
                }
            } else {
                for (int jj = 0; jj < nv; jj += 4)  // four columns per step; survivors are rare
                    scan4(jj, *(const float4*)(St + idx * TLD + half * 64 + jj), *(const float4*)(red + 128 + half * 64 + jj));
            }
        }
        __syncthreads();
        const int cnt = pcnt[0];
        if (cnt == 0) return;
        if (cnt > PLIST) {  // cannot happen for thr >= 4/128-ish tiles, but stay exact: recompute path
            if (t == 0) w.npre[g.N] = 1;
            return;
        }
        if (t == 0) pcnt[1] = atomicAdd(&w.npre[n], cnt);
        __syncthreads();
        const int base = pcnt[1];
        if (base + cnt > w.capp) {
            if (t == 0) w.npre[g.N] = 1;  // overflow: cm_cand_kernel<0> recomputes
            return;
        }
        for (int k = t; k < cnt; k += 256) w.pre[(size_t)n * w.capp + base + k] = plist[k];
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Pass A as a PERSISTENT ROW-PANEL kernel (16-bit features, C = 256, no padding masks): round 3.
//
// The tile-per-workgroup kernel above spends 14 us per 128 x 128 tile for 1 us of MFMAs: every tile re-stages both operand panels
// through five dependent L2 round trips, writes the fp32 similarity tile to LDS, re-reads it twice (rows, columns) behind five
// barriers and evaluates TWO exponentials per element (row and column softmax use different maxima).  Here
//   * a workgroup owns a 128-row panel of feat0 for a third of the columns: the A panel (64 KiB) is staged ONCE, the B tiles stream
//     through a single 64 KiB buffer -- tile t + 1 is fetched (LDS-DMA issued through inline asm, invisible to the compiler's
//     waits) while tile t's statistics run;
//   * the statistics come straight out of the MFMA accumulators (lane = feat0 row, registers = 32 of the wave's 64 columns):
//     row max / sum-exp in the lane (+ one exchange with lane ^ 32), ONE exponential per element e = exp(v - rowmax), and the
//     column sums as sum_i e_ij * exp(rowmax_i - ref) with one wave-wide reference `ref` -- the row weights cost two exps per
//     lane -- reduced over the 32 row lanes by a halving butterfly (31 exchanges instead of 32 x 5);
//   * row statistics are carried ONLINE across the panel's tiles (max, rescaled sum) and written once per panel third;
//   * pre-candidates are tested on the registers.
// Range guard: exp(v - ref) underflows when a wave tile's logits span more than ~87; at 80 the kernel raises npre[N + 1] and the
// tile-per-workgroup kernel (gated launch right behind this one) redoes the statistics.  exp() rounding: as above (fast_exp for
// the sums, expf for every decision).
constexpr int PJ = 3;                                   // column thirds per row panel: N x ntL x 3 workgroups (912 at 640x480 batch 8)
constexpr int P2_OFF_B = 4 * BM * KTB;                  // A panel: 4 K slabs of [128 rows][128 B]
constexpr int P2_OFF_X = P2_OFF_B + 4 * BN * KTB;       // B tile : 4 K slabs
struct P2X {
    float2 rowx[2][128];      // [column half wn][row]: (max, sum-exp) over the wave tile's 64 columns
    float2 colx[2][128];      // [row half wm][column]: (reference, sum-exp) over the wave tile's 64 rows
    float rowm[128], rowz[128], colm[128], colz[128];   // statistics of the whole tile (exact pre-candidate test)
    float trow[128], tcol[128];                          // cheap thresholds m + log(thr z)
    PreCand plist[PLIST];
    int pcnt[4];
};
constexpr int P2_SMEM = P2_OFF_X + (int)sizeof(P2X);
static_assert(P2_SMEM <= 160 * 1024, "row-panel kernel: LDS");

__global__ void __launch_bounds__(256) cm_stats_panel_kernel(const CmGeom g, const CmWs w, const int force_wide) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_t;
    const int n = blockIdx.y;
    const int mt = blockIdx.x / PJ, jc = blockIdx.x - mt * PJ;
    const int m0 = mt * BM;
    const int jt0 = jc * w.ntS / PJ, jt1 = (jc + 1) * w.ntS / PJ;
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 1, wn = wave & 1;
    P2X& X = *(P2X*)(smem + P2_OFF_X);
    const unsigned smem_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)smem);
    const unsigned rowb = (unsigned)g.ldf * 2u;
    const gim_u32x4_t rA = gim_make_rsrc((const char*)g.feat0 + (size_t)n * g.L * rowb, (unsigned)(((size_t)(g.L - 1) * g.ldf + g.C) * 2));
    const gim_u32x4_t rB = gim_make_rsrc((const char*)g.feat1 + (size_t)n * g.S * rowb, (unsigned)(((size_t)(g.S - 1) * g.ldf + g.C) * 2));
    // staging: a piece = 8 rows x 128 B; this wave fetches pieces wave, wave + 4, wave + 8, wave + 12 of every K slab; lane -> row
    // piece * 8 + (lane >> 3), LDS slot lane & 7 <- source slot (lane & 7) ^ ((row >> 1) & 7).  Rows beyond L / S lie beyond the
    // descriptor's bound and read as zeros.
    const int srow = lane >> 3, sslot = lane & 7;
    auto issue = [&](const gim_u32x4_t rs, const unsigned lds_base, const int r0) __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (i * 4 + wave) * 8 + srow;
                const unsigned voff = (unsigned)(r0 + row) * rowb + (unsigned)(kt * KTB) + (unsigned)((sslot ^ ((row >> 1) & 7)) << 4);
                gim_dma16(rs, lds_base + (unsigned)(kt * BM * KTB + (i * 4 + wave) * 1024), voff);
            }
    };
    issue(rA, smem_addr, m0);
    issue(rB, smem_addr + P2_OFF_B, jt0 * BN);
    if (force_wide && t == 0) w.npre[g.N + 1] = 1;   // tests: exercise the gated fallback
    float runM = -INFINITY, runZ = 0.f;            // threads 0..127: online row statistics of row m0 + t over this panel third
    const float thr_pre = g.thr * (1.0f - 1e-4f);  // slack: rounding must never drop a true candidate
    const int lswz = (l31 >> 1) & 7;
    const char* sA = smem + (wm * 64 + l31) * KTB;
    const char* sB = smem + P2_OFF_B + (wn * 64 + l31) * KTB;
    const int irow0 = m0 + wm * 64 + l31;           // this lane's rows: irow0, irow0 + 32
    for (int nt = jt0; nt < jt1; ++nt) {
        const int n0 = nt * BN;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                             // both operands of this tile have landed (every wave waited for its pieces)
        f32x16_t acc[2][2];                          // [column fragment ic][row fragment jr]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int so = ((2 * ks + lh) ^ lswz) << 4;
                const bf16x8_t a0 = *(const bf16x8_t*)(sA + kt * BM * KTB + so), a1 = *(const bf16x8_t*)(sA + kt * BM * KTB + 32 * KTB + so);
                const bf16x8_t b0 = *(const bf16x8_t*)(sB + kt * BN * KTB + so), b1 = *(const bf16x8_t*)(sB + kt * BN * KTB + 32 * KTB + so);
                acc[0][0] = mfma_h16_32x32x16(b0, a0, acc[0][0]);
                acc[0][1] = mfma_h16_32x32x16(b0, a1, acc[0][1]);
                acc[1][0] = mfma_h16_32x32x16(b1, a0, acc[1][0]);
                acc[1][1] = mfma_h16_32x32x16(b1, a1, acc[1][1]);
            }
        __syncthreads();                             // everybody is done with the B tile
        if (nt + 1 < jt1) issue(rB, smem_addr + P2_OFF_B, n0 + BN);   // lands under the statistics below
        // ---- statistics on the registers: acc[ic][jr][4 rg + e] = sim(row irow0 + 32 jr, column n0 + 64 wn + 32 ic + 8 rg + 4 lh + e) ----
        const bool edge = (m0 + BM > g.L) || (n0 + BN > g.S);   // block-uniform
        const int jcol0 = n0 + wn * 64 + lh * 4;
        float rmax[2], rsum[2], vmin = INFINITY;
#pragma unroll
        for (int jr = 0; jr < 2; ++jr) {
            const bool rok = irow0 + 32 * jr < g.L;
            float mx = -INFINITY;
#pragma unroll
            for (int ic = 0; ic < 2; ++ic)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[ic][jr][r] * g.inv_ct;   // (f0/sqrt C).(f1/sqrt C)/T as one multiply (see sim_tile_to_lds)
                    if (edge && !(rok && jcol0 + ic * 32 + (r >> 2) * 8 + (r & 3) < g.S)) v = -INFINITY;
                    acc[ic][jr][r] = v;
                    mx = fmaxf(mx, v);
                    vmin = fminf(vmin, v == -INFINITY ? INFINITY : v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            rmax[jr] = mx == -INFINITY ? 0.f : mx;       // a row without a valid column (beyond L): any finite reference
        }
        float mw = fmaxf(rmax[0], rmax[1]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, o, 64)); vmin = fminf(vmin, __shfl_xor(vmin, o, 64)); }
        vmin = fminf(vmin, __shfl_xor(vmin, 32, 64));
        if (mw - vmin > 80.f && lane == 0) w.npre[g.N + 1] = 1;   // shared exponentials would underflow: the gated fallback redoes it
        const float wg0 = fast_exp(rmax[0] - mw), wg1 = fast_exp(rmax[1] - mw);
        float cs[32];
        rsum[0] = rsum[1] = 0.f;
#pragma unroll
        for (int ic = 0; ic < 2; ++ic)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e0 = fast_exp(acc[ic][0][r] - rmax[0]), e1 = fast_exp(acc[ic][1][r] - rmax[1]);
                rsum[0] += e0; rsum[1] += e1;
                cs[ic * 16 + r] = e0 * wg0 + e1 * wg1;     // this lane's two rows of column (ic, r), relative to mw
            }
        rsum[0] += __shfl_xor(rsum[0], 32, 64);
        rsum[1] += __shfl_xor(rsum[1], 32, 64);
        // column sums over the 32 row lanes: halving butterfly, lane l31 ends up with column x = l31 of its 32 (ic = x >> 4, r = x & 15)
#pragma unroll
        for (int lvl = 16; lvl > 0; lvl >>= 1) {
            const bool up = (l31 & lvl) != 0;
#pragma unroll
            for (int x = 0; x < lvl; ++x) {
                float lo = cs[x], hi = cs[x + lvl];
                asm volatile("" : "+v"(lo), "+v"(hi));   // keep them values: hipcc otherwise folds the two selects into ONE
                                                         // dynamically indexed read of cs[] = a 32-deep compare/select chain each
                cs[x] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, lvl, 64);
            }
        }
        const int cloc = wn * 64 + (l31 >> 4) * 32 + ((l31 & 15) >> 2) * 8 + lh * 4 + (l31 & 3);   // column of this lane within the tile
        if (lh == 0) {
            X.rowx[wn][wm * 64 + l31] = make_float2(rmax[0], rsum[0]);
            X.rowx[wn][wm * 64 + 32 + l31] = make_float2(rmax[1], rsum[1]);
        }
        X.colx[wm][cloc] = make_float2(mw, cs[0]);
        if (t == 0) X.pcnt[0] = 0;
        __syncthreads();
        if (t < 128) {                                // tile row statistics + the online update of this panel third
            const float2 a = X.rowx[0][t], b = X.rowx[1][t];
            const float m = fmaxf(a.x, b.x);
            const float z = a.y * expf(a.x - m) + b.y * expf(b.x - m);
            X.rowm[t] = m; X.rowz[t] = z; X.trow[t] = m + logf(thr_pre * z);
            const float mn = fmaxf(runM, m);
            runZ = runZ * expf(runM - mn) + z * expf(m - mn);
            runM = mn;
        } else {
            const int c = t - 128;
            const float2 a = X.colx[0][c], b = X.colx[1][c];
            const float m = fmaxf(a.x, b.x);
            const float z = a.y * expf(a.x - m) + b.y * expf(b.x - m);
            X.colm[c] = m; X.colz[c] = z; X.tcol[c] = m + logf(thr_pre * z);
            if (n0 + c < g.S) w.colpart[((size_t)n * w.ntL + mt) * g.S + n0 + c] = make_float2(m, z);
        }
        __syncthreads();
        // ---- pre-candidates (same test as the tile kernel): cheap thresholds on the registers, the exact product for the survivors ----
#pragma unroll
        for (int jr = 0; jr < 2; ++jr) {
            const int il = wm * 64 + jr * 32 + l31;
            const float trow = X.trow[il];
            unsigned hit = 0u;
#pragma unroll
            for (int ic = 0; ic < 2; ++ic)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const float4 t4 = *(const float4*)(X.tcol + wn * 64 + ic * 32 + rg * 8 + lh * 4);
                    const f32x16_t& v = acc[ic][jr];
                    hit |= (v[rg * 4] > trow && v[rg * 4] > t4.x) ? 1u << (ic * 16 + rg * 4) : 0u;
                    hit |= (v[rg * 4 + 1] > trow && v[rg * 4 + 1] > t4.y) ? 1u << (ic * 16 + rg * 4 + 1) : 0u;
                    hit |= (v[rg * 4 + 2] > trow && v[rg * 4 + 2] > t4.z) ? 1u << (ic * 16 + rg * 4 + 2) : 0u;
                    hit |= (v[rg * 4 + 3] > trow && v[rg * 4 + 3] > t4.w) ? 1u << (ic * 16 + rg * 4 + 3) : 0u;
                }
            while (hit) {
                const int x = __ffs(hit) - 1;
                hit &= hit - 1;
                const int ic = x >> 4, r = x & 15;
                float sv = 0.f;                       // acc[ic][jr][r] with a run-time (ic, r): select, do not index (scratch)
#pragma unroll
                for (int q = 0; q < 32; ++q) sv = (q == x) ? acc[q >> 4][jr][q & 15] : sv;
                const int jl = wn * 64 + ic * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
                const float pr = expf(sv - X.rowm[il]) / X.rowz[il];
                const float pc = expf(sv - X.colm[jl]) / X.colz[jl];
                if (pr * pc > thr_pre) {
                    const int k = atomicAdd(&X.pcnt[0], 1);
                    if (k < PLIST) X.plist[k] = PreCand{m0 + il, n0 + jl, sv};
                }
            }
        }
        __syncthreads();
        const int cnt = X.pcnt[0];
        if (cnt > 0) {                                // block-uniform
            if (cnt > PLIST) {
                if (t == 0) w.npre[g.N] = 1;          // recompute path (cm_cand_kernel<0>)
            } else {
                if (t == 0) X.pcnt[1] = atomicAdd(&w.npre[n], cnt);
                __syncthreads();
                const int base = X.pcnt[1];
                if (base + cnt > w.capp) { if (t == 0) w.npre[g.N] = 1; }
                else for (int k = t; k < cnt; k += 256) w.pre[(size_t)n * w.capp + base + k] = X.plist[k];
            }
        }
    }
    if (t < 128 && m0 + t < g.L) w.rowpart[((size_t)n * w.ntS + jc) * g.L + m0 + t] = make_float2(runM, runZ);
}

// Pass B on the pre-candidate list: final statistics -> conf; candidates + row/column maxima
__global__ void cm_precand_kernel(const CmGeom g, const CmWs w) {
    const int n = blockIdx.y;
    if (w.npre[g.N]) return;  // overflowed: cm_cand_kernel<0> recomputes everything
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= min(w.npre[n], w.capp)) return;
    const PreCand c = w.pre[(size_t)n * w.capp + k];
    const float2 r = w.rowstat[(size_t)n * g.L + c.i], cs = w.colstat[(size_t)n * g.S + c.j];
    const float pr = expf(c.s - r.x) / r.y;    // softmax over j (dim=2)
    const float pc = expf(c.s - cs.x) / cs.y;  // softmax over i (dim=1)
    const float p = pc * pr;
    if (p > g.thr) {
        const unsigned pb = __float_as_uint(p);
        atomicMax(&w.rowmaxP[(size_t)n * g.L + c.i], pb);
        atomicMax(&w.colmaxP[(size_t)n * g.S + c.j], pb);
        const int q = atomicAdd(&w.ncand[n], 1);
        if (q < w.capc) w.cand[(size_t)n * w.capc + q] = Cand{c.i, c.j, p, 0};
    }
}

// stat[n][x] = combine over tiles:  m = max m_t ; z = sum_t z_t * exp(m_t - m)   (ascending tile order)
// `stride` = tile slots per pair in `part`; `ntile_panel` > 0: the row-panel kernel filled only that many of them -- unless its
// fallback ran (*wide), which filled all `ntile`
__global__ void cm_combine_kernel(const float2* __restrict__ part, float2* __restrict__ stat, int N, int len, int ntile_all,
                                  int ntile_panel, const int* __restrict__ wide) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * len) return;
    const int ntile = (ntile_panel > 0 && !*wide) ? ntile_panel : ntile_all;
    const size_t n = idx / len, x = idx - n * len;
    float m = -INFINITY;
    for (int t = 0; t < ntile; ++t) m = fmaxf(m, part[(n * ntile_all + t) * len + x].x);
    float z = 0.f;
    for (int t = 0; t < ntile; ++t) {
        const float2 p = part[(n * ntile_all + t) * len + x];
        z += p.y * expf(p.x - m);
    }
    stat[idx] = make_float2(m, z);
}

__global__ void cm_init_kernel(const CmWs w, int N, int L, int S) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)N * L) { w.rowmaxP[idx] = 0u; w.jsel[idx] = INT_MAX; w.psel[idx] = 0.f; }
    if (idx < (size_t)N * S) w.colmaxP[idx] = 0u;
    if (idx < (size_t)N) w.ncand[idx] = 0;
    if (idx <= (size_t)N + 1) w.npre[idx] = 0;
}

__global__ void cm_ktab_kernel(int* ktab, int C, int ge) {  // dense table: K group g -> channel ge * g (ge = 4 fp32 / 8 bf16 per 16 B); 2 padding slabs
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int ng = (C / 32 + 2) * 8;
    if (g < ng) ktab[g] = (g * ge < C) ? g * ge : (int)0xFF000000;
}

// MODE 0: emit candidates.  MODE 1: write the conf tile to `conf` (lazy data['conf_matrix']).
template <int MODE, bool BF16>
__global__ void __launch_bounds__(256) cm_cand_kernel(const CmGeom g, const CmWs w, float* __restrict__ conf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (MODE == 0 && !w.npre[g.N]) return;  // fallback only: the pre-candidate path did the work
    const int n = blockIdx.y;
    const int mt = blockIdx.x / w.ntS, nt = blockIdx.x - mt * w.ntS;
    const int m0 = mt * BM, n0 = nt * BN;
    sim_tile_to_lds<1, BF16>(g, w.ktab, n, m0, n0, smem);
    const float* St = (const float*)smem;
    float2* rs = (float2*)(smem + BM * TLD * 4);  // [128] row stats
    float2* cs = rs + 128;                         // [128] col stats
    const int t = threadIdx.x, idx = t & 127, half = t >> 7;
    if (half == 0) rs[idx] = (m0 + idx < g.L) ? w.rowstat[(size_t)n * g.L + m0 + idx] : make_float2(0.f, 1.f);
    else cs[idx] = (n0 + idx < g.S) ? w.colstat[(size_t)n * g.S + n0 + idx] : make_float2(0.f, 1.f);
    __syncthreads();
    const int i = m0 + idx;
    if (i >= g.L) return;
    const float2 r = rs[idx];
    for (int jj = 0; jj < 64; ++jj) {
        const int jl = half * 64 + jj, j = n0 + jl;
        if (j >= g.S) break;
        const float s = St[idx * TLD + jl];
        const float pr = expf(s - r.x) / r.y;  // softmax over j (dim=2)
        if (MODE == 1) {
            const float2 c = cs[jl];
            conf[((size_t)n * g.L + i) * g.S + j] = (expf(s - c.x) / c.y) * pr;
        } else if (pr > g.thr) {
            const float2 c = cs[jl];
            const float pc = expf(s - c.x) / c.y;  // softmax over i (dim=1)
            const float p = pc * pr;
            if (p > g.thr) {
                const unsigned pb = __float_as_uint(p);
                atomicMax(&w.rowmaxP[(size_t)n * g.L + i], pb);
                atomicMax(&w.colmaxP[(size_t)n * g.S + j], pb);
                const int k = atomicAdd(&w.ncand[n], 1);
                if (k < w.capc) w.cand[(size_t)n * w.capc + k] = Cand{i, j, p, 0};
            }
        }
    }
}

struct BorderGeom { int h0c, w0c, h1c, w1c, b; const int* ext; };  // ext: NULL or [N][4] = valid (h0, w0, h1, w1)

// cell idx of a map of row length `wd`; valid extent (h, w): off the border iff b <= y < h-b and b <= x < w-b
__device__ __forceinline__ bool off_border2(int idx, int wd, int h, int w, int b) {
    const int y = idx / wd, x = idx - y * wd;
    return y >= b && y < h - b && x >= b && x < w - b;
}

// valid extent of the padding masks, as the reference computes it: h = max over columns of the per-column
// count of valid cells (p_m.sum(1).max(-1)), w = max over rows of the per-row count (p_m.sum(-1).max(-1)).
__global__ void cm_mask_extent_kernel(const uint8_t* __restrict__ m0, const uint8_t* __restrict__ m1, int* __restrict__ ext,
                                      int h0c, int w0c, int h1c, int w1c) {
    __shared__ int best[4];
    const int n = blockIdx.x, t = threadIdx.x;
    if (t < 4) best[t] = 0;
    __syncthreads();
    for (int which = 0; which < 2; ++which) {
        const uint8_t* m = which ? m1 + (size_t)n * h1c * w1c : m0 + (size_t)n * h0c * w0c;
        const int h = which ? h1c : h0c, w = which ? w1c : w0c;
        for (int x = t; x < w; x += blockDim.x) {  // column sums -> valid height
            int c = 0;
            for (int y = 0; y < h; ++y) c += m[y * w + x] ? 1 : 0;
            atomicMax(&best[which * 2 + 0], c);
        }
        for (int y = t; y < h; y += blockDim.x) {  // row sums -> valid width
            int c = 0;
            for (int x = 0; x < w; ++x) c += m[y * w + x] ? 1 : 0;
            atomicMax(&best[which * 2 + 1], c);
        }
    }
    __syncthreads();
    if (t < 4) ext[n * 4 + t] = best[t];
}

// phase 0: jsel[i] = min j over surviving candidates; phase 1: psel[i] = conf of the selected one
template <int PHASE>
__global__ void cm_select_kernel(const CmWs w, int N, int L, int S, BorderGeom bg) {
    const int n = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int nc = min(w.ncand[n], w.capc);
    if (k >= nc) return;
    const Cand c = w.cand[(size_t)n * w.capc + k];
    const unsigned pb = __float_as_uint(c.p);
    if (pb != w.rowmaxP[(size_t)n * L + c.i] || pb != w.colmaxP[(size_t)n * S + c.j]) return;
    // mask_border (coarse_matching.py:9-26) or, with padding masks, mask_border_with_padding (:29-44):
    // the far borders are measured from the valid extent of each image
    const int h0 = bg.ext ? bg.ext[n * 4 + 0] : bg.h0c, w0 = bg.ext ? bg.ext[n * 4 + 1] : bg.w0c;
    const int h1 = bg.ext ? bg.ext[n * 4 + 2] : bg.h1c, w1 = bg.ext ? bg.ext[n * 4 + 3] : bg.w1c;
    if (!off_border2(c.i, bg.w0c, h0, w0, bg.b) || !off_border2(c.j, bg.w1c, h1, w1, bg.b)) return;
    if (PHASE == 0) atomicMin(&w.jsel[(size_t)n * L + c.i], c.j);
    else if (w.jsel[(size_t)n * L + c.i] == c.j) w.psel[(size_t)n * L + c.i] = c.p;
}

// one block (1024 threads) per pair: exclusive scan of the match flags over i
__global__ void __launch_bounds__(1024) cm_scan_kernel(const CmWs w, int L, int* __restrict__ count) {
    __shared__ int wtot[16];
    __shared__ int running;
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < L; base += 1024) {
        const int i = base + t;
        const bool f = i < L && w.jsel[(size_t)n * L + i] != INT_MAX;
        const unsigned long long bal = __ballot(f);
        const int excl = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wave] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int k = 0; k < wave; ++k) off += wtot[k];
        if (i < L) w.lpos[(size_t)n * L + i] = off + excl;
        __syncthreads();
        if (t == 0) {
            int s = 0;
            for (int k = 0; k < 16; ++k) s += wtot[k];
            running += s;
        }
        __syncthreads();
    }
    if (t == 0) count[1 + n] = running;
}

struct EmitArgs {
    int64_t *b_ids, *i_ids, *j_ids;
    float *mconf, *mk0, *mk1;
    const float *scale0, *scale1;
    int* count;
    int N, L, cap, w0c, w1c;
    float scale;
};

__global__ void cm_emit_kernel(const CmWs w, const EmitArgs e) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) {
        int s = 0;
        for (int k = 0; k < e.N; ++k) s += e.count[1 + k];
        e.count[0] = s < e.cap ? s : e.cap;  // the host sizes its views by count[0]: never beyond the output capacity
    }
    if (idx >= (size_t)e.N * e.L) return;
    const int n = (int)(idx / e.L), i = (int)(idx - (size_t)n * e.L);
    const int j = w.jsel[idx];
    if (j == INT_MAX) return;
    int o = w.lpos[idx];
    for (int k = 0; k < n; ++k) o += e.count[1 + k];
    if (o >= e.cap) return;
    e.b_ids[o] = n; e.i_ids[o] = i; e.j_ids[o] = j;
    e.mconf[o] = w.psel[idx];
    // mkpts = [idx % w, idx // w] * (scale * scale{0,1}[b])   (coarse_matching.py:237-245)
    const float s0x = e.scale0 ? e.scale * e.scale0[n * 2 + 0] : e.scale;
    const float s0y = e.scale0 ? e.scale * e.scale0[n * 2 + 1] : e.scale;
    const float s1x = e.scale1 ? e.scale * e.scale1[n * 2 + 0] : e.scale;
    const float s1y = e.scale1 ? e.scale * e.scale1[n * 2 + 1] : e.scale;
    e.mk0[2 * o + 0] = (float)(i % e.w0c) * s0x;
    e.mk0[2 * o + 1] = (float)(i / e.w0c) * s0y;
    e.mk1[2 * o + 0] = (float)(j % e.w1c) * s1x;
    e.mk1[2 * o + 1] = (float)(j / e.w1c) * s1y;
}

int validate(const gim_coarse_args& a) {
    GIM_REQUIRE(a.feat0 && a.feat1 && a.ws && a.count, "coarse_match: NULL pointer");
    GIM_REQUIRE(a.N > 0 && a.L > 0 && a.S > 0, "coarse_match: bad sizes");
    GIM_REQUIRE(a.feat_dtype == GIM_F32 || a.feat_dtype == GIM_H16, "coarse_match: feat_dtype %d", a.feat_dtype);
    GIM_REQUIRE(a.C > 0 && a.C % (a.feat_dtype == GIM_H16 ? 64 : 32) == 0, "coarse_match: C=%d must be a multiple of the 128-byte K slab", a.C);
    GIM_REQUIRE(a.ldf == 0 || (a.ldf >= a.C && a.ldf % (a.feat_dtype == GIM_H16 ? 8 : 4) == 0), "coarse_match: ldf=%d", a.ldf);
    GIM_REQUIRE(a.h0c * a.w0c == a.L && a.h1c * a.w1c == a.S, "coarse_match: hw0_c/hw1_c do not match L/S");
    GIM_REQUIRE((int64_t)a.L * (a.ldf ? a.ldf : a.C) * 4 < (int64_t)0xFFFFFFF0ll && (int64_t)a.S * (a.ldf ? a.ldf : a.C) * 4 < (int64_t)0xFFFFFFF0ll, "coarse_match: feature map too large");
    GIM_REQUIRE(a.temperature > 0.f, "coarse_match: temperature must be positive");
    GIM_REQUIRE(a.thr >= 0.05f, "coarse_match: thr=%g below 0.05 (the candidate buffers hold 20 entries per row = 1/0.05; gim_loftr uses 0.2)", (double)a.thr);
    GIM_REQUIRE((a.mask0 == nullptr) == (a.mask1 == nullptr), "coarse_match: mask0 and mask1 must be given together");
    return GIM_OK;
}

// GIM_CM_PANEL: 1 (default) = row-panel statistics kernel where it applies, 0 = tile-per-workgroup kernel always,
// 2 = row-panel kernel AND its fallback forced (tests: the gated kernel runs although the range guard did not trip)
// GIM_CM_PANEL: 0 (default) the tile-per-workgroup statistics kernel; 1 the row-panel kernel (+ gated fallback); 2 the row-panel
// kernel with the fallback forced (tests).  Measured on MI355X (profiles/r03_cm_panel.txt): the panel kernel is exact but SLOWER
// -- 410 us against 320 us on planted features, +0.7 ms per step on the bench's random-weight features (one wave per SIMD at 259
// VGPRs has nothing to hide its ~3000 VALU instructions per tile behind; 912 workgroups = 3.56 rounds of one workgroup per CU) --
// so it stays opt-in until it wins.
static int panel_mode() { static const int v = [] { const char* e = getenv("GIM_CM_PANEL"); return e ? atoi(e) : 0; }(); return v; }

template <typename K>
int set_smem(K kern) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_SMEM);
    if (e != hipSuccess) { gim_set_error("coarse_match: hipFuncSetAttribute: %s", hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
    return GIM_OK;
}

}  // namespace

extern "C" int64_t GIM_FN(gim_coarse_match_ws_bytes)(int N, int L, int S) {
    CmWs w;
    return (int64_t)carve(w, nullptr, N, L, S, 1024) + 256;
}

static int cm_prepare(const gim_coarse_args& a, CmWs& w, CmGeom& g) {
    carve(w, (char*)a.ws, a.N, a.L, a.S, a.C);
    g.feat0 = a.feat0; g.feat1 = a.feat1; g.bf16 = a.feat_dtype == GIM_H16; g.ldf = a.ldf ? a.ldf : a.C; g.mask0 = a.mask0; g.mask1 = a.mask1; g.N = a.N; g.L = a.L; g.S = a.S; g.C = a.C;
    g.inv_c = 1.0f / (float)a.C; g.temperature = a.temperature; g.thr = a.thr;
    g.inv_ct = 1.0f / ((float)a.C * a.temperature);
    static GimPerDevice attr;
    if (attr.needed()) {
        int rc = set_smem(cm_stats_kernel<false>);
        if (rc == GIM_OK) rc = set_smem(cm_stats_kernel<true>);
        if (rc == GIM_OK) rc = set_smem(cm_cand_kernel<0, false>);
        if (rc == GIM_OK) rc = set_smem(cm_cand_kernel<1, false>);
        if (rc == GIM_OK) rc = set_smem(cm_cand_kernel<0, true>);
        if (rc == GIM_OK) rc = set_smem(cm_cand_kernel<1, true>);
        if (rc == GIM_OK && hipFuncSetAttribute((const void*)cm_stats_panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P2_SMEM) != hipSuccess) {
            gim_set_error("coarse_match: hipFuncSetAttribute(panel kernel, %d B LDS)", P2_SMEM);
            rc = GIM_ERR_LAUNCH;
        }
        if (rc != GIM_OK) return rc;
        attr.done();
    }
    return GIM_OK;
}

#if !GIM_HALF_KIND
extern "C" int gim_coarse_match_f16(const gim_coarse_args* ap, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_coarse_match)(const gim_coarse_args* ap, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (ap && ap->feat_dtype == GIM_F16) return gim_coarse_match_f16(ap, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(ap, "coarse_match: NULL args");
    const gim_coarse_args& a = *ap;
    int rc = validate(a);
    if (rc != GIM_OK) return rc;
    GIM_REQUIRE(a.b_ids && a.i_ids && a.j_ids && a.mconf && a.mkpts0_c && a.mkpts1_c && a.cap > 0, "coarse_match: NULL output");
    CmWs w; CmGeom g;
    rc = cm_prepare(a, w, g);
    if (rc != GIM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t nmax = (size_t)a.N * (a.L > a.S ? a.L : a.S);
    hipLaunchKernelGGL(cm_init_kernel, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, s, w, a.N, a.L, a.S);
    hipLaunchKernelGGL(cm_ktab_kernel, dim3(1), dim3(256), 0, s, w.ktab, a.C, g.bf16 ? 8 : 4);
    dim3 tgrid((unsigned)(w.ntL * w.ntS), (unsigned)a.N);
    // row-panel statistics kernel: 16-bit features of 256 channels, no padding masks, enough column tiles for its three-way split
    const bool panel = g.bf16 && a.C == 256 && !a.mask0 && w.ntS >= 2 * PJ && panel_mode();
    if (panel) {
        hipLaunchKernelGGL(cm_stats_panel_kernel, dim3((unsigned)(w.ntL * PJ), (unsigned)a.N), dim3(256), P2_SMEM, s, g, w, panel_mode() == 2 ? 1 : 0);
        hipLaunchKernelGGL(cm_stats_kernel<true>, tgrid, dim3(256), TILE_SMEM, s, g, w, 1);   // gated fallback (wide logit range)
    } else if (g.bf16) hipLaunchKernelGGL(cm_stats_kernel<true>, tgrid, dim3(256), TILE_SMEM, s, g, w, 0);
    else hipLaunchKernelGGL(cm_stats_kernel<false>, tgrid, dim3(256), TILE_SMEM, s, g, w, 0);
    hipLaunchKernelGGL(cm_combine_kernel, dim3((unsigned)(((size_t)a.N * a.L + 255) / 256)), dim3(256), 0, s, w.rowpart, w.rowstat, a.N, a.L, w.ntS,
                       panel ? PJ : 0, w.npre + a.N + 1);
    hipLaunchKernelGGL(cm_combine_kernel, dim3((unsigned)(((size_t)a.N * a.S + 255) / 256)), dim3(256), 0, s, w.colpart, w.colstat, a.N, a.S, w.ntL,
                       0, w.npre + a.N + 1);
    hipLaunchKernelGGL(cm_precand_kernel, dim3((unsigned)((w.capp + 255) / 256), (unsigned)a.N), dim3(256), 0, s, g, w);
    if (g.bf16) hipLaunchKernelGGL((cm_cand_kernel<0, true>), tgrid, dim3(256), TILE_SMEM, s, g, w, (float*)nullptr);
    else hipLaunchKernelGGL((cm_cand_kernel<0, false>), tgrid, dim3(256), TILE_SMEM, s, g, w, (float*)nullptr);
    if (a.mask0) hipLaunchKernelGGL(cm_mask_extent_kernel, dim3((unsigned)a.N), dim3(256), 0, s, a.mask0, a.mask1, w.ext, a.h0c, a.w0c, a.h1c, a.w1c);
    BorderGeom bg{a.h0c, a.w0c, a.h1c, a.w1c, a.border_rm, a.mask0 ? w.ext : nullptr};
    dim3 cgrid((unsigned)((w.capc + 255) / 256), (unsigned)a.N);
    hipLaunchKernelGGL(cm_select_kernel<0>, cgrid, dim3(256), 0, s, w, a.N, a.L, a.S, bg);
    hipLaunchKernelGGL(cm_select_kernel<1>, cgrid, dim3(256), 0, s, w, a.N, a.L, a.S, bg);
    hipLaunchKernelGGL(cm_scan_kernel, dim3((unsigned)a.N), dim3(1024), 0, s, w, a.L, a.count);
    EmitArgs e{a.b_ids, a.i_ids, a.j_ids, a.mconf, a.mkpts0_c, a.mkpts1_c, a.scale0, a.scale1, a.count,
               a.N, a.L, a.cap, a.w0c, a.w1c, a.scale};
    hipLaunchKernelGGL(cm_emit_kernel, dim3((unsigned)(((size_t)a.N * a.L + 255) / 256)), dim3(256), 0, s, w, e);
    return gim_check_launch("coarse_match");
}

#if !GIM_HALF_KIND
extern "C" int gim_coarse_conf_matrix_f16(const gim_coarse_args* ap, float* conf, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_coarse_conf_matrix)(const gim_coarse_args* ap, float* conf, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (ap && ap->feat_dtype == GIM_F16) return gim_coarse_conf_matrix_f16(ap, conf, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(ap && conf, "coarse_conf_matrix: NULL args");
    const gim_coarse_args& a = *ap;
    int rc = validate(a);
    if (rc != GIM_OK) return rc;
    CmWs w; CmGeom g;
    rc = cm_prepare(a, w, g);
    if (rc != GIM_OK) return rc;
    dim3 tgrid((unsigned)(w.ntL * w.ntS), (unsigned)a.N);
    if (g.bf16) hipLaunchKernelGGL((cm_cand_kernel<1, true>), tgrid, dim3(256), TILE_SMEM, (hipStream_t)stream, g, w, conf);
    else hipLaunchKernelGGL((cm_cand_kernel<1, false>), tgrid, dim3(256), TILE_SMEM, (hipStream_t)stream, g, w, conf);
    return gim_check_launch("coarse_conf_matrix");
}

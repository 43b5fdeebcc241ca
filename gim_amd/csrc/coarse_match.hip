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
    int* npre;         // [N] pre-candidate counters, npre[N] = overflow flag
    PreCand* pre;      // [N][capp]
    int* ktab;         // dense K table for the mainloop
    int ntL, ntS, capc, capp;
    int* health;       // caller's count[1]: bit 0 = a non-finite similarity reached the statistics (inf / NaN features: the fp16 mode's
                       // range guard, loftr.py reads it with the match count); NULL when the caller has no count buffer
    int ntL64;         // column-partial slots of the 256-tile statistics kernel (64 rows each); colpart holds max(ntL, ntL64) slots per pair
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

size_t carve(CmWs& w, char* base, int N, int L, int S, int C, int per_row = 16) {
    w.ntL = (L + BM - 1) / BM;
    w.ntS = (S + BN - 1) / BN;
    w.ntL64 = (L + 63) / 64;
    w.capc = 20 * L + 64;  // < 1/thr entries of a row can exceed thr (sum_j softmax_j <= 1); validate() enforces thr >= 0.05
    // pre-candidate capacity: 16 entries per row (what gim_coarse_match_ws_bytes sizes the workspace for); gim_coarse_args.precand_per_row
    // lets a caller SHRINK it (tests force the overflow -> recompute fallback with -1 = none)
    w.capp = per_row * L + 1024;
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += al(bytes); return p; };
    w.rowpart = (float2*)take((size_t)N * w.ntS * L * 8);
    w.colpart = (float2*)take((size_t)N * w.ntL64 * S * 8);   // ntL64 >= ntL
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
    w.npre = (int*)take((size_t)(N + 1) * 4);
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
template <bool BF16>
__device__ __forceinline__ void cm_stats_tile(const CmGeom& g, const CmWs& w, const int n, const int mt, const int nt, char* smem) {
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
            if (m0 + idx < g.L) {
                w.rowpart[((size_t)n * w.ntS + nt) * g.L + m0 + idx] = make_float2(m, zz);
                if (!(zz < INFINITY) && w.health) atomicOr(w.health, 1);   // NaN / inf features (fp16 overflow upstream)
            }
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


// One workgroup per tile (grid = N * ntL * ntS)
template <bool BF16>
__global__ void __launch_bounds__(256) cm_stats_kernel(const CmGeom g, const CmWs w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per = w.ntL * w.ntS, total = per * g.N;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / per, r = tile - n * per, mt = r / w.ntS;
        cm_stats_tile<BF16>(g, w, n, mt, r - mt * w.ntS, smem);
        __syncthreads();   // (block-uniform early returns inside the tile function) the next tile's staging overwrites this one's LDS
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Pass A on 256 x 256 tiles, statistics straight from the accumulators (16-bit features, no padding masks): round 4.
//
// The 128 x 128 tile kernel above spends ~14 us per tile for 1 us of MFMAs (profiles/r03_final_kernel_stats.txt: 311 us per batch-8
// call): both operand panels of every tile are re-staged (64 flop per staged byte = 64 B/clk/CU at the full MFMA rate, twice what
// L2 -> LDS delivers), the fp32 tile goes to LDS and is re-read twice behind five barriers, and every element costs TWO
// exponentials (the row and the column softmax are taken against different maxima).  Here
//   * persistent 8-wave workgroups (one per CU) walk 256 x 256 tiles (128 flop per staged byte), pair = tile % N so that with
//     8 pairs the two feature maps of pair n (4.8 MB) stay in XCD n's L2; the first K slab of the NEXT tile is in flight during
//     the statistics of this one;
//   * a wave owns 64 rows x 128 columns in its 128 accumulator registers (lane = row, registers = columns) and takes the
//     statistics straight from them: row maxima / sums in the lane (+ one exchange with lane ^ 32), column maxima / sums by a
//     halving butterfly over the 32 row lanes (v_permlane16_swap, DPP row_ror / quad_perm, one ds_swizzle level) -- the exact
//     (max, sum-exp) pairs of the wave's sub-tile, the same arithmetic as the 128 x 128 kernel on a different partition.
//     (A first version took ONE exponential per element against a wave-wide reference: 253 us per call on narrow logits, but the
//     bench's features span > 69 between a wave tile's matches and its weakest rows -- shared exponentials underflow there, its
//     range guard sent every call to the fallback and the step got slower; profiles/r04_cm_stats.txt.)
//   * nothing of the statistics touches shared LDS: the column maxima / thresholds are broadcast through a wave-private buffer;
//   * pre-candidates: same superset argument as above with the WAVE sub-tile's softmax factors (over 128 columns / 64 rows:
//     still >= the true factors), tested on the registers; the exact product test is left to cm_precand.
// Row partials land in the 128-column slots of the tile kernel's layout; column partials have 64-row granularity (ntL64 slots).
// Two workgroup shapes of the same kernel (measured in profiles/r04_cm_stats.txt; WN2 = 1 is what the library launches):
//   WN2 = 2: 256 x 256 tile, 8 waves, K slabs double-buffered, one workgroup per CU (128 flop per staged byte).  The two waves of a
//            SIMD run the same phase: MFMA pipe and VALU take turns.
//   WN2 = 1: 256 x 128 tile, 4 waves, ONE stage buffer, two workgroups per CU (85 flop per staged byte).  A workgroup cannot hide its
//            own staging, but its neighbour is in another phase: one wave of every SIMD multiplies while the other runs statistics.
constexpr int PL2 = 512;                                // per-tile pre-candidate list
template <int WN2> struct Cm2 {
    static constexpr int BN2 = 128 * WN2, NT = 256 * WN2, NBUF = WN2;   // tile columns, threads, stage buffers
    typedef gim::Igemm<256, BN2, 4, WN2, true, true> G;
    static constexpr int STAGE = NBUF * G::STAGE;
};
struct S2X {
    float tcol[8][128];     // wave-private: column maxima, then column thresholds of the wave's 128 columns (accumulator units)
    PreCand plist[PL2];
    int pcnt[4];            // [0] = local count, [1] = global base
};
template <int WN2> constexpr int s2_smem() { return Cm2<WN2>::STAGE + (int)sizeof(S2X); }
static_assert(s2_smem<2>() <= 160 * 1024 && 2 * s2_smem<1>() <= 160 * 1024, "256-row statistics kernel: LDS");
typedef f32x16_t Acc2[4][2];                            // [column fragment][row fragment] of a wave's 64 x 128 sub-tile

__device__ __forceinline__ float lane_xor_dpp_1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor_dpp_2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float lane_xor_dpp_8(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xf, 0xf, true)); }  // row_ror:8
__device__ __forceinline__ float lane_xor_swz_4(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F)); }             // bit mode: xor 4
// v + (v of lane ^ 32): v_permlane32_swap(v, v) leaves (v.lo, v.lo) in one result and (v.hi, v.hi) in the other -- their sum is
// the pair total in every lane, whichever operand the instruction calls "first"
__device__ __forceinline__ float pair32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pair32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Sum of c[r] over the 32 lanes that share (lane >> 5): halving butterfly.  Returns the total of ONE column per lane; which one
// (register index r in 0..15) is whatever `tag` says after the same network ran on the register indices (cm256_column_tag).
struct BflyAdd { __device__ __forceinline__ float operator()(float keep, float recv) const { return keep + recv; } };
struct BflyMax { __device__ __forceinline__ float operator()(float keep, float recv) const { return fmaxf(keep, recv); } };
struct BflyTag { __device__ __forceinline__ float operator()(float keep, float) const { return keep; } };
template <typename OP>
__device__ __forceinline__ float cm256_butterfly(float (&c)[16], const int lane, OP op) {
    float d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // lane ^ 16: v_permlane16_swap exchanges the odd 16-lane rows of one operand with the even rows of the other
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(c[k]), __float_as_uint(c[k + 8]), false, false);
        d[k] = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
    float e[4], f[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // lane ^ 8
        float lo = d[k], hi = d[k + 4];
        asm volatile("" : "+v"(lo), "+v"(hi));   // keep them values (hipcc otherwise turns the two selects into an indexed array read)
        e[k] = op(b3 ? hi : lo, lane_xor_dpp_8(b3 ? lo : hi));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {   // lane ^ 4
        float lo = e[k], hi = e[k + 2];
        asm volatile("" : "+v"(lo), "+v"(hi));
        f[k] = op(b2 ? hi : lo, lane_xor_swz_4(b2 ? lo : hi));
    }
    float lo = f[0], hi = f[1];
    asm volatile("" : "+v"(lo), "+v"(hi));
    const float g = op(b1 ? hi : lo, lane_xor_dpp_2(b1 ? lo : hi));   // lane ^ 2
    return op(g, lane_xor_dpp_1(g));                                    // lane ^ 1: both lanes of a pair end with the total
}
// register index (0..15) of the column a lane holds after cm256_butterfly: the network run on the indices themselves, so the map
// does not depend on which operand v_permlane16_swap calls "first"
__device__ __forceinline__ int cm256_column_tag(const int lane) {
    float c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = (float)r;
    return (int)cm256_butterfly(c, lane, BflyTag());
}

// Statistics of one wave sub-tile of 64 rows x 32 NI columns held in the accumulators (lane = row, registers = columns):
//   acc[i][j][4 rg + e] = sim(row0 + 32 j, colw + 32 i + 8 rg + 4 lh + e) * C * T.
// Column partials (64-row slot `cslot`) are stored here; the row partials are stored into slot `rslot` (ROWSTORE) or handed back
// (rmax[j], rsum[j]: maximum in accumulator units and sum-exp of this lane's two rows, valid in both lanes of a pair) for the
// caller's online merge across tiles.  `tc`: wave-private LDS buffer of 32 NI floats.
template <bool EDGE, int NI, bool ROWSTORE>
__device__ __forceinline__ void cm_wave_stats(const CmGeom& g, const CmWs& w, f32x16_t (&acc)[NI][2], const int n, const int row0, const int colw,
                                              const int rslot, const int cslot, const bool wave_rows, float* tc, PreCand* plist, int* pcnt,
                                              const int tag, float (&rmax)[2], float (&rsum)[2]) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const float NEG = -INFINITY;
    const int coff = 8 * (tag >> 2) + 4 * lh + (tag & 3);   // the column (within a 32-column fragment) this lane owns after a butterfly
    // ---- references: exact maxima of the wave's sub-tile, per row (in the lane) and per column (butterfly over the row lanes) ----
    float rm0 = NEG, rm1 = NEG, cml[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (EDGE) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool rok = row0 + 32 * j < g.L;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!(rok && colw + 32 * i + 8 * (r >> 2) + 4 * lh + (r & 3) < g.S)) acc[i][j][r] = NEG;
            }
        }
        float c[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            rm0 = fmaxf(rm0, acc[i][0][r]);
            rm1 = fmaxf(rm1, acc[i][1][r]);
            c[r] = fmaxf(acc[i][0][r], acc[i][1][r]);
        }
        cml[i] = cm256_butterfly(c, lane, BflyMax());
        if (EDGE) cml[i] = cml[i] == NEG ? 0.f : cml[i];   // column beyond S (or no valid row): any finite reference
        if ((l31 & 1) == 0) tc[32 * i + coff] = cml[i];
    }
    rm0 = pair32_max(rm0);
    rm1 = pair32_max(rm1);
    if (EDGE) { rm0 = rm0 == NEG ? 0.f : rm0; rm1 = rm1 == NEG ? 0.f : rm1; }
    // ---- two exponentials per element (row softmax against the row maximum, column softmax against the column maximum) ----
    const float k2 = g.inv_ct * 1.44269504088896341f;   // exp(s - m) = exp2((acc - macc) * inv_ct * log2 e); acc - macc is exact where it matters
    float rs0 = 0.f, rs1 = 0.f, cs[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float c[16];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 m4 = *(const float4*)(tc + 32 * i + 8 * rg + 4 * lh);
            const float mq[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * rg + e;
                rs0 += __builtin_amdgcn_exp2f((acc[i][0][r] - rm0) * k2);
                rs1 += __builtin_amdgcn_exp2f((acc[i][1][r] - rm1) * k2);
                c[r] = __builtin_amdgcn_exp2f((acc[i][0][r] - mq[e]) * k2) + __builtin_amdgcn_exp2f((acc[i][1][r] - mq[e]) * k2);
            }
        }
        cs[i] = cm256_butterfly(c, lane, BflyAdd());
    }
    rs0 = pair32_sum(rs0);
    rs1 = pair32_sum(rs1);
    rmax[0] = rm0; rmax[1] = rm1; rsum[0] = rs0; rsum[1] = rs1;
    // ---- partials: (maximum, sum) in similarity units ----
    const bool wave_cols = colw < g.S;
    if (wave_cols && lh == 0) {
        bool bad = false;   // every element feeds a row sum: NaN / inf anywhere in the features (fp16 overflow upstream) shows here
        if (row0 < g.L) { if (ROWSTORE) w.rowpart[((size_t)n * w.ntS + rslot) * g.L + row0] = make_float2(rm0 * g.inv_ct, rs0); bad |= !(rs0 < INFINITY); }
        if (row0 + 32 < g.L) { if (ROWSTORE) w.rowpart[((size_t)n * w.ntS + rslot) * g.L + row0 + 32] = make_float2(rm1 * g.inv_ct, rs1); bad |= !(rs1 < INFINITY); }
        if (bad && w.health) atomicOr(w.health, 1);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int col = colw + 32 * i + coff;
        if (wave_rows && (l31 & 1) == 0 && col < g.S) w.colpart[((size_t)n * w.ntL64 + cslot) * g.S + col] = make_float2(cml[i] * g.inv_ct, cs[i]);
    }
    // ---- pre-candidates.  A row can only hold one if its LARGEST element passes the row test, i.e. if the wave-local softmax of
    // its maximum, 1 / rs, exceeds thr: two compares per lane screen the whole sub-tile.  Rows of tiles the matches do not run
    // through have rs >> 1 / thr; only the wave tiles that do hold a peaked row run the scan that names the elements ----
    const float thr_pre = g.thr * (1.0f - 1e-3f);   // slack: rounding must never drop a true candidate
    const bool hot0 = thr_pre * rs0 < 1.0f, hot1 = thr_pre * rs1 < 1.0f;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hot0 | hot1) != 0ull, 0)) {
        const float ct = (float)g.C * g.temperature;
        // thresholds in accumulator units: s > m + log(thr z)  <=>  acc > macc + log(thr z) * C * T; the column ones are broadcast
        // through the wave's buffer (the maxima in it were consumed above; LDS operations of one wave execute in order)
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if ((l31 & 1) == 0) tc[32 * i + coff] = cml[i] + __logf(thr_pre * cs[i]) * ct;
        const float tr0 = hot0 ? rm0 + __logf(thr_pre * rs0) * ct : INFINITY, tr1 = hot1 ? rm1 + __logf(thr_pre * rs1) * ct : INFINITY;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 t4 = *(const float4*)(tc + 32 * i + 8 * rg + 4 * lh);
                const float tq[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float tr = j ? tr1 : tr0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (acc[i][j][4 * rg + e] > fmaxf(tr, tq[e])) {
                            const int k = atomicAdd(&pcnt[0], 1);
                            if (k < PL2) plist[k] = PreCand{row0 + 32 * j, colw + 32 * i + 8 * rg + 4 * lh + e, acc[i][j][4 * rg + e] * g.inv_ct};
                        }
                }
            }
    }
}

template <bool EDGE, int WN2>
__device__ __forceinline__ void cm256_stats(const CmGeom& g, const CmWs& w, Acc2& acc, const int n, const int m0, const int n0,
                                            S2X& X, const int tag) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int wm, wn;
    Cm2<WN2>::G::wave_mn(wave, wm, wn);
    const int colw = n0 + wn * 128;
    float rmax[2], rsum[2];
    cm_wave_stats<EDGE, 4, true>(g, w, acc, n, m0 + wm * 64 + (threadIdx.x & 31), colw, colw >> 7, (m0 >> 6) + wm, m0 + wm * 64 < g.L,
                                 X.tcol[wave], X.plist, X.pcnt, tag, rmax, rsum);
}

template <int WN2>
__global__ void __launch_bounds__(256 * WN2, 2) cm_stats256_kernel(const CmGeom g, const CmWs w) {
    typedef Cm2<WN2> K;
    typedef typename K::G G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    S2X& X = *(S2X*)(smem + K::STAGE);
    const int t = threadIdx.x;
    const int ntL2 = (g.L + 255) >> 8, ntS2 = (g.S + K::BN2 - 1) / K::BN2;
    const int total = g.N * ntL2 * ntS2;
    const int nkt = g.C * 2 / KTB;
    const int tag = cm256_column_tag(t & 63);
    if (t == 0) X.pcnt[0] = 0;
    // (round 5: starting every second workgroup 2 ... 16 us late -- by blockIdx parity or by blockIdx / CUs -- so that the two co-resident
    //  workgroups of a CU alternate their MFMA and statistics phases instead of running them in lockstep changes nothing: 222.9 vs 224.5 us per
    //  call on random features, 849-864 us on planted ones, profiles/r05_cm_stats.txt)
    gim::MainloopArgs ml;
    ml.ktab = nullptr;
    ml.x_bytes = (unsigned)(((size_t)(g.L - 1) * g.ldf + g.C) * 2);
    ml.w_bytes = (unsigned)(((size_t)(g.S - 1) * g.ldf + g.C) * 2);
    ml.H = 1; ml.W = g.L; ml.Ho = 1; ml.Wo = g.L; ml.stride = 1; ml.pad = 0; ml.ldx = g.ldf;
    ml.kpad = g.C; ml.ldw = g.ldf; ml.M = g.L;
    // dense K table entry of this thread's staging slot: K group kt * 8 + (slot ^ swizzle) -> channel 8 * group, no tap offset
    const int kgrp = (t & 7) ^ (((t >> 3) >> 1) & 7);
    G gg;
    int n = 0, m0 = 0, n0 = 0;
    auto locate = [&](const int tile) __attribute__((always_inline)) {
        n = tile % g.N;
        const int r = tile / g.N, mt = r / ntS2;
        m0 = mt << 8; n0 = (r - mt * ntS2) * K::BN2;
        ml.x = (const char*)g.feat0 + (size_t)n * g.L * g.ldf * 2;
        ml.w = (const char*)g.feat1 + (size_t)n * g.S * g.ldf * 2;
        gg.decode(ml, m0, n0);
    };
    int tile = blockIdx.x, sc = 0;   // sc: running slab counter, slab sc lives in stage buffer sc & (NBUF - 1)
    if (tile < total) { locate(tile); gg.stage_issue(ml, smem, 0, 0, kgrp * 8); }
    while (tile < total) {
        Acc2 acc;
        G::zero(acc);
        for (int kt = 0; kt < nkt; ++kt, ++sc) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                        // slab sc has landed for everybody (two buffers: and everybody is done with slab sc - 1)
            if (K::NBUF == 2 && kt + 1 < nkt) gg.stage_issue(ml, smem, (sc + 1) & 1, kt + 1, ((kt + 1) * 8 + kgrp) * 8);
            G::compute(smem, sc & (K::NBUF - 1), acc);
            if (K::NBUF == 1) {
                __syncthreads();                    // one buffer: everybody is done with it before the next slab overwrites it
                if (kt + 1 < nkt) gg.stage_issue(ml, smem, 0, kt + 1, ((kt + 1) * 8 + kgrp) * 8);
            }
        }
        const int cn = n, cm0 = m0, cn0 = n0;
        const int next = tile + gridDim.x;
        // the target buffer is free (two buffers: last read one slab ago, a barrier since; one buffer: the barrier above): the next
        // tile's first slab travels during the statistics
        if (next < total) { locate(next); gg.stage_issue(ml, smem, sc & (K::NBUF - 1), 0, kgrp * 8); }
        if ((cm0 + 256 <= g.L) && (cn0 + K::BN2 <= g.S)) cm256_stats<false, WN2>(g, w, acc, cn, cm0, cn0, X, tag);   // block-uniform
        else cm256_stats<true, WN2>(g, w, acc, cn, cm0, cn0, X, tag);
        __syncthreads();
        const int cnt = X.pcnt[0];
        if (cnt > 0) {                                // block-uniform
            if (cnt > PL2) {
                if (t == 0) w.npre[g.N] = 1;          // recompute path (cm_cand_kernel<0>)
            } else {
                if (t == 0) X.pcnt[1] = atomicAdd(&w.npre[cn], cnt);
                __syncthreads();
                const int base = X.pcnt[1];
                if (base + cnt > w.capp) { if (t == 0) w.npre[g.N] = 1; }
                else for (int k = t; k < cnt; k += K::NT) w.pre[(size_t)cn * w.capp + base + k] = X.plist[k];
            }
            __syncthreads();
            if (t == 0) X.pcnt[0] = 0;                // the next pushes come behind the next tile's barriers
        }
        tile = next;
    }
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

// stat[n][x] = combine over the first `ntile` of a pair's `stride` partial slots:  m = max m_t ; z = sum_t z_t * exp(m_t - m)   (fixed order:
// deterministic).  Four threads share an element (slots t, t + 4, ...) so that a pair's 38 ... 75 dependent-latency loads become
// 10 ... 19: the two launches took 20 us each as one thread per element.
struct CombineJob { const float2* part; float2* stat; int len, ntile; };   // `ntile` partial slots per pair, all of them filled
constexpr int CMB_INFLIGHT = 20;   // slots per thread held in registers: covers 80 partial slots (4800 tokens: 38 row / 75 column slots)
// one launch for both directions (round 5): blockIdx.y = 0 the row statistics, 1 the column statistics
__global__ void __launch_bounds__(256) cm_combine_kernel(const CombineJob jr, const CombineJob jc, int N) {
    const CombineJob& J = blockIdx.y ? jc : jr;
    const float2* __restrict__ part = J.part;
    float2* __restrict__ stat = J.stat;
    const int len = J.len, ntile = J.ntile, stride = J.ntile;
    if ((size_t)blockIdx.x * 64 >= (size_t)N * len) return;   // block-uniform: the grid covers the longer direction
    __shared__ float2 sh[4][64];
    const int xq = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const size_t idx = (size_t)blockIdx.x * 64 + xq;
    const bool ok = idx < (size_t)N * len;
    float m = -INFINITY, z = 0.f;
    if (ok) {
        const size_t n = idx / len, x = idx - n * len;
        const float2* p = part + n * (size_t)stride * len + x;   // `stride` slots per pair, the first `ntile` of them filled
        if (ntile <= 4 * CMB_INFLIGHT) {
            // all of this thread's slots requested at once (one memory latency instead of 10 ... 19 dependent ones per pass), then the same
            // arithmetic in the same order
            float2 v[CMB_INFLIGHT];
#pragma unroll
            for (int q = 0; q < CMB_INFLIGHT; ++q) {
                const int t = tq + 4 * q;
                v[q] = t < ntile ? p[(size_t)t * len] : make_float2(-INFINITY, 0.f);
            }
#pragma unroll
            for (int q = 0; q < CMB_INFLIGHT; ++q) m = fmaxf(m, v[q].x);
#pragma unroll
            for (int q = 0; q < CMB_INFLIGHT; ++q)
                if (tq + 4 * q < ntile) z += v[q].y * expf(v[q].x - m);
        } else {
            for (int t = tq; t < ntile; t += 4) m = fmaxf(m, p[(size_t)t * len].x);
            for (int t = tq; t < ntile; t += 4) {
                const float2 v = p[(size_t)t * len];
                z += v.y * expf(v.x - m);
            }
        }
    }
    sh[tq][xq] = make_float2(m, z);
    __syncthreads();
    if (tq == 0 && ok) {
        float mm = m;
#pragma unroll
        for (int q = 1; q < 4; ++q) mm = fmaxf(mm, sh[q][xq].x);
        float zz = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float2 v = sh[q][xq]; zz += v.x == -INFINITY ? 0.f : v.y * expf(v.x - mm); }
        stat[idx] = make_float2(mm, zz);
    }
}

__global__ void cm_init_kernel(const CmWs w, int N, int L, int S, int C, int ge) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)N * L) { w.rowmaxP[idx] = 0u; w.jsel[idx] = INT_MAX; w.psel[idx] = 0.f; }
    if (idx < (size_t)N * S) w.colmaxP[idx] = 0u;
    if (idx < (size_t)N) w.ncand[idx] = 0;
    if (idx <= (size_t)N) w.npre[idx] = 0;
    if (idx == 0 && w.health) *w.health &= ~1;   // bits 1 (fine kernel of the PREVIOUS call on this buffer) and 2 (range guard of the kernels in front of this call) are sticky: the host clears them
    // dense K table (same launch since round 5): K group g -> channel ge * g (ge = 4 fp32 / 8 16-bit values per 16 B); 2 padding slabs
    const int ng = (C / 32 + 2) * 8;
    for (size_t g = idx; g < (size_t)ng; g += (size_t)gridDim.x * blockDim.x) w.ktab[g] = ((int)g * ge < C) ? (int)g * ge : (int)0xFF000000;
}

// MODE 0: emit candidates.  MODE 1: write the conf tile to `conf` (lazy data['conf_matrix']).
template <int MODE, bool BF16>
__device__ __forceinline__ void cm_cand_tile(const CmGeom& g, const CmWs& w, float* __restrict__ conf, const int n, const int mt, const int nt, char* smem) {
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

// MODE 0 is launched as a small persistent grid (it is the overflow fallback and normally leaves at once), MODE 1 with one
// workgroup per tile
template <int MODE, bool BF16>
__global__ void __launch_bounds__(256) cm_cand_kernel(const CmGeom g, const CmWs w, float* __restrict__ conf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (MODE == 0 && !w.npre[g.N]) return;  // fallback only: the pre-candidate path did the work
    const int per = w.ntL * w.ntS, total = per * g.N;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / per, r = tile - n * per, mt = r / w.ntS;
        cm_cand_tile<MODE, BF16>(g, w, conf, n, mt, r - mt * w.ntS, smem);
        __syncthreads();
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
    if (t == 0) count[2 + n] = running;
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
        for (int k = 0; k < e.N; ++k) s += e.count[2 + k];
        e.count[0] = s < e.cap ? s : e.cap;  // the host sizes its views by count[0]: never beyond the output capacity
    }
    if (idx >= (size_t)e.N * e.L) return;
    const int n = (int)(idx / e.L), i = (int)(idx - (size_t)n * e.L);
    const int j = w.jsel[idx];
    if (j == INT_MAX) return;
    int o = w.lpos[idx];
    for (int k = 0; k < n; ++k) o += e.count[2 + k];
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
    carve(w, (char*)a.ws, a.N, a.L, a.S, a.C, a.precand_per_row == 0 ? 16 : (a.precand_per_row < 0 ? 0 : (a.precand_per_row > 16 ? 16 : a.precand_per_row)));
    w.health = a.count ? a.count + 1 : nullptr;
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
        if (rc == GIM_OK && hipFuncSetAttribute((const void*)cm_stats256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, s2_smem<1>()) != hipSuccess) {
            gim_set_error("coarse_match: hipFuncSetAttribute(256-row statistics kernel, %d B LDS)", s2_smem<1>());
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
    hipLaunchKernelGGL(cm_init_kernel, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, s, w, a.N, a.L, a.S, a.C, g.bf16 ? 8 : 4);
    const unsigned ntiles = (unsigned)(w.ntL * w.ntS * a.N), nfallback = ntiles < 512u ? ntiles : 512u;
    // 256-tile statistics kernel: 16-bit features, no padding masks (K in whole 128-byte slabs is validate()'s rule already)
    const bool big = g.bf16 && !a.mask0;
    if (big) {
        constexpr int wn2 = 1;   // 256 x 128 tiles, two 4-wave workgroups per CU (the 256 x 256 / 8-wave shape lost its A/B: profiles/r04_cm_stats.txt)
        int ncu = 256;
        { int dev = 0; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); }
        // persistent: one (two) workgroup(s) per CU; a grid that is a multiple of N (and of 8) keeps pair = tile % N on one XCD per workgroup
        const unsigned tiles2 = (unsigned)(a.N * ((a.L + 255) / 256) * ((a.S + 128 * wn2 - 1) / (128 * wn2)));
        unsigned grid = (unsigned)ncu * (wn2 == 1 ? 2u : 1u);
        if (grid > tiles2) grid = tiles2;
        hipLaunchKernelGGL(cm_stats256_kernel<1>, dim3(grid), dim3(256), s2_smem<1>(), s, g, w);
    } else if (g.bf16) hipLaunchKernelGGL(cm_stats_kernel<true>, dim3(ntiles), dim3(256), TILE_SMEM, s, g, w);
    else hipLaunchKernelGGL(cm_stats_kernel<false>, dim3(ntiles), dim3(256), TILE_SMEM, s, g, w);
    hipLaunchKernelGGL(cm_combine_kernel, dim3((unsigned)((nmax + 63) / 64), 2u), dim3(256), 0, s, CombineJob{w.rowpart, w.rowstat, a.L, w.ntS},
                       CombineJob{w.colpart, w.colstat, a.S, big ? w.ntL64 : w.ntL}, a.N);
    hipLaunchKernelGGL(cm_precand_kernel, dim3((unsigned)((w.capp + 255) / 256), (unsigned)a.N), dim3(256), 0, s, g, w);
    if (g.bf16) hipLaunchKernelGGL((cm_cand_kernel<0, true>), dim3(nfallback), dim3(256), TILE_SMEM, s, g, w, (float*)nullptr);
    else hipLaunchKernelGGL((cm_cand_kernel<0, false>), dim3(nfallback), dim3(256), TILE_SMEM, s, g, w, (float*)nullptr);
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
    dim3 tgrid((unsigned)(w.ntL * w.ntS * a.N));
    if (g.bf16) hipLaunchKernelGGL((cm_cand_kernel<1, true>), tgrid, dim3(256), TILE_SMEM, (hipStream_t)stream, g, w, conf);
    else hipLaunchKernelGGL((cm_cand_kernel<1, false>), tgrid, dim3(256), TILE_SMEM, (hipStream_t)stream, g, w, conf);
    return gim_check_launch("coarse_conf_matrix");
}

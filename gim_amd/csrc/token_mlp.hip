// Fused token-wise tail of a LoFTREncoderLayer for the coarse level (d_model 256, bf16 operand mode) on gfx950:
//
//     msg  = norm1(merge(attention output))                                 transformer.py:52-53
//     msg  = norm2(mlp.2(relu(mlp.0(cat[x, msg]))))                          transformer.py:55-57
//     x   += msg                                                             transformer.py:58
//
// Optionally the attention's apply step runs in the prologue (msg = (Q KV) Z S from the query rows and the [32 x 32] per-head
// state, attentions.py:44-45), so the attention output never exists in memory either.
// ONE launch instead of merge GEMM -> LayerNorm -> mlp.0 GEMM -> mlp.2 GEMM -> LayerNorm+residual, and none of the five
// intermediate row buffers (merge output, norm1 output, 512-wide hidden layer, mlp.2 output) touches HBM.  Everything after
// the attention is row-local, so a 256-thread workgroup owns 64 token rows from the attention output to the updated stream:
//
//   LDS   A   [64 rows][256] bf16   attention output, later norm1(msg)        32 KiB
//         X   [64 rows][256] bf16   operand copy of x (the other half of cat)  32 KiB
//         H   [64 rows][128] bf16   one quarter of the hidden layer             16 KiB   (+ LayerNorm partial sums while it is idle)
//   two workgroups per CU (80 KiB each): one's epilogues / barrier waits run under the other's MFMAs.
//
// The 512-wide hidden layer is produced and consumed in four 128-column quarters: relu([x | msg] W0[q]) -> H -> out += H W2[:, q];
// `out` (64 x 256 fp32) stays in the accumulators of the 4 waves (64 columns each) across the quarters.
// Weights stream from L2 straight into registers in a per-wave fragment order fixed at pack time (gim_amd/packing.py::
// pack_token_mlp): 28 units of 8 fragments per wave = 896 KiB per workgroup, each unit requested right behind the MFMAs of the
// previous one.  v_mfma_f32_32x32x16_bf16, fp32 accumulate; LayerNorm statistics and the residual stream are fp32
// (x32 is read, updated and written once, through an LDS transposition so that every global access is a full 256-byte row
// segment); the operand copy of the new x is written as bf16.
//
// PROJECTION BLOCKS (round 3): the new x is the operand of the NEXT attention's q / k / v projections (transformer.py:42-44 of the
// following layer -- and, in a cross layer, the k / v of the same layer's second call), each a bias-free [256 x 256] Linear.  Up to
// six of them run here on the tile that is still in LDS: emit[b] = act_b(x_new W_b^T) (elu+1 on q and k, attentions.py:31-32),
// per block a row range (a self layer over both images projects k / v only for the side that is a source next), weights streamed
// the same way (4 units per block and wave).  The stand-alone projection GEMMs (K = 256: 4 slabs of prologue / epilogue per tile,
// 350 TFLOP/s) and their re-read of x disappear.
//
// LOCAL QUERIES (round 5): with `qwts` the elu + 1 query rows of the fused attention apply are not read from `msg` but projected HERE, from
// the operand copy of x the tile loads anyway (q = elu(x Wq^T) + 1, transformer.py:42, attentions.py:31): 4 weight units in front of the
// merge product, written to the A tile in the accumulator's row pattern.  The tail that updated the rows no longer emits a q block: 32 KiB
// of stores and 32 KiB of loads per tile less, the same MFMAs in the same order (the values are bit-identical to the emitted ones).
//
// FUSED KV STATE (round 5): the k and v rows of a sequence have ONE reader -- the linear attention's state reduction KV = K^T V / S,
// Ksum = K^T 1 (attentions.py:38-43) of the call they are the source of.  A (k, v) block pair flagged with `kv_part` is therefore not
// written at all: both projections are computed with the MFMA operands SWAPPED (tokens = A, weights = B), which leaves a lane holding one
// output CHANNEL and 16 TOKENS per accumulator -- and that is, register for register, the operand layout of the state product itself
// (lane = d resp. v, 8 consecutive registers = 8 k positions; K and V carry the same token in the same position, so the contraction
// over tokens needs no transposition).  Two packs per quad and 16 more MFMAs per wave give the tile's [8 heads][32 x 32 + 32] partial
// state, stored where gim_linear_attention_kv would have put the partial of a 64-row chunk; gim_linear_attention_finalize sums the
// partials of a sequence.  128 KiB of k / v stores per tile, their read-back by la_kv and the la_kv launch itself are gone.
#include "gim_common.h"

namespace {

constexpr int C = 256;
constexpr int ROWS = 64;
constexpr int ROWB = C * 2;          // bytes of one row of the A / X tiles
constexpr int HROWB = 128 * 2;       // bytes of one row of the hidden-quarter tile
constexpr int OFF_A = 0, OFF_X = ROWS * ROWB, OFF_H = 2 * ROWS * ROWB;
constexpr int SMEM = OFF_H + ROWS * HROWB;   // 80 KiB
static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");
constexpr int UNITS_PER_WAVE = 4 + 4 * (4 + 2);
constexpr int UNIT_U4 = 8 * 64;      // uint4 per unit (8 fragments x 64 lanes x 16 B)

constexpr int MAXBLK = GIM_TOKEN_EMIT_MAX;

// elu(x) + 1 for a 16-bit output: exp(x) for x <= 0 on v_exp_f32 (2 ulp in fp32, far below the output rounding; the libm expf of the
// projection GEMM's epilogue costs ~20 instructions per value -- 4 k cycles per projection block and wave)
__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v + 1.f : __expf(v); }

struct Args {
    const float* kv;             // NULL: `msg` is the attention output.  Else: [nb][8][32*32 + 32] fp32 KV / Ksum state of the linear
                                 // attention (gim_linear_attention_kv) and `msg` holds the elu+1 QUERY rows: the apply step runs here
    const unsigned char* qmask;  // optional [R] padding mask of the query rows (attentions.py:36)
    int L;                       // rows per sequence (kv != NULL: must be a multiple of 64, rows of a tile share one sequence)
    float slen;                  // source length S (values were divided by it, attentions.py:41,45)
    const unsigned short* msg;   // [R][ldm] bf16 attention output (or queries, see kv); not read when qwts != NULL
    const uint4* qwts;           // NULL or the query projection of THIS call: 4 waves x 4 units x 8 fragments (pack_token_emit([Wq])); needs kv
    // projection only, with the positional encoding in front (loftr.py:74-75): rows = feat + pe[row % pe_hw], written to x32 and xb as well
    const unsigned short* feat;  // NULL: the rows are read from xb
    const float* pe;             // [pe_hw][256] fp32
    int ldf, pe_hw;
    unsigned short* xb;          // [R][ldxb] bf16 operand copy of x (in: x, out: x + msg)
    float* x32;                  // [R][ldx32] fp32 residual stream (in / out)
    const uint4* wts;            // 4 waves x 28 units x 8 fragments
    const float* ln;             // [g1 | b1 | g2 | b2] x 256
    int R, ldm, ldxb, ldx32;
    float eps;
    // projection blocks of the new x
    int nblk;
    const uint4* ewts;           // 4 waves x nblk x 4 units x 8 fragments (packing.py::pack_token_emit)
    unsigned short* eout[MAXBLK];
    int eld[MAXBLK], eact[MAXBLK], elo[MAXBLK], ehi[MAXBLK];
    // fused KV state: block b = K, block b + 1 = V of a pair when ekv[b] != NULL (neither is stored)
    float* ekv[MAXBLK];          // partial states of the consuming call: [sequence][8][nchunk][32 * 32 + 32] fp32
    int ekv_nchunk[MAXBLK];      // 64-row tiles per sequence of the consumer's source
    int ekv_tile0[MAXBLK];       // consumer-relative index of the tile that starts at row elo[b]
    float ekv_inv_s[MAXBLK];     // 1 / source length (attentions.py:41)
};

struct W8 { bf16x8_t f[8]; };

// -DGIM_TOKEN_TIMING (development builds only: tools/token_timing.py): shader-clock stamps at the phase boundaries, per wave, for the
// first TT_WG workgroups of a launch
#ifdef GIM_TOKEN_TIMING
constexpr int TT_WG = 4096, TT_N = 10;
__device__ unsigned long long g_tt[TT_WG][4][TT_N];
#define TT(i) do { if (blockIdx.x < TT_WG && L.lane == 0) g_tt[blockIdx.x][L.w][i] = __builtin_readcyclecounter(); } while (0)
#else
#define TT(i) do { } while (0)
#endif

struct Lane {
    int lane, l31, lh, w, sw;
    int a8[8];    // l31 * 512 + (((2ks + lh) ^ sw) << 4), ks = 0..7: first 128 channels of an A / X row (+ 256 B for the next 128)
    int h8[8];    // l31 * 256 + (((2ks + lh) ^ sw) << 4): hidden-quarter rows
    __device__ __forceinline__ void tables(int key) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            a8[ks] = l31 * ROWB + (((2 * ks + lh) ^ key) << 4);
            h8[ks] = l31 * HROWB + (((2 * ks + lh) ^ key) << 4);
        }
    }
};

__device__ __forceinline__ void wload(W8& w, const uint4* __restrict__ p, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) w.f[i] = __builtin_bit_cast(bf16x8_t, p[i * 64 + lane]);
    __builtin_amdgcn_sched_barrier(0);
}

// unit of a 64-column product (merge, mlp.2): 4 k16 steps x 2 column fragments; acc[nf][j] += W-frag x rows 32j..
// `tile` + tab[ks] (+ koff bytes) addresses the operand rows
template <bool FIRST>
__device__ __forceinline__ void mma_n64(const char* tile, const int (&tab)[8], int ks0, int jstride, const W8& w, f32x16_t (&acc)[2][2]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const char* ab = tile + tab[ks0 + k];
        const bf16x8_t v0 = *(const bf16x8_t*)(ab), v1 = *(const bf16x8_t*)(ab + jstride);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
            if (FIRST && k == 0) {
                f32x16_t z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                acc[nf][0] = mfma_h16_32x32x16(w.f[2 * k + nf], v0, z);
                acc[nf][1] = mfma_h16_32x32x16(w.f[2 * k + nf], v1, z);
            } else {
                acc[nf][0] = mfma_h16_32x32x16(w.f[2 * k + nf], v0, acc[nf][0]);
                acc[nf][1] = mfma_h16_32x32x16(w.f[2 * k + nf], v1, acc[nf][1]);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// the same unit with the operands swapped (rows = A, weights = B): acc[nf][j][4 rg + e] = channel 32 nf + l31 of row 32 j + 8 rg + 4 lh + e
template <bool FIRST>
__device__ __forceinline__ void mma_n64t(const char* tile, const int (&tab)[8], int ks0, int jstride, const W8& w, f32x16_t (&acc)[2][2]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const char* ab = tile + tab[ks0 + k];
        const bf16x8_t v0 = *(const bf16x8_t*)(ab), v1 = *(const bf16x8_t*)(ab + jstride);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
            if (FIRST && k == 0) {
                f32x16_t z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                acc[nf][0] = mfma_h16_32x32x16(v0, w.f[2 * k + nf], z);
                acc[nf][1] = mfma_h16_32x32x16(v1, w.f[2 * k + nf], z);
            } else {
                acc[nf][0] = mfma_h16_32x32x16(v0, w.f[2 * k + nf], acc[nf][0]);
                acc[nf][1] = mfma_h16_32x32x16(v1, w.f[2 * k + nf], acc[nf][1]);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// unit of a 32-column product (one hidden quarter): 8 k16 steps x 1 column fragment
template <bool FIRST>
__device__ __forceinline__ void mma_n32(const char* tile, const int (&tab)[8], const W8& w, f32x16_t (&acc)[2]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const char* ab = tile + tab[k];
        const bf16x8_t v0 = *(const bf16x8_t*)(ab), v1 = *(const bf16x8_t*)(ab + 32 * ROWB);
        if (FIRST && k == 0) {
            f32x16_t z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            acc[0] = mfma_h16_32x32x16(w.f[k], v0, z);
            acc[1] = mfma_h16_32x32x16(w.f[k], v1, z);
        } else {
            acc[0] = mfma_h16_32x32x16(w.f[k], v0, acc[0]);
            acc[1] = mfma_h16_32x32x16(w.f[k], v1, acc[1]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// LayerNorm over the 256 channels of every row, in accumulator layout (wave w holds channels 64w..64w+63 of all 64 rows:
// acc[nf][j][4rg + e] = channel 64w + 32nf + 8rg + 4lh + e of row 32j + l31).  One workgroup barrier inside.
__device__ __forceinline__ void layernorm_rows(f32x16_t (&acc)[2][2], const float* __restrict__ gamma, const float* __restrict__ beta,
                                               float2* stat, float eps, const Lane& L) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s += acc[nf][j][r]; q = fmaf(acc[nf][j][r], acc[nf][j][r], q); }
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (L.lh == 0) stat[(32 * j + L.l31) * 4 + L.w] = make_float2(s, q);
    }
    __syncthreads();
    float mean[2], rstd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4 p0 = *(const float4*)(stat + (32 * j + L.l31) * 4), p1 = *(const float4*)(stat + (32 * j + L.l31) * 4 + 2);
        const float s = (p0.x + p0.z) + (p1.x + p1.z), q = (p0.y + p0.w) + (p1.y + p1.w);
        mean[j] = s * (1.0f / C);
        rstd[j] = rsqrtf(fmaxf(q * (1.0f / C) - mean[j] * mean[j], 0.f) + eps);
    }
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 g = *(const float4*)(gamma + 64 * L.w + 32 * nf + 8 * rg + 4 * L.lh);
            const float4 b = *(const float4*)(beta + 64 * L.w + 32 * nf + 8 * rg + 4 * L.lh);
            const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[nf][j][rg * 4 + e] = fmaf(fmaf(acc[nf][j][rg * 4 + e], rstd[j], -mean[j] * rstd[j]), gg[e], bb[e]);
        }
}

// PROJ = true ("projection only", round 5): the tile is just the 16-bit rows `xb`; nothing of the layer runs, only the projection blocks behind
// it -- the k / v pair of the FIRST layer, handed over as partial KV states like every later one (replaces the initial [k | v] projection GEMM,
// its rows and the la_kv launches that read them).
// the projection-block descriptors are indexed at run time: read through the kernel-argument segment (scalar loads at a uniform index) --
// indexing the by-value struct itself made hipcc copy it to scratch (192 B per lane until round 5)
typedef const __attribute__((address_space(4))) Args* KArgs;

template <bool PROJ>
__global__ void __launch_bounds__(256, 2) token_mlp_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    Lane L;
    L.lane = threadIdx.x & 63;
    L.l31 = L.lane & 31;
    L.lh = L.lane >> 5;
    L.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    L.sw = L.l31 & 15;
    L.tables(L.sw);
    const int r0 = blockIdx.x * ROWS;
    char* A = smem + OFF_A;
    char* X = smem + OFF_X;
    char* H = smem + OFF_H;
    float2* stat = (float2*)H;   // [64 rows][4 waves] partial (sum, sum of squares): H is idle whenever a LayerNorm runs
    const uint4* wp = a.wts + (size_t)L.w * UNITS_PER_WAVE * UNIT_U4;
    const uint4* ewp = a.ewts + (size_t)L.w * a.nblk * 4 * UNIT_U4;
    unsigned emask = 0u;         // projection blocks whose row range holds this tile (block-uniform)
    for (int b = 0; b < a.nblk; ++b) emask |= (r0 >= ka->elo[b] && r0 < ka->ehi[b]) ? 1u << b : 0u;
    // Weight units are requested TWO ahead of their use into alternating register sets (wa: even units, wb: odd units): with one
    // set, unit u + 1 could only be requested once unit u's MFMAs had issued, and every unit waited out an L2 round trip.
    // Unit sequence: the 28 base units, then 4 per active projection block.
    int pf = PROJ ? UNITS_PER_WAVE : 0;              // next base unit to request (projection only: none)
    int pb = emask ? __ffs(emask) - 1 : -1, pq = 0;  // next projection unit to request: block, unit
    const bool qloc = !PROJ && a.qwts != nullptr;    // the queries are projected here (4 units in front of the base stream)
    const uint4* qwp = a.qwts + (size_t)L.w * 4 * UNIT_U4;
    int pl = qloc ? 0 : 4;                           // next query-projection unit to request
    auto fetch = [&](W8& w) __attribute__((always_inline)) {
        if (pl < 4) { wload(w, qwp + (size_t)pl * UNIT_U4, L.lane); ++pl; return; }
        if (pf < UNITS_PER_WAVE) { wload(w, wp + (size_t)pf * UNIT_U4, L.lane); ++pf; return; }
        if (pb < 0) return;
        wload(w, ewp + (size_t)(4 * pb + pq) * UNIT_U4, L.lane);
        if (++pq == 4) {
            pq = 0;
            const unsigned rest = emask & ~((2u << pb) - 1u);
            pb = rest ? __ffs(rest) - 1 : -1;
        }
    };
    W8 wa, wb;
    TT(0);
    fetch(wa);   // merge (or query projection), units 0 and 1: in flight during the tile loads
    fetch(wb);
    // ---- A <- attention output rows, X <- operand copy of x (512 B rows: 32 lanes x 16 B, 8 rows per pass) ----------
    if (PROJ && a.feat) {
        // x = feat + pe (gim_posenc_add's arithmetic: 16-bit feature -> fp32, + fp32 table row, one rounding for the operand copy); every load
        // of the lane is in flight before the first store (the stores alias the loads for all the compiler knows)
        const int t = threadIdx.x, slot = t & 31;
        uint4 vf[ROWS / 8];
        float4 p0[ROWS / 8], p1[ROWS / 8];
#pragma unroll
        for (int pass = 0; pass < ROWS / 8; ++pass) {
            const int m = r0 + pass * 8 + (t >> 5);
            vf[pass] = make_uint4(0u, 0u, 0u, 0u);
            p0[pass] = p1[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < a.R) {
                vf[pass] = *(const uint4*)(a.feat + (size_t)m * a.ldf + slot * 8);
                const float* pp = a.pe + (size_t)(m % a.pe_hw) * C + slot * 8;
                p0[pass] = *(const float4*)pp;
                p1[pass] = *(const float4*)(pp + 4);
            }
        }
#pragma unroll
        for (int pass = 0; pass < ROWS / 8; ++pass) {
            const int row = pass * 8 + (t >> 5), m = r0 + row;
            const float4 lo = make_float4(h16_lo(vf[pass].x) + p0[pass].x, h16_hi(vf[pass].x) + p0[pass].y, h16_lo(vf[pass].y) + p0[pass].z, h16_hi(vf[pass].y) + p0[pass].w);
            const float4 hi = make_float4(h16_lo(vf[pass].z) + p1[pass].x, h16_hi(vf[pass].z) + p1[pass].y, h16_lo(vf[pass].w) + p1[pass].z, h16_hi(vf[pass].w) + p1[pass].w);
            const uint4 o = make_uint4(cvt_pk_h16(lo.x, lo.y), cvt_pk_h16(lo.z, lo.w), cvt_pk_h16(hi.x, hi.y), cvt_pk_h16(hi.z, hi.w));
            if (m < a.R) {
                *(float4*)(a.x32 + (size_t)m * a.ldx32 + slot * 8) = lo;
                *(float4*)(a.x32 + (size_t)m * a.ldx32 + slot * 8 + 4) = hi;
                *(uint4*)(a.xb + (size_t)m * a.ldxb + slot * 8) = o;
            }
            *(uint4*)(A + row * ROWB + ((slot ^ (row & 15)) << 4)) = (m < a.R) ? o : make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
        // all 16 row loads of a lane are in flight before the first LDS write: ONE HBM round trip (measured with the phase stamps
        // of -DGIM_TOKEN_TIMING: 14.3 k cycles for this phase when the loop was rolled 4 passes at a time)
        const int t = threadIdx.x, slot = t & 31;
        uint4 va[ROWS / 8], vx[ROWS / 8];
#pragma unroll
        for (int pass = 0; pass < ROWS / 8; ++pass) {
            const int m = r0 + pass * 8 + (t >> 5);
            va[pass] = vx[pass] = make_uint4(0u, 0u, 0u, 0u);
            if (m < a.R) {
                if (!PROJ && !qloc) va[pass] = *(const uint4*)(a.msg + (size_t)m * a.ldm + slot * 8);
                vx[pass] = *(const uint4*)(a.xb + (size_t)m * a.ldxb + slot * 8);
            }
        }
#pragma unroll
        for (int pass = 0; pass < ROWS / 8; ++pass) {
            const int row = pass * 8 + (t >> 5);
            const int off = row * ROWB + ((slot ^ (row & 15)) << 4);   // XOR on the low 4 slot bits: conflict-free b128 rows
            if (PROJ) { *(uint4*)(A + off) = vx[pass]; continue; }   // projection only: the rows ARE the operand tile of the blocks
            if (!qloc) *(uint4*)(A + off) = va[pass];
            *(uint4*)(X + off) = vx[pass];
        }
    }
    __syncthreads();
    f32x16_t acc[2][2];
    if constexpr (PROJ) {
        if (!emask) return;
    } else {   // ======== the layer itself (not re-indented: everything down to the operand tile of the new x) ========
    if (qloc) {
        // ---- queries of this call: A <- elu(x Wq^T) + 1 of the tile's rows, this wave's 64 columns (same units, same order as a projection block) ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char* tile = X + (q >> 1) * 256;
            W8& w = (q & 1) ? wb : wa;
            if (q == 0) mma_n64<true>(tile, L.a8, 0, 32 * ROWB, w, acc);
            else mma_n64<false>(tile, L.a8, (q & 1) * 4, 32 * ROWB, w, acc);
            fetch(w);
        }
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *(uint2*)(A + (32 * j + L.l31) * ROWB + (((8 * L.w + 4 * nf + rg) ^ L.sw) << 4) + L.lh * 8) =
                        make_uint2(cvt_pk_h16(elu1(acc[nf][j][rg * 4]), elu1(acc[nf][j][rg * 4 + 1])), cvt_pk_h16(elu1(acc[nf][j][rg * 4 + 2]), elu1(acc[nf][j][rg * 4 + 3])));
        __syncthreads();
    }
    TT(1);
    if (a.kv) {
        // ---- linear-attention apply (attentions.py:44-45), in place on A: this wave owns heads 2w, 2w+1 = channels 64w..64w+63, and
        // nothing else reads or writes those columns.  msg[row, 32h + v] = S * Z[row,h] * sum_d Q[row, 32h + d] KV[h][d][v]  on the
        // bf16 MFMA (KV rounded to bf16), Z = 1 / (Q . Ksum + eps) in fp32.
        const float* kvb = a.kv + (size_t)(r0 / a.L) * 8 * (32 * 32 + 32);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * L.w + hh;
            const float* KV = kvb + h * (32 * 32 + 32);
            bf16x8_t kf[2];
            float ksum[2][8];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float t[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    t[p] = KV[(16 * ks + 8 * L.lh + p) * 32 + L.l31];          // A operand: m = v (l31), k = d
                    ksum[ks][p] = KV[32 * 32 + 16 * ks + 8 * L.lh + p];
                }
                kf[ks] = __builtin_bit_cast(bf16x8_t, make_uint4(cvt_pk_h16(t[0], t[1]), cvt_pk_h16(t[2], t[3]), cvt_pk_h16(t[4], t[5]), cvt_pk_h16(t[6], t[7])));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16_t o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
                float z = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int k16 = 2 * h + ks;   // k16 step of the 256-channel row
                    const uint4 q = *(const uint4*)(A + (k16 >> 3) * 256 + (L.l31 * ROWB + (((2 * (k16 & 7) + L.lh) ^ L.sw) << 4)) + j * 32 * ROWB);   // = L.a8[k16 & 7], computed: k16 depends on the wave, and a run-time index put the whole table into scratch
                    const unsigned qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        z = fmaf(h16_lo(qq[p]), ksum[ks][2 * p], z);
                        z = fmaf(h16_hi(qq[p]), ksum[ks][2 * p + 1], z);
                    }
                    o = mfma_h16_32x32x16(kf[ks], __builtin_bit_cast(bf16x8_t, q), o);
                }
                z += __shfl_xor(z, 32, 64);
                const int m = r0 + 32 * j + L.l31;
                const bool valid = !a.qmask || (m < a.R && a.qmask[m]);
                const float sc = valid ? a.slen * __builtin_amdgcn_rcpf(z + 1e-6f) : 0.f;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *(uint2*)(A + (32 * j + L.l31) * ROWB + (((4 * h + rg) ^ L.sw) << 4) + L.lh * 8) =
                        make_uint2(cvt_pk_h16(o[rg * 4] * sc, o[rg * 4 + 1] * sc), cvt_pk_h16(o[rg * 4 + 2] * sc, o[rg * 4 + 3] * sc));
            }
        }
        __syncthreads();
    }
    TT(2);
    // ---- merge: [64 x 256] x W_merge^T, this wave's 64 output channels (transformer.py:52) ----------------------------
#pragma unroll
    for (int q = 0; q < 4; ++q) {          // 4 units x 4 k16 steps = K 256
        const char* tile = A + (q >> 1) * 256;   // k16 steps 0..7 -> first 256 B of the row, 8..15 -> second
        W8& w = (q & 1) ? wb : wa;
        if (q == 0) mma_n64<true>(tile, L.a8, 0, 32 * ROWB, w, acc);
        else mma_n64<false>(tile, L.a8, (q & 1) * 4, 32 * ROWB, w, acc);
        fetch(w);
    }
    TT(3);
    // ---- norm1 -> A (the attention output is consumed: barrier inside the LayerNorm) ---------------------------------
    layernorm_rows(acc, a.ln, a.ln + C, stat, a.eps, L);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                *(uint2*)(A + (32 * j + L.l31) * ROWB + (((8 * L.w + 4 * nf + rg) ^ L.sw) << 4) + L.lh * 8) =
                    make_uint2(cvt_pk_h16(acc[nf][j][rg * 4], acc[nf][j][rg * 4 + 1]), cvt_pk_h16(acc[nf][j][rg * 4 + 2], acc[nf][j][rg * 4 + 3]));
    __syncthreads();
    TT(4);
    // ---- mlp: four hidden quarters (transformer.py:55-56) --------------------------------------------------------------
    f32x16_t out[2][2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[nf][j][r] = 0.f;
#pragma unroll 1
    for (int hq = 0; hq < 4; ++hq) {
        f32x16_t hid[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // K = 512: [x | msg], 4 units of 8 k16 steps
            const char* tile = (q < 2 ? X : A) + (q & 1) * 256;
            W8& w = (q & 1) ? wb : wa;
            if (q == 0) mma_n32<true>(tile, L.a8, w, hid);
            else mma_n32<false>(tile, L.a8, w, hid);
            fetch(w);
        }
        if (hq > 0) __syncthreads();       // every wave is done reading the previous quarter
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                *(uint2*)(H + (32 * j + L.l31) * HROWB + (((4 * L.w + rg) ^ L.sw) << 4) + L.lh * 8) =
                    make_uint2(cvt_pk_h16(fmaxf(hid[j][rg * 4], 0.f), fmaxf(hid[j][rg * 4 + 1], 0.f)),
                               cvt_pk_h16(fmaxf(hid[j][rg * 4 + 2], 0.f), fmaxf(hid[j][rg * 4 + 3], 0.f)));
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {      // out += H x W2[:, quarter]: K = 128, 2 units of 4 k16 steps
            W8& w = (q & 1) ? wb : wa;
            mma_n64<false>(H, L.h8, q * 4, 32 * HROWB, w, out);
            fetch(w);                      // (the last quarter requests the first two projection units: they land under norm2)
        }
    }
    TT(5);
    __syncthreads();   // H is consumed: its space serves the LayerNorm partial sums
    // ---- norm2 (transformer.py:57); residual add + stores through a wave-private LDS transposition --------------------
    layernorm_rows(out, a.ln + 2 * C, a.ln + 3 * C, stat, a.eps, L);   // barrier inside: A and X are dead from here on
    TT(6);
    float* tr = (float*)(smem + L.w * 16384);   // this wave's [64 rows][64 channels] fp32, 16-byte slots XOR-swizzled by row
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = 32 * j + L.l31;
                *(float4*)((char*)tr + row * 256 + (((8 * nf + 2 * rg + L.lh) ^ (row & 15)) << 4)) =
                    make_float4(out[nf][j][rg * 4], out[nf][j][rg * 4 + 1], out[nf][j][rg * 4 + 2], out[nf][j][rg * 4 + 3]);
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave reads below: LDS executes a wave's accesses in order
    uint2 nx[16];   // this lane's share of the new x (16-bit): row it * 4 + rsub, channels 64w + 4 slot .. + 3
    {
        const int slot = L.lane & 15, rsub = L.lane >> 4;
        // every residual row of the lane is requested before the first store: the stores below alias the loads for all the compiler
        // knows, and a rolled read-modify-write waited out 16 HBM round trips one after the other (19.3 k cycles per tile, -DGIM_TOKEN_TIMING)
        float4 xin[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = r0 + it * 4 + rsub;
            xin[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < a.R) xin[it] = *(const float4*)(a.x32 + (size_t)m * a.ldx32 + 64 * L.w + 4 * slot);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + rsub, m = r0 + row;
            const float4 v = *(const float4*)((char*)tr + row * 256 + ((slot ^ (row & 15)) << 4));
            float4 x = xin[it];
            x.x += v.x; x.y += v.y; x.z += v.z; x.w += v.w;   // x + message
            nx[it] = make_uint2(cvt_pk_h16(x.x, x.y), cvt_pk_h16(x.z, x.w));
            if (m < a.R) {
                *(float4*)(a.x32 + (size_t)m * a.ldx32 + 64 * L.w + 4 * slot) = x;
                *(uint2*)(a.xb + (size_t)m * a.ldxb + 64 * L.w + 4 * slot) = nx[it];
            }
        }
    }
    TT(7);
    if (!emask) { TT(8); TT(9); return; }
    // ---- projection blocks of the new x ---------------------------------------------------------------------------------------
    __syncthreads();   // every wave is done with its transposition tile: the operand tile of the new x overwrites waves 0 / 1's
    {
        const int slot = L.lane & 15, rsub = L.lane >> 4;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + rsub;
            *(uint2*)(A + row * ROWB + (((8 * L.w + (slot >> 1)) ^ (row & 15)) << 4) + (slot & 1) * 8) = nx[it];
        }
    }
    __syncthreads();
    }   // ======== !PROJ ========
    TT(8);
    char* t2 = X + L.w * 4096;   // wave-private [32 rows][64 channels] 16-bit: 16-byte slot s of row r at slot s ^ ((r >> 1) & 7), the
                                 // slot's 8-byte halves swapped for rows 16..31 (32 accumulator rows x 8 B hit 32 distinct bank pairs)
    int b = __ffs(emask) - 1;
#pragma unroll 1
    while (b >= 0) {
        const unsigned rest = emask & ~((2u << b) - 1u);
        const int nb = rest ? __ffs(rest) - 1 : -1;
        if (ka->ekv[b]) {
            // ---- (k, v) pair -> partial KV state of this tile (block-uniform branch; the V block is the next active one: same row gate) ----
            // K^T as MFMA A operand, [head of the wave][row half][k16 step] x 8 tokens per lane: parked in LDS while the V block is projected
            // (32 registers the emit loop does not have; X's upper half and H are idle here: 8 KiB per wave, lane-linear)
            char* kpark = smem + OFF_X + 16384 + L.w * 8192 + L.lane * 16;
            float ksum[2] = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const char* tile = A + (q >> 1) * 256;
                W8& w = (q & 1) ? wb : wa;
                if (q == 0) mma_n64t<true>(tile, L.a8, 0, 32 * ROWB, w, acc);
                else mma_n64t<false>(tile, L.a8, (q & 1) * 4, 32 * ROWB, w, acc);
                fetch(w);
            }
            const bool kelu = ka->eact[b] == GIM_ACT_ELU1;
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        float v[8];
#pragma unroll
                        for (int p = 0; p < 8; ++p) v[p] = kelu ? elu1(acc[nf][j][8 * ks + p]) : acc[nf][j][8 * ks + p];
                        const unsigned kq[4] = {cvt_pk_h16(v[0], v[1]), cvt_pk_h16(v[2], v[3]), cvt_pk_h16(v[4], v[5]), cvt_pk_h16(v[6], v[7])};
                        *(uint4*)(kpark + ((nf * 2 + j) * 2 + ks) * 1024) = make_uint4(kq[0], kq[1], kq[2], kq[3]);
                        // Ksum[d] = sum over the tokens of the ROUNDED k values (what la_kv reads back), 16 of them in this lane
#pragma unroll
                        for (int p = 0; p < 4; ++p) ksum[nf] += h16_lo(kq[p]) + h16_hi(kq[p]);
                    }
            const int bv = nb;   // the pair's V block
            const unsigned restv = emask & ~((2u << bv) - 1u);
            const int nbv = restv ? __ffs(restv) - 1 : -1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const char* tile = A + (q >> 1) * 256;
                W8& w = (q & 1) ? wb : wa;
                if (q == 0) mma_n64t<true>(tile, L.a8, 0, 32 * ROWB, w, acc);
                else mma_n64t<false>(tile, L.a8, (q & 1) * 4, 32 * ROWB, w, acc);
                fetch(w);
            }
            const int tix = ka->ekv_tile0[b] + (r0 - ka->elo[b]) / ROWS, nch = ka->ekv_nchunk[b];
            const int seq = tix / nch, chunk = tix - seq * nch;
            const float inv_s = ka->ekv_inv_s[b];
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                f32x16_t kv;
#pragma unroll
                for (int r = 0; r < 16; ++r) kv[r] = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const f32x16_t& va = acc[nf][j];
                        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(cvt_pk_h16(va[8 * ks], va[8 * ks + 1]), cvt_pk_h16(va[8 * ks + 2], va[8 * ks + 3]),
                                                                                    cvt_pk_h16(va[8 * ks + 4], va[8 * ks + 5]), cvt_pk_h16(va[8 * ks + 6], va[8 * ks + 7])));
                        const bf16x8_t kk = *(const bf16x8_t*)(kpark + ((nf * 2 + j) * 2 + ks) * 1024);   // (same wave wrote it: LDS executes a wave's accesses in order)
                        kv = mfma_h16_32x32x16(kk, vf, kv);        // D[d][v] += sum_tokens K[token][d] V[token][v]
                    }
                // register r of lane (l31, lh) is element [d = 8 (r >> 2) + 4 lh + (r & 3)][v = l31] (the layout la_kv_h16_kernel stores)
                float* out = (float*)ka->ekv[b] + ((size_t)(seq * 8 + 2 * L.w + nf) * nch + chunk) * (32 * 32 + 32);
#pragma unroll
                for (int r = 0; r < 16; ++r) out[((r >> 2) * 8 + L.lh * 4 + (r & 3)) * 32 + L.l31] = kv[r] * inv_s;
                const float ksd = ksum[nf] + __shfl_xor(ksum[nf], 32, 64);   // lane (d = l31, lh) held the tokens 8 rg + 4 lh + e of both row halves
                if (L.lh == 0) out[32 * 32 + L.l31] = ksd;
            }
            b = nbv;
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char* tile = A + (q >> 1) * 256;
            W8& w = (q & 1) ? wb : wa;
            if (q == 0) mma_n64<true>(tile, L.a8, 0, 32 * ROWB, w, acc);
            else mma_n64<false>(tile, L.a8, (q & 1) * 4, 32 * ROWB, w, acc);
            fetch(w);
        }
        unsigned short* ob = (unsigned short*)ka->eout[b];
        const int ld = ka->eld[b];
        const bool elu = ka->eact[b] == GIM_ACT_ELU1;
        // the lane index behind an empty asm: everything this branch derives from it (row addresses, bounds masks) is then computed HERE -- hipcc
        // otherwise hoists ~24 registers of it in front of the emit loop and spills them around it, although the default 16-bit path (every block
        // a fused (k, v) pair) never takes this branch (round 6: the kernel's last scratch bytes)
        int ln = L.lane;
        asm volatile("" : "+v"(ln));
        const int l31p = ln & 31, lhp = ln >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float v0 = acc[nf][j][rg * 4], v1 = acc[nf][j][rg * 4 + 1], v2 = acc[nf][j][rg * 4 + 2], v3 = acc[nf][j][rg * 4 + 3];
                    if (elu) {
                        v0 = elu1(v0); v1 = elu1(v1); v2 = elu1(v2); v3 = elu1(v3);
                    }
                    *(uint2*)(t2 + l31p * 128 + (((4 * nf + rg) ^ ((l31p >> 1) & 7)) << 4) + ((lhp ^ (l31p >> 4)) << 3)) =
                        make_uint2(cvt_pk_h16(v0, v1), cvt_pk_h16(v2, v3));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave reads below (LDS executes a wave's accesses in order)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (ln >> 3), slot = ln & 7, m = r0 + 32 * j + row;
                uint4 v = *(const uint4*)(t2 + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
                if (row >> 4) v = make_uint4(v.z, v.w, v.x, v.y);
                if (m < a.R) *(uint4*)(ob + (size_t)m * ld + 64 * L.w + 8 * slot) = v;
            }
        }
        b = nb;
    }
    TT(9);
}

}  // namespace

extern "C" int64_t GIM_FN(gim_token_mlp_weight_bytes)(void) { return (int64_t)4 * UNITS_PER_WAVE * UNIT_U4 * 16; }

static int token_mlp_launch(const void* msg, void* xb, float* x32, const void* weights, const float* ln_params, const float* kv,
                            const uint8_t* q_mask, int R, int C_, int L, int S, int ldm, int ldxb, int ldx32, float ln_eps,
                            const gim_token_emit* em, gim_stream_t stream) {
    if (R == 0) return GIM_OK;
    const bool proj = em && em->project_only;
    GIM_REQUIRE((msg || proj || (em && em->q_weights)) && xb && (proj || (x32 && weights && ln_params)), "token_mlp: NULL pointer");
    GIM_REQUIRE(C_ == C, "token_mlp: built for d_model 256 (got %d)", C_);
    GIM_REQUIRE(R > 0 && (proj || (em && em->q_weights) || (ldm >= C && ldm % 8 == 0)) && ldxb >= C && ldxb % 8 == 0 && (proj || (ldx32 >= C && ldx32 % 4 == 0)), "token_mlp: bad strides");
    GIM_REQUIRE(!kv || (L > 0 && L % ROWS == 0 && R % L == 0 && S > 0), "token_mlp: fused attention apply needs L %% 64 == 0 and R %% L == 0 (L=%d R=%d)", L, R);
    static GimPerDevice attr;
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)token_mlp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)token_mlp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) { gim_set_error("token_mlp: hipFuncSetAttribute(%d B LDS): %s", SMEM, hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
        attr.done();
    }
    Args a;
    a.kv = kv; a.qmask = q_mask; a.L = L > 0 ? L : R; a.slen = (float)S;
    a.msg = (const unsigned short*)msg; a.xb = (unsigned short*)xb; a.x32 = x32; a.wts = (const uint4*)weights; a.ln = ln_params;
    a.R = R; a.ldm = ldm; a.ldxb = ldxb; a.ldx32 = ldx32; a.eps = ln_eps;
    a.nblk = 0; a.ewts = nullptr; a.qwts = nullptr;
    a.feat = nullptr; a.pe = nullptr; a.ldf = 0; a.pe_hw = 1;
    if (proj && em->pe) {
        GIM_REQUIRE(em->pe_feat && em->pe_hw > 0 && em->pe_ld >= C && em->pe_ld % 8 == 0 && x32 && ldx32 >= C && ldx32 % 4 == 0 && ((uintptr_t)em->pe_feat & 15) == 0 && ((uintptr_t)em->pe & 15) == 0,
                    "token_mlp: positional encoding in front of the projection: feature rows %p (stride %d), table %p x %d rows, x32 %p", em->pe_feat, em->pe_ld, em->pe, em->pe_hw, (void*)x32);
        a.feat = (const unsigned short*)em->pe_feat; a.pe = em->pe; a.ldf = em->pe_ld; a.pe_hw = em->pe_hw;
    }
    if (em && em->q_weights) {
        GIM_REQUIRE(kv, "token_mlp: q_weights (local query projection) needs the fused attention apply (kv)");
        a.qwts = (const uint4*)em->q_weights;
    }
    for (int b = 0; b < MAXBLK; ++b) {
        a.eout[b] = nullptr; a.eld[b] = 0; a.eact[b] = GIM_ACT_NONE; a.elo[b] = a.ehi[b] = 0;
        a.ekv[b] = nullptr; a.ekv_nchunk[b] = 1; a.ekv_tile0[b] = 0; a.ekv_inv_s[b] = 0.f;
    }
    if (em && em->nblk > 0) {
        GIM_REQUIRE(em->nblk <= MAXBLK && em->weights, "token_mlp: %d projection blocks (at most %d), weights %p", em->nblk, MAXBLK, em->weights);
        a.nblk = em->nblk; a.ewts = (const uint4*)em->weights;
        for (int b = 0; b < em->nblk; ++b) {
            const bool pair_k = em->kv_part[b] != nullptr, pair_v = b > 0 && em->kv_part[b - 1] != nullptr;
            if (pair_k) {
                GIM_REQUIRE(!pair_v && b + 1 < em->nblk && !em->kv_part[b + 1] && em->row_lo[b + 1] == em->row_lo[b] && em->row_hi[b + 1] == em->row_hi[b],
                            "token_mlp: fused KV state: block %d (K) must be followed by its V block with the same row range", b);
                GIM_REQUIRE(em->kv_nchunk[b] > 0 && em->kv_tile0[b] >= 0 && em->kv_len[b] > 0.f && em->row_hi[b] <= R && em->row_hi[b] % ROWS == 0 && ((uintptr_t)em->kv_part[b] & 15) == 0,
                            "token_mlp: fused KV state of block %d: %d tiles per sequence, first tile %d, source length %g, rows up to %d of %d",
                            b, em->kv_nchunk[b], em->kv_tile0[b], (double)em->kv_len[b], em->row_hi[b], R);
                a.ekv[b] = em->kv_part[b]; a.ekv_nchunk[b] = em->kv_nchunk[b]; a.ekv_tile0[b] = em->kv_tile0[b]; a.ekv_inv_s[b] = 1.0f / em->kv_len[b];
            }
            GIM_REQUIRE(pair_k || pair_v || (em->out[b] && ((uintptr_t)em->out[b] & 15) == 0), "token_mlp: projection block %d: output %p", b, em->out[b]);
            GIM_REQUIRE(pair_k || pair_v || (em->ld[b] >= C && em->ld[b] % 8 == 0),
                        "token_mlp: projection block %d: output %p, row stride %d", b, em->out[b], em->ld[b]);
            GIM_REQUIRE(em->act[b] == GIM_ACT_NONE || em->act[b] == GIM_ACT_ELU1, "token_mlp: projection block %d: activation %d", b, em->act[b]);
            GIM_REQUIRE(em->row_lo[b] % ROWS == 0 && em->row_lo[b] <= em->row_hi[b], "token_mlp: projection block %d: rows [%d, %d) (the start must be a multiple of %d)",
                        b, em->row_lo[b], em->row_hi[b], ROWS);
            a.eout[b] = (unsigned short*)em->out[b]; a.eld[b] = em->ld[b]; a.eact[b] = em->act[b]; a.elo[b] = em->row_lo[b]; a.ehi[b] = em->row_hi[b];
        }
    }
    if (em && em->project_only) {
        GIM_REQUIRE(a.nblk > 0 && !a.qwts && !kv, "token_mlp: project_only runs projection blocks only (nblk > 0, no q_weights, no kv)");
        hipLaunchKernelGGL(token_mlp_kernel<true>, dim3((unsigned)((R + ROWS - 1) / ROWS)), dim3(256), SMEM, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(token_mlp_kernel<false>, dim3((unsigned)((R + ROWS - 1) / ROWS)), dim3(256), SMEM, (hipStream_t)stream, a);
    }
    return gim_check_launch("token_mlp");
}

#ifdef GIM_TOKEN_TIMING
extern "C" int GIM_FN(gim_token_mlp_timing)(unsigned long long* host, int n_wg) {   // copies the stamps of the last launch
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tt), (size_t)(n_wg < TT_WG ? n_wg : TT_WG) * 4 * TT_N * 8) == hipSuccess ? GIM_OK : GIM_ERR_LAUNCH;
}
#endif

extern "C" int GIM_FN(gim_token_mlp)(const void* msg, void* xb, float* x32, const void* weights, const float* ln_params, const float* kv,
                             const uint8_t* q_mask, int R, int C_, int L, int S, int ldm, int ldxb, int ldx32, float ln_eps,
                             gim_stream_t stream) {
    return token_mlp_launch(msg, xb, x32, weights, ln_params, kv, q_mask, R, C_, L, S, ldm, ldxb, ldx32, ln_eps, nullptr, stream);
}

extern "C" int GIM_FN(gim_token_mlp_emit)(const void* msg, void* xb, float* x32, const void* weights, const float* ln_params, const float* kv,
                                  const uint8_t* q_mask, int R, int C_, int L, int S, int ldm, int ldxb, int ldx32, float ln_eps,
                                  const gim_token_emit* emit, gim_stream_t stream) {
    return token_mlp_launch(msg, xb, x32, weights, ln_params, kv, q_mask, R, C_, L, S, ldm, ldxb, ldx32, ln_eps, emit, stream);
}

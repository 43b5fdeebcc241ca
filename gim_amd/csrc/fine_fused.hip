// Fused fine level of gim_loftr (bf16 operand mode) for gfx950: ONE kernel per forward instead of
// gather -> 21 GEMM / attention / LayerNorm launches -> fine_match, and no activation ever leaves the CU.
//
// Replaces, per coarse match (reference file:line):
//   networks/loftr/submodules/fine_preprocess.py:40-47   the two 5x5x128 windows (only the M matched ones)
//   networks/loftr/submodules/transformer.py:35-58,80-101 LocalFeatureTransformer(['self','cross'], d=128, 8 heads)
//   networks/loftr/submodules/attentions.py:20-47        LinearAttention on 25-token sequences
//   networks/loftr/utils/fine_matching.py:43-74          centre-row correlation, softmax, DSNT expectation, std
//
// One 256-thread workgroup owns G = 2 matches and two workgroups are resident per CU (2 waves per SIMD): one workgroup's
// epilogue VALU / LDS latencies / barrier waits overlap the other's MFMAs, and since hardware barriers are per workgroup the two
// never wait for each other (measured: 8 waves x 4 matches in ONE workgroup 1.02 ms, 2 x (4 waves x 2 matches) 0.89 ms at 12 000
// matches).  A match is one 32-row MFMA fragment (25 window tokens + 7 zero rows), so each token stream is a [64 rows x 128
// channels] bf16 operand tile that lives in LDS for the whole kernel:
//
//   LDS  X0, X1   operand copies of the two token streams                                              2 x 16 KiB
//        T1, T2   temporaries (K^T / Q / LN1(msg) / hidden 128..255   and   V^T / msg / hidden 0..127)  2 x 16 KiB
//        scratch  per-match K sums, LayerNorm partial sums                                              3 KiB
//        LO       low halves (x - bf16(x), as bf16) of the token stream that is not being updated       12.5 KiB
//   VGPR          the fp32 master copy of the token stream being updated (residual adds stay fp32, as in the unfused
//                 path); the other stream is parked as bf16 hi (its operand tile) + bf16 lo (LO): 2^-17 relative
//
// Every matrix product is [64 rows] x [128 out] x K (128 or 256) on v_mfma_f32_32x32x16_bf16; wave wn owns output columns
// 32wn..32wn+31 for all rows, so a workgroup fetches each weight element once -- straight from L2 into registers in a pre-packed
// fragment order (16 B per lane, 1 KiB per wave instruction), never through LDS, requested one 8-fragment unit ahead of its use
// (behind the MFMAs of the previous unit, so the fetch runs under that unit's epilogue: -8 % time).
// The 256-wide MLP hidden layer lives in the two temporaries (its second half replaces LN1(msg) once that is consumed).
//
// Linear attention on the matrix cores (two waves per match, two 32-channel groups = 4 heads each):
//   K and V are produced TRANSPOSED ([channel][token], operands swapped in the MFMA) with the 7 padding tokens zeroed;
//   KV = K^T V is one 32x32 fragment per channel group (contraction over the 32 token slots), masked to its two 16x16
//   head blocks; msg = Q KV consumes that accumulator directly as the next MFMA's operand -- the contraction order of
//   the second product is permuted to the accumulator layout (Q is read as two 8-byte pieces per lane), so the fragment
//   never moves between lanes.  Z = 1 / (Q . sum_s K + eps) on the VALU.  (v / S ... * S of attentions.py:41,45
//   cancels and is dropped.)
//
// Numerics: bf16 operands, fp32 accumulation, fp32 residual stream, fp32 LayerNorm / softmax statistics -- the same
// rounding points as the unfused bf16 path (q, k, v, msg, LN outputs and hidden activations are bf16 there too).
#include "gim_common.h"

namespace {

constexpr int C = 128;             // fine d_model
constexpr int ROWB = C * 2;        // bytes of one activation row (bf16)
constexpr int WW = 25, G = 2, NH = 8;   // G matches per workgroup
constexpr int ROWS = 32 * G;        // token rows of one stream (a match = one 32-row MFMA fragment)
constexpr int BUF = ROWS * ROWB;    // 16 KiB: 64 rows x 128 channels
constexpr int TROWB = ROWS * 2;     // bytes of one row of a TRANSPOSED tile [128 channels][64 token slots] (8 x 16 B, XOR key (ch >> 1) & 7:
                                    // two 128-byte rows share a 256-byte bank row, so the key skips the channel's low bit)
static_assert(G == 2, "tile addressing below is written for 2 matches (64 rows) per workgroup");
constexpr int OFF_X0 = 0, OFF_X1 = BUF, OFF_T1 = 2 * BUF, OFF_T2 = 3 * BUF, OFF_SCR = 4 * BUF;
constexpr int SCR_KSUM = 0, SCR_STAT = G * C * 4, SCR_BYTES = SCR_STAT + ROWS * 4 * 8;
constexpr int OFF_LO = 4 * BUF + SCR_BYTES;   // parked low halves of one fp32 token stream: [4 x 25 valid rows][128] bf16
constexpr int SMEM = OFF_LO + G * 25 * ROWB;
static_assert(2 * SMEM <= 160 * 1024, "LDS budget: two workgroups per CU");
constexpr int FIN_LD = 132;        // floats per row of the final fp32 image-1 tokens (bank-conflict-free float4 rows)
static_assert(G * WW * FIN_LD * 4 <= 2 * BUF, "final image-1 tokens must fit in T1|T2");

// weight stream of one layer, in 16-byte units (uint4): [Wq | Wk | Wv | Wm | W0a | W0b | W2a | W2b]
constexpr int U128 = 128 * 128 * 2 / 16, U256 = 128 * 256 * 2 / 16;
constexpr int W_Q = 0, W_K = U128, W_V = 2 * U128, W_M = 3 * U128, W_0A = 4 * U128, W_0B = W_0A + U256,
              W_2A = W_0B + U256, W_2B = W_2A + U128, W_LAYER = W_2B + U128;

struct FineArgs {
    const unsigned short* f0;   // fine maps, NHWC bf16
    const unsigned short* f1;
    const int64_t *b_ids, *i_ids, *j_ids;
    const float* mkpts1_c;
    const float* scale1;
    const uint4* wts;           // 2 layers x W_LAYER
    const float* ln;            // 2 layers x [g1 | b1 | g2 | b2] x 128
    float* expec_f;
    float* mkpts1_f;
    float* dbg0;                // optional [M, 25, 128] fp32 dumps of the transformer output (tests)
    float* dbg1;
    int M, hf0, wf0, hf1, wf1, ldf, w0c, w1c, stride;
    const int* count;           // NULL, or the device-side match count: the launch covers M = the capacity of the lists and only the first
                                // min(M, *count) matches are processed (no host round trip between coarse matching and this kernel)
    float fscale, eps;
    int has_scale0;
    int dbg_stage;
};

#ifdef FF_NOSYNC   // timing experiment only (results are wrong): how much of the kernel is barrier waiting
#define FF_SYNC() do { } while (0)
#else
#define FF_SYNC() __syncthreads()
#endif

struct Lane {
    int lane, l31, lh, wn, sw, tsw;  // wave wn owns all 64 rows (both matches) x columns 32wn..32wn+31;  tsw = (l31 >> 1) & 7: key of transposed tiles
                                     // sw = l31 & 15: the XOR swizzle key of this lane's row
    // byte offsets inside a 32 KiB tile that every access is "one of these + a compile-time constant" of (the XOR swizzle
    // cannot be folded into an instruction offset; without the tables the compiler materialises and hoists one address
    // register per (buffer, row fragment, k step) and spills a hundred of them)
#ifdef FF_DEBUG_STAGES
    int dbg_stage, dbg_call, dbg_m_base, dbg_M; float* dbg_out;
#endif
    int a8[8];    // MFMA operand rows: l31 * 256 + (((2ks + lh) ^ sw) << 4)
    int st4[4];   // accumulator-layout 8-byte pieces: l31 * 256 + (((4wn + rg) ^ sw) << 4) + 8lh
    __device__ __forceinline__ void tables(int key) {   // key = sw, passed through an opaque asm so that the tables (and
#pragma unroll                                           // everything derived from them) are rebuilt per call, not kept alive
        for (int ks = 0; ks < 8; ++ks) a8[ks] = l31 * ROWB + (((2 * ks + lh) ^ key) << 4);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) st4[rg] = l31 * ROWB + (((4 * wn + rg) ^ key) << 4) + lh * 8;
    }
};

struct W8 { bf16x8_t f[8];           // 8 weight fragments (one 128-deep K range of this wave's 32 output columns)
#ifdef FF_NOPREFETCH
            const uint4* p;
#endif
};

__device__ __forceinline__ float elu1(float v) { return fmaxf(v, 0.f) + __expf(fminf(v, 0.f)); }   // elu(v) + 1

// issue the 8 fragment loads of one weight unit (1 KiB per wave instruction, straight from L2); consumed one unit later
__device__ __forceinline__ void wload(W8& w, const uint4* __restrict__ p, const Lane& L) {
#ifdef FF_NOPREFETCH
    w.p = p; return;
#endif
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w.f[ks] = __builtin_bit_cast(bf16x8_t, p[ks * 64 + L.lane]);
    __builtin_amdgcn_sched_barrier(0);   // keep the requests here: hoisted above the MFMAs they would need a second register set
}

// acc[j] (+)= rows [32j ..] of the LDS operand tile `a` (128 channels = 8 k16 steps) x this wave's 8 fragments
template <bool SWAP, bool FIRST>
__device__ __forceinline__ void mma8(const char* a, const W8& w, f32x16_t (&acc)[2], const Lane& L) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FIRST) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const char* ab = a + L.a8[ks];
#ifdef FF_NOPREFETCH
        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, w.p[ks * 64 + L.lane]);
#else
        const bf16x8_t wf = w.f[ks];
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8_t v = *(const bf16x8_t*)(ab + j * 32 * ROWB);
            if constexpr (SWAP) acc[j] = mfma_h16_32x32x16(v, wf, acc[j]);
            else acc[j] = mfma_h16_32x32x16(wf, v, acc[j]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// accumulators in row orientation (lane = token row 32j + l31, channels 32wn + 8rg + 4lh + e) -> bf16 rows in LDS
template <int ACT>  // 0 none, 1 relu, 2 elu+1
__device__ __forceinline__ void store_rows(char* buf, const f32x16_t (&acc)[2], const Lane& L) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = acc[j][rg * 4 + e];
                v[e] = ACT == 1 ? fmaxf(x, 0.f) : (ACT == 2 ? elu1(x) : x);
            }
            *(uint2*)(buf + L.st4[rg] + j * 32 * ROWB) = make_uint2(cvt_pk_h16(v[0], v[1]), cvt_pk_h16(v[2], v[3]));
        }
}

// accumulators in swapped orientation (lane = channel 32wn + l31, tokens of match j: 8rg + 4lh + e) -> [channel][token]
// in LDS, padding tokens (>= 25 of each match) written as exact zeros
template <int ACT>
__device__ __forceinline__ void store_transposed(char* buf, const f32x16_t (&acc)[2], const Lane& L) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = acc[j][rg * 4 + e];
                v[e] = (8 * rg + 4 * L.lh + e < WW) ? (ACT == 2 ? elu1(x) : x) : 0.f;
            }
            *(uint2*)(buf + (32 * L.wn + L.l31) * TROWB + (((4 * j + rg) ^ L.tsw) << 4) + L.lh * 8) =
                make_uint2(cvt_pk_h16(v[0], v[1]), cvt_pk_h16(v[2], v[3]));
        }
}

// LayerNorm over the 128 channels of every row, in accumulator layout (the 4 waves hold 32 channels each).
// Contains one workgroup barrier.  transformer.py:53,57 (nn.LayerNorm, biased variance, eps 1e-5).
__device__ __forceinline__ void layernorm_rows(f32x16_t (&acc)[2], const float* __restrict__ gamma, const float* __restrict__ beta,
                                               float2* stat, float eps, const Lane& L) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s += acc[j][r]; q = fmaf(acc[j][r], acc[j][r], q); }
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (L.lh == 0) stat[(32 * j + L.l31) * 4 + L.wn] = make_float2(s, q);
    }
    float g[16], b[16];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const float4 gg = *(const float4*)(gamma + 32 * L.wn + 8 * rg + 4 * L.lh);
        const float4 bb = *(const float4*)(beta + 32 * L.wn + 8 * rg + 4 * L.lh);
        g[rg * 4] = gg.x; g[rg * 4 + 1] = gg.y; g[rg * 4 + 2] = gg.z; g[rg * 4 + 3] = gg.w;
        b[rg * 4] = bb.x; b[rg * 4 + 1] = bb.y; b[rg * 4 + 2] = bb.z; b[rg * 4 + 3] = bb.w;
    }
    FF_SYNC();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4 p0 = *(const float4*)(stat + (32 * j + L.l31) * 4), p1 = *(const float4*)(stat + (32 * j + L.l31) * 4 + 2);
        const float s = (p0.x + p0.z) + (p1.x + p1.z), q = (p0.y + p0.w) + (p1.y + p1.w);
        const float mean = s * (1.0f / C);
        const float var = fmaxf(q * (1.0f / C) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        const float nm = -mean * rstd;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = fmaf(fmaf(acc[j][r], rstd, nm), g[r], b[r]);
    }
}

// fp32 master of a token stream in accumulator layout <- its bf16 operand tile (+ the parked low halves)
template <bool WITH_LO>
__device__ __forceinline__ void load_master(f32x16_t (&xm)[2], const char* xb, const char* lo, const Lane& L) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const uint2 u = *(const uint2*)(xb + L.st4[rg] + j * 32 * ROWB);
            xm[j][rg * 4] = h16_lo(u.x); xm[j][rg * 4 + 1] = h16_hi(u.x);
            xm[j][rg * 4 + 2] = h16_lo(u.y); xm[j][rg * 4 + 3] = h16_hi(u.y);
            if constexpr (WITH_LO) {
                if (L.l31 < WW) {
                    const int cr = j * WW + L.l31;
                    const uint2 v = *(const uint2*)(lo + cr * ROWB + (((4 * L.wn + rg) ^ (cr & 15)) << 4) + L.lh * 8);
                    xm[j][rg * 4] += h16_lo(v.x); xm[j][rg * 4 + 1] += h16_hi(v.x);
                    xm[j][rg * 4 + 2] += h16_lo(v.y); xm[j][rg * 4 + 3] += h16_hi(v.y);
                }
            }
        }
}

// low halves x - bf16(x) of a master, packed as bf16 (the high halves are the operand tile store_rows wrote)
__device__ __forceinline__ void pack_lo(uint2 (&lo)[2][4], const f32x16_t (&xm)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const unsigned h0 = cvt_pk_h16(xm[j][rg * 4], xm[j][rg * 4 + 1]), h1 = cvt_pk_h16(xm[j][rg * 4 + 2], xm[j][rg * 4 + 3]);
            lo[j][rg] = make_uint2(cvt_pk_h16(xm[j][rg * 4] - h16_lo(h0), xm[j][rg * 4 + 1] - h16_hi(h0)),
                                   cvt_pk_h16(xm[j][rg * 4 + 2] - h16_lo(h1), xm[j][rg * 4 + 3] - h16_hi(h1)));
        }
}
__device__ __forceinline__ void store_lo(char* lob, const uint2 (&lo)[2][4], const Lane& L) {
    if (L.l31 < WW) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int cr = j * WW + L.l31;
                *(uint2*)(lob + cr * ROWB + (((4 * L.wn + rg) ^ (cr & 15)) << 4) + L.lh * 8) = lo[j][rg];
            }
    }
}

#ifdef FF_DEBUG_STAGES
#define FF_STAGE(ID, BUFPTR, TRANSPOSED)                                                                              \
    if (L.dbg_stage == L.dbg_call * 10 + (ID)) {                                                                                          \
        FF_SYNC();                                                                                              \
        for (int e = threadIdx.x; e < ROWS * 128; e += 256) {                                                         \
            const int r = e >> 7, c = e & 127;   /* r: token slot, c: channel */                                      \
            const int row = (TRANSPOSED) ? c : r, col = (TRANSPOSED) ? r : c; /* tile[row][col] */                    \
            const unsigned short hv = (TRANSPOSED) ? *(const unsigned short*)((BUFPTR) + row * TROWB + ((((col >> 3)) ^ ((row >> 1) & 7)) << 4) + (col & 7) * 2) \
                                                   : *(const unsigned short*)((BUFPTR) + row * ROWB + ((((col >> 3)) ^ (row & 15)) << 4) + (col & 7) * 2); \
            const int tokrow = (TRANSPOSED) ? col : row, ch = (TRANSPOSED) ? row : col;                               \
            const int mq = tokrow >> 5, tok = tokrow & 31;                                                            \
            if (tok < WW && L.dbg_m_base + mq < L.dbg_M) L.dbg_out[((size_t)(L.dbg_m_base + mq) * WW + tok) * C + ch] = h16_to_f32(hv); \
        }                                                                                                             \
        L.dbg_stage = -1;                                                                                             \
    }
#else
#define FF_STAGE(ID, BUFPTR, TRANSPOSED)
#endif

// LoFTREncoderLayer.forward(x, source) (transformer.py:35-58) for the workgroup's 2 matches.  bx / bs: LDS operand tiles of
// x / source; xm: fp32 master of x in accumulator layout.  The 8 fragments of the NEXT weight unit are requested into `w` right
// behind the MFMAs of the current one, so the fetch runs under the epilogue / barrier (on entry `w` holds this call's Wk, on exit
// `wnext`, the next call's Wk).  Ends with a barrier.
__device__ __forceinline__ void encoder_layer(char* smem, char* bx, const char* bs, const uint4* __restrict__ wl, const uint4* __restrict__ wnext,
                                              const float* __restrict__ ln, f32x16_t (&xm)[2], W8& w, float eps, Lane& L) {
    {
        int key = L.sw;
        asm volatile("" : "+v"(key));   // opaque: address tables are per call (see Lane)
        L.sw = key;
        L.tables(key);
    }
    char* T1 = smem + OFF_T1;
    char* T2 = smem + OFF_T2;
    float* ksum = (float*)(smem + OFF_SCR + SCR_KSUM);
    float2* stat = (float2*)(smem + OFF_SCR + SCR_STAT);
    const uint4* w128 = wl + L.wn * 8 * 64;    // this wave's stream inside a [128 x 128] unit
    const uint4* w256 = wl + L.wn * 16 * 64;   // ... inside a [128 x 256] unit (two consecutive 8-fragment halves)
    f32x16_t acc[2];
    // ---- K^T = elu1(S Wk)^T -> T1,  V^T = (S Wv)^T -> T2 ------------------------------------------------------
    mma8<true, true>(bs, w, acc, L);
    wload(w, w128 + W_V, L);
    store_transposed<2>(T1, acc, L);
    mma8<true, true>(bs, w, acc, L);
    wload(w, w128 + W_Q, L);
    store_transposed<0>(T2, acc, L);
    FF_SYNC();
    FF_STAGE(1, T1, true)
    FF_STAGE(2, T2, true)
    // ---- two waves per match (mm), two 32-channel groups (= 4 heads) each: K sums, KV = K^T V masked to the head blocks
    const int mm = L.wn >> 1, ig0 = 2 * (L.wn & 1);
    {
        const int ch = 64 * (L.wn & 1) + L.lane;
        float s = 0.f;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const uint4 u = *(const uint4*)(T1 + ch * TROWB + (((4 * mm + sl) ^ ((ch >> 1) & 7)) << 4));
            const unsigned uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) s += h16_lo(uu[t]) + h16_hi(uu[t]);
        }
        ksum[mm * C + ch] = s;
    }
    bf16x8_t kvp[2][2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        f32x16_t kv;
#pragma unroll
        for (int r = 0; r < 16; ++r) kv[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = (32 * (ig0 + ii) + L.l31) * TROWB + (((4 * mm + 2 * ks + L.lh) ^ L.tsw) << 4);
            const bf16x8_t a = *(const bf16x8_t*)(T1 + off), b = *(const bf16x8_t*)(T2 + off);
            kv = mfma_h16_32x32x16(a, b, kv);  // lane: v-channel l31, k-channels 8rg+4lh+e
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bool keep = (L.l31 >> 4) == ks;  // same head: k-channels 16ks.. with v-channels 16ks..
            unsigned p[4];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int rg = 2 * ks + h2;
                p[h2 * 2] = keep ? cvt_pk_h16(kv[rg * 4], kv[rg * 4 + 1]) : 0u;
                p[h2 * 2 + 1] = keep ? cvt_pk_h16(kv[rg * 4 + 2], kv[rg * 4 + 3]) : 0u;
            }
            kvp[ii][ks] = __builtin_bit_cast(bf16x8_t, make_uint4(p[0], p[1], p[2], p[3]));
        }
    }
    FF_SYNC();  // every wave is done with K^T / V^T
    // ---- Q = elu1(X Wq) -> T1 (row layout) -----------------------------------------------------------------------
    mma8<false, true>(bx, w, acc, L);
    wload(w, w128 + W_M, L);
    store_rows<2>(T1, acc, L);
    FF_SYNC();
    FF_STAGE(3, T1, false)
    // ---- Z = 1 / (Q . Ksum + eps) for this wave's 4 heads, msg = (Q KV) Z -> T2 rows of match mm ---------------------
    {
        const int row = 32 * mm + L.l31, hb = 4 * (L.wn & 1);
        float zq[4];   // Z of heads hb .. hb+3 for this lane's token
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {  // this lane: heads hb + 2lh + {0,1}; the partner lane (xor 32) has the other two
            const int h = hb + 2 * L.lh + hq;
            float d = 0.f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const uint4 u = *(const uint4*)(T1 + row * ROWB + (((2 * h + half) ^ L.sw) << 4));
                const float4 k0 = *(const float4*)(ksum + mm * C + 16 * h + 8 * half), k1 = *(const float4*)(ksum + mm * C + 16 * h + 8 * half + 4);
                d = fmaf(h16_lo(u.x), k0.x, d); d = fmaf(h16_hi(u.x), k0.y, d);
                d = fmaf(h16_lo(u.y), k0.z, d); d = fmaf(h16_hi(u.y), k0.w, d);
                d = fmaf(h16_lo(u.z), k1.x, d); d = fmaf(h16_hi(u.z), k1.y, d);
                d = fmaf(h16_lo(u.w), k1.z, d); d = fmaf(h16_hi(u.w), k1.w, d);
            }
            const float zz = 1.0f / (d + 1e-6f);
            const float other = __shfl_xor(zz, 32, 64);
            zq[hq] = L.lh == 0 ? zz : other;
            zq[2 + hq] = L.lh == 0 ? other : zz;
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = ig0 + ii;
            f32x16_t o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // contraction order follows the KV accumulator: position p <-> k-channel 16ks + 8(p/4) + 4lh + p%4
                const uint2 lo = *(const uint2*)(T1 + row * ROWB + (((4 * i + 2 * ks) ^ L.sw) << 4) + L.lh * 8);
                const uint2 hi = *(const uint2*)(T1 + row * ROWB + (((4 * i + 2 * ks + 1) ^ L.sw) << 4) + L.lh * 8);
                const bf16x8_t qf = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
                o = mfma_h16_32x32x16(kvp[ii][ks], qf, o);  // lane: token l31, v-channels 8rg+4lh+e
            }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float zz = zq[2 * ii + (rg >> 1)];
                *(uint2*)(T2 + row * ROWB + (((4 * i + rg) ^ L.sw) << 4) + L.lh * 8) =
                    make_uint2(cvt_pk_h16(o[rg * 4] * zz, o[rg * 4 + 1] * zz), cvt_pk_h16(o[rg * 4 + 2] * zz, o[rg * 4 + 3] * zz));
            }
        }
    }
    FF_SYNC();
    FF_STAGE(4, T2, false)
    // ---- merge + norm1 -> T1 (transformer.py:52-53) ----------------------------------------------------------------
    mma8<false, true>(T2, w, acc, L);
    wload(w, w256 + W_0A, L);
    layernorm_rows(acc, ln, ln + C, stat, eps, L);   // barrier inside: every wave is done reading Q (T1) and msg (T2)
    store_rows<0>(T1, acc, L);
    FF_SYNC();
    FF_STAGE(5, T1, false)
    // ---- mlp: relu([x | msg] W0) W2 (transformer.py:55-56): hidden columns 0..127 -> T2, 128..255 -> T1 (LN1(msg) is dead
    // once both halves have read it), then ONE accumulation over both -- a single accumulator set is live at a time
    mma8<false, true>(bx, w, acc, L);
    wload(w, w256 + W_0A + 8 * 64, L);
    mma8<false, false>(T1, w, acc, L);
    wload(w, w256 + W_0B, L);
    store_rows<1>(T2, acc, L);
    mma8<false, true>(bx, w, acc, L);
    wload(w, w256 + W_0B + 8 * 64, L);
    mma8<false, false>(T1, w, acc, L);
    wload(w, w128 + W_2A, L);
    FF_SYNC();  // LN1(msg) consumed by every wave
    FF_STAGE(6, T2, false)
    store_rows<1>(T1, acc, L);
    FF_SYNC();
    FF_STAGE(7, T1, false)
    f32x16_t out[2];
    mma8<false, true>(T2, w, out, L);
    wload(w, w128 + W_2B, L);
    mma8<false, false>(T1, w, out, L);
    wload(w, wnext + L.wn * 8 * 64, L);   // the next call's Wk
    // ---- norm2, residual add in fp32, new operand copy of x (transformer.py:57-58) ------------------------------------
    layernorm_rows(out, ln + 2 * C, ln + 3 * C, stat, eps, L);  // barrier inside: every wave is done reading X
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) xm[j][r] += out[j][r];
    store_rows<0>(bx, xm, L);
    FF_SYNC();
    FF_STAGE(8, bx, false)
#ifdef FF_DEBUG_STAGES
    L.dbg_call++;
#endif
}

__global__ void __launch_bounds__(256, 2) fine_fused_kernel(const FineArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lane L;
    L.lane = threadIdx.x & 63;
    L.l31 = L.lane & 31;
    L.lh = L.lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    L.wn = wave;
    L.sw = L.l31 & 15;
    L.tsw = (L.l31 >> 1) & 7;
    const int m_base = blockIdx.x * G;
    const int Mv = a.count ? min(a.M, *a.count) : a.M;   // block-uniform
    if (m_base >= Mv) return;
#ifdef FF_DEBUG_STAGES
    L.dbg_stage = a.dbg_stage; L.dbg_call = 0; L.dbg_m_base = m_base; L.dbg_M = Mv; L.dbg_out = a.dbg0;
#endif
    W8 w;
    wload(w, a.wts + W_K + L.wn * 8 * 64, L);   // layer 0, Wk: in flight during the gather
    // ---- gather: 2 sides x 4 matches x 32 token slots x 256 B (fine_preprocess.py:40-47; border -> zeros like F.unfold padding)
    {
        // side and match of a row are uniform per pass (16 rows per pass, 32 token slots per match): the ids come through scalar
        // loads once, and all 8 window rows of a lane are requested before the first LDS write -- ONE round trip instead of
        // (ids -> row) x 2 batches of dependent ones
        const int t = threadIdx.x, slot = t & 15;
        static_assert(2 * ROWS / 16 == 8 && G == 2, "gather schedule: 8 passes = 2 sides x 2 matches x 2 half windows");
        int bq[G], ci[G], cj[G];
#pragma unroll
        for (int mm = 0; mm < G; ++mm) {
            const int m = m_base + mm;
            const bool ok = m < Mv;
            bq[mm] = ok ? (int)a.b_ids[m] : 0;
            ci[mm] = ok ? (int)a.i_ids[m] : 0;
            cj[mm] = ok ? (int)a.j_ids[m] : 0;
        }
        uint4 v[8];
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int side = pass >> 2, mm = (pass >> 1) & 1, tok = (pass & 1) * 16 + (t >> 4);
            const int m = m_base + mm;
            v[pass] = make_uint4(0u, 0u, 0u, 0u);
            if (tok < WW && m < Mv) {
                const int cell = side ? cj[mm] : ci[mm];
                const int wc = side ? a.w1c : a.w0c, hf = side ? a.hf1 : a.hf0, wf = side ? a.wf1 : a.wf0;
                const int cy = cell / wc, cx = cell - cy * wc;
                const int y = cy * a.stride - 2 + tok / 5, x = cx * a.stride - 2 + tok % 5;
                if (y >= 0 && y < hf && x >= 0 && x < wf)
                    v[pass] = *(const uint4*)((side ? a.f1 : a.f0) + (((size_t)bq[mm] * hf + y) * wf + x) * a.ldf + slot * 8);
            }
        }
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int side = pass >> 2, rr = (pass & 3) * 16 + (t >> 4);
            *(uint4*)(smem + (side ? OFF_X1 : OFF_X0) + rr * ROWB + ((slot ^ (rr & 15)) << 4)) = v[pass];
        }
    }
    FF_SYNC();
    char* X0 = smem + OFF_X0;
    char* X1 = smem + OFF_X1;
    char* LO = smem + OFF_LO;
    float* fin1 = (float*)(smem + OFF_T1);   // [4 x 25][FIN_LD] final fp32 tokens of image 1 (over the dead temporaries T1 | T2)
    // final centre tokens of image 0, 512 B per match: parked in padding rows 25, 26 of the match's X0 block -- the last call only
    // READS X0 (as the source), and whatever its padding rows hold is replaced by exact zeros in K^T / V^T (store_transposed)
    auto fin0 = [&](int mq) { return (float*)(smem + OFF_X0 + (32 * mq + WW) * ROWB); };
    f32x16_t xm[2];   // fp32 master of the stream being updated
    L.tables(L.sw);
    load_master<false>(xm, X0, LO, L);
    // layer 0 'self' (transformer.py:91-93), layer 1 'cross': feat0 first, feat1 against the UPDATED feat0 (:94-96)
#pragma unroll 1
    for (int layer = 0; layer < 2; ++layer) {
        const uint4* wl = a.wts + (size_t)layer * W_LAYER;
        const uint4* wk_next_layer = a.wts + (size_t)(layer == 0 ? W_LAYER : 0) + W_K;   // (after layer 1: a harmless re-fetch)
        const float* ln = a.ln + layer * 4 * C;
        encoder_layer(smem, X0, layer == 0 ? X0 : X1, wl, wl + W_K, ln, xm, w, a.eps, L);        // stream 0
        if (layer == 0) {
            uint2 lo[2][4];
            pack_lo(lo, xm);
            store_lo(LO, lo, L);                      // LO = low halves of stream 0
            load_master<false>(xm, X1, LO, L);        // stream 1 is still exactly its bf16 tile
        } else {
            // stream 0 is final: centre tokens for the fine matching, optional dump; then un-park stream 1
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int mq = j, m = m_base + mq;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int ch = 32 * L.wn + 8 * rg + 4 * L.lh;
                    const float4 v = make_float4(xm[j][rg * 4], xm[j][rg * 4 + 1], xm[j][rg * 4 + 2], xm[j][rg * 4 + 3]);
                    if (L.l31 == WW / 2) *(float4*)(fin0(mq) + ch) = v;
                    if (a.dbg0 && a.dbg_stage == 0 && m < Mv && L.l31 < WW) *(float4*)(a.dbg0 + ((size_t)m * WW + L.l31) * C + ch) = v;
                }
            }
            load_master<true>(xm, X1, LO, L);
        }
        encoder_layer(smem, X1, layer == 0 ? X1 : X0, wl, wk_next_layer, ln, xm, w, a.eps, L);   // stream 1
        if (layer == 0) {
            uint2 lo[2][4];
            pack_lo(lo, xm);                          // low halves of stream 1 ...
            load_master<true>(xm, X0, LO, L);         // ... swap places with those of stream 0 (lane-private LDS words)
            store_lo(LO, lo, L);
        }
    }
    // ---- stream 1 is final: optional dump, fp32 tokens -> LDS for the fine matching (fine_matching.py:43-74) ---------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int mq = j, m = m_base + mq;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int ch = 32 * L.wn + 8 * rg + 4 * L.lh;
            const float4 v = make_float4(xm[j][rg * 4], xm[j][rg * 4 + 1], xm[j][rg * 4 + 2], xm[j][rg * 4 + 3]);
            if (L.l31 < WW) *(float4*)(fin1 + (mq * WW + L.l31) * FIN_LD + ch) = v;   // T1 / T2 are dead: the last call ended with a barrier
            if (a.dbg1 && m < Mv && L.l31 < WW) *(float4*)(a.dbg1 + ((size_t)m * WW + L.l31) * C + ch) = v;
        }
    }
    FF_SYNC();
    const int m = m_base + wave;
    if (wave >= G || m >= Mv) return;
    float s = -INFINITY;
    if (L.lane < WW) {
        const float* kr = fin1 + (wave * WW + L.lane) * FIN_LD;
        const float* q = fin0(wave);
        float acc = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; c += 4) {
            const float4 x = *(const float4*)(q + c), y = *(const float4*)(kr + c);
            acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc);
            acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        }
        s = (1.0f / sqrtf((float)C)) * acc;  // softmax_temp * sim_matrix
    }
    const float mx = wave_max(s);
    const float e = L.lane < WW ? expf(s - mx) : 0.f;
    const float heat = e / wave_sum(e);
    const float gx = L.lane < WW ? -1.0f + 0.5f * (float)(L.lane % 5) : 0.f;   // create_meshgrid(5, 5, normalized): {-1,-.5,0,.5,1}
    const float gy = L.lane < WW ? -1.0f + 0.5f * (float)(L.lane / 5) : 0.f;
    const float cx = wave_sum(gx * heat), cy = wave_sum(gy * heat);
    const float vx = wave_sum(gx * gx * heat) - cx * cx, vy = wave_sum(gy * gy * heat) - cy * cy;
    if (L.lane == 0) {
        const float sd = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f));
        a.expec_f[3 * m + 0] = cx; a.expec_f[3 * m + 1] = cy; a.expec_f[3 * m + 2] = sd;
        if (a.count && !(fabsf(cx) + fabsf(cy) + sd < INFINITY)) atomicOr((int*)a.count + 1, 2);   // health word of the coarse count buffer: non-finite fine output
        float s1x = a.fscale, s1y = a.fscale;
        if (a.has_scale0) {  // quirk preserved: keyed on scale0, multiplies scale1 (fine_matching.py:68)
            const int b = (int)a.b_ids[m];
            s1x = a.fscale * a.scale1[2 * b + 0];
            s1y = a.fscale * a.scale1[2 * b + 1];
        }
        a.mkpts1_f[2 * m + 0] = a.mkpts1_c[2 * m + 0] + cx * 2.0f * s1x;   // W // 2 = 2
        a.mkpts1_f[2 * m + 1] = a.mkpts1_c[2 * m + 1] + cy * 2.0f * s1y;
    }
}

}  // namespace

extern "C" int64_t GIM_FN(gim_fine_fused_weight_bytes)(void) { return (int64_t)2 * W_LAYER * 16; }

static int fine_fused_launch(const void* feat_f0, const void* feat_f1, const int64_t* b_ids, const int64_t* i_ids,
                              const int64_t* j_ids, const float* mkpts1_c, const float* scale1, const void* weights,
                              const float* ln_params, float* expec_f, float* mkpts1_f, float* dbg_fine0, float* dbg_fine1,
                              int M, int hf0, int wf0, int hf1, int wf1, int C_, int ldf, int w0c, int w1c, int stride, int W,
                              float scale, float ln_eps, int has_scale0, const int* count, gim_stream_t stream) {
    if (M == 0) return GIM_OK;
    GIM_REQUIRE(feat_f0 && feat_f1 && b_ids && i_ids && j_ids && mkpts1_c && weights && ln_params && expec_f && mkpts1_f, "fine_fused: NULL pointer");
    GIM_REQUIRE(C_ == C && W == 5, "fine_fused: built for d_model 128 / 5x5 windows (got C=%d W=%d)", C_, W);
    GIM_REQUIRE(M > 0 && hf0 > 0 && wf0 > 0 && hf1 > 0 && wf1 > 0 && stride > 0 && ldf >= C && ldf % 8 == 0, "fine_fused: bad geometry");
    GIM_REQUIRE((dbg_fine0 == nullptr) == (dbg_fine1 == nullptr), "fine_fused: give both debug outputs or none");
    GIM_REQUIRE(!has_scale0 || scale1, "fine_fused: scale1 required when has_scale0");
    static GimPerDevice attr;
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)fine_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) { gim_set_error("fine_fused: hipFuncSetAttribute(%d B LDS): %s", SMEM, hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
        attr.done();
    }
    FineArgs a;
    a.f0 = (const unsigned short*)feat_f0; a.f1 = (const unsigned short*)feat_f1;
    a.b_ids = b_ids; a.i_ids = i_ids; a.j_ids = j_ids; a.mkpts1_c = mkpts1_c; a.scale1 = scale1;
    a.wts = (const uint4*)weights; a.ln = ln_params; a.expec_f = expec_f; a.mkpts1_f = mkpts1_f; a.dbg0 = dbg_fine0; a.dbg1 = dbg_fine1;
    a.M = M; a.hf0 = hf0; a.wf0 = wf0; a.hf1 = hf1; a.wf1 = wf1; a.ldf = ldf; a.w0c = w0c; a.w1c = w1c; a.stride = stride;
    a.fscale = scale; a.eps = ln_eps; a.has_scale0 = has_scale0; a.count = count;
    a.dbg_stage = 0;
#ifdef FF_DEBUG_STAGES
    a.dbg_stage = FF_DEBUG_STAGES;   // development builds only: -DFF_DEBUG_STAGES=<stage> (no environment variable is read in csrc/)
#endif
    hipLaunchKernelGGL(fine_fused_kernel, dim3((unsigned)((M + G - 1) / G)), dim3(256), SMEM, (hipStream_t)stream, a);
    return gim_check_launch("fine_fused");
}

extern "C" int GIM_FN(gim_fine_fused)(const void* feat_f0, const void* feat_f1, const int64_t* b_ids, const int64_t* i_ids,
                              const int64_t* j_ids, const float* mkpts1_c, const float* scale1, const void* weights,
                              const float* ln_params, float* expec_f, float* mkpts1_f, float* dbg_fine0, float* dbg_fine1,
                              int M, int hf0, int wf0, int hf1, int wf1, int C_, int ldf, int w0c, int w1c, int stride, int W,
                              float scale, float ln_eps, int has_scale0, gim_stream_t stream) {
    return fine_fused_launch(feat_f0, feat_f1, b_ids, i_ids, j_ids, mkpts1_c, scale1, weights, ln_params, expec_f, mkpts1_f, dbg_fine0, dbg_fine1,
                             M, hf0, wf0, hf1, wf1, C_, ldf, w0c, w1c, stride, W, scale, ln_eps, has_scale0, nullptr, stream);
}

extern "C" int GIM_FN(gim_fine_fused_dev)(const void* feat_f0, const void* feat_f1, const int64_t* b_ids, const int64_t* i_ids,
                                  const int64_t* j_ids, const float* mkpts1_c, const float* scale1, const void* weights,
                                  const float* ln_params, float* expec_f, float* mkpts1_f, int M_cap, const int* count_dev,
                                  int hf0, int wf0, int hf1, int wf1, int C_, int ldf, int w0c, int w1c, int stride, int W,
                                  float scale, float ln_eps, int has_scale0, gim_stream_t stream) {
    GIM_REQUIRE(count_dev, "fine_fused_dev: NULL count");
    return fine_fused_launch(feat_f0, feat_f1, b_ids, i_ids, j_ids, mkpts1_c, scale1, weights, ln_params, expec_f, mkpts1_f, nullptr, nullptr,
                             M_cap, hf0, wf0, hf1, wf1, C_, ldf, w0c, w1c, stride, W, scale, ln_eps, has_scale0, count_dev, stream);
}

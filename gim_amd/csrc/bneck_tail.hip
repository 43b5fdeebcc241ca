// Tail of a ResNet Bottleneck fused with the head of the next block, planes P = 128 / 256 (layers 2 and 3 of the gim_loftr backbone,
// networks/loftr/backbone/resnet.py:109-126), for gfx950:
//
//     x'  = relu(bn3(conv3_1x1(t2)) + identity)        P -> 4P          (resnet.py:117-124)
//     t1' = relu(bn1'(conv1'_1x1(x')))                4P -> N1          (the NEXT block's resnet.py:109-111)
//
// Two implicit-GEMM launches become one: x' (the widest tensor of the block, 315 MB at 16 x 120 x 160 x 512) is written once and
// NOT read back for conv1'.  Both convolutions are 1x1, so a "tile" is simply 256 (128) consecutive pixel rows; wave w of the
// 8-wave (4-wave: layer 3, see Cfg) workgroup owns 32 of them for both products.
//
// The chain stays in registers exactly as in bneck_fused.hip: every MFMA is computed transposed (weights = A operand, pixels = B
// operand), a lane's accumulator quad is 4 consecutive channels of ONE pixel, and two v_cvt_pk per quad turn it into the next
// product's B fragment -- conv1's K axis is packed in that accumulator order (packing.py::pack_bneck_tail).
//
// The weights (conv3 4P x P, conv1' N1 x 4P: 256 KiB at N1 = 128) do not fit LDS next to anything else, so they STREAM: the 4P
// output channels of conv3 are walked in chunks of CH = 64 (P = 128) or 32 (P = 256); chunk q needs W3[CH q .. CH q + CH - 1][:] (16 KiB)
// and W1'[:, CH q .. CH q + CH - 1] (16 / 32 KiB), fetched by LDS-DMA into a ring of chunk buffers ahead of their use (L2-resident after the first workgroups).  Per chunk and
// wave: 16 MFMAs of conv3, the + identity / relu / store of 64 channels of x', 16 (32) MFMAs of conv1'.
//
// Vector memory operations return IN ORDER: anything issued behind a store cannot be waited for without waiting for that
// store's acknowledgement.  Hence the order inside a chunk -- next chunk's weight DMA and next chunk's identity rows are issued
// BEFORE this chunk's stores of x', and the wait at the top of the next chunk is a counted vmcnt(4) that leaves exactly those
// four stores in flight.
#include "gim_common.h"

namespace {

// NW waves per workgroup, 32 pixel rows each: 8 (one workgroup per CU) or 4 (two per CU: layer 3, whose 76 800 rows are 300 tiles of
// 256 -- 1.17 rounds of 256 CUs -- but 600 tiles of 128 on 512 slots with a short second round; the two co-resident workgroups also
// run their chunk barriers independently, so one's LDS phase overlaps the other's MFMAs)
// DS (round 5; planes 128): the block's `downsample` branch -- identity = bn(conv1x1 stride 2 (x_in)), resnet.py:120-124 -- is computed HERE as
// extra K of conv3: x' = relu([W3 | Wds] [t2 ; x_in(2y, 2x)] + b3 + bds).  The wave's 32 strided input pixels (2 P channels each) live in
// registers beside t2; the weight rows are P + 2P long, so the chunks are 32 channels wide; no identity tensor exists (its launch, its
// write and its read are gone) and the identity DMA of the chunk loop is simply not issued -- the counted waits are the same.
template <int P, int N1, int NW = 8, bool DS = false> struct Cfg {
    static constexpr int ROWS = 32 * NW;
    static constexpr int LDS_LIMIT = (NW == 8 ? 160 : 80) * 1024;
    static constexpr int C4 = 4 * P;
    static constexpr int KD = DS ? 2 * P : 0;                // channels of the downsample branch's input
    static constexpr int CH = (P == 128 && !DS) ? 64 : 32;   // conv3 output channels per chunk
    static constexpr int NCHUNK = C4 / CH;                   // 8 / 32 (DS: 16)
    static constexpr int W3ROW = (P + KD) * 2;               // bytes of a conv3 weight row (K = P, DS: 3 P): 256 / 512 (768)
    static constexpr int W3C = CH * W3ROW;                   // 16 KiB
    static constexpr int W1ROW = CH * 2;                     // bytes of a conv1' weight row of one chunk (K = CH): 128 / 64
    static constexpr int W1C = N1 * W1ROW;                   // 16 / 32 KiB
    static constexpr int BUF = W3C + W1C;
    static constexpr int PROW = CH * 2;                      // bytes of one pixel row of the per-wave patch: 128 / 64
    static constexpr int PATCH = 32 * PROW;                  // [32 px][PROW]
    static constexpr int NBUF = (3 * BUF + NW * PATCH + (C4 + N1) * 4 <= LDS_LIMIT) ? 3 : 2;   // chunk buffers: NBUF - 1 chunks ahead
    static constexpr int OFF_PATCH = NBUF * BUF;
    static constexpr int OFF_BIAS = OFF_PATCH + NW * PATCH;  // b3 [C4] then b1' [N1], fp32
    static constexpr int SMEM = OFF_BIAS + (C4 + N1) * 4;
    static constexpr int PIECES = BUF / 1024;                // LDS-DMA instructions per chunk (32 / 48)
    static constexpr int PPW = PIECES / NW;                  // ... per wave
    static constexpr int IPC = PATCH / 1024;                 // identity pieces (= x' stores) per wave and chunk: 4 / 2
    static_assert(SMEM <= LDS_LIMIT && PIECES % NW == 0 && (P == 128 || P == 256) && (NW == 8 || NW == 4) && NW * 4096 <= NBUF * BUF && (!DS || P == 128) &&
                  W3C % 1024 == 0, "LDS map");
};

struct Args {
    const unsigned short* t2;    // [M][P]    conv2 output (after bn2 + relu)
    const unsigned short* res;   // [M][4P]   identity / downsample branch
    unsigned short* xo;          // [M][4P]   x'  (NULL: not stored -- nobody reads the last block's output but the fused conv1')
    unsigned short* t1n;         // [M][N1]   t1'
    const unsigned short* w3;    // [4P][P], K in natural channel order
    const unsigned short* w1n;   // [chunks][N1][CH], K of a chunk in accumulator order
    const float* b3;             // [4P]
    const float* b1n;            // [N1]
    int M;
    int act1;                    // activation of conv1': GIM_ACT_RELU (next block's conv1) or GIM_ACT_NONE
    unsigned w3_bytes, w1n_bytes;
    int* health;                 // fp16 range guard word (gim_common.h) or NULL
    const unsigned short* xin;   // DS: [B][Hin][Win][2P] the block's input; output pixel (b, y, x) of the Ho x Wo map reads (b, 2y, 2x)
    int Ho, Wo, Hin, Win;
};

typedef __attribute__((address_space(3))) void lds_t;
GIM_TT_DECL(bneck_tail)

// The weight and identity streams use gim_dma16 (gim_common.h): LDS-DMA through inline asm, counted by hand (tail_wait below).
typedef gim_u32x4_t u32x4_t;
__device__ __forceinline__ void dma16(const u32x4_t rsrc, unsigned lds_addr, unsigned voff) { gim_dma16(rsrc, lds_addr, voff); }
__device__ __forceinline__ u32x4_t make_rsrc(const void* p, unsigned bytes) { return gim_make_rsrc(p, bytes); }
template <int N> __device__ __forceinline__ void tail_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// LDS images (LDS-DMA is lane-linear, so every swizzle is applied to the SOURCE slot and again on the ds_read side):
//   rows of 512 / 256 B (32 / 16 slots of 16 B): slot' = (slot & ~15) | ((slot & 15) ^ (row & 15))
//   rows of 128 B (8 slots):  slot' = slot ^ ((row >> 1) & 7)          (two rows share a 256-byte bank row)
//   rows of  64 B (4 slots):  slot' = slot ^ ((row >> 2) & 3)          (four rows share one)
template <int ROWB> __device__ __forceinline__ int swz(int row, int slot) {
    if constexpr (ROWB >= 256) return (slot & ~15) | ((slot & 15) ^ (row & 15));
    else if constexpr (ROWB == 128) return slot ^ ((row >> 1) & 7);
    else return slot ^ ((row >> 2) & 3);
}
// the per-wave patch [32 px][PROW]: slot' = slot ^ (px & 7) for 128-byte rows, slot ^ ((px >> 2) & 3) for 64-byte rows
template <int PROW> __device__ __forceinline__ int pswz(int px, int slot) {
    if constexpr (PROW == 128) return slot ^ (px & 7);
    else return slot ^ ((px >> 2) & 3);
}

// chunk q -> LDS buffer `buf`: this wave's share of the pieces
template <int P, int N1, int NW, bool DS>
__device__ __forceinline__ void issue_chunk(const u32x4_t rw3, const u32x4_t rw1, unsigned smem_addr, int buf, int q, int w, int lane) {
    typedef Cfg<P, N1, NW, DS> C;
    constexpr int S3 = C::W3ROW / 16, S1 = C::W1ROW / 16;   // slots per row
    const unsigned base = smem_addr + (unsigned)(buf * C::BUF);
#pragma unroll
    for (int k = 0; k < C::PPW; ++k) {
        const int pc = w + NW * k;                         // wave-uniform piece index
        if (pc < C::W3C / 1024) {
            const int idx = pc * 64 + lane, n = idx / S3, d = idx % S3;
            const unsigned voff = (unsigned)((q * C::CH + n) * C::W3ROW + (swz<C::W3ROW>(n, d) << 4));
            dma16(rw3, base + (unsigned)(pc * 1024), voff);
        } else {
            const int p1 = pc - C::W3C / 1024;
            const int idx = p1 * 64 + lane, n = idx / S1, d = idx % S1;
            const unsigned voff = (unsigned)((q * N1 + n) * C::W1ROW + (swz<C::W1ROW>(n, d) << 4));
            dma16(rw1, base + (unsigned)(C::W3C + p1 * 1024), voff);
        }
    }
}

// No vector memory load of the chunk loop has a VGPR destination -- the identity rows take the same road as the weights: LDS-DMA
// straight into this wave's patch (row layout).  A load hipcc can see beside LDS-DMA makes it wait vmcnt(0); a load hidden in inline
// asm has its destination registers copied (v_mov) by the register allocator BEFORE the hand-placed wait whenever that wait sits in
// more than one branch (measured: garbage identity rows in some tiles of launches with more workgroups than CUs).
template <int P, int N1, int NW, bool DS = false>
__global__ void __launch_bounds__(NW * 64, 2) bneck_tail_kernel(const Args a) {
    typedef Cfg<P, N1, NW, DS> C;
    constexpr int ROWS = C::ROWS;
    constexpr int C4 = C::C4, CH = C::CH, NCHUNK = C::NCHUNK, PROW = C::PROW;
    constexpr int NF = N1 / 32;                            // conv1' output fragments per wave
    constexpr int TF = CH / 32;                            // conv3 output fragments per chunk (2 / 1)
    constexpr int LPP = PROW / 16;                         // lanes per pixel in row layout (8 / 4)
    constexpr int PPI = 64 / LPP;                          // pixels per wave instruction in row layout (8 / 16)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bool hbig = false;  // fp16 range guard: a residual-stream value this thread converts is beyond the fp16 range (gim_common.h)
    const size_t prow0 = (size_t)blockIdx.x * ROWS + w * 32;   // first pixel row of this wave
    char* patch = smem + C::OFF_PATCH + w * C::PATCH;

    GIM_TT(bneck_tail, w, 0);
    unsigned long long tt_wait = 0, tt_c3 = 0, tt_epi = 0, tt_c1 = 0, tt_a = 0, tt_b = 0;   // GIM_TIMING: per-phase totals over the chunks
    (void)tt_wait; (void)tt_c3; (void)tt_epi; (void)tt_c1; (void)tt_a; (void)tt_b;
    const u32x4_t rw3 = make_rsrc(a.w3, a.w3_bytes), rw1 = make_rsrc(a.w1n, a.w1n_bytes);
    const unsigned smem_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)smem);   // LDS byte address of the dynamic array
    issue_chunk<P, N1, NW, DS>(rw3, rw1, smem_addr, 0, 0, w, lane);
    // biases -> LDS (a global bias load inside the chunk loop makes the compiler wait vmcnt(0): it would drain the weight DMA)
    float* bias = (float*)(smem + C::OFF_BIAS);
    for (int t = threadIdx.x; t < (C4 + N1) / 4; t += NW * 64) {
        if (t < C4 / 4) *(float4*)(bias + 4 * t) = *(const float4*)(a.b3 + 4 * t);
        else *(float4*)(bias + 4 * t) = *(const float4*)(a.b1n + 4 * t - C4);
    }
    // conv3's pixel operand: P / 16 k16 steps, lane (pixel l31, half lh) holds channels 16s + 8lh .. + 7
    bf16x8_t t2[P / 16];
    {
        const unsigned short* tp = a.t2 + (prow0 + l31) * P + 8 * lh;
#pragma unroll
        for (int s = 0; s < P / 16; ++s) t2[s] = *(const bf16x8_t*)(tp + 16 * s);
    }
    // opaque to the optimiser from here on: left alone, hipcc RE-LOADS these fragments from global memory in every chunk
    // (rematerialisation beats the live VGPRs in its cost model) and waits vmcnt(0) for them -- draining the weight DMA each time
#pragma unroll
    for (int s = 0; s < P / 16; ++s) asm volatile("" : "+v"(t2[s]));
    // DS: the downsample branch's pixel operand, 2 P / 16 more k16 steps of the same lane layout, from the strided source pixel
    bf16x8_t xi[DS ? C::KD / 16 : 1];
    if constexpr (DS) {
        const size_t m = prow0 + l31;
        const int ox = (int)(m % (size_t)a.Wo);
        const size_t r_ = m / (size_t)a.Wo;
        const int oy = (int)(r_ % (size_t)a.Ho);
        const size_t ib = r_ / (size_t)a.Ho;
        const unsigned short* xp = a.xin + ((ib * a.Hin + 2 * oy) * a.Win + 2 * ox) * C::KD + 8 * lh;
#pragma unroll
        for (int s = 0; s < C::KD / 16; ++s) xi[s] = *(const bf16x8_t*)(xp + 16 * s);
#pragma unroll
        for (int s = 0; s < C::KD / 16; ++s) asm volatile("" : "+v"(xi[s]));
    }
    // identity rows of chunk q: 32 px x PROW bytes = IPC pieces; lane i of piece k -> pixel PPI k + i / LPP, patch slot i % LPP <-
    // source slot pswz(pixel, i % LPP)
    const u32x4_t rres = make_rsrc(DS ? (const void*)a.t2 : (const void*)a.res, DS ? 16u : (unsigned)((size_t)a.M * C4 * 2));   // (DS: never used)
    const unsigned patch_addr = smem_addr + (unsigned)(C::OFF_PATCH + w * C::PATCH);
    const int ipx = lane / LPP, isl = lane % LPP;
    const unsigned id_voff = (unsigned)((prow0 + ipx) * (C4 * 2)) + (unsigned)(pswz<PROW>(ipx, isl) << 4);   // (px + PPI k) keeps its key
    auto issue_identity = [&](const int q) __attribute__((always_inline)) {
        if constexpr (!DS) {
#pragma unroll
            for (int k = 0; k < C::IPC; ++k) dma16(rres, patch_addr + (unsigned)(k * 1024), id_voff + (unsigned)(k * PPI * C4 * 2 + q * CH * 2));
        }
    };
    issue_identity(0);
    if constexpr (C::NBUF == 3) issue_chunk<P, N1, NW, DS>(rw3, rw1, smem_addr, 1, 1, w, lane);   // behind what chunk 0 needs: may stay in flight
    __syncthreads();             // biases are in LDS (the compiler knows nothing of the DMA in flight: no drain)

    GIM_TT(bneck_tail, w, 1);
    f32x16_t c1[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 bb = *(const float4*)(bias + C4 + 32 * f + 8 * rg + 4 * lh);
            c1[f][rg * 4] = bb.x; c1[f][rg * 4 + 1] = bb.y; c1[f][rg * 4 + 2] = bb.z; c1[f][rg * 4 + 3] = bb.w;
        }
    }

#pragma unroll 1
    for (int q = 0; q < NCHUNK; ++q) {
        // chunk q's weights (this wave's pieces) and identity rows have landed.  Younger operations that may fly on: the IPC stores
        // of the previous chunk and, with three buffers, the weight DMA of chunk q + 1 (issued behind the identity DMA of chunk q).
        // The count must be exact -- vmcnt(n) only guarantees that all but the n YOUNGEST operations are complete.
        tt_a = GIM_TT_NOW();
        const bool dma_ahead = C::NBUF == 3 && q + 1 < NCHUNK;
        const bool stores_behind = q > 0 && a.xo != nullptr;
        if (dma_ahead) { if (stores_behind) tail_wait<C::PPW + C::IPC>(); else tail_wait<C::PPW>(); }
        else { if (stores_behind) tail_wait<C::IPC>(); else tail_wait<0>(); }
        __syncthreads();          // everybody's pieces are visible, and everybody is done with the buffer of chunk q - 1
        const char* wb3 = smem + (q % C::NBUF) * C::BUF;
        const char* wb1 = wb3 + C::W3C;
        tt_b = GIM_TT_NOW(); tt_wait += tt_b - tt_a;

        // ---- conv3, CH output channels: D[m = channel][n = pixel] over K = P ---------------------------------------------------
        f32x16_t c3[TF];
#pragma unroll
        for (int f = 0; f < TF; ++f) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 bb = *(const float4*)(bias + q * CH + 32 * f + 8 * rg + 4 * lh);
                c3[f][rg * 4] = bb.x; c3[f][rg * 4 + 1] = bb.y; c3[f][rg * 4 + 2] = bb.z; c3[f][rg * 4 + 3] = bb.w;
            }
        }
#pragma unroll
        for (int s = 0; s < P / 16; ++s) {
#pragma unroll
            for (int f = 0; f < TF; ++f) {
                const int n = 32 * f + l31;
                const bf16x8_t wv = *(const bf16x8_t*)(wb3 + n * C::W3ROW + (swz<C::W3ROW>(n, 2 * s + lh) << 4));
                c3[f] = mfma_h16_32x32x16(wv, t2[s], c3[f]);
            }
        }
        if constexpr (DS) {
#pragma unroll
            for (int s = 0; s < C::KD / 16; ++s) {
#pragma unroll
                for (int f = 0; f < TF; ++f) {
                    const int n = 32 * f + l31;
                    const bf16x8_t wv = *(const bf16x8_t*)(wb3 + n * C::W3ROW + (swz<C::W3ROW>(n, 2 * (P / 16 + s) + lh) << 4));
                    c3[f] = mfma_h16_32x32x16(wv, xi[s], c3[f]);
                }
            }
        }
        tt_a = GIM_TT_NOW(); tt_c3 += tt_a - tt_b;
        // ---- + identity (DMA'd into the patch in row layout), relu; x' chunk out; operand of conv1' ----------------------------------
#pragma unroll
        for (int f = 0; f < TF; ++f) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                if constexpr (DS) {   // the identity branch is already inside the accumulators
#pragma unroll
                    for (int e = 0; e < 4; ++e) c3[f][rg * 4 + e] = fmaxf(c3[f][rg * 4 + e], 0.f);
                } else {
                    const uint2 r = *(const uint2*)(patch + l31 * PROW + (pswz<PROW>(l31, 4 * f + rg) << 4) + lh * 8);
                    c3[f][rg * 4] = fmaxf(c3[f][rg * 4] + h16_lo(r.x), 0.f);
                    c3[f][rg * 4 + 1] = fmaxf(c3[f][rg * 4 + 1] + h16_hi(r.x), 0.f);
                    c3[f][rg * 4 + 2] = fmaxf(c3[f][rg * 4 + 2] + h16_lo(r.y), 0.f);
                    c3[f][rg * 4 + 3] = fmaxf(c3[f][rg * 4 + 3] + h16_hi(r.y), 0.f);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16x8_t xq[2 * TF];   // conv1' operand of this chunk: k16 step s = 2f + t (accumulator order)
#pragma unroll
        for (int f = 0; f < TF; ++f) {
            unsigned u[8];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                u[2 * rg] = cvt_pk_h16(c3[f][rg * 4], c3[f][rg * 4 + 1]);
                u[2 * rg + 1] = cvt_pk_h16(c3[f][rg * 4 + 2], c3[f][rg * 4 + 3]);
                *(uint2*)(patch + l31 * PROW + (pswz<PROW>(l31, 4 * f + rg) << 4) + lh * 8) = make_uint2(u[2 * rg], u[2 * rg + 1]);
            }
            xq[2 * f] = __builtin_bit_cast(bf16x8_t, make_uint4(u[0], u[1], u[2], u[3]));
            xq[2 * f + 1] = __builtin_bit_cast(bf16x8_t, make_uint4(u[4], u[5], u[6], u[7]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // named registers, not an array: behind "memory"-clobbering asm statements an array lives in scratch (bneck_fused.hip)
        const char* prd = patch + ipx * PROW + (pswz<PROW>(ipx, isl) << 4);       // row layout: pixel ipx + PPI k keeps its key
        uint4 x0 = *(const uint4*)(prd), x1 = *(const uint4*)(prd + PPI * PROW), x2 = x0, x3 = x0;
        if constexpr (C::IPC == 4) { x2 = *(const uint4*)(prd + 2 * PPI * PROW); x3 = *(const uint4*)(prd + 3 * PPI * PROW); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is free: the next chunk's identity rows may land in it
        h16_range_track(hbig, x0); h16_range_track(hbig, x1);   // x': the un-normalised residual stream
        if constexpr (C::IPC == 4) { h16_range_track(hbig, x2); h16_range_track(hbig, x3); }
        // in THIS order: identity of chunk q + 1, weights of chunk q + NBUF - 1 (into the buffer chunk q - 1 used, free since this
        // chunk's barrier), and only then this chunk's stores -- nothing the next chunks wait for sits behind a store
        if (q + 1 < NCHUNK) issue_identity(q + 1);
        if (q + C::NBUF - 1 < NCHUNK) issue_chunk<P, N1, NW, DS>(rw3, rw1, smem_addr, (q + C::NBUF - 1) % C::NBUF, q + C::NBUF - 1, w, lane);
        if (a.xo != nullptr) {
            unsigned short* xp = a.xo + (prow0 + ipx) * C4 + CH * q + isl * 8;
            *(uint4*)(xp) = x0; *(uint4*)(xp + (size_t)PPI * C4) = x1;
            if constexpr (C::IPC == 4) { *(uint4*)(xp + (size_t)2 * PPI * C4) = x2; *(uint4*)(xp + (size_t)3 * PPI * C4) = x3; }
        }
        tt_b = GIM_TT_NOW(); tt_epi += tt_b - tt_a;
        // ---- conv1' of the next block, this chunk's CH input channels: CH / 16 k16 steps x NF fragments ------------------------------
#pragma unroll
        for (int s = 0; s < 2 * TF; ++s) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int n = 32 * f + l31;
                const bf16x8_t wv = *(const bf16x8_t*)(wb1 + n * C::W1ROW + (swz<C::W1ROW>(n, 2 * s + lh) << 4));
                c1[f] = mfma_h16_32x32x16(wv, xq[s], c1[f]);
            }
        }
        tt_a = GIM_TT_NOW(); tt_c1 += tt_a - tt_b;
    }
    GIM_TT(bneck_tail, w, 2);
    GIM_TT_SET(bneck_tail, w, 4, tt_wait); GIM_TT_SET(bneck_tail, w, 5, tt_c3); GIM_TT_SET(bneck_tail, w, 6, tt_epi); GIM_TT_SET(bneck_tail, w, 7, tt_c1);
    // ---- t1' out: 64 channels per pass through a [32 px][128 B] transposition tile.  The per-wave patch is only 2 KiB when P = 256,
    // so every wave takes 4 KiB of the (now idle) weight ring instead ---------------------------------------------------------------
    __syncthreads();             // all waves are past their last weight reads and no DMA is in flight: the ring is free
    char* opatch = smem + w * 4096;
    const bool relu1 = a.act1 == GIM_ACT_RELU;
#pragma unroll
    for (int h2 = 0; h2 < NF / 2; ++h2) {
#pragma unroll
        for (int ff = 0; ff < 2; ++ff)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x16_t& c = c1[2 * h2 + ff];
                const float lo = relu1 ? 0.f : -INFINITY;
                *(uint2*)(opatch + l31 * 128 + (((4 * ff + rg) ^ (l31 & 7)) << 4) + lh * 8) =
                    make_uint2(cvt_pk_h16(fmaxf(c[rg * 4], lo), fmaxf(c[rg * 4 + 1], lo)), cvt_pk_h16(fmaxf(c[rg * 4 + 2], lo), fmaxf(c[rg * 4 + 3], lo)));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = it * 8 + (lane >> 3), sl = lane & 7;
            const uint4 v = *(const uint4*)(opatch + px * 128 + ((sl ^ (px & 7)) << 4));
            *(uint4*)(a.t1n + (prow0 + px) * N1 + 64 * h2 + sl * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    h16_range_flag(a.health, hbig);
    GIM_TT(bneck_tail, w, 3);
}

template <int P, int N1, int NW = 8, bool DS = false>
int launch_tail(const Args& a, hipStream_t s) {
    typedef Cfg<P, N1, NW, DS> C;
    static GimPerDevice attr;
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)bneck_tail_kernel<P, N1, NW, DS>, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) { gim_set_error("bneck_tail: hipFuncSetAttribute(%d B LDS): %s", C::SMEM, hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
        attr.done();
    }
    hipLaunchKernelGGL((bneck_tail_kernel<P, N1, NW, DS>), dim3((unsigned)(a.M / C::ROWS)), dim3(NW * 64), C::SMEM, s, a);
    return gim_check_launch("bneck_tail");
}

int tail_entry(int P, const void* t2, const void* res, void* x_out, void* t1_next, const void* w3, const void* w1n,
               const float* b3, const float* b1n, int M, int n_next, int act_next, int32_t* health, gim_stream_t stream) {
    GIM_REQUIRE(t2 && res && t1_next && w3 && w1n && b3 && b1n, "bneck_tail: NULL pointer");
    GIM_REQUIRE(act_next == GIM_ACT_RELU || act_next == GIM_ACT_NONE, "bneck_tail: activation of the next conv1 must be relu or none");
    GIM_REQUIRE(M > 0 && M % 256 == 0, "bneck_tail: the pixel row count must be a multiple of 256 (got %d)", M);
    GIM_REQUIRE((int64_t)M * 4 * P * 2 < (int64_t)0xFFFFFFF0ll, "bneck_tail: tensor too large for 32-bit buffer offsets");
    Args a;
    a.t2 = (const unsigned short*)t2; a.res = (const unsigned short*)res; a.xo = (unsigned short*)x_out; a.t1n = (unsigned short*)t1_next;
    a.w3 = (const unsigned short*)w3; a.w1n = (const unsigned short*)w1n; a.b3 = b3; a.b1n = b1n; a.M = M; a.act1 = act_next;
    a.w3_bytes = (unsigned)(4 * P * P * 2); a.w1n_bytes = (unsigned)(n_next * 4 * P * 2); a.health = (int*)health;
    a.xin = nullptr; a.Ho = a.Wo = a.Hin = a.Win = 0;
    if (P == 128) {
        GIM_REQUIRE(n_next == 128 || n_next == 256, "bneck_tail128: n_next must be 128 or 256 (got %d)", n_next);
        return n_next == 128 ? launch_tail<128, 128>(a, (hipStream_t)stream) : launch_tail<128, 256>(a, (hipStream_t)stream);
    }
    GIM_REQUIRE(n_next == 256, "bneck_tail256: n_next must be 256 (got %d)", n_next);
    // 4-wave workgroups (two per CU) when the 256-row tiles are between one and three rounds of the chip (A/B: DESIGN.md section 4, round 3).  A launch of at
    // most one round -- layer 3 of one image chain of the batch-8 benchmark, 150 tiles -- is better off with 8 waves and the third chunk buffer (round 6, two
    // chains, same box: 9.54-9.57 vs 9.57-9.63 ms per step)
    if (M / 256 > 256 && M / 256 < 3 * 256) return launch_tail<256, 256, 4>(a, (hipStream_t)stream);
    return launch_tail<256, 256>(a, (hipStream_t)stream);
}

}  // namespace

extern "C" int GIM_FN(gim_bneck_tail128)(const void* t2, const void* res, void* x_out, void* t1_next, const void* w3, const void* w1n,
                                         const float* b3, const float* b1n, int M, int n_next, int act_next, int32_t* health, gim_stream_t stream) {
    GIM_REQUIRE(x_out, "bneck_tail128: NULL x_out");
    return tail_entry(128, t2, res, x_out, t1_next, w3, w1n, b3, b1n, M, n_next, act_next, health, stream);
}

extern "C" int GIM_FN(gim_bneck_tail128_ds)(const void* t2, const void* x_in, void* x_out, void* t1_next, const void* w3ds, const void* w1n,
                                            const float* b3ds, const float* b1n, int B, int Ho, int Wo, int Hin, int Win, int n_next, int act_next,
                                            int32_t* health, gim_stream_t stream) {
    GIM_REQUIRE(t2 && x_in && x_out && t1_next && w3ds && w1n && b3ds && b1n, "bneck_tail128_ds: NULL pointer");
    GIM_REQUIRE(act_next == GIM_ACT_RELU || act_next == GIM_ACT_NONE, "bneck_tail128_ds: activation of the next conv1 must be relu or none");
    GIM_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && Hin >= 2 * Ho - 1 && Win >= 2 * Wo - 1, "bneck_tail128_ds: the input map must cover the stride-2 samples (%d x %d -> %d x %d)", Hin, Win, Ho, Wo);
    const int64_t M = (int64_t)B * Ho * Wo;
    GIM_REQUIRE(M % 256 == 0, "bneck_tail128_ds: the pixel row count must be a multiple of 256 (got %lld)", (long long)M);
    GIM_REQUIRE(M * 512 * 2 < (int64_t)0xFFFFFFF0ll, "bneck_tail128_ds: tensor too large for 32-bit buffer offsets");
    GIM_REQUIRE(n_next == 128, "bneck_tail128_ds: n_next must be 128 (got %d)", n_next);
    Args a;
    a.t2 = (const unsigned short*)t2; a.res = nullptr; a.xo = (unsigned short*)x_out; a.t1n = (unsigned short*)t1_next;
    a.w3 = (const unsigned short*)w3ds; a.w1n = (const unsigned short*)w1n; a.b3 = b3ds; a.b1n = b1n; a.M = (int)M; a.act1 = act_next;
    a.w3_bytes = (unsigned)(512 * 384 * 2); a.w1n_bytes = (unsigned)(n_next * 512 * 2); a.health = (int*)health;
    a.xin = (const unsigned short*)x_in; a.Ho = Ho; a.Wo = Wo; a.Hin = Hin; a.Win = Win;
    return launch_tail<128, 128, 8, true>(a, (hipStream_t)stream);
}

extern "C" int GIM_FN(gim_bneck_tail256)(const void* t2, const void* res, void* x_out, void* t1_next, const void* w3, const void* w1n,
                                         const float* b3, const float* b1n, int M, int n_next, int act_next, int32_t* health, gim_stream_t stream) {
    return tail_entry(256, t2, res, x_out, t1_next, w3, w1n, b3, b1n, M, n_next, act_next, health, stream);
}

// Tail of a ResNet Bottleneck fused with the head of the next block, planes P = 128 (layer 2 of the gim_loftr backbone,
// networks/loftr/backbone/resnet.py:109-126), for gfx950:
//
//     x'  = relu(bn3(conv3_1x1(t2)) + identity)        P -> 4P          (resnet.py:117-124)
//     t1' = relu(bn1'(conv1'_1x1(x')))                4P -> N1          (the NEXT block's resnet.py:109-111)
//
// Two implicit-GEMM launches become one: x' (the widest tensor of the block, 315 MB at 16 x 120 x 160 x 512) is written once and
// NOT read back for conv1'.  Both convolutions are 1x1, so a "tile" is simply 256 consecutive pixel rows; wave w of the
// 512-thread workgroup owns 32 of them for both products.
//
// The chain stays in registers exactly as in bneck_fused.hip: every MFMA is computed transposed (weights = A operand, pixels = B
// operand), a lane's accumulator quad is 4 consecutive channels of ONE pixel, and two v_cvt_pk per quad turn it into the next
// product's B fragment -- conv1's K axis is packed in that accumulator order (packing.py::pack_bneck_tail).
//
// The weights (conv3 4P x P, conv1' N1 x 4P: 256 KiB at N1 = 128) do not fit LDS next to anything else, so they STREAM: the 4P
// output channels of conv3 are walked in chunks of 64; chunk q needs W3[64q .. 64q+63][:] (16 KiB) and W1'[:, 64q .. 64q+63]
// (16 / 32 KiB), fetched by LDS-DMA into a double buffer one chunk ahead (L2-resident after the first workgroups).  Per chunk and
// wave: 16 MFMAs of conv3, the + identity / relu / store of 64 channels of x', 16 (32) MFMAs of conv1'.
//
// Vector memory operations return IN ORDER: anything issued behind a store cannot be waited for without waiting for that
// store's acknowledgement.  Hence the order inside a chunk -- next chunk's weight DMA and next chunk's identity rows are issued
// BEFORE this chunk's stores of x', and the wait at the top of the next chunk is a counted vmcnt(4) that leaves exactly those
// four stores in flight.
#include "gim_common.h"

namespace {

constexpr int PT = 128;              // planes
constexpr int C4 = 4 * PT;           // 512
constexpr int CH = 64;               // conv3 output channels per chunk
constexpr int NCHUNK = C4 / CH;      // 8
constexpr int ROWS = 256;            // pixel rows per workgroup (8 waves x 32)
constexpr int W3C = CH * PT * 2;     // bytes of a conv3 weight chunk: [64 rows][256 B]
constexpr int PATCH = 4096;          // per-wave transposition patch [32 px][128 B]

template <int N1> struct Cfg {
    static constexpr int W1C = N1 * CH * 2;                  // conv1' weight chunk: [N1 rows][128 B]
    static constexpr int BUF = W3C + W1C;
    static constexpr int NBUF = (3 * BUF + 8 * PATCH <= 160 * 1024) ? 3 : 2;   // weight chunks in LDS: fetched NBUF - 1 chunks ahead
    static constexpr int OFF_PATCH = NBUF * BUF;
    static constexpr int OFF_BIAS = OFF_PATCH + 8 * PATCH;   // b3 [512] then b1' [N1], fp32
    static constexpr int SMEM = OFF_BIAS + (C4 + N1) * 4;
    static constexpr int PIECES = BUF / 1024;                // LDS-DMA instructions per chunk (32 / 48)
    static constexpr int PPW = PIECES / 8;                   // ... per wave
    static_assert(SMEM <= 160 * 1024 && PIECES % 8 == 0, "LDS map");
};

struct Args {
    const unsigned short* t2;    // [M][128]  conv2 output (after bn2 + relu)
    const unsigned short* res;   // [M][512]  identity / downsample branch
    unsigned short* xo;          // [M][512]  x'
    unsigned short* t1n;         // [M][N1]   t1'
    const unsigned short* w3;    // [512][128], K in natural channel order
    const unsigned short* w1n;   // [8 chunks][N1][64], K of a chunk in accumulator order
    const float* b3;             // [512]
    const float* b1n;            // [N1]
    int M;
    int act1;                    // activation of conv1': GIM_ACT_RELU (next block's conv1) or GIM_ACT_NONE
    unsigned w3_bytes, w1n_bytes;
};

typedef __attribute__((address_space(3))) void lds_t;

// One LDS-DMA instruction (64 lanes x 16 B -> 1 KiB at LDS byte address `lds_addr`, lane-linear) through inline asm: hipcc makes the
// first LDS access behind a DMA it can see (__builtin_amdgcn_raw_ptr_buffer_load_lds) wait vmcnt(0) -- it assumes every ds_read /
// ds_write may alias the DMA's destination -- which here would drain the two-chunks-ahead weight stream inside every chunk.
// Invisible to the compiler, the DMA is counted by hand (TAIL_WAIT below).  M0 carries the LDS address and is written in the same
// statement that reads it (cdna_hip_programming.md section 5.7).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const u32x4_t rsrc, unsigned lds_addr, unsigned voff) {
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);   // wave-uniform by construction; the "s" constraint needs it provable
    // s_nop 4: the descriptor SGPRs come out of v_readfirstlane (VALU writes SGPR -> VMEM reads it: 5 wait states, which hipcc
    // cannot insert for an instruction inside an asm string).  Without it the first pieces of a workgroup that starts on a CU with
    // a warm instruction cache fetched through a stale descriptor: wrong weights in SOME tiles of SOME launches once the grid
    // exceeded one workgroup per CU (tests/test_gpu_bneck_tail.py::test_bneck_tail_many_tiles_and_repeatability).
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ u32x4_t make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    u32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

// chunk q -> LDS buffer `buf`: this wave's share of the pieces.  W3 rows are 256 B = 16 slots, swizzled slot ^ (row & 15);
// W1' rows are 128 B = 8 slots, swizzled slot ^ ((row >> 1) & 7).  LDS-DMA is lane-linear, so the swizzle is on the SOURCE slot.
template <int N1>
__device__ __forceinline__ void issue_chunk(const u32x4_t rw3, const u32x4_t rw1, unsigned smem_addr, int buf, int q, int w, int lane) {
    typedef Cfg<N1> C;
    const unsigned base = smem_addr + (unsigned)(buf * C::BUF);
#pragma unroll
    for (int k = 0; k < C::PIECES / 8; ++k) {
        const int pc = w + 8 * k;                          // wave-uniform piece index
        if (pc < W3C / 1024) {
            const int idx = pc * 64 + lane, n = idx >> 4, d = idx & 15;
            const unsigned voff = (unsigned)((q * CH + n) * (PT * 2) + ((d ^ (n & 15)) << 4));
            dma16(rw3, base + (unsigned)(pc * 1024), voff);
        } else {
            const int p1 = pc - W3C / 1024;
            const int idx = p1 * 64 + lane, n = idx >> 3, d = idx & 7;
            const unsigned voff = (unsigned)((q * N1 + n) * (CH * 2) + ((d ^ ((n >> 1) & 7)) << 4));
            dma16(rw1, base + (unsigned)(W3C + p1 * 1024), voff);
        }
    }
}

// The identity rows take the same road as the weights: LDS-DMA straight into this wave's patch (row layout, XOR swizzle on the source
// slot), issued through inline asm and counted by hand.  No vector memory load of the chunk loop has a VGPR destination: a load
// hipcc can see beside LDS-DMA makes it wait vmcnt(0), and a load hidden in inline asm has its destination registers copied
// (v_mov) by the register allocator BEFORE the hand-placed wait whenever the wait sits in more than one branch (measured: garbage
// identity rows in some tiles of launches with more workgroups than CUs).
#define TAIL_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

template <int N1>
__global__ void __launch_bounds__(512, 2) bneck_tail_kernel(const Args a) {
    typedef Cfg<N1> C;
    constexpr int NF = N1 / 32;                            // conv1' output fragments per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t prow0 = (size_t)blockIdx.x * ROWS + w * 32;   // first pixel row of this wave
    char* patch = smem + C::OFF_PATCH + w * PATCH;

    const u32x4_t rw3 = make_rsrc(a.w3, a.w3_bytes), rw1 = make_rsrc(a.w1n, a.w1n_bytes);
    const unsigned smem_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_t*)smem);   // LDS byte address of the dynamic array
    issue_chunk<N1>(rw3, rw1, smem_addr, 0, 0, w, lane);
    // biases -> LDS (a global bias load inside the chunk loop makes the compiler wait vmcnt(0): it would drain the weight DMA)
    float* bias = (float*)(smem + C::OFF_BIAS);
    {
        const int t = threadIdx.x;
        if (t < C4 / 4) *(float4*)(bias + 4 * t) = *(const float4*)(a.b3 + 4 * t);
        else if (t < (C4 + N1) / 4) *(float4*)(bias + 4 * t) = *(const float4*)(a.b1n + 4 * t - C4);
    }
    // conv3's pixel operand: 8 k16 steps, lane (pixel l31, half lh) holds channels 16s + 8lh .. + 7
    bf16x8_t t2[PT / 16];
    {
        const unsigned short* tp = a.t2 + (prow0 + l31) * PT + 8 * lh;
#pragma unroll
        for (int s = 0; s < PT / 16; ++s) t2[s] = *(const bf16x8_t*)(tp + 16 * s);
    }
    // opaque to the optimiser from here on: left alone, hipcc RE-LOADS these 8 fragments from global memory in every chunk
    // (rematerialisation beats 32 live VGPRs in its cost model) and waits vmcnt(0) for them -- draining the weight DMA each time
#pragma unroll
    for (int s = 0; s < PT / 16; ++s) asm volatile("" : "+v"(t2[s]));
    // identity rows of chunk q: 32 px x 128 B = 4 pieces; lane i of piece k -> pixel 8k + (i >> 3), patch slot i & 7 <- source slot
    // (i & 7) ^ (pixel & 7)
    const u32x4_t rres = make_rsrc(a.res, (unsigned)((size_t)a.M * C4 * 2));
    const unsigned patch_addr = smem_addr + (unsigned)(C::OFF_PATCH + w * PATCH);
    const unsigned id_voff = (unsigned)((prow0 + (lane >> 3)) * (C4 * 2)) + (unsigned)((((lane & 7) ^ ((lane >> 3) & 7))) << 4);
    auto issue_identity = [&](const int q) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dma16(rres, patch_addr + (unsigned)(k * 1024), id_voff + (unsigned)(k * 8 * C4 * 2 + q * CH * 2));
    };
    issue_identity(0);
    if constexpr (C::NBUF == 3) issue_chunk<N1>(rw3, rw1, smem_addr, 1, 1, w, lane);   // behind what chunk 0 needs: may stay in flight
    __syncthreads();             // biases are in LDS (the compiler knows nothing of the DMA / identity loads in flight: no drain)

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
        // chunk q's weights (this wave's pieces) and identity rows have landed.  Younger operations that may fly on: the four stores
        // of the previous chunk and, with three buffers, the weight DMA of chunk q + 1 (issued behind the identity DMA of chunk q).
        // The count must be exact -- vmcnt(n) only guarantees that all but the n YOUNGEST operations are complete.
        const bool dma_ahead = C::NBUF == 3 && q + 1 < NCHUNK;
        static_assert(C::PPW == 4 || C::PPW == 6, "the counted waits below spell out PPW and PPW + 4");
        if (q == 0) {
            if (dma_ahead) { if constexpr (C::PPW == 4) TAIL_WAIT(4); else TAIL_WAIT(6); }
            else TAIL_WAIT(0);
        } else {
            if (dma_ahead) { if constexpr (C::PPW == 4) TAIL_WAIT(8); else TAIL_WAIT(10); }
            else TAIL_WAIT(4);
        }
        __syncthreads();          // everybody's pieces are visible, and everybody is done with the buffer of chunk q - 1
        const char* wb3 = smem + (q % C::NBUF) * C::BUF;
        const char* wb1 = wb3 + W3C;

        // ---- conv3, 64 output channels: D[m = channel][n = pixel] over K = 128 ------------------------------------------
        f32x16_t c3[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 bb = *(const float4*)(bias + q * CH + 32 * f + 8 * rg + 4 * lh);
                c3[f][rg * 4] = bb.x; c3[f][rg * 4 + 1] = bb.y; c3[f][rg * 4 + 2] = bb.z; c3[f][rg * 4 + 3] = bb.w;
            }
        }
#pragma unroll
        for (int s = 0; s < PT / 16; ++s) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int n = 32 * f + l31;
                const bf16x8_t wv = *(const bf16x8_t*)(wb3 + n * (PT * 2) + (((2 * s + lh) ^ (n & 15)) << 4));
                c3[f] = mfma_h16_32x32x16(wv, t2[s], c3[f]);
            }
        }
        // ---- + identity (DMA'd into the patch in row layout), relu; x' chunk out; operand of conv1' ----------------------------------
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const uint2 r = *(const uint2*)(patch + l31 * 128 + (((4 * f + rg) ^ (l31 & 7)) << 4) + lh * 8);
                c3[f][rg * 4] = fmaxf(c3[f][rg * 4] + h16_lo(r.x), 0.f);
                c3[f][rg * 4 + 1] = fmaxf(c3[f][rg * 4 + 1] + h16_hi(r.x), 0.f);
                c3[f][rg * 4 + 2] = fmaxf(c3[f][rg * 4 + 2] + h16_lo(r.y), 0.f);
                c3[f][rg * 4 + 3] = fmaxf(c3[f][rg * 4 + 3] + h16_hi(r.y), 0.f);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16x8_t xq[4];   // conv1' operand of this chunk: k16 step s = 2f + t (accumulator order)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            unsigned u[8];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                u[2 * rg] = cvt_pk_h16(c3[f][rg * 4], c3[f][rg * 4 + 1]);
                u[2 * rg + 1] = cvt_pk_h16(c3[f][rg * 4 + 2], c3[f][rg * 4 + 3]);
                *(uint2*)(patch + l31 * 128 + (((4 * f + rg) ^ (l31 & 7)) << 4) + lh * 8) = make_uint2(u[2 * rg], u[2 * rg + 1]);
            }
            xq[2 * f] = __builtin_bit_cast(bf16x8_t, make_uint4(u[0], u[1], u[2], u[3]));
            xq[2 * f + 1] = __builtin_bit_cast(bf16x8_t, make_uint4(u[4], u[5], u[6], u[7]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // four named registers, not an array: behind "memory"-clobbering asm statements an array lives in scratch (bneck_fused.hip)
        const int spx = lane >> 3, ssl = lane & 7;
        const char* prd = patch + spx * 128 + ((ssl ^ (spx & 7)) << 4);       // (px + 8k) & 7 == px & 7
        const uint4 x0 = *(const uint4*)(prd), x1 = *(const uint4*)(prd + 8 * 128), x2 = *(const uint4*)(prd + 16 * 128), x3 = *(const uint4*)(prd + 24 * 128);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is free: the next chunk's identity rows may land in it
        // in THIS order: identity of chunk q + 1, weights of chunk q + NBUF - 1 (into the buffer chunk q - 1 used, free since this
        // chunk's barrier), and only then this chunk's stores -- nothing the next chunks wait for sits behind a store
        if (q + 1 < NCHUNK) issue_identity(q + 1);
        if (q + C::NBUF - 1 < NCHUNK) issue_chunk<N1>(rw3, rw1, smem_addr, (q + C::NBUF - 1) % C::NBUF, q + C::NBUF - 1, w, lane);
        {
            unsigned short* xp = a.xo + (prow0 + spx) * C4 + CH * q + ssl * 8;
            *(uint4*)(xp) = x0; *(uint4*)(xp + 8 * C4) = x1; *(uint4*)(xp + 16 * C4) = x2; *(uint4*)(xp + 24 * C4) = x3;
        }
        // ---- conv1' of the next block, this chunk's 64 input channels: 4 k16 steps x NF fragments ------------------------------
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int n = 32 * f + l31;
                const bf16x8_t wv = *(const bf16x8_t*)(wb1 + n * (CH * 2) + (((2 * s + lh) ^ ((n >> 1) & 7)) << 4));
                c1[f] = mfma_h16_32x32x16(wv, xq[s], c1[f]);
            }
        }
    }
    // ---- t1' out: 64 channels per pass through the patch ------------------------------------------------------------------------
    const bool relu1 = a.act1 == GIM_ACT_RELU;
#pragma unroll
    for (int h2 = 0; h2 < NF / 2; ++h2) {
#pragma unroll
        for (int ff = 0; ff < 2; ++ff)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x16_t& c = c1[2 * h2 + ff];
                const float lo = relu1 ? 0.f : -INFINITY;
                *(uint2*)(patch + l31 * 128 + (((4 * ff + rg) ^ (l31 & 7)) << 4) + lh * 8) =
                    make_uint2(cvt_pk_h16(fmaxf(c[rg * 4], lo), fmaxf(c[rg * 4 + 1], lo)), cvt_pk_h16(fmaxf(c[rg * 4 + 2], lo), fmaxf(c[rg * 4 + 3], lo)));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = it * 8 + (lane >> 3), sl = lane & 7;
            const uint4 v = *(const uint4*)(patch + px * 128 + ((sl ^ (px & 7)) << 4));
            *(uint4*)(a.t1n + (prow0 + px) * N1 + 64 * h2 + sl * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int N1>
int launch_tail(const Args& a, hipStream_t s) {
    typedef Cfg<N1> C;
    static GimPerDevice attr;
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute((const void*)bneck_tail_kernel<N1>, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) { gim_set_error("bneck_tail: hipFuncSetAttribute(%d B LDS): %s", C::SMEM, hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
        attr.done();
    }
    hipLaunchKernelGGL(bneck_tail_kernel<N1>, dim3((unsigned)(a.M / ROWS)), dim3(512), C::SMEM, s, a);
    return gim_check_launch("bneck_tail");
}

}  // namespace

extern "C" int GIM_FN(gim_bneck_tail128)(const void* t2, const void* res, void* x_out, void* t1_next, const void* w3, const void* w1n,
                                         const float* b3, const float* b1n, int M, int n_next, int act_next, gim_stream_t stream) {
    GIM_REQUIRE(t2 && res && x_out && t1_next && w3 && w1n && b3 && b1n, "bneck_tail128: NULL pointer");
    GIM_REQUIRE(n_next == 128 || n_next == 256, "bneck_tail128: n_next must be 128 or 256 (got %d)", n_next);
    GIM_REQUIRE(act_next == GIM_ACT_RELU || act_next == GIM_ACT_NONE, "bneck_tail128: activation of the next conv1 must be relu or none");
    GIM_REQUIRE(M > 0 && M % ROWS == 0, "bneck_tail128: the pixel row count must be a multiple of %d (got %d)", ROWS, M);
    GIM_REQUIRE((int64_t)M * C4 * 2 < (int64_t)0xFFFFFFF0ll, "bneck_tail128: tensor too large for 32-bit buffer offsets");
    Args a;
    a.t2 = (const unsigned short*)t2; a.res = (const unsigned short*)res; a.xo = (unsigned short*)x_out; a.t1n = (unsigned short*)t1_next;
    a.w3 = (const unsigned short*)w3; a.w1n = (const unsigned short*)w1n; a.b3 = b3; a.b1n = b1n; a.M = M; a.act1 = act_next;
    a.w3_bytes = C4 * PT * 2; a.w1n_bytes = (unsigned)n_next * C4 * 2;
    return n_next == 128 ? launch_tail<128>(a, (hipStream_t)stream) : launch_tail<256>(a, (hipStream_t)stream);
}

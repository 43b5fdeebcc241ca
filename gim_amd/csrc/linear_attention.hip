// LinearAttention core (networks/loftr/submodules/attentions.py:20-47) for gfx950.
//
//   KV[b,h] = sum_s K[b,s,h,:]^T (V[b,s,h,:] / S)        [D x D]      (attentions.py:42-43)
//   Ksum[b,h] = sum_s K[b,s,h,:]                         [D]          (attentions.py:44)
//   out[b,l,h,:] = (Q[b,l,h,:] KV[b,h]) * 1/(Q[b,l,h,:].Ksum[b,h] + eps) * S   (attentions.py:44-45)
//
// q/k arrive already mapped through elu(x)+1 (fused into the projection GEMM epilogue).  The whole
// stage is HBM-bound (reads q,k,v once, writes out once); the D x D state lives in LDS.  Partial
// sums over S chunks are combined in a fixed order (no float atomics -> run-to-run deterministic).
#include "gim_common.h"

namespace {

constexpr int CH = 128;   // rows of S per block in the KV reduction (VALU kernel)
constexpr int CHM = 256;  // ... of the MFMA kernel (fewer partials: the finalize pass is pure latency)

template <int D, bool BF16>
__global__ void __launch_bounds__(256)
la_kv_kernel(const void* __restrict__ k, const void* __restrict__ v, const uint8_t* __restrict__ kv_mask,
             float* __restrict__ part, int S, int H, int ldk, int ldv, int nchunk) {
    __shared__ __attribute__((aligned(16))) float Ks[CH][D];
    __shared__ __attribute__((aligned(16))) float Vs[CH][D];
    const int bh = blockIdx.x, chunk = blockIdx.y;
    const int b = bh / H, h = bh - b * H;
    const int s0 = chunk * CH;
    const int t = threadIdx.x;
    constexpr int Q4 = D / 4;            // float4 per row
    constexpr int RPP = 256 / Q4;        // rows per pass
    const float slen = (float)S;
    for (int r = t / Q4; r < CH; r += RPP) {
        const int s = s0 + r, c4 = (t % Q4) * 4;
        float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
        if (s < S && (!kv_mask || kv_mask[(size_t)b * S + s])) {  // K * kv_mask, values * kv_mask (attentions.py:38-39)
            kk = ElemIO<BF16>::ld4(k, ((size_t)b * S + s) * ldk + h * D + c4);
            vv = ElemIO<BF16>::ld4(v, ((size_t)b * S + s) * ldv + h * D + c4);
            vv.x = vv.x / slen; vv.y = vv.y / slen; vv.z = vv.z / slen; vv.w = vv.w / slen;  // values / v_length
        }
        *(float4*)&Ks[r][c4] = kk;
        *(float4*)&Vs[r][c4] = vv;
    }
    __syncthreads();
    float* out = part + ((size_t)bh * nchunk + chunk) * (D * D + D);
    const int nrow = min(CH, S - s0);
    if (t < D * Q4) {
        const int d = t / Q4, v0 = (t % Q4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < nrow; ++r) {
            const float kd = Ks[r][d];
            const float4 vv = *(const float4*)&Vs[r][v0];
            acc.x = fmaf(kd, vv.x, acc.x); acc.y = fmaf(kd, vv.y, acc.y);
            acc.z = fmaf(kd, vv.z, acc.z); acc.w = fmaf(kd, vv.w, acc.w);
        }
        *(float4*)(out + d * D + v0) = acc;
    }
    if (t < D) {
        float sk = 0.f;
        for (int r = 0; r < nrow; ++r) sk += Ks[r][t];
        out[D * D + t] = sk;
    }
}

// final[bh][:] = sum_chunk part[bh][chunk][:].  Round 5: the first version gave one thread one element and walked the chunks in a loop of
// dependent load -> add steps: 19-38 memory latencies in a row, 16.3 us per launch for 5-10 MB (profiles/r05_kernel_stats.txt; 32 launches per
// forward).  Now a workgroup owns 64 elements, its four waves take the chunks c = wave, wave + 4, ... -- up to eight independent loads in flight
// per thread and round -- and the four partial sums meet in LDS.  The order of the additions is a fixed function of nchunk alone (not of nb), so
// the state of a sequence stays independent of the batch it is computed in (tests/test_gpu_kernels.py::test_linear_attention_state_is_batch_invariant).
__global__ void __launch_bounds__(256)
la_kv_finalize_kernel(const float* __restrict__ part, float* __restrict__ fin, int per, int nchunk, size_t total) {
    __shared__ float red[256];
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t idx = (size_t)blockIdx.x * 64 + el;
    float s = 0.f;
    if (idx < total) {
        const size_t bh = idx / per, e = idx - bh * per;
        const float* p = part + bh * nchunk * per + e;
        for (int c0 = g; c0 < nchunk; c0 += 32) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c0 + 4 * i;
                v[i] = c < nchunk ? p[(size_t)c * per] : 0.f;
            }
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && idx < total) fin[idx] = (red[el] + red[64 + el]) + (red[128 + el] + red[192 + el]);
}

template <int D, bool BF16, bool OUT_BF16>
__global__ void __launch_bounds__(256)
la_apply_kernel(const void* __restrict__ q, const uint8_t* __restrict__ q_mask, const float* __restrict__ kvfin,
                void* __restrict__ out, int L, int S, int H, int ldq, int ldo) {
    extern __shared__ __attribute__((aligned(16))) float kv[];  // [H][D*D + D]
    const int b = blockIdx.x, l0 = blockIdx.y * 64;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int PER = D * D + D;
    for (int i = t; i < H * PER; i += 256) kv[i] = kvfin[(size_t)b * H * PER + i];
    __syncthreads();
    const int l = l0 + lane;
    if (l >= L) return;
    const float slen = (float)S;
    const bool qvalid = !q_mask || q_mask[(size_t)b * L + l];  // Q * q_mask (attentions.py:36): masked row -> Q = 0
    for (int h = wave; h < H; h += 4) {
        float qv[D];
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            float4 x = ElemIO<BF16>::ld4(q, ((size_t)b * L + l) * ldq + h * D + c);
            if (!qvalid) x = make_float4(0.f, 0.f, 0.f, 0.f);
            qv[c] = x.x; qv[c + 1] = x.y; qv[c + 2] = x.z; qv[c + 3] = x.w;
        }
        const float* KV = kv + h * PER;
        float z = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) z = fmaf(qv[d], KV[D * D + d], z);
        const float Z = 1.0f / (z + 1e-6f);
#pragma unroll
        for (int v0 = 0; v0 < D; v0 += 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float4 w = *(const float4*)(KV + d * D + v0);
                acc.x = fmaf(qv[d], w.x, acc.x); acc.y = fmaf(qv[d], w.y, acc.y);
                acc.z = fmaf(qv[d], w.z, acc.z); acc.w = fmaf(qv[d], w.w, acc.w);
            }
            acc.x = acc.x * Z * slen; acc.y = acc.y * Z * slen; acc.z = acc.z * Z * slen; acc.w = acc.w * Z * slen;
            ElemIO<OUT_BF16>::st4(out, ((size_t)b * L + l) * ldo + h * D + v0, acc);
        }
    }
}

// Short sequences (the fine level: 25-token windows, tens of thousands of sequences): one wave per sequence does
// the whole LinearAttention -- KV/Ksum reduction and the apply step -- with the D x D state exchanged through a
// private LDS slice; no workspace, no second launch.  Lane = (head h = lane/8, j = lane%8): in the reduction it
// owns rows d = j*D/8 .. of KV_h, in the apply step columns v = j*D/8 .. of the output.  H must be 8.
template <int D, bool BF16, bool OUT_BF16>
__global__ void __launch_bounds__(256)
la_short_kernel(const void* __restrict__ q, const void* __restrict__ k, const void* __restrict__ v,
                const uint8_t* __restrict__ q_mask, const uint8_t* __restrict__ kv_mask, void* __restrict__ out,
                int nb, int L, int S, int ldq, int ldk, int ldv, int ldo) {
    constexpr int H = 8, PL = D / 8, PER = D * D + D;  // PL: KV rows (reduction) / output columns (apply) per lane
    __shared__ __attribute__((aligned(16))) float kvs[4][H][PER];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= nb) return;
    const int h = lane >> 3, j = lane & 7;
    const float slen = (float)S;
    // ---- KV_h[d][:] = sum_s K[s,h,d] * V[s,h,:]/S ,  Ksum_h[d] = sum_s K[s,h,d]     (attentions.py:41-44) ----
    float acc[PL][D], ks[PL];
#pragma unroll
    for (int x = 0; x < PL; ++x) {
        ks[x] = 0.f;
#pragma unroll
        for (int y = 0; y < D; ++y) acc[x][y] = 0.f;
    }
    for (int s = 0; s < S; ++s) {
        if (kv_mask && !kv_mask[(size_t)b * S + s]) continue;
        const size_t kr = ((size_t)b * S + s) * ldk + h * D, vr = ((size_t)b * S + s) * ldv + h * D;
        float kk[PL], vv[D];
#pragma unroll
        for (int x = 0; x < PL; ++x) kk[x] = ElemIO<BF16>::ld(k, kr + j * PL + x);
#pragma unroll
        for (int y = 0; y < D; y += 4) {
            const float4 t = ElemIO<BF16>::ld4(v, vr + y);
            vv[y] = t.x / slen; vv[y + 1] = t.y / slen; vv[y + 2] = t.z / slen; vv[y + 3] = t.w / slen;
        }
#pragma unroll
        for (int x = 0; x < PL; ++x) {
            ks[x] += kk[x];
#pragma unroll
            for (int y = 0; y < D; ++y) acc[x][y] = fmaf(kk[x], vv[y], acc[x][y]);
        }
    }
    float* KV = &kvs[wave][h][0];
#pragma unroll
    for (int x = 0; x < PL; ++x) {
        KV[D * D + j * PL + x] = ks[x];
#pragma unroll
        for (int y = 0; y < D; y += 4)
            *(float4*)(KV + (j * PL + x) * D + y) = make_float4(acc[x][y], acc[x][y + 1], acc[x][y + 2], acc[x][y + 3]);
    }
    // LDS operations of one wave execute in order: the reads below see the writes above (all lanes of this head's
    // 8-lane group belong to this wave); only the data dependency needs a wait.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float kvc[D][PL], ksum[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ksum[d] = KV[D * D + d];
#pragma unroll
        for (int e = 0; e < PL; ++e) kvc[d][e] = KV[d * D + j * PL + e];
    }
    // ---- out[l,h,:] = (Q[l,h,:] KV_h) / (Q[l,h,:].Ksum_h + eps) * S                  (attentions.py:44-45) ----
    for (int l = 0; l < L; ++l) {
        const bool qvalid = !q_mask || q_mask[(size_t)b * L + l];
        const size_t qr = ((size_t)b * L + l) * ldq + h * D;
        float qv[D];
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            float4 t = ElemIO<BF16>::ld4(q, qr + d);
            if (!qvalid) t = make_float4(0.f, 0.f, 0.f, 0.f);
            qv[d] = t.x; qv[d + 1] = t.y; qv[d + 2] = t.z; qv[d + 3] = t.w;
        }
        float z = 0.f, o[PL];
#pragma unroll
        for (int e = 0; e < PL; ++e) o[e] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            z = fmaf(qv[d], ksum[d], z);
#pragma unroll
            for (int e = 0; e < PL; ++e) o[e] = fmaf(qv[d], kvc[d][e], o[e]);
        }
        const float Z = 1.0f / (z + 1e-6f);
        const size_t oo = ((size_t)b * L + l) * ldo + h * D + j * PL;
#pragma unroll
        for (int e = 0; e < PL; ++e) ElemIO<OUT_BF16>::st(out, oo + e, o[e] * Z * slen);
    }
}


// ---- MFMA versions for the coarse level (D = 32, H = 8) -----------------------------------------------------
// Counters / arithmetic: the VALU kernels above fetch one LDS operand per 4 FMAs and ran at 1.1 TB/s of HBM
// traffic (la_kv 35 us, la_apply 35 us per layer call at M = 38 400) although both are [rows x 32] x [32 x 32]
// contractions.  v_mfma_f32_32x32x2_f32 (exact fp32, serves the parity mode too) does them at one LDS / register
// operand per 2048 flops, which leaves both kernels HBM-bound.
//
//   la_kv_mfma2:   KV_h[d][dv] = sum_s K[s][h,d] * V[s][h,dv]/S : A = K (m = d), B = V (n = dv), k = s -- a wave's row-major
//                  head slice in LDS feeds both operands without any transposition (lane = channel).
//   la_apply_mfma: out[row][dv] = sum_d Q[row][h,d] KV_h[d][dv]   : A = Q straight from global (lane = row; the k
//                  order is permuted to d = 16*(lane>>5) + step so that every lane reads 16 contiguous channels),
//                  B = KV_h from LDS; the row normaliser z = Q.Ksum is a 16-term dot product per lane, moved to the
//                  accumulator layout with one shuffle per register.
// KV reduction for fp32 operands (the parity mode).  Round 4: the first MFMA kernel ran 304 / 608 workgroups per coarse-level call, each a
// serial chain of four (global load -> barrier -> LDS -> barrier -> 32 MFMAs) steps: ~1.2 - 2.4 workgroups per CU cannot hide a chain
// of four memory round trips.  Here
//   * 8 waves per workgroup = (head, half of the 256-row chunk): twice the waves, half the MFMA chain per wave;
//   * every wave streams ITS OWN head slice (128 B per row) of its 128 rows: ALL of its global loads are issued before the
//     first one is consumed (one memory round trip per wave instead of four);
//   * the slices go through a wave-private LDS region (lane = (row, 16-byte piece) on the way in, lane = channel on the way out: the
//     fp32 MFMA wants one scalar per lane and row) -- no workgroup barrier in the loop, a wave's own LDS accesses execute in order;
//   * the two halves of a head are added in a fixed order through LDS at the end (lower half + upper half): deterministic.
// Arithmetic per row: K exact, V / S in fp32, fp32 MFMA (attentions.py:42-43).
template <bool BF16, int CHK>   // CHK rows per workgroup (256 is what the dispatch below instantiates, for fp32 operands)
__global__ void __launch_bounds__(512)
la_kv_mfma2_kernel(const void* __restrict__ k, const void* __restrict__ v, const uint8_t* __restrict__ kv_mask,
                   float* __restrict__ part, int S, int ldk, int ldv, int nchunk) {
    constexpr int D = 32, H = 8, HG = 4, ES = BF16 ? 2 : 4, PER = D * D + D;
    constexpr int SEG = D * ES;                   // bytes of one head's slice of a row (64 / 128)
    constexpr int STROWS = 64, NST = CHK / 2 / STROWS;   // a wave's CHK / 2 rows go through its LDS region in stages of 64
    constexpr int PPR = SEG / 16, RPI = 64 / PPR, LPS = STROWS / RPI;   // 16-byte pieces per row, rows per wave load, loads per stage and operand
    constexpr int WREG = 2 * STROWS * SEG;        // a wave's LDS region: K slice, V slice (8 / 16 KiB)
    static_assert(17 * 64 * 4 <= WREG, "the combine tile (16 accumulator registers + Ksum, per lane) lives in the upper wave's region");
    extern __shared__ __attribute__((aligned(16))) char la_smem[];
    const int b = blockIdx.x, chunk = blockIdx.y, hg = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, lh = lane >> 5;
    const int hf = wave >> 2, h = hg * HG + (wave & 3);
    const int s0 = chunk * CHK + hf * (CHK / 2);
    char* Kw = la_smem + wave * WREG;
    char* Vw = Kw + STROWS * SEG;
    const float slen = (float)S, inv_s = 1.0f / slen;
    const int prow = lane / PPR, ppc = lane % PPR;
    const char* kb = (const char*)k + (size_t)h * SEG + ppc * 16;
    const char* vb = (const char*)v + (size_t)h * SEG + ppc * 16;
    // which of this lane's rows hold data: inside the sequence and not masked (K * kv_mask, values * kv_mask, attentions.py:38-39).
    // All mask bytes are requested before the first K / V load (a mask byte in front of every load would drain the queue each time).
    unsigned okm = 0u;
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int i = 0; i < LPS; ++i)
            okm |= (s0 + st * STROWS + i * RPI + prow < S ? 1u : 0u) << (st * LPS + i);
    if (kv_mask) {
        unsigned mm = 0u;
#pragma unroll
        for (int st = 0; st < NST; ++st)
#pragma unroll
            for (int i = 0; i < LPS; ++i)
                mm |= (kv_mask[(size_t)b * S + min(s0 + st * STROWS + i * RPI + prow, S - 1)] ? 1u : 0u) << (st * LPS + i);
        okm &= mm;
    }
    // straight-line loads: a row without data reads a clamped address and is zeroed on its way into LDS (behind a branch around
    // every load pair the register allocator parks loaded values in other registers -- and waits for them in the middle of the sequence)
    uint4 rk[NST][LPS], rv[NST][LPS];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const size_t srow = (size_t)b * S + min(s0 + st * STROWS + i * RPI + prow, S - 1);
            rk[st][i] = *(const uint4*)(kb + srow * ldk * ES);
            rv[st][i] = *(const uint4*)(vb + srow * ldv * ES);
        }
    __builtin_amdgcn_sched_barrier(0);   // every load is issued before the first one is waited for (left alone, the scheduler waits for
                                         // the first stage's rows in front of the second stage's requests: two round trips)
    f32x16_t acc;
    float ks = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) {   // lane (row, piece) of load i sits at byte i * 1024 + lane * 16 of the [64 rows][SEG] slice
            const unsigned m = ((okm >> (st * LPS + i)) & 1u) ? 0xffffffffu : 0u;
            *(uint4*)(Kw + i * 1024 + lane * 16) = make_uint4(rk[st][i].x & m, rk[st][i].y & m, rk[st][i].z & m, rk[st][i].w & m);
            *(uint4*)(Vw + i * 1024 + lane * 16) = make_uint4(rv[st][i].x & m, rv[st][i].y & m, rv[st][i].z & m, rv[st][i].w & m);
        }
        // (no barrier: the region is this wave's own, and LDS executes a wave's accesses in order)
#pragma unroll 8
        for (int x = 0; x < STROWS / 2; ++x) {
            const int row = 2 * x + lh;
            float a, bv;
            if constexpr (BF16) {
                a = h16_to_f32(*(const unsigned short*)(Kw + row * SEG + l31 * ES));
                bv = h16_to_f32(*(const unsigned short*)(Vw + row * SEG + l31 * ES)) * inv_s;
            } else {
                a = *(const float*)(Kw + row * SEG + l31 * ES);
                bv = *(const float*)(Vw + row * SEG + l31 * ES) / slen;  // values / v_length (attentions.py:42)
            }
            ks += a;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
        }
    }
    // upper half -> LDS (its own region: its reads are done), lower half adds and stores
    float* comb = (float*)(la_smem + (wave | 4) * WREG);
    if (hf == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) comb[r * 64 + lane] = acc[r];
        comb[16 * 64 + lane] = ks;
    }
    __syncthreads();
    if (hf == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += comb[r * 64 + lane];
    ks += comb[16 * 64 + lane];
    float* out = part + ((size_t)(b * H + h) * nchunk + chunk) * PER;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r >> 2) * 8 + lh * 4 + (r & 3)) * D + l31] = acc[r];
    const float kt = ks + __shfl_xor(ks, 32, 64);
    if (lh == 0) out[D * D + l31] = kt;
}

// 16-bit operands.  Measured (profiles/r04_la_kv.txt): the wave-private streaming above takes the call
// from 27.8 to 24.3 us only -- 153 600 v_mfma_f32_32x32x2_f32 of 64 cycles each are 4 waves x 4 096 cycles on the SIMDs of a CU that
// holds two workgroups: the fp32 MFMA (2 rows per instruction) is the chain, not the memory.  16-bit K and V multiply exactly in fp32,
// so the 16-bit MFMA (16 rows per 32-cycle instruction) computes the same sums:
//   KV_h = K_h^T V_h: A = K^T (lane = channel d, 8 consecutive ROWS per lane: read as 8 x ds_read_u16 down a column of the wave's
//   row-major LDS slice), B = V the same way; 1 / S is applied once to the fp32 sums (attentions.py:41-43 divides V first: the two differ
//   in the last bit of the fp32 result); Ksum = K^T 1 comes from a second MFMA against a fragment of ones (every column of that
//   accumulator is Ksum).  Per 16 rows and wave: 16 LDS reads, 8 packs, 2 MFMAs = 64 MFMA cycles instead of 512.
// CHK rows per workgroup: 256, or 512 for the 16-sequence (self-attention) calls so that one round of <= 512 resident workgroups covers
// the launch (a workgroup holds 64 KiB of LDS: two per CU).  The partials are those of 256-ROW chunks in BOTH shapes -- a 256-row partial is
// (sum of its first 128 rows + sum of its last 128 rows) / S: in the 256-row shape the two sums belong to the two wave halves and meet in LDS,
// in the 512-row shape one wave owns a whole 256-row chunk and keeps two accumulator sets -- so a sequence's state is bit-identical whatever
// batch it travels in (round 5; the round-4 kernel summed 256 rows in one accumulator when the batch was large).
template <int CHK>
__global__ void __launch_bounds__(512)
la_kv_h16_kernel(const void* __restrict__ k, const void* __restrict__ v, const uint8_t* __restrict__ kv_mask,
                 float* __restrict__ part, int S, int ldk, int ldv, int nchunk) {
    constexpr int D = 32, H = 8, HG = 4, ES = 2, PER = D * D + D;
    constexpr int SEG = D * ES;                   // 64 B: one head's slice of a row
    constexpr int STROWS = 64, NST = CHK / 2 / STROWS;
    constexpr int RPI = 16, LPS = STROWS / RPI;   // 4 pieces of 16 B per row, 16 rows per wave load, 4 loads per stage and operand
    constexpr int WREG = 2 * STROWS * SEG;        // 8 KiB: K slice, V slice
    static_assert((16 * 64 + 32) * 4 <= WREG, "combine tile");
    extern __shared__ __attribute__((aligned(16))) char la_smem[];
    const int b = blockIdx.x, chunk = blockIdx.y, hg = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, lh = lane >> 5;
    const int hf = wave >> 2, h = hg * HG + (wave & 3);
    const int s0 = chunk * CHK + hf * (CHK / 2);
    char* Kw = la_smem + wave * WREG;
    char* Vw = Kw + STROWS * SEG;
    const int prow = lane >> 2, ppc = lane & 3;
    const char* kb = (const char*)k + (size_t)h * SEG + ppc * 16;
    const char* vb = (const char*)v + (size_t)h * SEG + ppc * 16;
    unsigned okm = 0u;   // which of this lane's rows hold data (inside the sequence, not masked: attentions.py:38-39)
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int i = 0; i < LPS; ++i)
            okm |= (s0 + st * STROWS + i * RPI + prow < S ? 1u : 0u) << (st * LPS + i);
    if (kv_mask) {
        unsigned mm = 0u;
#pragma unroll
        for (int st = 0; st < NST; ++st)
#pragma unroll
            for (int i = 0; i < LPS; ++i)
                mm |= (kv_mask[(size_t)b * S + min(s0 + st * STROWS + i * RPI + prow, S - 1)] ? 1u : 0u) << (st * LPS + i);
        okm &= mm;
    }
    uint4 rk[NST][LPS], rv[NST][LPS];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const size_t srow = (size_t)b * S + min(s0 + st * STROWS + i * RPI + prow, S - 1);
            rk[st][i] = *(const uint4*)(kb + srow * ldk * ES);
            rv[st][i] = *(const uint4*)(vb + srow * ldv * ES);
        }
    __builtin_amdgcn_sched_barrier(0);   // every load is in flight before the first one is waited for
    // acc / aks: rows [0, 128) of this wave's share, acc2 / aks2: rows [128, 256) (CHK = 512 only)
    f32x16_t acc, aks, acc2, aks2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = aks[r] = acc2[r] = aks2[r] = 0.f;
    const unsigned one2 = cvt_pk_h16(1.f, 1.f);
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(one2, one2, one2, one2));
#pragma unroll
    for (int st = 0; st < NST; ++st) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const unsigned m = ((okm >> (st * LPS + i)) & 1u) ? 0xffffffffu : 0u;
            *(uint4*)(Kw + i * 1024 + lane * 16) = make_uint4(rk[st][i].x & m, rk[st][i].y & m, rk[st][i].z & m, rk[st][i].w & m);
            *(uint4*)(Vw + i * 1024 + lane * 16) = make_uint4(rv[st][i].x & m, rv[st][i].y & m, rv[st][i].z & m, rv[st][i].w & m);
        }
        // (no barrier: the region is this wave's own, and LDS executes a wave's accesses in order)
#pragma unroll
        for (int g = 0; g < STROWS / 16; ++g) {
            const char* kc = Kw + (16 * g + 8 * lh) * SEG + l31 * ES;   // column l31, rows 16 g + 8 lh ... + 7
            const char* vc = Vw + (16 * g + 8 * lh) * SEG + l31 * ES;
            unsigned ka[4], va[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ka[q] = (unsigned)*(const unsigned short*)(kc + (2 * q) * SEG) | ((unsigned)*(const unsigned short*)(kc + (2 * q + 1) * SEG) << 16);
                va[q] = (unsigned)*(const unsigned short*)(vc + (2 * q) * SEG) | ((unsigned)*(const unsigned short*)(vc + (2 * q + 1) * SEG) << 16);
            }
            const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, make_uint4(ka[0], ka[1], ka[2], ka[3]));
            const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(va[0], va[1], va[2], va[3]));
            if (CHK == 512 && st >= NST / 2) {   // compile-time: the stage loop is unrolled
                acc2 = mfma_h16_32x32x16(kf, vf, acc2);
                aks2 = mfma_h16_32x32x16(kf, ones, aks2);
            } else {
                acc = mfma_h16_32x32x16(kf, vf, acc);     // D[d][dv] += sum_rows K[row][d] V[row][dv]
                aks = mfma_h16_32x32x16(kf, ones, aks);   // D[d][*]  += sum_rows K[row][d]
            }
        }
    }
    // Accumulator register r of lane (l31, lh) is element [d = (r >> 2) * 8 + lh * 4 + (r & 3)][dv = l31]; Ksum[d] is any column of aks: the
    // lanes with l31 == 0 carry it.
    const float inv_s = 1.0f / (float)S;
    if constexpr (CHK == 512) {
        // this wave owns the 256-row chunk 2 chunk + hf whole: (first 128 rows + last 128 rows) / S, exactly the 256-row shape's sum
        const int vchunk = 2 * chunk + hf;
        if (vchunk >= nchunk) return;             // wave-uniform: the sequence ends before this chunk (S % 512 <= 256)
        float* out = part + ((size_t)(b * H + h) * nchunk + vchunk) * PER;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((r >> 2) * 8 + lh * 4 + (r & 3)) * D + l31] = (acc[r] + acc2[r]) * inv_s;
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) out[D * D + (r >> 2) * 8 + lh * 4 + (r & 3)] = aks[r] + aks2[r];
        }
    } else {
        // upper half -> LDS (its own region: its reads are done), lower half adds, scales and stores
        float* comb = (float*)(la_smem + (wave | 4) * WREG);
        if (hf == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) comb[r * 64 + lane] = acc[r];
            if (l31 == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) comb[16 * 64 + lh * 16 + r] = aks[r];
            }
        }
        __syncthreads();
        if (hf == 1) return;
        float* out = part + ((size_t)(b * H + h) * nchunk + chunk) * PER;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((r >> 2) * 8 + lh * 4 + (r & 3)) * D + l31] = (acc[r] + comb[r * 64 + lane]) * inv_s;
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) out[D * D + (r >> 2) * 8 + lh * 4 + (r & 3)] = aks[r] + comb[16 * 64 + lh * 16 + r];
        }
    }
}

template <bool BF16, bool OUT_BF16>
__global__ void __launch_bounds__(256)
la_apply_mfma_kernel(const void* __restrict__ q, const uint8_t* __restrict__ q_mask, const float* __restrict__ kvfin,
                     void* __restrict__ out, int L, int S, int ldq, int ldo) {
    // workgroup = 64 rows; wave = (32-row half, group of 4 heads).  All four heads' Q rows are requested before the
    // first MFMA chain so that only one HBM round trip is exposed per wave.
    constexpr int D = 32, H = 8, HG = 4, PER = D * D + D;
    __shared__ __attribute__((aligned(16))) float kv[H * PER];
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = t; i < H * PER / 4; i += 256) *(float4*)(kv + i * 4) = *(const float4*)(kvfin + (size_t)b * H * PER + i * 4);
    const int row0 = blockIdx.y * 64 + (wave & 1) * 32, h0 = (wave >> 1) * HG;
    const int row = row0 + l31, rowc = min(row, L - 1);
    const bool qvalid = row < L && (!q_mask || q_mask[(size_t)b * L + rowc]);  // Q * q_mask (attentions.py:36)
    const float slen = (float)S;
    float qv[HG][16];
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
        const size_t qo = ((size_t)b * L + rowc) * ldq + (h0 + hh) * D + lh * 16;
#pragma unroll
        for (int c = 0; c < 16; c += 4) {
            float4 x = ElemIO<BF16>::ld4(q, qo + c);
            if (!qvalid) x = make_float4(0.f, 0.f, 0.f, 0.f);
            qv[hh][c] = x.x; qv[hh][c + 1] = x.y; qv[hh][c + 2] = x.z; qv[hh][c + 3] = x.w;
        }
    }
    __syncthreads();
    if (row0 >= L) return;
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
        const int h = h0 + hh;
        const float* KV = kv + h * PER;
        float z = 0.f;
#pragma unroll
        for (int c = 0; c < 16; c += 4) {
            const float4 kq = *(const float4*)(KV + D * D + lh * 16 + c);
            z = fmaf(qv[hh][c], kq.x, z); z = fmaf(qv[hh][c + 1], kq.y, z);
            z = fmaf(qv[hh][c + 2], kq.z, z); z = fmaf(qv[hh][c + 3], kq.w, z);
        }
        z += __shfl_xor(z, 32, 64);  // every lane of row `l31` now holds Q[row].Ksum
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 16; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[hh][st], KV[(lh * 16 + st) * D + l31], acc, 0, 0, 0);
        // acc[r] = out[row0 + rr][h*32 + l31],  rr = (r/4)*8 + lh*4 + r%4
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r >> 2) * 8 + lh * 4 + (r & 3);
            const float zr = __shfl(z, rr, 64);
            const float Z = __builtin_amdgcn_rcpf(zr + 1e-6f);  // v_rcp_f32, 1 ulp (an IEEE division is ~10 instructions x 64 per wave)
            if (row0 + rr < L) ElemIO<OUT_BF16>::st(out, ((size_t)b * L + row0 + rr) * ldo + h * D + l31, acc[r] * Z * slen);
        }
    }
}

inline int nchunks(int S) { return (S + CH - 1) / CH; }

}  // namespace

extern "C" int64_t GIM_FN(gim_linear_attention_ws_bytes)(int nb, int S, int H, int D) {
    const int64_t per = (int64_t)D * D + D;
    const int64_t nc = nchunks(S);
    return (int64_t)nb * H * per * 4 * (nc > 1 ? nc + 1 : 1);
}

#if !GIM_HALF_KIND
extern "C" int gim_linear_attention_kv_f16(const void* k, const void* v, const uint8_t* kv_mask, float* kv_ws, int nb,
                                       int S, int H, int D, int ldk, int ldv, int dtype, gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_linear_attention_kv)(const void* k, const void* v, const uint8_t* kv_mask, float* kv_ws, int nb,
                                       int S, int H, int D, int ldk, int ldv, int dtype, gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16) return gim_linear_attention_kv_f16(k, v, kv_mask, kv_ws, nb, S, H, D, ldk, ldv, dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(k && v && kv_ws && nb > 0 && S > 0 && H > 0, "linear_attention_kv: bad args");
    GIM_REQUIRE(D == 32 || D == 16, "linear_attention_kv: head dim %d unsupported (16 or 32)", D);
    GIM_REQUIRE(ldk % 4 == 0 && ldv % 4 == 0, "linear_attention_kv: ld alignment");
    hipStream_t s = (hipStream_t)stream;
    const bool mfma_path = D == 32 && H == 8 && (ldk * (dtype == GIM_H16 ? 2 : 4)) % 16 == 0 &&
                           (ldv * (dtype == GIM_H16 ? 2 : 4)) % 16 == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0;
    // coarse level (D = 32, H = 8): la_kv_h16_kernel for 16-bit operands, la_kv_mfma2_kernel for fp32 ones (profiles/r04_la_kv.txt); partials
    // are per 256-row chunk.  16-bit operands: 512 rows per WORKGROUP when 256-row workgroups would need more than one round of 512 resident
    // ones (the 16-sequence calls of the benchmark) -- the partials stay those of 256-row chunks, so the state of a sequence does not depend on nb
    const bool bf = dtype == GIM_H16;
    const bool wg512 = mfma_path && bf && (int64_t)nb * ((S + 255) / 256) * 2 > 512;
    const int nc = mfma_path ? (S + CHM - 1) / CHM : nchunks(S);
    const int per = D * D + D;
    float* fin = kv_ws;
    float* part = nc > 1 ? kv_ws + (size_t)nb * H * per : kv_ws;
    dim3 grid((unsigned)(nb * H), (unsigned)nc);
    if (mfma_path) {
        const dim3 g2((unsigned)nb, (unsigned)(wg512 ? (S + 511) / 512 : nc), 2u);
        if (bf) {
            static GimPerDevice attr3;
            if (attr3.needed()) {
                hipError_t e = hipFuncSetAttribute((const void*)la_kv_h16_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * 64 * 32 * 2);
                if (e == hipSuccess) e = hipFuncSetAttribute((const void*)la_kv_h16_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * 64 * 32 * 2);
                if (e != hipSuccess) { gim_set_error("linear_attention: hipFuncSetAttribute: %s", hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
                attr3.done();
            }
            if (wg512) hipLaunchKernelGGL(la_kv_h16_kernel<512>, g2, dim3(512), 8 * 2 * 64 * 32 * 2, s, k, v, kv_mask, part, S, ldk, ldv, nc);
            else hipLaunchKernelGGL(la_kv_h16_kernel<256>, g2, dim3(512), 8 * 2 * 64 * 32 * 2, s, k, v, kv_mask, part, S, ldk, ldv, nc);
        } else {
            const int smem2 = 8 * 2 * 64 * 32 * 4;
            static GimPerDevice attr2;
            if (attr2.needed()) {
                hipError_t e = hipFuncSetAttribute((const void*)la_kv_mfma2_kernel<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
                if (e != hipSuccess) { gim_set_error("linear_attention: hipFuncSetAttribute: %s", hipGetErrorString(e)); return GIM_ERR_LAUNCH; }
                attr2.done();
            }
            hipLaunchKernelGGL((la_kv_mfma2_kernel<false, 256>), g2, dim3(512), smem2, s, k, v, kv_mask, part, S, ldk, ldv, nc);
        }
    } else if (D == 32) {
        if (bf) hipLaunchKernelGGL((la_kv_kernel<32, true>), grid, dim3(256), 0, s, k, v, kv_mask, part, S, H, ldk, ldv, nc);
        else hipLaunchKernelGGL((la_kv_kernel<32, false>), grid, dim3(256), 0, s, k, v, kv_mask, part, S, H, ldk, ldv, nc);
    } else {
        if (bf) hipLaunchKernelGGL((la_kv_kernel<16, true>), grid, dim3(256), 0, s, k, v, kv_mask, part, S, H, ldk, ldv, nc);
        else hipLaunchKernelGGL((la_kv_kernel<16, false>), grid, dim3(256), 0, s, k, v, kv_mask, part, S, H, ldk, ldv, nc);
    }
    int rc = gim_check_launch("la_kv");
    if (rc != GIM_OK) return rc;
    if (nc > 1) {
        const size_t total = (size_t)nb * H * per;
        hipLaunchKernelGGL(la_kv_finalize_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, part, fin, per, nc, total);
        rc = gim_check_launch("la_kv_finalize");
    }
    return rc;
}

#if !GIM_HALF_KIND
// Second half of gim_linear_attention_kv alone, for partial states written by somebody else (gim_token_mlp_emit's fused KV state: one
// partial per 64-row tile): kv_ws = [state nb x H x (D D + D)][partials nb x H x nchunk x (D D + D)]; state = sum over the chunks.
extern "C" int64_t gim_linear_attention_ws_bytes_chunks(int nb, int H, int D, int nchunk) {
    return (int64_t)nb * H * ((int64_t)D * D + D) * 4 * ((int64_t)nchunk + 1);
}
extern "C" int gim_linear_attention_finalize(float* kv_ws, int nb, int H, int D, int nchunk, gim_stream_t stream) {
    GIM_REQUIRE(kv_ws && nb > 0 && H > 0 && D > 0 && nchunk > 0, "linear_attention_finalize: bad args");
    const int per = D * D + D;
    const size_t total = (size_t)nb * H * per;
    hipLaunchKernelGGL(la_kv_finalize_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, (hipStream_t)stream, kv_ws + total, kv_ws, per, nchunk, total);
    return gim_check_launch("la_kv_finalize");
}
#endif

#if !GIM_HALF_KIND
extern "C" int gim_linear_attention_apply_f16(const void* q, const uint8_t* q_mask, const float* kv_ws, void* out, int nb,
                                          int L, int S, int H, int D, int ldq, int ldo, int dtype, int out_dtype,
                                          gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_linear_attention_apply)(const void* q, const uint8_t* q_mask, const float* kv_ws, void* out, int nb,
                                          int L, int S, int H, int D, int ldq, int ldo, int dtype, int out_dtype,
                                          gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16 || out_dtype == GIM_F16) return gim_linear_attention_apply_f16(q, q_mask, kv_ws, out, nb, L, S, H, D, ldq, ldo, dtype, out_dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(q && kv_ws && out && nb > 0 && L > 0 && S > 0 && H > 0, "linear_attention_apply: bad args");
    GIM_REQUIRE(D == 32 || D == 16, "linear_attention_apply: head dim %d unsupported (16 or 32)", D);
    GIM_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0, "linear_attention_apply: ld alignment");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)nb, (unsigned)((L + 63) / 64));
    const size_t smem = (size_t)H * (D * D + D) * 4;
    GIM_REQUIRE(smem <= 64 * 1024, "linear_attention_apply: H=%d too large", H);
    const bool bf = dtype == GIM_H16, obf = out_dtype == GIM_H16;
    if (D == 32 && H == 8) {  // coarse level: fp32-MFMA kernel, 64 rows per workgroup
        const dim3 g2((unsigned)nb, (unsigned)((L + 63) / 64));
#define LA_APPLY_M(A, B) hipLaunchKernelGGL((la_apply_mfma_kernel<A, B>), g2, dim3(256), 0, s, q, q_mask, kv_ws, out, L, S, ldq, ldo)
        if (bf && obf) LA_APPLY_M(true, true);
        else if (bf) LA_APPLY_M(true, false);
        else if (obf) LA_APPLY_M(false, true);
        else LA_APPLY_M(false, false);
#undef LA_APPLY_M
        return gim_check_launch("la_apply_mfma");
    }
#define LA_APPLY(DD, A, B) hipLaunchKernelGGL((la_apply_kernel<DD, A, B>), grid, dim3(256), smem, s, q, q_mask, kv_ws, out, L, S, H, ldq, ldo)
    if (D == 32) {
        if (bf && obf) LA_APPLY(32, true, true);
        else if (bf) LA_APPLY(32, true, false);
        else if (obf) LA_APPLY(32, false, true);
        else LA_APPLY(32, false, false);
    } else {
        if (bf && obf) LA_APPLY(16, true, true);
        else if (bf) LA_APPLY(16, true, false);
        else if (obf) LA_APPLY(16, false, true);
        else LA_APPLY(16, false, false);
    }
#undef LA_APPLY
    return gim_check_launch("la_apply");
}

#if !GIM_HALF_KIND
extern "C" int gim_linear_attention_short_f16(const void* q, const void* k, const void* v, const uint8_t* q_mask,
                                          const uint8_t* kv_mask, void* out, int nb, int L, int S, int H, int D,
                                          int ldq, int ldk, int ldv, int ldo, int dtype, int out_dtype,
                                          gim_stream_t stream);
#endif
extern "C" int GIM_FN(gim_linear_attention_short)(const void* q, const void* k, const void* v, const uint8_t* q_mask,
                                          const uint8_t* kv_mask, void* out, int nb, int L, int S, int H, int D,
                                          int ldq, int ldk, int ldv, int ldo, int dtype, int out_dtype,
                                          gim_stream_t stream) {
#if !GIM_HALF_KIND
    if (dtype == GIM_F16 || out_dtype == GIM_F16) return gim_linear_attention_short_f16(q, k, v, q_mask, kv_mask, out, nb, L, S, H, D, ldq, ldk, ldv, ldo, dtype, out_dtype, stream);   // the fp16 objects of this file
#endif
    GIM_REQUIRE(q && k && v && out && nb > 0 && L > 0 && S > 0, "linear_attention_short: bad args");
    GIM_REQUIRE(H == 8 && (D == 16 || D == 32), "linear_attention_short: needs H == 8 and D in {16, 32} (got H=%d D=%d)", H, D);
    GIM_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "linear_attention_short: ld alignment");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((nb + 3) / 4));
    const bool bf = dtype == GIM_H16, obf = out_dtype == GIM_H16;
#define LA_SHORT(DD, A, B) hipLaunchKernelGGL((la_short_kernel<DD, A, B>), grid, dim3(256), 0, s, q, k, v, q_mask, kv_mask, out, nb, L, S, ldq, ldk, ldv, ldo)
    if (D == 16) {
        if (bf && obf) LA_SHORT(16, true, true);
        else if (bf) LA_SHORT(16, true, false);
        else if (obf) LA_SHORT(16, false, true);
        else LA_SHORT(16, false, false);
    } else {
        if (bf && obf) LA_SHORT(32, true, true);
        else if (bf) LA_SHORT(32, true, false);
        else if (obf) LA_SHORT(32, false, true);
        else LA_SHORT(32, false, false);
    }
#undef LA_SHORT
    return gim_check_launch("la_short");
}

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

constexpr int CH = 128;  // rows of S per block in the KV reduction

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

// final[bh][:] = sum_chunk part[bh][chunk][:] (ascending chunk order)
__global__ void la_kv_finalize_kernel(const float* __restrict__ part, float* __restrict__ fin, int per, int nchunk,
                                      size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t bh = idx / per, e = idx - bh * per;
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += part[(bh * nchunk + c) * per + e];
    fin[idx] = s;
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

inline int nchunks(int S) { return (S + CH - 1) / CH; }

}  // namespace

extern "C" int64_t gim_linear_attention_ws_bytes(int nb, int S, int H, int D) {
    const int64_t per = (int64_t)D * D + D;
    const int64_t nc = nchunks(S);
    return (int64_t)nb * H * per * 4 * (nc > 1 ? nc + 1 : 1);
}

extern "C" int gim_linear_attention_kv(const void* k, const void* v, const uint8_t* kv_mask, float* kv_ws, int nb,
                                       int S, int H, int D, int ldk, int ldv, int dtype, gim_stream_t stream) {
    GIM_REQUIRE(k && v && kv_ws && nb > 0 && S > 0 && H > 0, "linear_attention_kv: bad args");
    GIM_REQUIRE(D == 32 || D == 16, "linear_attention_kv: head dim %d unsupported (16 or 32)", D);
    GIM_REQUIRE(ldk % 4 == 0 && ldv % 4 == 0, "linear_attention_kv: ld alignment");
    hipStream_t s = (hipStream_t)stream;
    const int nc = nchunks(S);
    const int per = D * D + D;
    float* fin = kv_ws;
    float* part = nc > 1 ? kv_ws + (size_t)nb * H * per : kv_ws;
    dim3 grid((unsigned)(nb * H), (unsigned)nc);
    const bool bf = dtype == GIM_BF16;
    if (D == 32) {
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
        hipLaunchKernelGGL(la_kv_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, fin, per, nc, total);
        rc = gim_check_launch("la_kv_finalize");
    }
    return rc;
}

extern "C" int gim_linear_attention_apply(const void* q, const uint8_t* q_mask, const float* kv_ws, void* out, int nb,
                                          int L, int S, int H, int D, int ldq, int ldo, int dtype, int out_dtype,
                                          gim_stream_t stream) {
    GIM_REQUIRE(q && kv_ws && out && nb > 0 && L > 0 && S > 0 && H > 0, "linear_attention_apply: bad args");
    GIM_REQUIRE(D == 32 || D == 16, "linear_attention_apply: head dim %d unsupported (16 or 32)", D);
    GIM_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0, "linear_attention_apply: ld alignment");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)nb, (unsigned)((L + 63) / 64));
    const size_t smem = (size_t)H * (D * D + D) * 4;
    GIM_REQUIRE(smem <= 64 * 1024, "linear_attention_apply: H=%d too large", H);
    const bool bf = dtype == GIM_BF16, obf = out_dtype == GIM_BF16;
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

extern "C" int gim_linear_attention_short(const void* q, const void* k, const void* v, const uint8_t* q_mask,
                                          const uint8_t* kv_mask, void* out, int nb, int L, int S, int H, int D,
                                          int ldq, int ldk, int ldv, int ldo, int dtype, int out_dtype,
                                          gim_stream_t stream) {
    GIM_REQUIRE(q && k && v && out && nb > 0 && L > 0 && S > 0, "linear_attention_short: bad args");
    GIM_REQUIRE(H == 8 && (D == 16 || D == 32), "linear_attention_short: needs H == 8 and D in {16, 32} (got H=%d D=%d)", H, D);
    GIM_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "linear_attention_short: ld alignment");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((nb + 3) / 4));
    const bool bf = dtype == GIM_BF16, obf = out_dtype == GIM_BF16;
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

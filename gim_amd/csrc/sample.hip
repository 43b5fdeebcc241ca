// Weighted sampling without replacement for RegressionMatcher.sample (networks/dkm/models/dkm.py:603-620, the two
// torch.multinomial(..., replacement=False) draws over up to 1152 x 3072 certainties).
//
// Exponential clocks: r_i = w_i / e_i with e_i ~ Exp(1) i.i.d.; the k largest r_i are a draw of k items without
// replacement with probabilities proportional to w (the construction torch.multinomial itself uses on the GPU).  The
// e_i come from a counter-based hash of (seed, i), so a call is reproducible from its seed; the top-k is a three-pass
// radix select over the float bit patterns (12 + 12 + 8 bits) with workgroup-local LDS histograms -- a handful of
// HBM-speed passes over n keys instead of a full sort (torch: ~250 ms for k = 20 000 of n = 3.5 M).
// The result is a SET (unordered) -- nothing downstream depends on the order of the draws -- and the set itself is a
// function of (weights, seed) only: keys tied with the threshold key are taken by ascending index (tie list + one
// small sorting workgroup), not in atomic arrival order.
#include "gim_common.h"

namespace {

struct WsState {       // device-side state of the radix select
    unsigned prefix;   // bits decided so far (high bits)
    int need;          // items still to take from the current bucket
    int count;         // output cursor
    int ties;          // keys equal to the threshold key seen by ws_compact
};
constexpr int TIE_CAP = 4096;  // tie list capacity (entries beyond it fall back to arrival order)

__device__ __forceinline__ unsigned hash32(unsigned x) {  // murmur3 finaliser
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) ws_keys_kernel(const float* __restrict__ w, unsigned* __restrict__ keys, unsigned* __restrict__ hist,
                                                      int n, unsigned seed) {
    __shared__ unsigned h[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0u;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float wi = w[i];
        unsigned key = 0u;
        if (wi > 0.f) {
            const unsigned r = hash32(hash32((unsigned)i ^ (seed * 0x9e3779b9u)) + seed);
            // 23 random bits + half an ulp: every value is exactly representable and strictly inside (0, 1)
            // ((r >> 8) + 0.5 needs 25 bits at the top end and rounds up to 1.0 -> e = 0 -> key = +inf)
            const float u = ((float)(r >> 9) + 0.5f) * (1.0f / 8388608.0f);
            const float e = -logf(u);
            key = __float_as_uint(wi / e);                                       // > 0: bit order = value order
        }
        keys[i] = key;
        atomicAdd(&h[key >> 20], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// later passes: histogram of the next `bits` bits among keys whose decided high bits match the prefix
__global__ void __launch_bounds__(256) ws_hist_kernel(const unsigned* __restrict__ keys, unsigned* __restrict__ hist, const WsState* __restrict__ st,
                                                      int n, int shift, int bits, int decided_shift) {
    __shared__ unsigned h[4096];
    const int nb = 1 << bits;
    for (int i = threadIdx.x; i < nb; i += 256) h[i] = 0u;
    __syncthreads();
    const unsigned prefix = st->prefix;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned key = keys[i];
        if ((key >> decided_shift) == (prefix >> decided_shift)) atomicAdd(&h[(key >> shift) & (nb - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// one workgroup: walk the histogram from the top, find the bucket that contains the need-th largest key
__global__ void __launch_bounds__(256) ws_pick_kernel(unsigned* __restrict__ hist, WsState* __restrict__ st, int bits, int shift, int first, int k) {
    __shared__ unsigned h[4096];
    const int nb = 1 << bits;
    for (int i = threadIdx.x; i < nb; i += 256) { h[i] = hist[i]; hist[i] = 0u; }   // leave the histogram clean for the next pass
    __syncthreads();
    if (threadIdx.x == 0) {
        int need = first ? k : st->need;
        int b = nb - 1;
        for (; b > 0; --b) {
            if ((int)h[b] >= need) break;
            need -= (int)h[b];
        }
        st->prefix = (first ? 0u : st->prefix) | ((unsigned)b << shift);
        st->need = need;
        if (first) { st->count = 0; st->ties = 0; }
    }
}

__global__ void __launch_bounds__(256) ws_compact_kernel(const unsigned* __restrict__ keys, WsState* __restrict__ st, int64_t* __restrict__ out,
                                                         int* __restrict__ tielist, int n, int k) {
    const unsigned T = st->prefix;   // the k-th largest key; st->need = how many keys equal to T belong to the sample
    const int need = st->need;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned key = keys[i];
        bool take = key > T;
        if (key == T && key != 0u) {
            const int q = atomicAdd(&st->ties, 1);
            if (q < TIE_CAP) tielist[q] = i;      // resolved by ws_ties_kernel (ascending index)
            else take = q - TIE_CAP < need - TIE_CAP;  // list overflow (pathological weights): the list supplies TIE_CAP, the rest in arrival order
        }
        if (take) {
            const int pos = atomicAdd(&st->count, 1);
            if (pos < k) out[pos] = i;
        }
    }
}

// one workgroup: the `need` smallest indices of the tie list join the sample (rank by counting; ties are few)
__global__ void __launch_bounds__(256) ws_ties_kernel(WsState* __restrict__ st, int64_t* __restrict__ out, const int* __restrict__ tielist, int k) {
    __shared__ int tl[TIE_CAP];
    const int nt = min(st->ties, TIE_CAP);
    const int need = min(st->need, nt);
    for (int q = threadIdx.x; q < nt; q += 256) tl[q] = tielist[q];
    __syncthreads();
    const int base = st->count;
    for (int q = threadIdx.x; q < nt; q += 256) {
        const int me = tl[q];
        int rank = 0;
        for (int r = 0; r < nt; ++r) rank += tl[r] < me ? 1 : 0;
        if (rank < need && base + rank < k) out[base + rank] = me;
    }
}

}  // namespace

extern "C" int64_t gim_weighted_sample_ws_bytes(int n) { return (int64_t)n * 4 + 4096 * 4 + 256 + TIE_CAP * 4; }

extern "C" int gim_weighted_sample(const float* w, int64_t* out, void* ws, int n, int k, uint32_t seed, gim_stream_t stream) {
    GIM_REQUIRE(w && out && ws && n > 0 && k > 0 && k <= n, "weighted_sample: bad args (n=%d k=%d)", n, k);
    hipStream_t s = (hipStream_t)stream;
    unsigned* keys = (unsigned*)ws;
    unsigned* hist = keys + n;
    WsState* st = (WsState*)(hist + 4096);
    if (hipMemsetAsync(hist, 0, 4096 * 4 + 256, s) != hipSuccess) return gim_check_launch("weighted_sample memset");
    const int blocks = n < 256 * 1024 ? (n + 255) / 256 : 1024;
    hipLaunchKernelGGL(ws_keys_kernel, dim3(blocks), dim3(256), 0, s, w, keys, hist, n, seed);
    hipLaunchKernelGGL(ws_pick_kernel, dim3(1), dim3(256), 0, s, hist, st, 12, 20, 1, k);
    hipLaunchKernelGGL(ws_hist_kernel, dim3(blocks), dim3(256), 0, s, keys, hist, st, n, 8, 12, 20);
    hipLaunchKernelGGL(ws_pick_kernel, dim3(1), dim3(256), 0, s, hist, st, 12, 8, 0, k);
    hipLaunchKernelGGL(ws_hist_kernel, dim3(blocks), dim3(256), 0, s, keys, hist, st, n, 0, 8, 8);
    hipLaunchKernelGGL(ws_pick_kernel, dim3(1), dim3(256), 0, s, hist, st, 8, 0, 0, k);
    int* tielist = (int*)((char*)(hist + 4096) + 256);
    hipLaunchKernelGGL(ws_compact_kernel, dim3(blocks), dim3(256), 0, s, keys, st, out, tielist, n, k);
    hipLaunchKernelGGL(ws_ties_kernel, dim3(1), dim3(256), 0, s, st, out, tielist, k);
    return gim_check_launch("weighted_sample");
}

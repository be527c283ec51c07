// Labeled priority sampling of the merged candidate list on gfx950: one 256-thread workgroup per query row.
//
// Replaces (paths relative to /root/reference/src/vod_dataloaders/core):
//   _labeled_priority_sampling_2d_ / _1d_   sample.py:245-352
//   _priority_sampling_1d                   sample.py:160-219
//   log_softmax_1d_ / max_1d / _logsumexp_1d numpy_ops.py:162-216
//
// Per row and per label class (positives, then negatives) the reference does: temperature scaling, optional
// support truncation, log-softmax, priority keys `log_p - log(Exp(1) noise)`, top-(k+1) by key, importance
// log-weights `log_p - log(1 - exp(-exp(log_p - tau)))`, self-normalisation.  Here the row (<= 4096 candidates)
// lives in LDS and both order statistics (the support threshold and the top-(k+1) keys) come from one bitonic
// sort of packed 64-bit (value, column) composites each.  Latency-bound, < 1 MB per batch: no MFMA.
//
// Deliberate, documented choices: sums are tree reductions (the reference accumulates sequentially in float32:
// results agree to ~1e-6, tolerance stated in the tests); ties between equal priority keys go to the smaller
// column (numpy's introsort order among equal keys is unspecified).
#include "vodhip_internal.h"

namespace vodhip {

constexpr int SM_THREADS = 256;
typedef unsigned long long u64;

__device__ __forceinline__ float sm_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float sm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float sm_block_max(float v, float* red) {
    v = sm_wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float sm_block_sum(float v, float* red) {
    v = sm_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// order-preserving image of a float; NaN maps to `nan_image`
__device__ __forceinline__ unsigned ord32(float v, unsigned nan_image) {
    if (v != v) return nan_image;
    unsigned u = __float_as_uint(v + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void sm_sort_desc(u64* keys, int P, int tid) {
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (P >> 1); t += SM_THREADS) {
                const int pos = 2 * t - (t & (stride - 1));
                const u64 a = keys[pos], b = keys[pos + stride];
                const bool desc = (pos & size) == 0;
                if ((a < b) == desc) {
                    keys[pos] = b;
                    keys[pos + stride] = a;
                }
            }
        }
    }
    __syncthreads();
}

// In-place log-softmax over `n` LDS floats selected by `member(i)` (numpy_ops.py:198-204 semantics: NaN -> -inf,
// max falls back to 0 when it is -inf).  Non-members are left untouched.  Returns nothing; all threads call it.
template <typename F>
__device__ void sm_log_softmax(float* x, int n, F member, float* red) {
    const int tid = threadIdx.x;
    float mx = -__builtin_inff();
    for (int i = tid; i < n; i += SM_THREADS)
        if (member(i)) {
            float v = x[i];
            if (v != v) v = -__builtin_inff();
            x[i] = v;
            mx = fmaxf(mx, v);
        }
    mx = sm_block_max(mx, red);
    if (__builtin_isinf(mx) && mx < 0) mx = 0.f;
    float se = 0.f;
    for (int i = tid; i < n; i += SM_THREADS)
        if (member(i)) {
            const float v = x[i] - mx;
            x[i] = v;
            se += expf(v);
        }
    se = sm_block_sum(se, red);
    const float lse = logf(se);
    for (int i = tid; i < n; i += SM_THREADS)
        if (member(i)) x[i] = x[i] - lse;
    __syncthreads();
}

__global__ __launch_bounds__(SM_THREADS) void priority_sample_kernel(
    const float* __restrict__ scores, const uint8_t* __restrict__ labels, const float* __restrict__ noise, int width,
    int P, int k_positive, int k_total_in, float temperature, int max_support, int normalized,
    int64_t* __restrict__ out_samples, float* __restrict__ out_logw, uint8_t* __restrict__ out_labels,
    float* __restrict__ out_lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = (u64*)smem;                  // [P] sort buffer
    float* lp = (float*)(keys + P);          // [width] log-probabilities of the current class
    float* wsel = lp + width;                // [k_total] weights of the selected samples
    int* isel = (int*)(wsel + k_total_in);   // [k_total] selected columns
    float* red = (float*)(isel + k_total_in);  // [4]

    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    const float* sc = scores + row * width;
    const uint8_t* lb = labels + row * width;
    const float* nz = noise + row * width;
    const int n = width;

    // class sizes and finite-negative count (sample.py:259-264)
    int c_pos = 0, c_negfin = 0;
    for (int i = tid; i < n; i += SM_THREADS) {
        const bool pos = lb[i] != 0;
        c_pos += pos;
        c_negfin += (!pos && !__builtin_isinf(sc[i]));
    }
    const int m_pos = (int)sm_block_sum((float)c_pos, red);
    const int n_neg_finite = (int)sm_block_sum((float)c_negfin, red);
    const int m_neg = n - m_pos;
    const int k_total = k_total_in > n ? n : k_total_in;  // :268
    int k_pos = k_positive;
    if (n_neg_finite < k_total - k_pos) k_pos = k_total - n_neg_finite;  // :276-277

    const float t_inv = temperature > 0.f ? temperature : 1.0f;  // :171 (the reference multiplies by it)
    int out_cursor = 0;
    int n_pos_selected = 0;
    for (int cls = 0; cls < 2; ++cls) {
        const bool want_pos = cls == 0;
        const int m = want_pos ? m_pos : m_neg;
        const int k = want_pos ? k_pos : k_total - n_pos_selected;
        auto member = [&](int i) { return (lb[i] != 0) == want_pos; };
        // log_p = scores * T
        for (int i = tid; i < n; i += SM_THREADS) lp[i] = member(i) ? sc[i] * t_inv : -__builtin_inff();
        __syncthreads();
        // support truncation: mask everything >= the max_support-th largest value (Q8), NaN sorts as the largest
        if (max_support > 0 && m > max_support) {
            for (int i = tid; i < P; i += SM_THREADS)
                keys[i] = (i < n && member(i)) ? (((u64)ord32(lp[i], 0xFFFFFFFFu) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
            sm_sort_desc(keys, P, tid);
            const unsigned col = 0xFFFFFFFFu - (unsigned)(keys[max_support - 1] & 0xFFFFFFFFull);
            const float thr = lp[col];
            __syncthreads();
            for (int i = tid; i < n; i += SM_THREADS)
                if (member(i) && lp[i] >= thr) lp[i] = -__builtin_inff();  // false for a NaN threshold
            __syncthreads();
        }
        sm_log_softmax(lp, n, member, red);
        // normalising constant log(sum(exp(log_p)))  (:183)
        float se = 0.f;
        for (int i = tid; i < n; i += SM_THREADS)
            if (member(i)) se += expf(lp[i]);
        se = sm_block_sum(se, red);
        const float log_norm = logf(se);
        if (tid == 0) out_lse[row * 2 + cls] = log_norm;
        // priority keys, sorted descending; NaN keys rank below every other member, non-members below those
        for (int i = tid; i < P; i += SM_THREADS) {
            u64 kv = 0ull;
            if (i < n && member(i)) {
                const float key = temperature > 0.f ? lp[i] - logf(nz[i]) : lp[i];
                kv = ((u64)ord32(key, 1u) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
                if ((kv >> 32) == 0) kv |= (1ull << 32);  // keep members above the non-member image 0
            }
            keys[i] = kv;
        }
        sm_sort_desc(keys, P, tid);
        const int n_sel = k < m ? (k < 0 ? 0 : k) : m;
        float log_tau = -__builtin_inff();
        if (k >= 0 && k < m) {
            const unsigned col = 0xFFFFFFFFu - (unsigned)(keys[k] & 0xFFFFFFFFull);
            log_tau = temperature > 0.f ? lp[col] - logf(nz[col]) : lp[col];
        }
        for (int j = tid; j < n_sel; j += SM_THREADS) {
            const unsigned col = 0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull);
            const float log_pi = lp[col];
            float w = log_pi;
            if (log_tau > -__builtin_inff()) w = log_pi - log1pf(-expf(-expf(log_pi - log_tau)));  // :209-213
            isel[j] = (int)col;
            wsel[j] = w;
        }
        __syncthreads();
        if (normalized && n_sel > 0) sm_log_softmax(wsel, n_sel, [](int) { return true; }, red);
        for (int j = tid; j < n_sel; j += SM_THREADS) {
            out_samples[row * k_total_in + out_cursor + j] = isel[j];
            out_logw[row * k_total_in + out_cursor + j] = wsel[j];
            out_labels[row * k_total_in + out_cursor + j] = want_pos ? 1 : 0;
        }
        out_cursor += n_sel;
        if (want_pos) n_pos_selected = n_sel;
        __syncthreads();
    }
    for (int j = out_cursor + tid; j < k_total_in; j += SM_THREADS) {
        out_samples[row * k_total_in + j] = -1;
        out_logw[row * k_total_in + j] = -__builtin_inff();
        out_labels[row * k_total_in + j] = 0;
    }
}

hipError_t launch_priority_sample(const float* scores, const uint8_t* labels, const float* noise, int64_t nq, int width,
                                  int k_positive, int k_total, float temperature, int max_support_size, int normalized,
                                  int64_t* out_samples, float* out_log_weights, uint8_t* out_labels, float* out_lse,
                                  hipStream_t stream) {
    int P = 64;
    while (P < width) P <<= 1;
    const size_t lds = (size_t)P * 8 + (size_t)width * 4 + (size_t)k_total * 8 + 16 + 16;
    hipLaunchKernelGGL(priority_sample_kernel, dim3((unsigned)nq), dim3(SM_THREADS), lds, stream, scores, labels, noise,
                       width, P, k_positive, k_total, temperature, max_support_size, normalized, out_samples,
                       out_log_weights, out_labels, out_lse);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// gather by id: the flattening of the sampled sections into one in-batch section set
// ------------------------------------------------------------------------------------------------
// Replaces gather_values_by_indices / _nopy_gather_values_{1d,2d} (numpy_ops.py:24-143) as used by flatten_samples
// (in_batch_negatives.py:10-52): for every row b and every query id u of a list shared by all rows, the value at
// the FIRST position j with keys[b, j] == u, or the fill value.  One workgroup per row; the row's keys and up to 8
// value arrays are read from LDS; integer compares, a few KB per row - latency-bound, no MFMA.
struct GatherArgs {
    const float* values[8];
    float* outs[8];
    float fill[8];
};

__global__ __launch_bounds__(256) void gather_by_id_kernel(const int64_t* __restrict__ queries, int n_queries,
                                                           const int64_t* __restrict__ keys, int n_keys, int n_values,
                                                           GatherArgs args) {
    extern __shared__ __attribute__((aligned(16))) char g_smem[];
    int64_t* k_lds = (int64_t*)g_smem;
    const int row = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < n_keys; j += 256) k_lds[j] = keys[(size_t)row * n_keys + j];
    __syncthreads();
    for (int u = tid; u < n_queries; u += 256) {
        const int64_t want = queries[u];
        int pos = -1;
        for (int j = 0; j < n_keys; ++j) {
            if (k_lds[j] == want) {
                pos = j;
                break;
            }
        }
        for (int v = 0; v < n_values; ++v)
            args.outs[v][(size_t)row * n_queries + u] = pos >= 0 ? args.values[v][(size_t)row * n_keys + pos] : args.fill[v];
    }
}

hipError_t launch_gather_by_id(const int64_t* queries, int64_t n_queries, const int64_t* keys, int64_t n_rows, int n_keys,
                               int n_values, const float* const* values, const float* fill, float* const* outs,
                               hipStream_t stream) {
    GatherArgs args{};
    for (int v = 0; v < n_values; ++v) {
        args.values[v] = values[v];
        args.outs[v] = outs[v];
        args.fill[v] = fill[v];
    }
    hipLaunchKernelGGL(gather_by_id_kernel, dim3((unsigned)n_rows), dim3(256), (size_t)n_keys * 8, stream, queries,
                       (int)n_queries, keys, n_keys, n_values, args);
    return hipGetLastError();
}

}  // namespace vodhip

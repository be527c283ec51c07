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
#include "wg_sort.h"

namespace vodhip {

#ifdef VODHIP_ABLATION
__device__ long long g_probe_sample[256];
__device__ long long g_probe_flatten[256];
#endif
#define SM_PROBE(i) VODHIP_PROBE(g_probe_sample, i)
#define FL_PROBE(i) VODHIP_PROBE(g_probe_flatten, i)

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

// In-place log-softmax over `n` LDS floats (numpy_ops.py:198-204 semantics: NaN -> -inf, max falls back to 0 when it is -inf).
// All threads call it.
__device__ __forceinline__ void sm_log_softmax(float* x, int n, float* red) {
    const int tid = threadIdx.x;
    float mx = -__builtin_inff();
    for (int i = tid; i < n; i += SM_THREADS) {
        float v = x[i];
        if (v != v) v = -__builtin_inff();
        x[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = sm_block_max(mx, red);
    if (__builtin_isinf(mx) && mx < 0) mx = 0.f;
    float se = 0.f;
    for (int i = tid; i < n; i += SM_THREADS) {
        const float v = x[i] - mx;
        x[i] = v;
        se += expf(v);
    }
    se = sm_block_sum(se, red);
    const float lse = logf(se);
    for (int i = tid; i < n; i += SM_THREADS) x[i] = x[i] - lse;
    __syncthreads();
}

// Descending sort of the first `m` packed keys of `keys` (LDS; the entries [m, P) are zeroed here: they sort last).  Classes of
// <= 64 members (the positives of a training batch: a handful of gold sections) are sorted by ONE wavefront in registers;
// larger ones by the workgroup network of wg_sort.h.  Returns with the sorted keys visible to every thread.
__device__ __forceinline__ void sm_sort_desc(u64* keys, int m, int tid) {
    if (m <= 64) {
        if (tid < 64) {
            u64 v = tid < m ? keys[tid] : 0ull;
            for (int size = 2; size <= 64; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const u64 o = wg_shfl_xor64(v, stride);
                    const bool keep_max = ((tid & stride) == 0) == ((tid & size) == 0);  // descending overall
                    v = ((o > v) == keep_max) ? o : v;
                }
            keys[tid] = v;
        }
        __syncthreads();
        return;
    }
    int P = 256;
    while (P < m) P <<= 1;
    for (int i = m + tid; i < P; i += SM_THREADS) keys[i] = 0ull;
    __syncthreads();
    (void)wg_sort_lds_256<true, u64>(keys, P, tid);
}

// Arguments of the sampling kernel (by value).  Round 3: the inputs may be the full-stride outputs of `merge_hybrid_kernel`
// with the width still ON THE DEVICE (`merge_width`), and an epilogue gathers the sampled ids / scores / raw engine scores
// and the rank diagnostic of `sample_search_results` (sample.py:52-70) - the collate pipeline never visits the host.
struct SampleArgs {
    const float* scores;
    const void* labels;          // uint8 (non-zero = positive) or int64 (> 0 = positive)
    int labels_i64;
    const float* noise;
    int64_t in_stride;           // row stride (elements) of scores / labels / ids / raw
    int64_t noise_stride;
    int width;                   // >= 0: columns in use; < 0: derive it from merge_width (the reference's cut, merge.py:160-162)
    const int* merge_width;      // device int32 [n_engines]: `out_width` of vodhip_merge_hybrid, or NULL
    const int* row_cursor;       // device int32 [nq, 4]: `out_row_cursor` of vodhip_merge_hybrid (the maximum is taken here), or NULL
    int nq;
    int k_lookup, n_engines, engine_k[4];
    int P_cap;                   // capacity of the LDS sort buffer (power of two >= the largest possible width)
    int k_positive, k_total;
    float temperature;
    int max_support, normalized;
    int keep_top;                // support truncation keeps the `max_support` best entries (the corrected mode) instead of removing them (Q8)
    int64_t* out_samples;
    float* out_logw;
    uint8_t* out_labels;
    float* out_lse;              // lse of class c of row r at out_lse[r * lse_row_stride + c * lse_cls_stride]
    int64_t lse_row_stride, lse_cls_stride;
    // gather epilogue (ids == NULL: none)
    const int64_t* ids;
    int64_t* out_ids;
    float* out_scores;
    int n_raw;
    const float* raw[4];
    float* out_raw[4];
    float* out_max_sampling_id;
};

__global__ __launch_bounds__(SM_THREADS) void priority_sample_kernel(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int k_total_in = a.k_total;
    u64* keys = (u64*)smem;                        // [P_cap] sort buffer
    float* lp = (float*)(keys + a.P_cap);          // [P_cap] log-probabilities of the current class
    float* wsel = lp + a.P_cap;                    // [k_total] weights of the selected samples of the current class
    int* isel = (int*)(wsel + k_total_in);         // [k_total] selected columns of the current class
    int* sel_all = isel + k_total_in;              // [k_total] every selected column of the row (-1 = pad)
    int* order = sel_all + k_total_in;             // [P_cap] the columns of the positives, then of the negatives, ascending
    float* red = (float*)(order + a.P_cap);        // [4]
    int* red_i = (int*)(red + 4);                  // [16]

    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    SM_PROBE(0);
    int n = a.width;
    if (n < 0) {  // the reference's cut (merge.py:160-162): W = k_lookup; for every engine: W = min(max_cursor_e + 1, W + k_e)
        int mx[4] = {0, 0, 0, 0};
        if (a.merge_width) {
#pragma unroll
            for (int e = 0; e < 4; ++e) mx[e] = e < a.n_engines ? a.merge_width[e] : 0;
        } else {  // maximum over the rows of the merge's per-row cursors (no atomics, no cleared buffer on the merge's side)
            const int4* rc = (const int4*)a.row_cursor;  // one 16-byte load per row: all four engines
            for (int r = tid; r < a.nq; r += SM_THREADS) {
                const int4 c = rc[r];
                mx[0] = max(mx[0], c.x), mx[1] = max(mx[1], c.y), mx[2] = max(mx[2], c.z), mx[3] = max(mx[3], c.w);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx[e] = max(mx[e], __shfl_xor(mx[e], o));
            if ((tid & 63) == 0)
#pragma unroll
                for (int e = 0; e < 4; ++e) red_i[(tid >> 6) * 4 + e] = mx[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) mx[e] = max(max(red_i[e], red_i[4 + e]), max(red_i[8 + e], red_i[12 + e]));
            __syncthreads();  // red_i is reused below
        }
        n = a.k_lookup;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < a.n_engines) n = min(mx[e] + 1, n + a.engine_k[e]);
    }
    const float* sc = a.scores + row * a.in_stride;
    const uint8_t* lb8 = a.labels_i64 ? nullptr : (const uint8_t*)a.labels + row * a.in_stride;
    const int64_t* lb64 = a.labels_i64 ? (const int64_t*)a.labels + row * a.in_stride : nullptr;
    const float* nz = a.noise + row * a.noise_stride;
    auto is_pos = [&](int i) { return lb64 ? lb64[i] > 0 : lb8[i] != 0; };
    const float temperature = a.temperature;
    const int max_support = a.max_support;
    int64_t* out_samples = a.out_samples;
    float* out_logw = a.out_logw;
    uint8_t* out_labels = a.out_labels;

    SM_PROBE(1);
    // ---- class sizes, finite-negative count (sample.py:259-264) and an ORDER-PRESERVING compaction of the two classes:
    //      order[0 .. m_pos) = the positive columns ascending, order[m_pos .. n) = the negative columns ascending.  Everything
    //      below works on a class's dense slots: no label is read again, and sums run over the members only, in column order ----
    const int chunk = (n + SM_THREADS - 1) / SM_THREADS;
    const int c_lo = min(n, tid * chunk), c_hi = min(n, c_lo + chunk);
    int c_pos = 0, c_negfin = 0;
    for (int i = c_lo; i < c_hi; ++i) {
        const bool pos = is_pos(i);
        c_pos += pos;
        c_negfin += (!pos && !__builtin_isinf(sc[i]));
    }
    int incl = c_pos;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += v;
    }
    __syncthreads();
    if ((tid & 63) == 63) red_i[tid >> 6] = incl;
    __syncthreads();
    int pos_before = incl - c_pos;
    for (int w = 0; w < (tid >> 6); ++w) pos_before += red_i[w];
    const int m_pos = red_i[0] + red_i[1] + red_i[2] + red_i[3];
    {
        int p_slot = pos_before, n_slot = m_pos + (c_lo - pos_before);
        for (int i = c_lo; i < c_hi; ++i) {
            if (is_pos(i)) order[p_slot++] = i;
            else order[n_slot++] = i;
        }
    }
    const int n_neg_finite = (int)sm_block_sum((float)c_negfin, red);  // (its barriers also publish order[])
    const int m_neg = n - m_pos;
    const int k_total = k_total_in > n ? n : k_total_in;  // :268
    int k_pos = a.k_positive;
    if (n_neg_finite < k_total - k_pos) k_pos = k_total - n_neg_finite;  // :276-277

    SM_PROBE(2);
    const float t_inv = temperature > 0.f ? temperature : 1.0f;  // :171 (the reference multiplies by it)
    int out_cursor = 0;
    int n_pos_selected = 0;
    for (int cls = 0; cls < 2; ++cls) {
        const bool want_pos = cls == 0;
        const int m = want_pos ? m_pos : m_neg;
        const int k = want_pos ? k_pos : k_total - n_pos_selected;
        const int* col_of = order + (want_pos ? 0 : m_pos);  // slot -> column of the merged row
        if (m <= 64 && !(max_support > 0 && m > max_support)) {
            // A class of <= 64 members (the positives of a training batch are a handful of gold sections): ONE wavefront does the
            // whole class in registers - log-softmax, priority keys, sort, weights - with wave shuffles; the other wavefronts wait
            // at a single barrier instead of the ~14 barrier pairs of the general path.
            const int n_sel = k < m ? (k < 0 ? 0 : k) : m;
            if (tid < 64) {
                const bool has = tid < m;
                const int col = has ? col_of[tid] : 0;
                float x = has ? sc[col] * t_inv : -__builtin_inff();
                if (x != x) x = -__builtin_inff();  // numpy_ops.py:198-204: NaN -> -inf
                float mxv = sm_wave_max(x);
                if (__builtin_isinf(mxv) && mxv < 0) mxv = 0.f;
                const float v = x - mxv;
                const float lpv = v - logf(sm_wave_sum(has ? expf(v) : 0.f));
                const float log_norm = logf(sm_wave_sum(has ? expf(lpv) : 0.f));
                if (tid == 0) a.out_lse[row * a.lse_row_stride + cls * a.lse_cls_stride] = log_norm;
                const float key = temperature > 0.f ? lpv - logf(has ? nz[col] : 1.f) : lpv;
                u64 kv = 0ull;
                if (has) {
                    kv = ((u64)ord32(key, 1u) << 32) | (u64)(0xFFFFFFFFu - (unsigned)tid);
                    if ((kv >> 32) == 0) kv |= (1ull << 32);
                }
                for (int size = 2; size <= 64; size <<= 1)
                    for (int stride = size >> 1; stride > 0; stride >>= 1) {
                        const u64 o = wg_shfl_xor64(kv, stride);
                        const bool keep_max = ((tid & stride) == 0) == ((tid & size) == 0);
                        kv = ((o > kv) == keep_max) ? o : kv;
                    }
                // lane j now holds the j-th largest key: fetch that member's log_p, key and column from its original lane
                const int src = (int)(0xFFFFFFFFu - (unsigned)(kv & 0xFFFFFFFFull)) & 63;
                const float s_lp = __shfl(lpv, src), s_key = __shfl(key, src);
                const int s_col = __shfl(col, src);
                const float tau_key = __shfl(s_key, (k >= 0 && k < m) ? k : 0);
                const float log_tau = (k >= 0 && k < m) ? tau_key : -__builtin_inff();
                float w = s_lp;
                if (log_tau > -__builtin_inff()) w = s_lp - log1pf(-expf(-expf(s_lp - log_tau)));  // :209-213
                const bool sel = tid < n_sel;
                if (a.normalized && n_sel > 0) {  // log-softmax of the selected weights
                    float wv = sel ? w : -__builtin_inff();
                    if (wv != wv) wv = -__builtin_inff();
                    float wm = sm_wave_max(wv);
                    if (__builtin_isinf(wm) && wm < 0) wm = 0.f;
                    const float d = wv - wm;
                    w = d - logf(sm_wave_sum(sel ? expf(d) : 0.f));
                }
                if (sel) {
                    out_samples[row * k_total_in + out_cursor + tid] = s_col;
                    out_logw[row * k_total_in + out_cursor + tid] = w;
                    out_labels[row * k_total_in + out_cursor + tid] = want_pos ? 1 : 0;
                    sel_all[out_cursor + tid] = s_col;
                }
            }
            out_cursor += n_sel;
            if (want_pos) n_pos_selected = n_sel;
            __syncthreads();
            SM_PROBE(7 + cls * 5);
            continue;
        }
        // log_p = scores * T
        for (int s_ = tid; s_ < m; s_ += SM_THREADS) lp[s_] = sc[col_of[s_]] * t_inv;
        __syncthreads();
        // support truncation: mask everything >= the max_support-th largest value (Q8: the reference REMOVES its best entries), or -
        // `keep_top`, the corrected mode - everything below it; NaN sorts as the largest; equal values rank by the smaller column
        if (max_support > 0 && m > max_support) {
            for (int s_ = tid; s_ < m; s_ += SM_THREADS) keys[s_] = ((u64)ord32(lp[s_], 0xFFFFFFFFu) << 32) | (u64)(0xFFFFFFFFu - (unsigned)s_);
            __syncthreads();
            sm_sort_desc(keys, m, tid);
            const unsigned slot = 0xFFFFFFFFu - (unsigned)(keys[max_support - 1] & 0xFFFFFFFFull);
            const float thr = lp[slot];
            __syncthreads();
            for (int s_ = tid; s_ < m; s_ += SM_THREADS)
                if (a.keep_top ? lp[s_] < thr : lp[s_] >= thr) lp[s_] = -__builtin_inff();  // false for a NaN threshold
            __syncthreads();
        }
        SM_PROBE(3 + cls * 5);
        sm_log_softmax(lp, m, red);
        SM_PROBE(4 + cls * 5);
        // normalising constant log(sum(exp(log_p)))  (:183)
        float se = 0.f;
        for (int s_ = tid; s_ < m; s_ += SM_THREADS) se += expf(lp[s_]);
        se = sm_block_sum(se, red);
        const float log_norm = logf(se);
        if (tid == 0) a.out_lse[row * a.lse_row_stride + cls * a.lse_cls_stride] = log_norm;
        // priority keys, sorted descending; NaN keys rank below every other member
        for (int s_ = tid; s_ < m; s_ += SM_THREADS) {
            const float key = temperature > 0.f ? lp[s_] - logf(nz[col_of[s_]]) : lp[s_];
            u64 kv = ((u64)ord32(key, 1u) << 32) | (u64)(0xFFFFFFFFu - (unsigned)s_);
            if ((kv >> 32) == 0) kv |= (1ull << 32);  // keep members above the padding image 0
            keys[s_] = kv;
        }
        __syncthreads();
        SM_PROBE(5 + cls * 5);
        sm_sort_desc(keys, m, tid);
        SM_PROBE(6 + cls * 5);
        const int n_sel = k < m ? (k < 0 ? 0 : k) : m;
        float log_tau = -__builtin_inff();
        if (k >= 0 && k < m) {
            const unsigned slot = 0xFFFFFFFFu - (unsigned)(keys[k] & 0xFFFFFFFFull);
            log_tau = temperature > 0.f ? lp[slot] - logf(nz[col_of[slot]]) : lp[slot];
        }
        for (int j = tid; j < n_sel; j += SM_THREADS) {
            const unsigned slot = 0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull);
            const float log_pi = lp[slot];
            float w = log_pi;
            if (log_tau > -__builtin_inff()) w = log_pi - log1pf(-expf(-expf(log_pi - log_tau)));  // :209-213
            isel[j] = col_of[slot];
            wsel[j] = w;
        }
        __syncthreads();
        if (a.normalized && n_sel > 0) sm_log_softmax(wsel, n_sel, red);
        for (int j = tid; j < n_sel; j += SM_THREADS) {
            out_samples[row * k_total_in + out_cursor + j] = isel[j];
            out_logw[row * k_total_in + out_cursor + j] = wsel[j];
            out_labels[row * k_total_in + out_cursor + j] = want_pos ? 1 : 0;
            sel_all[out_cursor + j] = isel[j];
        }
        out_cursor += n_sel;
        if (want_pos) n_pos_selected = n_sel;
        __syncthreads();
        SM_PROBE(7 + cls * 5);
    }
    for (int j = out_cursor + tid; j < k_total_in; j += SM_THREADS) {
        out_samples[row * k_total_in + j] = -1;
        out_logw[row * k_total_in + j] = -__builtin_inff();
        out_labels[row * k_total_in + j] = 0;
        sel_all[j] = -1;
    }
    if (!a.ids) return;
    __syncthreads();
    // ---- epilogue: take_along_axis of ids / scores / raw scores by the sampled columns (-1 wraps to the last column, as
    //      NumPy indexes it: sample.py:52-60) and the rank diagnostic (sample.py:64-70) ----
    float min_neg = __builtin_inff();
    for (int j = tid; j < k_total_in; j += SM_THREADS) {
        int g = sel_all[j];
        if (g < 0) g += n;
        const bool ok = g >= 0 && g < n;  // n = 0: nothing to take (the reference raises there)
        const float s = ok ? sc[g] : -__builtin_inff();
        a.out_ids[row * k_total_in + j] = ok ? a.ids[row * a.in_stride + g] : -1;
        a.out_scores[row * k_total_in + j] = s;
        for (int e = 0; e < a.n_raw; ++e) a.out_raw[e][row * k_total_in + j] = ok ? a.raw[e][row * a.in_stride + g] : __builtin_nanf("");
        const bool neg = j >= n_pos_selected;  // sampled label <= 0 (pads carry label 0)
        if (neg && !(__builtin_isinf(s) || s != s)) min_neg = fminf(min_neg, s);
    }
    min_neg = -sm_block_max(-min_neg, red);
    int larger = 0;
    for (int i = tid; i < n; i += SM_THREADS) {
        const float s = sc[i];
        larger += (!is_pos(i) && !(__builtin_isinf(s) || s != s) && s >= min_neg);
    }
    const float total = sm_block_sum((float)larger, red);
    if (tid == 0 && a.out_max_sampling_id) a.out_max_sampling_id[row] = total;
    SM_PROBE(13);
}

static hipError_t launch_sample_args(SampleArgs& a, int64_t nq, int max_width, hipStream_t stream) {
    // more positives than samples cannot be drawn (the reference's numba loop would write past its [k_total] output row there)
    if (a.k_positive > a.k_total) a.k_positive = a.k_total;
    int P = 256;
    while (P < max_width) P <<= 1;
    a.P_cap = P;
    const size_t lds = (size_t)P * 8 + (size_t)P * 4 + (size_t)a.k_total * 12 + (size_t)P * 4 + 16 + 64 + 16;
    if (lds > 64 * 1024) {
        hipError_t e = allow_dynamic_lds((const void*)priority_sample_kernel, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(priority_sample_kernel, dim3((unsigned)nq), dim3(SM_THREADS), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_priority_sample(const float* scores, const uint8_t* labels, const float* noise, int64_t nq, int width,
                                  int k_positive, int k_total, float temperature, int max_support_size, int normalized,
                                  int64_t* out_samples, float* out_log_weights, uint8_t* out_labels, float* out_lse,
                                  hipStream_t stream) {
    SampleArgs a{};
    a.scores = scores;
    a.labels = labels;
    a.noise = noise;
    a.in_stride = a.noise_stride = width;
    a.width = width;
    a.k_positive = k_positive;
    a.k_total = k_total;
    a.temperature = temperature;
    a.max_support = max_support_size;
    a.normalized = normalized & 1;
    a.keep_top = (normalized >> 1) & 1;
    a.out_samples = out_samples;
    a.out_logw = out_log_weights;
    a.out_labels = out_labels;
    a.out_lse = out_lse;
    a.lse_row_stride = 2;
    a.lse_cls_stride = 1;
    return launch_sample_args(a, nq, width, stream);
}

hipError_t launch_priority_sample_merged(const SampleMergedArgs& m, hipStream_t stream) {
    SampleArgs a{};
    a.scores = m.scores;
    a.labels = m.labels;
    a.labels_i64 = 1;
    a.noise = m.noise;
    a.in_stride = m.stride;
    a.noise_stride = m.noise_stride;
    a.width = m.width;
    a.merge_width = m.merge_width;
    a.row_cursor = m.row_cursor;
    a.nq = (int)m.nq;
    a.k_lookup = m.k_lookup;
    a.n_engines = m.n_engines;
    for (int e = 0; e < 4; ++e) a.engine_k[e] = m.engine_k[e];
    a.k_positive = m.k_positive;
    a.k_total = m.k_total;
    a.temperature = m.temperature;
    a.max_support = m.max_support;
    a.normalized = m.normalized & 1;
    a.keep_top = (m.normalized >> 1) & 1;
    a.out_samples = m.out_samples;
    a.out_logw = m.out_logw;
    a.out_labels = m.out_labels;
    a.out_lse = m.out_lse;
    a.lse_row_stride = m.lse_row_stride;
    a.lse_cls_stride = m.lse_cls_stride;
    a.ids = m.ids;
    a.out_ids = m.out_ids;
    a.out_scores = m.out_scores;
    a.n_raw = m.n_raw;
    for (int e = 0; e < 4; ++e) {
        a.raw[e] = m.raw[e];
        a.out_raw[e] = m.out_raw[e];
    }
    a.out_max_sampling_id = m.out_max_sampling_id;
    return launch_sample_args(a, m.nq, m.width >= 0 ? m.width : (int)m.stride, stream);
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

// ------------------------------------------------------------------------------------------------
// in-batch flattening in ONE launch (round 3): sorted unique ids of the whole batch + the per-row gather
// ------------------------------------------------------------------------------------------------
// Replaces flatten_samples (in_batch_negatives.py:10-52) = np.unique over the B x n sampled ids, padded to B * n entries with the
// reference's constant 1, then gather_values_by_indices of every value array onto that list.  Every
// workgroup (one per row) sorts the U = B * n ids itself in LDS (a bitonic sort of <= 8192 int64: cheaper than a second launch
// and a device-wide dependency), compacts the distinct values, and gathers ITS row; workgroup 0 also writes the id list.
constexpr int FL_THREADS = 1024;  // 16 wavefronts: the sort network is a latency chain per stage, so wide and shallow (2 keys per thread at U = 2048)

__global__ __launch_bounds__(FL_THREADS) void flatten_inbatch_kernel(const int64_t* __restrict__ ids, int n_keys, int U, int P, int n_values,
                                                                     GatherArgs args, const uint8_t* __restrict__ labels,
                                                                     uint8_t* __restrict__ out_labels, int64_t* __restrict__ out_unique,
                                                                     int* __restrict__ out_n_unique) {
    extern __shared__ __attribute__((aligned(16))) char f_smem[];
    int64_t* srt = (int64_t*)f_smem;          // [P] sort buffer, later the row's keys
    int64_t* uq = srt + P;                    // [U] distinct ids ascending, then the padding
    float* vals = (float*)(uq + U);           // [n_values][n_keys] the row's value arrays
    uint8_t* lab = (uint8_t*)(vals + 8 * n_keys);  // [n_keys] the row's labels
    int* s_w = (int*)(lab + ((n_keys + 3) & ~3));  // [16] wave sums
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    FL_PROBE(0);
    for (int i = tid; i < P; i += FL_THREADS) srt[i] = i < U ? ids[i] : 0x7FFFFFFFFFFFFFFFll;
    // the row's own value arrays and labels go to LDS now: the gather below never waits for global memory
    for (int j = tid; j < n_keys * n_values; j += FL_THREADS) {
        const int v = j / n_keys, c = j - v * n_keys;
        vals[v * n_keys + c] = args.values[v][(size_t)row * n_keys + c];
    }
    if (labels)
        for (int j = tid; j < n_keys; j += FL_THREADS) lab[j] = labels[(size_t)row * n_keys + j];
    __syncthreads();
    FL_PROBE(1);
    (void)wg_sort_lds_1024<false, long long>((long long*)srt, P, tid);  // ascending, like np.unique
    FL_PROBE(2);
    const int chunk = (U + FL_THREADS - 1) / FL_THREADS;
    const int lo = min(U, tid * chunk), hi = min(U, lo + chunk);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += (i == 0 || srt[i] != srt[i - 1]);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int run = incl - cnt, n_unique = 0;
#pragma unroll
    for (int w = 0; w < FL_THREADS / 64; ++w) {
        run += w < wave ? s_w[w] : 0;
        n_unique += s_w[w];
    }
    for (int i = lo; i < hi; ++i)
        if (i == 0 || srt[i] != srt[i - 1]) uq[run++] = srt[i];
    __syncthreads();
    for (int i = n_unique + tid; i < U; i += FL_THREADS) uq[i] = 1;  // np.ones padding (Q7)
    for (int j = tid; j < n_keys; j += FL_THREADS) srt[j] = ids[(size_t)row * n_keys + j];
    __syncthreads();
    FL_PROBE(3);
    if (row == 0) {
        for (int i = tid; i < U; i += FL_THREADS) out_unique[i] = uq[i];
        if (tid == 0 && out_n_unique) *out_n_unique = n_unique;
    }
    for (int u = tid; u < U; u += FL_THREADS) {
        const int64_t want = uq[u];
        int pos = -1;
        for (int j = 0; j < n_keys; ++j) {
            if (srt[j] == want) {
                pos = j;
                break;
            }
        }
        for (int v = 0; v < n_values; ++v) args.outs[v][(size_t)row * U + u] = pos >= 0 ? vals[v * n_keys + pos] : args.fill[v];
        if (labels) out_labels[(size_t)row * U + u] = pos >= 0 ? (lab[pos] != 0) : 0;  // fill_value=0
    }
    FL_PROBE(4);
}

hipError_t launch_flatten_inbatch(const int64_t* ids, int64_t n_rows, int n_keys, int n_values, const float* const* values,
                                  const float* fill, float* const* outs, const uint8_t* labels, uint8_t* out_labels,
                                  int64_t* out_unique, int* out_n_unique, hipStream_t stream) {
    GatherArgs args{};
    for (int v = 0; v < n_values; ++v) {
        args.values[v] = values[v];
        args.outs[v] = outs[v];
        args.fill[v] = fill[v];
    }
    const int U = (int)(n_rows * n_keys);
    int P = 1024;
    while (P < U) P <<= 1;
    const size_t lds = (size_t)P * 8 + (size_t)U * 8 + (size_t)n_keys * 8 * 4 + (size_t)((n_keys + 3) & ~3) + 16 * 4 + 16;
    if (lds > 64 * 1024) {
        hipError_t e = allow_dynamic_lds((const void*)flatten_inbatch_kernel, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(flatten_inbatch_kernel, dim3((unsigned)n_rows), dim3(FL_THREADS), lds, stream, ids, n_keys, U, P, n_values, args,
                       labels, out_labels, out_unique, out_n_unique);
    return hipGetLastError();
}

hipError_t read_probe_sample(int which, long long* out) {
#ifdef VODHIP_ABLATION
    return which == 1 ? hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe_sample), sizeof(long long) * 256)
                      : hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe_flatten), sizeof(long long) * 256);
#else
    (void)which;
    (void)out;
    return hipErrorNotSupported;
#endif
}

}  // namespace vodhip

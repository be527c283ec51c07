/* Oracle (test infrastructure, never linked into the product): the reference's numba loops of the collate-side chain restated in
 * plain C - the CPU figure reported beside the C5 kernels (bench.py `side` entry C5, `cpu_baseline`), and a second, independent
 * restatement checked against the same reference-generated fixtures as oracle/hybrid.py / oracle/sampling.py
 * (tests/test_oracle_golden.py).  Built by vod_amd.build.build_oracle(): gcc -O3 -fopenmp.
 *
 * Follows (relative to /root/reference/src/vod_dataloaders/core):
 *   normalize.py:6-20       _subtract_min_score (row minimum over the finite entries)
 *   merge.py:71-105         _search_1d_arr / _write_1d_arr (linear search of the output row, accumulate on a hit)
 *   merge.py:108-164        _nopy_merge_two_search_results (rows in a parallel loop; cut to max cursor + 1 after every fold)
 *   merge.py:31-62          _merge_n_search_results (fold order, raw scores and labels gathered by id)
 *   numpy_ops.py:24-143     gather_values_by_indices (first match wins, NaN / -1 / 0 fill)
 *   numpy_ops.py:162-216    max_1d, _logsumexp_1d, log_softmax_1d_ (sequential float32 accumulation)
 *   sample.py:160-219       _priority_sampling_1d (quirk Q8: the support truncation masks the entries >= the threshold)
 *   sample.py:245-320       _labeled_priority_sampling_1d_ ; :323-352 the row-parallel 2-D driver
 *   sample.py:56-70         take_along_axis of the sampled columns + the rank diagnostic
 *   in_batch_negatives.py:10-52  flatten_samples (np.unique, padding with 1s: quirk Q7, gathers by id)
 * float32 arithmetic throughout (the reference's arrays are float32); NaN / inf follow IEEE as NumPy's do.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_ENGINES 4

#ifdef _OPENMP
#include <omp.h>
void vodref_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); } /* the cgroup's grant, not the cores the host shows */
#else
void vodref_set_threads(int n) { (void)n; }
#endif

static int search_1d(const int64_t* arr, int n, int64_t x) { /* merge.py:71-82 */
    for (int i = 0; i < n; ++i)
        if (arr[i] == x) return i;
    return -1;
}

static int write_1d(int64_t* indices, float* scores, int width, int64_t index, float score, int cursor) { /* merge.py:85-105 */
    if (index < 0) return cursor;
    const int found = search_1d(indices, width, index);
    if (found < 0) {
        scores[cursor] = score;
        indices[cursor] = index;
        return cursor + 1;
    }
    scores[found] = score + scores[found];
    return cursor;
}

static float row_min_finite(const float* s, int n) { /* normalize.py:17-20: inf / NaN entries count as +inf */
    float mn = INFINITY;
    for (int i = 0; i < n; ++i) {
        const float v = (isinf(s[i]) || isnan(s[i])) ? INFINITY : s[i];
        if (v < mn) mn = v;
    }
    return mn;
}

/* core/search.py:79-125 = normalise, weight (lookup weight 0, its scores discarded), fold pairwise, gather raw scores and labels.
 * Outputs are rows of `stride` = k_lookup + sum(e_k) + 1 columns; *out_width receives the reference's final width. */
int vodref_merge_hybrid(const int64_t* l_idx, const int64_t* l_lbl, int k_lookup, int n_engines, const int64_t* const* e_idx,
                        const float* const* e_scr, const int* e_k, const float* e_w, int64_t nq, int64_t* out_idx, float* out_scr,
                        int64_t* out_lbl, float* const* out_raw, int stride, int* out_width) {
    if (n_engines < 1 || n_engines > MAX_ENGINES) return -1;
    int max_cursor[MAX_ENGINES] = {0, 0, 0, 0};
    /* the width of the buffer each fold allocates: a.width + b.width, with a.width cut to max_cursor + 1 after the previous fold.
     * The cut only removes untouched (-1, -inf) columns, so every row can be folded in one pass and the widths derived afterwards;
     * the linear searches run over the fold's allocated width like the reference's. */
    int fold_width[MAX_ENGINES];
    {
        int w = k_lookup;
        for (int e = 0; e < n_engines; ++e) {
            fold_width[e] = w + e_k[e];
            w = fold_width[e]; /* upper bound (the true cut is <= this); searching a few more -1 columns finds nothing more */
        }
    }
#pragma omp parallel
    {
        int local_max[MAX_ENGINES] = {0, 0, 0, 0};
        float* norm = (float*)malloc(sizeof(float) * (size_t)stride);
#pragma omp for schedule(static)
        for (int64_t r = 0; r < nq; ++r) {
            int64_t* oi = out_idx + r * stride;
            float* os = out_scr + r * stride;
            for (int c = 0; c < stride; ++c) {
                oi[c] = -1;
                os[c] = -INFINITY;
            }
            int cursor = 0;
            /* `a` of the first fold = the lookup result x weight 0 (scores were filled with 0 and min-subtracted: 0) */
            for (int j = 0; j < k_lookup; ++j) cursor = write_1d(oi, os, fold_width[0], l_idx[r * k_lookup + j], 0.0f * 0.0f, cursor);
            for (int e = 0; e < n_engines; ++e) {
                const int k = e_k[e];
                const float* s = e_scr[e] + r * k;
                const int64_t* id = e_idx[e] + r * k;
                const float mn = k ? row_min_finite(s, k) : 0.0f;
                for (int j = 0; j < k; ++j) norm[j] = (s[j] - mn) + 0.0f;
                for (int j = 0; j < k; ++j) cursor = write_1d(oi, os, fold_width[e], id[j], norm[j] * e_w[e], cursor);
                if (cursor > local_max[e]) local_max[e] = cursor;
                /* raw scores of this engine: gathered AFTER all folds in the reference; the ids of the columns filled so far do not
                 * change later, so the gather is done at the end (below) over the final row */
            }
            /* gathers by id (numpy_ops.py:24-37): first match in the engine's own list, the pad columns (-1) included */
            for (int e = 0; e < n_engines; ++e) {
                const int k = e_k[e];
                const float* s = e_scr[e] + r * k;
                const int64_t* id = e_idx[e] + r * k;
                const float mn = k ? row_min_finite(s, k) : 0.0f;
                float* raw = out_raw[e] + r * stride;
                for (int c = 0; c < stride; ++c) {
                    const int f = search_1d(id, k, oi[c]);
                    raw[c] = f < 0 ? NAN : (s[f] - mn) + 0.0f;
                }
            }
            int64_t* ol = out_lbl + r * stride;
            for (int c = 0; c < stride; ++c) {
                const int f = search_1d(l_idx + r * k_lookup, k_lookup, oi[c]);
                ol[c] = f < 0 ? -1 : (l_lbl ? l_lbl[r * k_lookup + f] : -1);
            }
        }
        free(norm);
#pragma omp critical
        for (int e = 0; e < n_engines; ++e)
            if (local_max[e] > max_cursor[e]) max_cursor[e] = local_max[e];
    }
    int w = k_lookup;
    for (int e = 0; e < n_engines; ++e) {
        const int cut = max_cursor[e] + 1;
        w = cut < w + e_k[e] ? cut : w + e_k[e];
    }
    *out_width = w;
    return 0;
}

/* ---- sampling ---------------------------------------------------------------------------------------------------------------- */
static void log_softmax_1d(float* x, int n) { /* numpy_ops.py:162-216 */
    for (int i = 0; i < n; ++i)
        if (isnan(x[i])) x[i] = -INFINITY;
    float xm = -INFINITY;
    for (int i = 0; i < n; ++i)
        if (x[i] > xm) xm = x[i];
    if (!isfinite(xm) && xm < 0) xm = 0.0f;
    for (int i = 0; i < n; ++i) x[i] += -xm;
    float lse = 0.0f;
    for (int i = 0; i < n; ++i) lse += expf(x[i]);
    const float l = logf(lse);
    for (int i = 0; i < n; ++i) x[i] += -l;
}

typedef struct {
    float key;
    int idx;
} keyed_t;

static int cmp_desc(const void* a, const void* b) { /* argsort(-keys): descending key, NaN last; ties by position (stable) */
    const keyed_t* x = (const keyed_t*)a;
    const keyed_t* y = (const keyed_t*)b;
    const int xn = isnan(x->key), yn = isnan(y->key);
    if (xn || yn) return xn != yn ? xn - yn : x->idx - y->idx;
    if (x->key > y->key) return -1;
    if (x->key < y->key) return 1;
    return x->idx - y->idx;
}

static int cmp_float_asc(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    const int xn = isnan(x), yn = isnan(y);
    if (xn || yn) return xn - yn;
    return (x > y) - (x < y);
}

/* sample.py:160-219.  scores / noise: the n members of one label class.  Returns the number of samples (<= k). */
static int priority_sampling_1d(const float* scores, const float* noise, int n, int k, float temperature, int max_support, int* out_ids,
                                float* out_logw, float* out_lse, float* log_p, keyed_t* keys, float* tmp) {
    const float t_inv = temperature > 0 ? temperature : 1.0f;
    for (int i = 0; i < n; ++i) log_p[i] = scores[i] * t_inv;
    if (max_support > 0 && n > max_support) {
        memcpy(tmp, log_p, sizeof(float) * (size_t)n);
        qsort(tmp, (size_t)n, sizeof(float), cmp_float_asc);
        const float thr = tmp[n - max_support];
        for (int i = 0; i < n; ++i)
            if (log_p[i] >= thr) log_p[i] = -INFINITY; /* Q8 */
    }
    log_softmax_1d(log_p, n);
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) sum += expf(log_p[i]);
    *out_lse = logf(sum);
    for (int i = 0; i < n; ++i) {
        keys[i].key = temperature > 0 ? log_p[i] - logf(noise[i]) : log_p[i];
        keys[i].idx = i;
    }
    qsort(keys, (size_t)n, sizeof(keyed_t), cmp_desc);
    const int take = k < n ? k : n;
    float log_tau = -INFINITY;
    if (k < n) log_tau = keys[k].key; /* the (k+1)-th largest key */
    for (int j = 0; j < take; ++j) {
        const float log_pi = log_p[keys[j].idx];
        out_ids[j] = keys[j].idx;
        if (log_tau > -INFINITY) {
            const float qz = log1pf(-expf(-expf(log_pi + (-log_tau))));
            out_logw[j] = log_pi - qz;
        } else {
            out_logw[j] = log_pi;
        }
    }
    return take;
}

/* sample.py:22-84 on rows of `width` columns (the cut merged batch), with the Exp(1) draw of sample.py:398 passed in.
 * labels > 0 = positive.  Outputs [nq, k_total]; lse [nq, 2]; max_sampling_id [nq]. */
int vodref_sample_search_results(const int64_t* ids, const float* scores, const int64_t* labels, int n_raw, const float* const* raw,
                                 const float* noise, int64_t nq, int width, int k_positive, int k_total, float temperature,
                                 int max_support, int64_t* out_local, int64_t* out_ids, float* out_scores, float* out_logw,
                                 uint8_t* out_labels, float* const* out_raw, float* out_lse, float* out_max_sampling_id) {
    if (max_support == 0) max_support = -1;
    if (max_support >= 0 && max_support < k_total) max_support = k_total; /* sample.py:123-128 */
#pragma omp parallel
    {
        float* cs = (float*)malloc(sizeof(float) * (size_t)width * 5);
        float *cn = cs + width, *lp = cn + width, *tmp = lp + width, *lw = tmp + width;
        int* members = (int*)malloc(sizeof(int) * (size_t)width * 2);
        int* picked = members + width;
        keyed_t* keys = (keyed_t*)malloc(sizeof(keyed_t) * (size_t)width);
#pragma omp for schedule(static)
        for (int64_t r = 0; r < nq; ++r) {
            const float* s = scores + r * width;
            const float* nz = noise + r * width;
            const int64_t* lb = labels + r * width;
            int n_neg_finite = 0;
            for (int c = 0; c < width; ++c) n_neg_finite += !(lb[c] > 0) && !isinf(s[c]);
            const int kt = k_total > width ? width : k_total;
            int kp = k_positive;
            if (n_neg_finite < kt - kp) kp = kt - n_neg_finite;
            int j = 0;
            for (int c = 0; c < k_total; ++c) {
                out_local[r * k_total + c] = -1;
                out_logw[r * k_total + c] = -INFINITY;
                out_labels[r * k_total + c] = 0;
            }
            int n_pos_taken = 0;
            for (int cls = 1; cls >= 0; --cls) {
                int n = 0;
                for (int c = 0; c < width; ++c)
                    if ((lb[c] > 0) == cls) {
                        members[n] = c;
                        cs[n] = s[c];
                        cn[n] = nz[c];
                        ++n;
                    }
                const int k = cls ? kp : kt - n_pos_taken;
                const int got = priority_sampling_1d(cs, cn, n, k < 0 ? 0 : k, temperature, max_support, picked, lw, out_lse + r * 2 + (cls ? 0 : 1), lp,
                                                     keys, tmp);
                if (got > 0) log_softmax_1d(lw, got); /* normalized = True */
                for (int i = 0; i < got; ++i) {
                    out_local[r * k_total + j] = members[picked[i]];
                    out_logw[r * k_total + j] = lw[i];
                    out_labels[r * k_total + j] = (uint8_t)cls;
                    ++j;
                }
                if (cls) n_pos_taken = got;
            }
            /* take_along_axis (a -1 pad takes the LAST column) + the rank diagnostic (sample.py:56-70) */
            float min_neg = INFINITY;
            for (int c = 0; c < k_total; ++c) {
                const int64_t l = out_local[r * k_total + c];
                const int col = l < 0 ? width - 1 : (int)l;
                out_ids[r * k_total + c] = ids[r * width + col];
                out_scores[r * k_total + c] = s[col];
                for (int e = 0; e < n_raw; ++e) out_raw[e][r * k_total + c] = raw[e][r * width + col];
                if (!out_labels[r * k_total + c] && isfinite(s[col]) && s[col] < min_neg) min_neg = s[col];
            }
            float larger = 0.0f;
            for (int c = 0; c < width; ++c) larger += (!(lb[c] > 0) && isfinite(s[c]) && s[c] >= min_neg) ? 1.0f : 0.0f;
            out_max_sampling_id[r] = larger;
        }
        free(cs);
        free(members);
        free(keys);
    }
    return 0;
}

/* ---- in-batch flattening (in_batch_negatives.py:10-52) ----------------------------------------------------------------------- */
static int cmp_i64(const void* a, const void* b) {
    const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}

/* ids [n_rows, n_keys]; values[v] float32 [n_rows, n_keys] gathered onto the padded unique id list (U = n_rows * n_keys entries; NaN
 * fill), labels uint8 gathered with fill 0.  Returns the number of distinct ids. */
int vodref_flatten_samples(const int64_t* ids, int64_t n_rows, int n_keys, int n_values, const float* const* values, float* const* outs,
                           const uint8_t* labels, uint8_t* out_labels, int64_t* out_unique) {
    const int64_t U = n_rows * n_keys;
    int64_t* sorted = (int64_t*)malloc(sizeof(int64_t) * (size_t)U);
    memcpy(sorted, ids, sizeof(int64_t) * (size_t)U);
    qsort(sorted, (size_t)U, sizeof(int64_t), cmp_i64);
    int64_t n_unique = 0;
    for (int64_t i = 0; i < U; ++i)
        if (i == 0 || sorted[i] != sorted[i - 1]) out_unique[n_unique++] = sorted[i];
    for (int64_t i = n_unique; i < U; ++i) out_unique[i] = 1; /* Q7 */
    free(sorted);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t* keys = ids + r * n_keys;
        for (int64_t c = 0; c < U; ++c) {
            const int f = search_1d(keys, n_keys, out_unique[c]);
            for (int v = 0; v < n_values; ++v) outs[v][r * U + c] = f < 0 ? NAN : values[v][r * n_keys + f];
            if (labels) out_labels[r * U + c] = f < 0 ? 0 : labels[r * n_keys + f];
        }
    }
    return (int)n_unique;
}

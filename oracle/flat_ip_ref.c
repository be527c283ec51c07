/*
 * Oracle (test infrastructure, never linked into the product): plain-C restatement of the
 * CPU path the reference reaches through `faiss_index.search(query_vec, k)`
 * (/root/reference/src/vod_search/faiss_search/server.py:72,84) for a "Flat",
 * METRIC_INNER_PRODUCT index (src/vod_search/faiss_search/build.py:17,60;
 * src/vod_configs/search.py:128-130): float32 corpus, float32 queries, float32 dot products,
 * a per-query bounded min-heap of the k best, output sorted by score descending.
 *
 * PARITY UNPINNED: faiss-cpu 1.7.4 (requirements.txt:42) is not vendored and not installable
 * in this image; this file follows the published IndexFlatIP contract (exhaustive scan + heap
 * for small batches) with two stated choices: ties -> smaller id first, pad = (-inf, -1).
 *
 * Build: gcc -O3 -fopenmp -shared -fPIC flat_ip_ref.c -o _build/liboracle_flat_ip.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float s;
    int64_t id;
} ent_t;

/* "a is worse than b": lower score, or equal score and larger id */
static inline int worse(ent_t a, ent_t b) { return (a.s < b.s) || (a.s == b.s && a.id > b.id); }

static void sift_down(ent_t* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && worse(h[l], h[m])) m = l;
        if (r < n && worse(h[r], h[m])) m = r;
        if (m == i) return;
        ent_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

static int cmp_best_first(const void* pa, const void* pb) {
    ent_t a = *(const ent_t*)pa, b = *(const ent_t*)pb;
    if (worse(b, a)) return -1;
    if (worse(a, b)) return 1;
    return 0;
}

/* q [nq,d], x [n,d] row-major float32; out_s [nq,k] float32, out_i [nq,k] int64 */
int oracle_flat_ip_f32(const float* q, const float* x, int64_t nq, int64_t n, int64_t d, int64_t k, int64_t id_base,
                       float* out_s, int64_t* out_i) {
    if (k <= 0 || d <= 0 || nq < 0 || n < 0) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t iq = 0; iq < nq; ++iq) {
        ent_t* heap = (ent_t*)malloc(sizeof(ent_t) * (size_t)k);
        int hn = 0;
        const float* qv = q + iq * d;
        for (int64_t j = 0; j < n; ++j) {
            const float* xv = x + j * d;
            float acc = 0.f;
            for (int64_t t = 0; t < d; ++t) acc += qv[t] * xv[t];
            if (acc != acc) continue; /* NaN never enters */
            ent_t e = {acc, j};
            if (hn < k) {
                heap[hn++] = e;
                if (hn == k)
                    for (int i = hn / 2 - 1; i >= 0; --i) sift_down(heap, hn, i);
            } else if (worse(heap[0], e)) {
                heap[0] = e;
                sift_down(heap, hn, 0);
            }
        }
        qsort(heap, (size_t)hn, sizeof(ent_t), cmp_best_first);
        for (int64_t t = 0; t < k; ++t) {
            if (t < hn) {
                out_s[iq * k + t] = heap[t].s;
                out_i[iq * k + t] = heap[t].id + id_base;
            } else {
                out_s[iq * k + t] = -INFINITY;
                out_i[iq * k + t] = -1;
            }
        }
        free(heap);
    }
    return 0;
}

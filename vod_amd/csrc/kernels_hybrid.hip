// Hybrid score merge on gfx950: lookup + N scored engines -> union in first-seen order with weighted,
// min-subtracted score accumulation, lookup labels and per-engine raw scores.
//
// Replaces (paths relative to /root/reference/src/vod_dataloaders/core):
//   _merge_search_results          search.py:79-125
//   _subtract_min_score            normalize.py:17-20
//   _nopy_merge_two_search_results merge.py:108-164 (folded pairwise over the engines, merge.py:31-43)
//   gather_values_by_indices       numpy_ops.py:24-143 (raw scores + labels, merge.py:45-60)
//
// One 256-thread workgroup per query row; the row's concatenated (id, weighted score) list lives in LDS.
// The reference's sequential fold is reproduced exactly: an entry is a "first occurrence" when no
// earlier entry of the concatenation carries its id; its output column is the number of first
// occurrences before it; its score is the left fold, in sequence order, of every occurrence's weighted
// score.  All float arithmetic uses the non-contracting intrinsics so that results are bit-identical to
// the reference's float32 NumPy arithmetic (no FMA contraction).  This is HBM/latency-bound integer
// work (< 1 MB per batch); it is deliberately not shaped into a GEMM.
#include "vodhip_internal.h"

namespace vodhip {

constexpr int HY_THREADS = 256;

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(HY_THREADS) void merge_hybrid_kernel(HybridArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    int W = a.k_lookup;
    for (int e = 0; e < a.n_engines; ++e) W += a.engine_k[e];
    // LDS carve-up
    int64_t* ids = (int64_t*)smem;                 // [W] concatenated ids
    int64_t* oids = ids + W;                       // [W+1] output ids (column -> id), -1 beyond the cursor
    float* wsc = (float*)(oids + W + 1);           // [W] weighted, min-subtracted scores
    float* nsc = wsc + W;                          // [W] min-subtracted (unweighted) scores
    int* first = (int*)(nsc + W);                  // [W] first-occurrence flag
    float(*s_min)[4] = (float(*)[4])(first + W);   // [4 waves][4 engines] partial minima
    int& s_cursor = *(int*)(first + W + 16);       // all LDS lives in the one dynamic array (16-B aligned base)

    // ---- 1. per-engine row minimum over finite scores (normalize.py:17-20) ----
    for (int e = 0; e < a.n_engines; ++e) {
        const float* sc = a.engine_scr[e] + row * a.engine_k[e];
        float m = __builtin_inff();
        for (int j = tid; j < a.engine_k[e]; j += HY_THREADS) {
            const float v = sc[j];
            if (!(__builtin_isinf(v) || v != v)) m = fminf(m, v);
        }
        m = wave_min(m);
        if (lane == 0) s_min[wave][e] = m;
    }
    __syncthreads();

    // ---- 2. concatenate: lookup (score 0 * 0), then each engine ((s - min) * w) ----
    for (int j = tid; j < a.k_lookup; j += HY_THREADS) {
        ids[j] = a.lookup_idx[row * a.k_lookup + j];
        wsc[j] = 0.0f;  // lookup scores are zeroed, min-subtracted (0 - 0) and weighted by 0.0 (search.py:92,117)
        nsc[j] = 0.0f;
    }
    int off = a.k_lookup;
    for (int e = 0; e < a.n_engines; ++e) {
        const float mn = fminf(fminf(s_min[0][e], s_min[1][e]), fminf(s_min[2][e], s_min[3][e]));
        const int ke = a.engine_k[e];
        const float w = a.engine_w[e];
        for (int j = tid; j < ke; j += HY_THREADS) {
            const float n = __fsub_rn(a.engine_scr[e][row * ke + j], mn);  // offset 0.0 adds nothing
            ids[off + j] = a.engine_idx[e][row * ke + j];
            nsc[off + j] = n;
            wsc[off + j] = __fmul_rn(n, w);
        }
        off += ke;
    }
    for (int c = tid; c <= W; c += HY_THREADS) oids[c] = -1;
    if (tid == 0) s_cursor = 0;
    __syncthreads();

    // ---- 3. first-occurrence flags ----
    for (int p = tid; p < W; p += HY_THREADS) {
        const int64_t id = ids[p];
        int f = id >= 0;
        for (int r = 0; f && r < p; ++r) f = ids[r] != id;
        first[p] = f;
    }
    __syncthreads();

    // ---- 4. scatter first occurrences to their column, folding later occurrences in order ----
    int64_t* o_idx = a.out_idx + row * a.out_stride;
    float* o_scr = a.out_scr + row * a.out_stride;
    for (int p = tid; p < W; p += HY_THREADS) {
        if (!first[p]) continue;
        int pos = 0;
        for (int r = 0; r < p; ++r) pos += first[r];
        const int64_t id = ids[p];
        float acc = wsc[p];
        for (int r = p + 1; r < W; ++r)
            if (ids[r] == id) acc = __fadd_rn(wsc[r], acc);  // scores[found] = score + scores[found]
        o_idx[pos] = id;
        o_scr[pos] = acc;
        oids[pos] = id;
        atomicMax(&s_cursor, pos + 1);
    }
    __syncthreads();
    const int cursor = s_cursor;
    for (int c = cursor + tid; c < a.out_stride; c += HY_THREADS) {
        o_idx[c] = -1;
        o_scr[c] = -__builtin_inff();
    }
    // The reference folds the engines pairwise and cuts the buffer to `[: max_cursor + 1]` after EVERY fold
    // (merge.py:160-162), so the final width depends on the per-stage maxima over rows of the cursor.
    // stage e = "after engine e has been folded in": cursor_e = #first occurrences among the first end_e entries.
    if (tid < a.n_engines) {
        int end = a.k_lookup;
        for (int e = 0; e <= tid; ++e) end += a.engine_k[e];
        int c = 0;
        for (int p = 0; p < end; ++p) c += first[p];
        atomicMax(&a.out_width[tid], c);
    }
    __syncthreads();

    // ---- 5. labels from the lookup, raw (min-subtracted) scores per engine: first match wins ----
    for (int c = tid; c < a.out_stride; c += HY_THREADS) {
        const int64_t id = c <= W ? oids[c] : -1;
        if (a.out_lbl) {
            int64_t lbl = -1;
            if (a.lookup_lbl) {
                for (int j = 0; j < a.k_lookup; ++j)
                    if (ids[j] == id) {
                        lbl = a.lookup_lbl[row * a.k_lookup + j];
                        break;
                    }
            }
            a.out_lbl[row * a.out_stride + c] = lbl;
        }
        int eo = a.k_lookup;
        for (int e = 0; e < a.n_engines; ++e) {
            if (a.out_raw[e]) {
                float v = __builtin_nanf("");
                for (int j = 0; j < a.engine_k[e]; ++j)
                    if (ids[eo + j] == id) {
                        v = nsc[eo + j];
                        break;
                    }
                a.out_raw[e][row * a.out_stride + c] = v;
            }
            eo += a.engine_k[e];
        }
    }
}

hipError_t launch_merge_hybrid(const HybridArgs& a, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(a.out_width, 0, sizeof(int32_t) * 4, stream);
    if (e != hipSuccess) return e;
    if (a.nq == 0) return hipSuccess;
    int W = a.k_lookup;
    for (int i = 0; i < a.n_engines; ++i) W += a.engine_k[i];
    const size_t lds = (size_t)W * (8 + 8 + 4 + 4 + 4) + 8 + 17 * 4 + 12;
    hipLaunchKernelGGL(merge_hybrid_kernel, dim3((unsigned)a.nq), dim3(HY_THREADS), lds, stream, a);
    return hipGetLastError();
}

}  // namespace vodhip

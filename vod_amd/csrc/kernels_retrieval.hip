// In-batch retrieval scoring + log-prob / loss combination, forward and backward, on gfx950.
//
// Replaces /root/reference/src/vod_models/vod_gradients/retrieval.py:
//   RetrievalGradients.__call__ :30-92, _compute_retriever_scores :186-203 (einsum bh,dh->bd | bh,bdh->bd,
//   masked_fill -inf), _cast_data_targets :206-215, _compute_loss :153-177, _compute_kld :225-243,
//   and the autograd backward of those (log-softmax backward, masked_fill_ backward, the two einsum grads).
//
// Forward: one 256-thread workgroup per query row computes the row of scores, its log-softmax, the
// targets / n_positives fallback, the row loss, dLoss/dScores (up to the 1/n_rows factor) and the three
// KL(q || p) diagnostics in ONE pass; a small finalize kernel reduces the per-row partials and applies
// 1/n_rows.  Sizes here are tiny (B=64 rows x D<=2048 sections x H<=1024): the stage is launch-latency
// bound, so the design goal is "2 launches instead of ~15", not MFMA utilisation.
#include "vodhip_internal.h"

#include <algorithm>

namespace vodhip {

constexpr int RT_THREADS = 256;
constexpr int WS_STRIDE = 8;  // floats of workspace per row: loss, has_pos, kl_score, kl_sparse, kl_dense

template <int DT>
__device__ __forceinline__ float ld_enc(const void* p, int64_t i) {
    if constexpr (DT == 2) {
        return ((const float*)p)[i];
    } else if constexpr (DT == 0) {
        return (float)((const _Float16*)p)[i];
    } else {
        return (float)((const __bf16*)p)[i];
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// block-wide reductions through a 4-float LDS scratch (256 threads = 4 waves)
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ bool finite_f(float v) { return !(__builtin_isinf(v) || v != v); }

// KL(q || p) over entries finite in both; p_lp is already the renormalised model log-prob (LDS).
__device__ float row_kld(const float* __restrict__ ref_row, const float* p_lp, int D, float* red) {
    const int tid = threadIdx.x;
    float m = -__builtin_inff();
    for (int d = tid; d < D; d += RT_THREADS) {
        const float v = ref_row[d];
        if (finite_f(v)) m = fmaxf(m, v);
    }
    m = block_max(m, red);
    float se = 0.f;
    for (int d = tid; d < D; d += RT_THREADS) {
        const float v = ref_row[d];
        if (finite_f(v)) se += expf(v - m);
    }
    se = block_sum(se, red);
    const float lse = m + logf(se);  // all-masked row: -inf + log(0) -> contributes nothing below
    float acc = 0.f;
    for (int d = tid; d < D; d += RT_THREADS) {
        const float v = ref_row[d];
        const float pl = p_lp[d];
        if (finite_f(v) && finite_f(pl)) {
            const float ql = v - lse;
            acc += expf(ql) * (ql - pl);
        }
    }
    return block_sum(acc, red);
}

template <int DT, bool S3D>
__global__ __launch_bounds__(RT_THREADS) void retrieval_forward_kernel(
    const void* __restrict__ q, const void* __restrict__ s, int D, int H, const float* __restrict__ score,
    const int64_t* __restrict__ relevance, const float* __restrict__ sparse, const float* __restrict__ dense,
    float* __restrict__ retriever_scores, float* __restrict__ d_scores, float* __restrict__ workspace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qrow = (float*)smem;   // [H]
    float* S = qrow + H;          // [D] scores -> log-probs
    float* red = S + D;           // [4]
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int h = tid; h < H; h += RT_THREADS) qrow[h] = ld_enc<DT>(q, b * H + h);
    __syncthreads();

    // 1. scores: one wavefront per section, lanes split the hidden dimension (coalesced reads of s)
    const int64_t s_base = S3D ? b * (int64_t)D * H : 0;
    for (int d = wave; d < D; d += RT_THREADS / 64) {
        const int64_t off = s_base + (int64_t)d * H;
        float acc = 0.f;
        for (int h = lane; h < H; h += 64) acc = fmaf(qrow[h], ld_enc<DT>(s, off + h), acc);
        acc = wave_sum(acc);
        if (lane == 0) S[d] = acc;
    }
    __syncthreads();

    // 2. padding mask, masked scores out
    const float* score_row = score + b * D;
    float mx = -__builtin_inff();
    int n_nonpad = 0, n_true = 0;
    for (int d = tid; d < D; d += RT_THREADS) {
        const float sc = score_row[d];
        const bool pad = __builtin_isinf(sc) && sc < 0;
        const float v = pad ? -__builtin_inff() : S[d];
        S[d] = v;
        retriever_scores[b * D + d] = v;
        mx = fmaxf(mx, v);
        n_nonpad += !pad;
        n_true += (!pad && relevance[b * D + d] > 0);
    }
    mx = block_max(mx, red);
    const float f_nonpad = block_sum((float)n_nonpad, red);
    const float f_true = block_sum((float)n_true, red);
    const float npos = f_true == 0.f ? f_nonpad : f_true;  // retrieval.py:57
    const bool has_pos = npos > 0.f;

    // 3. log-softmax (an all-padded row gives NaN log-probs, exactly as torch; it is masked below)
    float se = 0.f;
    for (int d = tid; d < D; d += RT_THREADS) se += expf(S[d] - mx);
    se = block_sum(se, red);
    const float lse = logf(se);

    // 4. w = (p - t) / n_pos ; row loss = sum_{~pad} w * logp ; dS = w - p * sum(w)   (x 1/n_rows later)
    float loss_acc = 0.f, g_acc = 0.f;
    for (int d = tid; d < D; d += RT_THREADS) {
        const float sc = score_row[d];
        const bool pad = __builtin_isinf(sc) && sc < 0;
        const float lp = S[d] - mx - lse;
        S[d] = lp;  // keep the log-prob for the KL terms
        if (!pad) {
            const float p = expf(lp);
            const float t = relevance[b * D + d] > 0 ? 1.f : 0.f;
            const float w = (p - t) / npos;
            loss_acc += w * lp;
            g_acc += w;
        }
    }
    const float row_loss = block_sum(loss_acc, red);
    const float gsum = block_sum(g_acc, red);
    for (int d = tid; d < D; d += RT_THREADS) {
        const float sc = score_row[d];
        const bool pad = __builtin_isinf(sc) && sc < 0;
        float g = 0.f;
        if (!pad && has_pos) {
            const float lp = S[d];
            const float p = expf(lp);
            const float t = relevance[b * D + d] > 0 ? 1.f : 0.f;
            g = (p - t) / npos - p * gsum;
        }
        d_scores[b * D + d] = g;
    }

    // 5. KL diagnostics against the sampling distributions (retrieval.py:79-86,225-243)
    //    p side: log-softmax of the finite model log-probs
    float pm = -__builtin_inff();
    for (int d = tid; d < D; d += RT_THREADS)
        if (finite_f(S[d])) pm = fmaxf(pm, S[d]);
    pm = block_max(pm, red);
    float pse = 0.f;
    for (int d = tid; d < D; d += RT_THREADS)
        if (finite_f(S[d])) pse += expf(S[d] - pm);
    pse = block_sum(pse, red);
    const float plse = pm + logf(pse);
    __syncthreads();
    for (int d = tid; d < D; d += RT_THREADS) {
        const float lp = S[d];
        S[d] = finite_f(lp) ? lp - plse : -__builtin_inff();
    }
    __syncthreads();
    const float kl0 = row_kld(score_row, S, D, red);
    const float kl1 = sparse ? row_kld(sparse + b * D, S, D, red) : __builtin_nanf("");
    const float kl2 = dense ? row_kld(dense + b * D, S, D, red) : __builtin_nanf("");
    if (tid == 0) {
        float* w = workspace + b * WS_STRIDE;
        w[0] = has_pos ? row_loss : 0.f;
        w[1] = has_pos ? 1.f : 0.f;
        w[2] = kl0;
        w[3] = kl1;
        w[4] = kl2;
    }
}

__global__ __launch_bounds__(RT_THREADS) void retrieval_finalize_kernel(const float* __restrict__ workspace, int B,
                                                                        int64_t n_elems, float* __restrict__ d_scores,
                                                                        float* __restrict__ loss, float* __restrict__ kl) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float l = 0.f, n = 0.f, k0 = 0.f, k1 = 0.f, k2 = 0.f;
    for (int b = tid; b < B; b += RT_THREADS) {
        const float* w = workspace + (int64_t)b * WS_STRIDE;
        l += w[0];
        n += w[1];
        k0 += w[2];
        k1 += w[3];
        k2 += w[4];
    }
    l = block_sum(l, red);
    n = block_sum(n, red);
    k0 = block_sum(k0, red);
    k1 = block_sum(k1, red);
    k2 = block_sum(k2, red);
    const float inv = n > 0.f ? 1.f / n : __builtin_nanf("");
    if (blockIdx.x == 0 && tid == 0) {
        loss[0] = n > 0.f ? l / n : __builtin_nanf("");  // retrieval.py:171-176
        kl[0] = k0 / B;
        kl[1] = k1 / B;
        kl[2] = k2 / B;
    }
    for (int64_t i = (int64_t)blockIdx.x * RT_THREADS + tid; i < n_elems; i += (int64_t)gridDim.x * RT_THREADS)
        d_scores[i] *= inv;
}

// dq[b,h] = go * sum_d dS[b,d] * s[(b,)d,h]
template <int DT, bool S3D>
__global__ __launch_bounds__(RT_THREADS) void retrieval_dq_kernel(const void* __restrict__ s, int D, int H,
                                                                  const float* __restrict__ d_scores,
                                                                  const float* __restrict__ grad_out, float* __restrict__ dq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* g = (float*)smem;  // [D]
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const float go = grad_out[0];
    for (int d = tid; d < D; d += RT_THREADS) g[d] = d_scores[b * D + d];
    __syncthreads();
    const int64_t s_base = S3D ? b * (int64_t)D * H : 0;
    for (int h = tid; h < H; h += RT_THREADS) {
        float acc = 0.f;
        for (int d = 0; d < D; ++d) {
            const float gd = g[d];
            if (gd != 0.f) acc = fmaf(gd, ld_enc<DT>(s, s_base + (int64_t)d * H + h), acc);
        }
        dq[b * H + h] = go * acc;
    }
}

// ds[d,h] = go * sum_b dS[b,d] * q[b,h]   (2-D sections)   |   ds[b,d,h] = go * dS[b,d] * q[b,h]   (3-D)
template <int DT, bool S3D>
__global__ __launch_bounds__(RT_THREADS) void retrieval_ds_kernel(const void* __restrict__ q, int B, int D, int H,
                                                                  const float* __restrict__ d_scores,
                                                                  const float* __restrict__ grad_out, float* __restrict__ ds) {
    const int tid = threadIdx.x;
    const float go = grad_out[0];
    if constexpr (S3D) {
        const int64_t bd = blockIdx.x;  // b * D + d
        const int64_t b = bd / D;
        const float g = go * d_scores[bd];
        for (int h = tid; h < H; h += RT_THREADS) ds[bd * H + h] = g * ld_enc<DT>(q, b * H + h);
    } else {
        const int64_t d = blockIdx.x;
        for (int h = tid; h < H; h += RT_THREADS) {
            float acc = 0.f;
            for (int b = 0; b < B; ++b) acc = fmaf(d_scores[(int64_t)b * D + d], ld_enc<DT>(q, (int64_t)b * H + h), acc);
            ds[d * H + h] = go * acc;
        }
    }
}

hipError_t launch_retrieval_forward(const void* q, const void* s, int enc_dtype, int sections_3d, int64_t B, int64_t D,
                                    int64_t H, const float* score, const int64_t* relevance, const float* sparse,
                                    const float* dense, float* retriever_scores, float* d_scores, float* loss, float* kl,
                                    float* workspace, hipStream_t stream) {
    const size_t lds = (size_t)(H + D + 4) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#define VOD_FWD(DT, S3)                                                                                              \
    if (enc_dtype == DT && (sections_3d != 0) == S3) {                                                               \
        auto kern = retrieval_forward_kernel<DT, S3>;                                                                \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (e != hipSuccess) return e;                                                                               \
        hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(RT_THREADS), lds, stream, q, s, (int)D, (int)H, score,      \
                           relevance, sparse, dense, retriever_scores, d_scores, workspace);                         \
    }
    VOD_FWD(0, false) VOD_FWD(0, true) VOD_FWD(1, false) VOD_FWD(1, true) VOD_FWD(2, false) VOD_FWD(2, true)
#undef VOD_FWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t n_elems = B * D;
    const unsigned blocks = (unsigned)std::min<int64_t>(1024, (n_elems + RT_THREADS - 1) / RT_THREADS);
    hipLaunchKernelGGL(retrieval_finalize_kernel, dim3(blocks), dim3(RT_THREADS), 0, stream, workspace, (int)B, n_elems,
                       d_scores, loss, kl);
    return hipGetLastError();
}

hipError_t launch_retrieval_backward(const void* q, const void* s, int enc_dtype, int sections_3d, int64_t B, int64_t D,
                                     int64_t H, const float* d_scores, const float* grad_out, float* dq, float* ds,
                                     hipStream_t stream) {
    const size_t lds = (size_t)D * sizeof(float);
#define VOD_BWD(DT, S3)                                                                                               \
    if (enc_dtype == DT && (sections_3d != 0) == S3) {                                                                \
        hipLaunchKernelGGL((retrieval_dq_kernel<DT, S3>), dim3((unsigned)B), dim3(RT_THREADS), lds, stream, s, (int)D, \
                           (int)H, d_scores, grad_out, dq);                                                           \
        hipLaunchKernelGGL((retrieval_ds_kernel<DT, S3>), dim3((unsigned)(S3 ? B * D : D)), dim3(RT_THREADS), 0, stream, \
                           q, (int)B, (int)D, (int)H, d_scores, grad_out, ds);                                        \
    }
    VOD_BWD(0, false) VOD_BWD(0, true) VOD_BWD(1, false) VOD_BWD(1, true) VOD_BWD(2, false) VOD_BWD(2, true)
#undef VOD_BWD
    return hipGetLastError();
}

}  // namespace vodhip

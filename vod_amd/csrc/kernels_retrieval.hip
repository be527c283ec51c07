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
constexpr int WS_STRIDE = 8;       // floats of workspace per row: loss, has_pos, kl_score, kl_sparse, kl_dense
constexpr int WS_STRIDE_AUX = 16;  // ... + huber sum, huber count, cross entropy, row counted, score^2 sum, finite count

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

template <int DT, bool S3D, bool PRE = false>
__global__ __launch_bounds__(RT_THREADS) void retrieval_forward_kernel(
    const void* __restrict__ q, const void* __restrict__ s, int D, int H, const float* __restrict__ score,
    const int64_t* __restrict__ relevance, const float* __restrict__ sparse, const float* __restrict__ dense,
    // the two outputs carry no `restrict`: in the PRE instantiation `pre_slabs` may point INTO them (the contraction's slab(s)
    // when the caller's workspace has no room); the row is read into LDS, a barrier follows, then it is written
    float* retriever_scores, float* d_scores, float* __restrict__ workspace, RetrievalAux aux,
    const float* pre_slabs = nullptr /* may alias the two outputs: read into LDS first */, int n_slabs = 0, int64_t slab_stride = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qrow = (float*)smem;   // [H]
    float* S = qrow + H;          // [D] scores -> log-probs
    float* red = S + D;           // [4]
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if constexpr (PRE) {
        // 1'. the contraction was done by small_gemm_kernel (MFMA) in `n_slabs` split-K slabs: summed here in slab order
        //     (fixed order: bitwise reproducible), no separate reduction launch
        for (int d = tid; d < D; d += RT_THREADS) {
            float acc = pre_slabs[b * D + d];
            for (int z = 1; z < n_slabs; ++z) acc += pre_slabs[(int64_t)z * slab_stride + b * D + d];
            S[d] = acc;
        }
        __syncthreads();
    } else {
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
    }

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

    // 4b. auxiliary losses (retrieval.py:94-150), all functions of the row of scores / log-probs that is in LDS now.
    //     Each writes its row partials and its UNNORMALISED gradient w.r.t. the scores; the finalize kernel divides by
    //     the batch-wide counts and adds the weighted terms to the loss and to d_scores.
    const int ws_stride = aux.enabled ? WS_STRIDE_AUX : WS_STRIDE;
    if (aux.enabled) {
        const int64_t n_el = (int64_t)gridDim.x * D;
        float* w = workspace + b * ws_stride;
        // guidance: huber(logp - ref) over entries finite in both, ref = sparse scores | zeros (:116-126,180-183)
        if (aux.w_guidance > 0.f) {
            float hs = 0.f, hc = 0.f, gh = 0.f;
            for (int d = tid; d < D; d += RT_THREADS) {
                const float lp = S[d];
                const float ref = aux.guidance_type == 1 ? sparse[b * D + d] : 0.f;
                if (finite_f(lp) && finite_f(ref)) {
                    const float x = lp - ref, ax = fabsf(x);
                    hs += ax < 1.f ? 0.5f * x * x : ax - 0.5f;
                    hc += 1.f;
                    gh += fminf(1.f, fmaxf(-1.f, x));
                }
            }
            hs = block_sum(hs, red);
            hc = block_sum(hc, red);
            gh = block_sum(gh, red);
            for (int d = tid; d < D; d += RT_THREADS) {  // log-softmax backward: g - p * sum(g)
                const float lp = S[d];
                const float ref = aux.guidance_type == 1 ? sparse[b * D + d] : 0.f;
                const float g = (finite_f(lp) && finite_f(ref)) ? fminf(1.f, fmaxf(-1.f, lp - ref)) : 0.f;
                const float p = finite_f(lp) ? expf(lp) : 0.f;
                aux.grad[b * D + d] = g - p * gh;
            }
            if (tid == 0) { w[8] = hs; w[9] = hc; }
        }
        // self-supervision: cross entropy of the positives' log-probs against their own arg-max (:129-140)
        if (aux.w_self > 0.f) {
            float lm = -__builtin_inff();
            for (int d = tid; d < D; d += RT_THREADS) {
                const float sc = score_row[d];
                const bool pos = !(__builtin_isinf(sc) && sc < 0) && relevance[b * D + d] > 0;
                if (pos) lm = fmaxf(lm, S[d]);
            }
            lm = block_max(lm, red);
            float first = -1e30f, se2 = 0.f, n_p = 0.f;
            for (int d = tid; d < D; d += RT_THREADS) {
                const float sc = score_row[d];
                const bool pos = !(__builtin_isinf(sc) && sc < 0) && relevance[b * D + d] > 0;
                if (pos) {
                    n_p += 1.f;
                    se2 += expf(S[d] - lm);
                    if (S[d] == lm) first = fmaxf(first, -(float)d);  // first arg-max = smallest index
                }
            }
            first = block_max(first, red);
            se2 = block_sum(se2, red);
            n_p = block_sum(n_p, red);
            const int idx = (int)(-first);
            const bool in_r = npos > 0.f;  // `n_positives > 0` AFTER the fallback (:57,138): a row without positives stays in
            const float ce = n_p > 0.f ? logf(se2) : __builtin_nanf("");
            for (int d = tid; d < D; d += RT_THREADS) {
                const float sc = score_row[d];
                const bool pos = !(__builtin_isinf(sc) && sc < 0) && relevance[b * D + d] > 0;
                float g = 0.f;
                // (a row without positives makes the LOSS NaN - log-softmax of an all -inf row - but sends no gradient:
                //  `torch.where(targets > 0, logp, -inf)` selects none of its log-probs)
                if (in_r && n_p > 0.f && pos) g = expf(S[d] - lm) / se2 - (d == idx ? 1.f : 0.f);
                aux.grad[n_el + b * D + d] = g;  // sums to zero over the row: the log-softmax backward leaves it unchanged
            }
            if (tid == 0) { w[10] = in_r ? ce : 0.f; w[11] = in_r ? 1.f : 0.f; }
        }
        // score decay: mean of the squared finite scores (:143-145)
        if (aux.w_decay > 0.f) {
            float ss = 0.f, sn = 0.f;
            for (int d = tid; d < D; d += RT_THREADS) {
                const float v = retriever_scores[b * D + d];
                const bool fin = finite_f(v);
                ss += fin ? v * v : 0.f;
                sn += fin ? 1.f : 0.f;
                aux.grad[2 * n_el + b * D + d] = fin ? 2.f * v : 0.f;
            }
            ss = block_sum(ss, red);
            sn = block_sum(sn, red);
            if (tid == 0) { w[12] = ss; w[13] = sn; }
        }
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
        float* w = workspace + b * ws_stride;
        w[0] = has_pos ? row_loss : 0.f;
        w[1] = has_pos ? 1.f : 0.f;
        w[2] = kl0;
        w[3] = kl1;
        w[4] = kl2;
    }
}

__global__ __launch_bounds__(RT_THREADS) void retrieval_finalize_kernel(const float* __restrict__ workspace, int B,
                                                                        int64_t n_elems, float* __restrict__ d_scores,
                                                                        float* __restrict__ loss, float* __restrict__ kl,
                                                                        RetrievalAux aux) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int stride = aux.enabled ? WS_STRIDE_AUX : WS_STRIDE;
    float l = 0.f, n = 0.f, k0 = 0.f, k1 = 0.f, k2 = 0.f;
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // huber sum / count, cross-entropy sum / rows, score^2 sum / count
    for (int b = tid; b < B; b += RT_THREADS) {
        const float* w = workspace + (int64_t)b * stride;
        l += w[0];
        n += w[1];
        k0 += w[2];
        k1 += w[3];
        k2 += w[4];
        if (aux.enabled) {
            if (aux.w_guidance > 0.f) { a[0] += w[8]; a[1] += w[9]; }
            if (aux.w_self > 0.f) { a[2] += w[10]; a[3] += w[11]; }
            if (aux.w_decay > 0.f) { a[4] += w[12]; a[5] += w[13]; }
        }
    }
    l = block_sum(l, red);
    n = block_sum(n, red);
    k0 = block_sum(k0, red);
    k1 = block_sum(k1, red);
    k2 = block_sum(k2, red);
    const float inv = n > 0.f ? 1.f / n : __builtin_nanf("");
    float c_g = 0.f, c_s = 0.f, c_d = 0.f;  // weight / count of each auxiliary mean
    float total = n > 0.f ? l / n : __builtin_nanf("");  // retrieval.py:171-176
    if (aux.enabled) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i] = block_sum(a[i], red);
        const float nan = __builtin_nanf("");
        const float lg = a[1] > 0.f ? a[0] / a[1] : nan, ls = a[3] > 0.f ? a[2] / a[3] : nan, ld = a[5] > 0.f ? a[4] / a[5] : nan;
        if (aux.w_guidance > 0.f) { total += aux.w_guidance * lg; c_g = a[1] > 0.f ? aux.w_guidance / a[1] : nan; }
        if (aux.w_self > 0.f) { total += aux.w_self * ls; c_s = a[3] > 0.f ? aux.w_self / a[3] : nan; }
        if (aux.w_decay > 0.f) { total += aux.w_decay * ld; c_d = a[5] > 0.f ? aux.w_decay / a[5] : nan; }
        if (blockIdx.x == 0 && tid == 0) {
            aux.out[0] = aux.w_guidance > 0.f ? lg : nan;
            aux.out[1] = aux.w_self > 0.f ? ls : nan;
            aux.out[2] = aux.w_decay > 0.f ? ld : nan;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        loss[0] = total;
        kl[0] = k0 / B;
        kl[1] = k1 / B;
        kl[2] = k2 / B;
    }
    for (int64_t i = (int64_t)blockIdx.x * RT_THREADS + tid; i < n_elems; i += (int64_t)gridDim.x * RT_THREADS) {
        float g = d_scores[i] * inv;
        if (aux.enabled) {
            if (aux.w_guidance > 0.f) g += c_g * aux.grad[i];
            if (aux.w_self > 0.f) g += c_s * aux.grad[n_elems + i];
            if (aux.w_decay > 0.f) g += c_d * aux.grad[2 * n_elems + i];
        }
        d_scores[i] = g;
    }
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

// ------------------------------------------------------------------------------------------------
// small_gemm_kernel: C[M,N] (f32) = alpha * A[M,K] * B[K,N] with arbitrary element strides, f32 accumulate on
// v_mfma_f32_32x32x2_f32 (exact f32 products and sums, MI355X_MICROARCH "FP32-input MFMA").
// Used for the in-batch contraction einsum("bh,dh->bd") and its two gradients when the section encodings are the
// flattened in-batch set (D = B * n_sections ~ 2048): 0.2 GFLOP, launch-latency bound - the point is one MFMA
// launch instead of every workgroup re-reading the whole section matrix.
// Workgroup = 4 waves = 64 x 64 tile of C (wave (wm, wn) owns 32 x 32); K tile 64 staged through LDS as f32.
// ------------------------------------------------------------------------------------------------
typedef float g_f32x16 __attribute__((ext_vector_type(16)));

template <int DTA, int DTB>
__global__ __launch_bounds__(256) void small_gemm_kernel(const void* __restrict__ A, int64_t sa_m, int64_t sa_k,
                                                         const void* __restrict__ Bm, int64_t sb_k, int64_t sb_n,
                                                         float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                         const float* __restrict__ alpha_ptr, int k_per_split,
                                                         int64_t c_split_stride) {
    // K tile 64.  Staging moves 4 consecutive elements of the unit-stride dimension per load (a 16-byte load for f32, 8 bytes for
    // f16 / bf16) where the tile is interior and the row pitch keeps them aligned; the scalar loop handles edges and odd pitches.
    constexpr int TMN = 64, KT = 64;
    __shared__ float As[TMN][KT + 1];
    __shared__ float Bs[KT][TMN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * TMN, n0 = blockIdx.x * TMN;
    const float alpha = alpha_ptr ? alpha_ptr[0] : 1.0f;
    g_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool a_k_fast = sa_k == 1, b_n_fast = sb_n == 1;
    // split-K: blockIdx.z owns K range [z * k_per_split, ...) and writes its own slab of C (summed in a fixed order by the
    // consumer: bitwise reproducible, unlike float atomics)
    const int k_begin = blockIdx.z * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);
    C += (int64_t)blockIdx.z * c_split_stride;
    K = k_end;
    // element (r, c) of a [R][Ccols] tile whose unit-stride dimension is `c`: src[(r0 + r) * pitch + c0 + c]
    auto stage_vec = [&](auto dt_tag, const void* src, int64_t pitch, int r0, int c0, int R_lim, int C_lim, auto store) {
        constexpr int DT = decltype(dt_tag)::value;
        const bool vec_ok = (pitch % 4 == 0) && (c0 % 4 == 0) && r0 + TMN <= R_lim && c0 + KT <= C_lim && TMN == KT;
        if (vec_ok) {
            for (int e = tid; e < TMN * (KT / 4); e += 256) {
                const int r = e / (KT / 4), c = (e % (KT / 4)) * 4;
                const int64_t off = (int64_t)(r0 + r) * pitch + c0 + c;
                float v[4];
                if constexpr (DT == 2) {
                    const float4 t = *(const float4*)((const float*)src + off);
                    v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
                } else {
                    const uint2 t = *(const uint2*)((const uint16_t*)src + off);
                    const unsigned w[2] = {t.x, t.y};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint16_t h = (uint16_t)(w[u >> 1] >> (16 * (u & 1)));
                        if constexpr (DT == 0) v[u] = (float)__builtin_bit_cast(_Float16, h);
                        else v[u] = __builtin_bit_cast(float, (unsigned)h << 16);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) store(r, c + u, v[u]);
            }
        } else {
            for (int e = tid; e < TMN * KT; e += 256) {
                const int r = e / KT, c = e % KT;
                store(r, c, (r0 + r < R_lim && c0 + c < C_lim) ? ld_enc<DT>(src, (int64_t)(r0 + r) * pitch + c0 + c) : 0.f);
            }
        }
    };
    for (int k0 = k_begin; k0 < k_end; k0 += KT) {
        // A tile [64 m][64 k]: rows = m when k is the unit-stride dimension, rows = k when m is
        if (a_k_fast) stage_vec(std::integral_constant<int, DTA>{}, A, sa_m, m0, k0, M, K, [&](int r, int c, float v) { As[r][c] = v; });
        else if (sa_m == 1) stage_vec(std::integral_constant<int, DTA>{}, A, sa_k, k0, m0, K, M, [&](int r, int c, float v) { As[c][r] = v; });
        else
            for (int e = tid; e < TMN * KT; e += 256) {
                const int m = e % TMN, k = e / TMN;
                As[m][k] = (m0 + m < M && k0 + k < K) ? ld_enc<DTA>(A, (m0 + m) * sa_m + (k0 + k) * sa_k) : 0.f;
            }
        // B tile [64 k][64 n]
        if (b_n_fast) stage_vec(std::integral_constant<int, DTB>{}, Bm, sb_k, k0, n0, K, N, [&](int r, int c, float v) { Bs[r][c] = v; });
        else if (sb_k == 1) stage_vec(std::integral_constant<int, DTB>{}, Bm, sb_n, n0, k0, N, K, [&](int r, int c, float v) { Bs[c][r] = v; });
        else
            for (int e = tid; e < KT * TMN; e += 256) {
                const int k = e % KT, n = e / KT;
                Bs[k][n] = (k0 + k < K && n0 + n < N) ? ld_enc<DTB>(Bm, (k0 + k) * sb_k + (n0 + n) * sb_n) : 0.f;
            }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KT; kk += 2) {
            // 32x32x2 f32 operands: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]
            const float a = As[wm * 32 + (lane & 31)][kk + (lane >> 5)];
            const float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int gn = n0 + wn * 32 + (lane & 31);
        if (gm < M && gn < N) C[(int64_t)gm * ldc + gn] = alpha * acc[r];
    }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int64_t count, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float acc = 0.f;
    for (int z = 0; z < n_slabs; ++z) acc += slabs[(int64_t)z * count + i];
    out[i] = acc;
}

static hipError_t launch_small_gemm(int dta, int dtb, const void* A, int64_t sa_m, int64_t sa_k, const void* B, int64_t sb_k,
                                    int64_t sb_n, float* C, int64_t ldc, int M, int N, int K, const float* alpha,
                                    hipStream_t stream, int n_splits = 1, int64_t c_split_stride = 0) {
    const int k_per_split = ((K + n_splits - 1) / n_splits + 63) / 64 * 64;
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)n_splits);
#define VOD_SG(X, Y)                                                                                              \
    if (dta == X && dtb == Y) {                                                                                   \
        hipLaunchKernelGGL((small_gemm_kernel<X, Y>), grid, dim3(256), 0, stream, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, M, N, K, \
                           alpha, k_per_split, c_split_stride);                                                   \
        return hipGetLastError();                                                                                 \
    }
    VOD_SG(0, 0) VOD_SG(1, 1) VOD_SG(2, 2) VOD_SG(2, 0) VOD_SG(2, 1)
#undef VOD_SG
    return hipErrorInvalidValue;
}

hipError_t launch_retrieval_forward(const void* q, const void* s, int enc_dtype, int sections_3d, int64_t B, int64_t D,
                                    int64_t H, const float* score, const int64_t* relevance, const float* sparse,
                                    const float* dense, float* retriever_scores, float* d_scores, float* loss, float* kl,
                                    float* workspace, const RetrievalAux& aux, hipStream_t stream, int64_t workspace_floats) {
    const size_t lds = (size_t)(H + D + 4) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (!sections_3d) {
        // einsum("bh,dh->bd"): scores[b,d] = sum_h q[b,h] * s[d,h]  ->  A = q [B,H] (k fast), B = s^T (k = h fast)
        // A 64 x 2048 output is 32 workgroups of 64 x 64: each would walk all of K alone (12 global -> LDS -> MFMA rounds, ~60 us,
        // 7/8 of the chip idle).  K is split over 4 workgroups per tile into slabs of a stream-ordered temporary; the row kernel
        // sums the slabs in order while it loads its row.
        // The slabs live in the caller's workspace behind the 16 B row words when it is large enough (a stream-ordered temporary was
        // tried: hipMallocAsync + hipFreeAsync cost ~150 us per call); else two slabs: the two [B, D] outputs, which the row kernel
        // reads (its own row, into LDS) before it writes them.
        int n_splits = 1;
        float* slabs = retriever_scores;
        int64_t slab_stride = B * D;
        if (H >= 512) {
            if (workspace_floats >= 16 * B + 4 * B * D) {
                n_splits = 4;
                slabs = workspace + 16 * B;
            } else if (d_scores == retriever_scores + B * D || retriever_scores == d_scores + B * D) {
                n_splits = 2;  // adjacent outputs: slab z = the z-th of the two buffers
                slabs = retriever_scores < d_scores ? retriever_scores : d_scores;
            }
        }
        hipError_t e = launch_small_gemm(enc_dtype, enc_dtype, q, H, 1, s, 1, H, slabs, D, (int)B, (int)D, (int)H, nullptr, stream, n_splits,
                                         slab_stride);
        if (e != hipSuccess) return e;
#define VOD_FWDP(DT)                                                                                                  \
    if (enc_dtype == DT) {                                                                                            \
        auto kern = retrieval_forward_kernel<DT, false, true>;                                                        \
        e = allow_dynamic_lds((const void*)kern, 160 * 1024); /* cached: no driver call on the launch path */          \
        if (e != hipSuccess) return e;                                                                                \
        hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(RT_THREADS), lds, stream, q, s, (int)D, (int)H, score,       \
                           relevance, sparse, dense, retriever_scores, d_scores, workspace, aux, (const float*)slabs, \
                           n_splits, slab_stride);                                                                    \
    }
        VOD_FWDP(0) VOD_FWDP(1) VOD_FWDP(2)
#undef VOD_FWDP
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int64_t n_el = B * D;
        const unsigned blk = (unsigned)std::min<int64_t>(1024, (n_el + RT_THREADS - 1) / RT_THREADS);
        hipLaunchKernelGGL(retrieval_finalize_kernel, dim3(blk), dim3(RT_THREADS), 0, stream, workspace, (int)B, n_el, d_scores,
                           loss, kl, aux);
        return hipGetLastError();
    }
#define VOD_FWD(DT, S3)                                                                                              \
    if (enc_dtype == DT && (sections_3d != 0) == S3) {                                                               \
        auto kern = retrieval_forward_kernel<DT, S3>;                                                                \
        hipError_t e = allow_dynamic_lds((const void*)kern, 160 * 1024);                                               \
        if (e != hipSuccess) return e;                                                                               \
        hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(RT_THREADS), lds, stream, q, s, (int)D, (int)H, score,      \
                           relevance, sparse, dense, retriever_scores, d_scores, workspace, aux, (const float*)nullptr, 0, \
                           (int64_t)0);                                                                              \
    }
    VOD_FWD(0, false) VOD_FWD(0, true) VOD_FWD(1, false) VOD_FWD(1, true) VOD_FWD(2, false) VOD_FWD(2, true)
#undef VOD_FWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t n_elems = B * D;
    const unsigned blocks = (unsigned)std::min<int64_t>(1024, (n_elems + RT_THREADS - 1) / RT_THREADS);
    hipLaunchKernelGGL(retrieval_finalize_kernel, dim3(blocks), dim3(RT_THREADS), 0, stream, workspace, (int)B, n_elems,
                       d_scores, loss, kl, aux);
    return hipGetLastError();
}

hipError_t launch_retrieval_backward(const void* q, const void* s, int enc_dtype, int sections_3d, int64_t B, int64_t D,
                                     int64_t H, const float* d_scores, const float* grad_out, float* dq, float* ds,
                                     hipStream_t stream) {
    const size_t lds = (size_t)D * sizeof(float);
    if (!sections_3d) {
        // dq[b,h] = go * sum_d dS[b,d] s[d,h]   : A = dS [B,D] (k = d fast), B = s [D,H] (n = h fast)
        // K = D is long and the output tiny: split K over workgroups into slabs parked in the (not yet written) `ds`
        // buffer, then sum them in order
        int n_splits = (int)std::min<int64_t>(16, D / std::max<int64_t>(B, 1));
        if (n_splits < 2 || D < 512) n_splits = 1;
        hipError_t e;
        if (n_splits == 1) {
            e = launch_small_gemm(2, enc_dtype, d_scores, D, 1, s, H, 1, dq, H, (int)B, (int)H, (int)D, grad_out, stream);
        } else {
            e = launch_small_gemm(2, enc_dtype, d_scores, D, 1, s, H, 1, ds, H, (int)B, (int)H, (int)D, grad_out, stream, n_splits,
                                  B * H);
            if (e != hipSuccess) return e;
            const int64_t count = B * H;
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, ds, n_splits,
                               count, dq);
            e = hipGetLastError();
        }
        if (e != hipSuccess) return e;
        // ds[d,h] = go * sum_b dS[b,d] q[b,h]   : A = dS^T [D,B] (m = d fast), B = q [B,H] (n = h fast)
        return launch_small_gemm(2, enc_dtype, d_scores, 1, D, q, H, 1, ds, H, (int)D, (int)H, (int)B, grad_out, stream);
    }
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

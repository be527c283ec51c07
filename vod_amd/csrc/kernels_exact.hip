// Exact-fp32 mode of the store (round 5): results equal a float32 brute force over the UNROUNDED inputs.
//
// The reference keeps and searches float32 (faiss IndexFlat filled by `index.add(vectors.astype(float32))`,
// /root/reference/src/vod_search/faiss_search/build.py:65-73; `faiss_index.search(float32 queries, k)`, server.py:71-72,81-84; the
// vectors on disk are `<f4`, src/vod_tools/ts_factory/ts_factory.py:71-76).  The MFMA scan of this library works on fp16 / bf16
// copies; with VODHIP_EXACT_F32 the store keeps the float32 rows in a second plane and the scan becomes a FILTER with a rigorous
// error bound:
//
//   s(q, x)   = the float32 dot product of the unrounded query and row, summed in the fixed order of `exact_dot` below - the
//               score this mode returns, whatever path, shard or batch computed it;
//   s~(q, x)  = what the scan computes: fp32-accumulated products of the ROUNDED query q~ and row x~;
//   |s - s~| <= eps_q = |q - q~| max|x| + |q~| max|x - x~| + (accumulation slack)       (Cauchy-Schwarz, norms taken from the
//               actual data: the row maxima are tracked at ingest, the query norms computed per query).
//
//   1. the scan returns its top-k' list by s~ (k' > k: `exact_kx`), rows gathered from the float32 plane are re-scored, the k
//      best by (s desc, id asc) leave;
//   2. a row outside the list has s~ <= s~_(k'), hence s <= s~_(k') + eps_q: if that is below the k-th re-scored score the
//      result is complete - checked per query ON THE DEVICE;
//   3. a query that fails the check is searched again as a BAND pass: every row with s~ >= s_(k) - eps_q (s_(k) = the k-th
//      re-scored score, a lower bound of the true k-th best) is a candidate and is re-scored - complete by construction;
//      candidate lists that overflow split the pass into more stages, down to dense chunks that cannot overflow.
//
// One kernel does the re-scoring in both shapes (a top-k' list / a stage's candidate list folded into a running top-k).
//
// Round 6 - input range and outliers.  (a) finite float32 values beyond the store dtype's range SATURATE in the scan copy
// (mips_common.h: `saturate_for_store`), rows and queries alike, so x - x~ and q - q~ stay finite and enter the bound.  (b) the maxima
// of the bound are taken over the ORDINARY rows only: every row's |x|^2 and |x - x~|^2 are kept (8 bytes per row); before the first
// search after rows were added the maxima are re-derived: with M the largest value, the rows above M / 4^j for the largest j that
// leaves at most EXACT_MAX_OUTLIERS of them are OUTLIERS (separately for the two statistics; j = 0: none).  Outliers are not bounded,
// they are SCORED: every query re-scores all of them next to its list (LIST) / its first stage's candidates (CAND), and a list entry
// or candidate that is an outlier is dropped in favour of that copy.  One row of norm 1e5 in a store of norm-30 rows therefore costs
// every query one more float32 dot product instead of widening every query's eps 3000-fold (round 5: every query fell to the BAND pass).
#include <algorithm>

#include "mips_common.h"
#include "wg_sort.h"

namespace vodhip {

namespace {

// every lane ends with the same value: the pairs of a butterfly step add the same two numbers, and the step order is fixed
__device__ __forceinline__ float wave_sum_fixed(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// THE score function of exact mode: lane l accumulates columns 4l .. 4l+3 (mod 256) in increasing column order with one fma chain
// per component, the four chains are added as (0 + 1) + (2 + 3), the 64 lane sums by the butterfly above.  Zero padded columns add
// exact zeros, so the value depends on the row and the query alone.  NC rows at once: their loads are in flight together.
template <int NC>
__device__ __forceinline__ void exact_dot(const float* __restrict__ plane, int64_t stride, const int (&rows)[NC], const float* qs,
                                          int dim_pad, int lane, float (&out)[NC]) {
    float acc[NC][4];
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f;
    for (int c = lane * 4; c < dim_pad; c += 256) {
        const float4 qv = *reinterpret_cast<const float4*>(qs + c);
        float4 xv[NC];
#pragma unroll
        for (int n = 0; n < NC; ++n)
            xv[n] = rows[n] >= 0 ? *reinterpret_cast<const float4*>(plane + (size_t)rows[n] * stride + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            acc[n][0] = __builtin_fmaf(xv[n].x, qv.x, acc[n][0]);
            acc[n][1] = __builtin_fmaf(xv[n].y, qv.y, acc[n][1]);
            acc[n][2] = __builtin_fmaf(xv[n].z, qv.z, acc[n][2]);
            acc[n][3] = __builtin_fmaf(xv[n].w, qv.w, acc[n][3]);
        }
    }
#pragma unroll
    for (int n = 0; n < NC; ++n) out[n] = wave_sum_fixed((acc[n][0] + acc[n][1]) + (acc[n][2] + acc[n][3]));
}

// the same function with every load of the NC rows issued before the first fma (dim_pad <= 256 * NI): one memory latency per group of
// rows instead of one per 256 columns.  Same fma order per component -> the same bits as exact_dot.
template <int NC, int NI>
__device__ __forceinline__ void exact_dot_preload(const float* __restrict__ plane, int64_t stride, const int (&rows)[NC], const float* qs,
                                                  int dim_pad, int lane, float (&out)[NC]) {
    float4 xv[NI][NC];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int c = lane * 4 + it * 256;
#pragma unroll
        for (int n = 0; n < NC; ++n)
            xv[it][n] = (c < dim_pad && rows[n] >= 0) ? *reinterpret_cast<const float4*>(plane + (size_t)rows[n] * stride + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float acc[NC][4];
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < dim_pad) {
            const float4 qv = *reinterpret_cast<const float4*>(qs + c);
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                acc[n][0] = __builtin_fmaf(xv[it][n].x, qv.x, acc[n][0]);
                acc[n][1] = __builtin_fmaf(xv[it][n].y, qv.y, acc[n][1]);
                acc[n][2] = __builtin_fmaf(xv[it][n].z, qv.z, acc[n][2]);
                acc[n][3] = __builtin_fmaf(xv[it][n].w, qv.w, acc[n][3]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NC; ++n) out[n] = wave_sum_fixed((acc[n][0] + acc[n][1]) + (acc[n][2] + acc[n][3]));
}

__device__ __forceinline__ float round_to_store(float f, int store_dtype) {  // what mips_prepare_kernel stages for the scan
    float r;
    (void)store_bits(f, store_dtype, &r);
    return r;
}

constexpr int XT = 512;  // threads of the re-scoring workgroup
#ifndef VODHIP_EXACT_NCR
#define VODHIP_EXACT_NCR 4  // (8 measured equal on C2 / C4-shard exact-f32, round 6: the gather is at its practical rate)
#endif
constexpr int NCR = VODHIP_EXACT_NCR;   // float32 rows a wave has in flight, every column of them requested at once: 8 waves x 4 rows = 32 rows x dim floats per query

// descending sort of P (a power of two, 64 .. 4096) keys in LDS: up to 512 keys ONE wavefront sorts them in its registers (shuffles only,
// no barrier: the 28-45 barrier-separated LDS stages this replaces were a third of the re-scoring launch), above that the
// workgroup-wide register network of wg_sort.h (3 of its levels touch LDS)
__device__ __forceinline__ void sort_desc_lds(key_t64* keys, int P, int tid) {
    __syncthreads();
    switch (P) {
        case 64: if (tid < 64) wave_sort_regs<1, true>(keys, tid); break;
        case 128: if (tid < 64) wave_sort_regs<2, true>(keys, tid); break;
        case 256: if (tid < 64) wave_sort_regs<4, true>(keys, tid); break;
        case 512: if (tid < 64) wave_sort_regs<8, true>(keys, tid); break;
        case 1024: wg_sort_regs<XT, 2, true>(keys, tid); break;
        case 2048: wg_sort_regs<XT, 4, true>(keys, tid); break;
        default: wg_sort_regs<XT, 8, true>(keys, tid); break;  // 4096
    }
    __syncthreads();
}

}  // namespace

// ---- ingest ---------------------------------------------------------------------------------------------------------------
// One wavefront per row: the float32 plane, the fp16 / bf16 plane (round-to-nearest-even) and the two row statistics the error
// bound needs - max |x|^2 and max |x - x~|^2 over the rows ever added (bit patterns of non-negative floats order like integers:
// atomicMax on the words).  Rows with a non-finite norm do not move the maxima (their scores are inf / NaN in any arithmetic).
template <int SRC, int DST>
__global__ __launch_bounds__(256) void ingest_exact_kernel(const void* __restrict__ src, int64_t n_rows, int64_t dim,
                                                           uint16_t* __restrict__ dst16, float* __restrict__ dst32, int64_t stride,
                                                           unsigned int* __restrict__ stats, float* __restrict__ row_n2,
                                                           float* __restrict__ row_d2) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    float n2 = 0.f, d2 = 0.f;
    const bool vec = (dim & 7) == 0 && ((uintptr_t)src & 31) == 0;  // whole 8-element groups, 16/32-byte aligned: vector loads
    for (int64_t c0 = (int64_t)lane * 8; c0 < stride; c0 += 512) {
        uint16_t h[8];
        float f[8];
        float in[8];
        if (vec && c0 < dim) {
            if constexpr (SRC == 2) {
                const float4 a = *reinterpret_cast<const float4*>((const float*)src + row * dim + c0);
                const float4 b = *reinterpret_cast<const float4*>((const float*)src + row * dim + c0 + 4);
                in[0] = a.x, in[1] = a.y, in[2] = a.z, in[3] = a.w, in[4] = b.x, in[5] = b.y, in[6] = b.z, in[7] = b.w;
            } else {
                const uint4 raw = *reinterpret_cast<const uint4*>((const uint16_t*)src + row * dim + c0);
                const uint16_t* hw = reinterpret_cast<const uint16_t*>(&raw);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (SRC == 0) in[e] = (float)__builtin_bit_cast(_Float16, hw[e]);
                    else in[e] = (float)__builtin_bit_cast(__bf16, hw[e]);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t c = c0 + e;
                float v = 0.f;
                if (c < dim) {
                    if constexpr (SRC == 2) v = ((const float*)src)[row * dim + c];
                    else if constexpr (SRC == 0) v = (float)(((const _Float16*)src)[row * dim + c]);
                    else v = (float)(((const __bf16*)src)[row * dim + c]);
                }
                in[e] = v;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = in[e];
            f[e] = v;
            float r;
            h[e] = store_bits(v, DST, &r);  // (finite values beyond the store dtype's range saturate: x - x~ stays finite)
            n2 = __builtin_fmaf(v, v, n2);
            d2 = __builtin_fmaf(v - r, v - r, d2);
        }
        *reinterpret_cast<uint4*>(dst16 + row * stride + c0) = *reinterpret_cast<const uint4*>(h);
        *reinterpret_cast<float4*>(dst32 + row * stride + c0) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(dst32 + row * stride + c0 + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
    n2 = wave_sum_fixed(n2);
    d2 = wave_sum_fixed(d2);
    // (atomics on ONE address execute one after the other, ~3 ns each: only a row that would raise a maximum issues one - after the
    // first few hundred rows almost none does)
    if (lane == 0) {
        row_n2[row] = n2;
        row_d2[row] = d2;
    }
    if (lane == 0 && n2 < __builtin_inff()) {  // (false for NaN too)
        if (__float_as_uint(n2) > __hip_atomic_load(stats + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(stats + 0, __float_as_uint(n2));
        if (d2 < __builtin_inff() && __float_as_uint(d2) > __hip_atomic_load(stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(stats + 1, __float_as_uint(d2));
    }
}

hipError_t launch_ingest_exact(const void* src, int src_dtype, int64_t n_rows, int64_t dim, void* dst16, int dst_dtype, float* dst32,
                               int64_t stride, unsigned int* stats, float* row_n2, float* row_d2, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n_rows + 3) / 4);
#define VOD_ING(S, D)                                                                                                          \
    if (src_dtype == S && dst_dtype == D) {                                                                                    \
        hipLaunchKernelGGL((ingest_exact_kernel<S, D>), dim3(blocks), dim3(256), 0, stream, src, n_rows, dim, (uint16_t*)dst16, \
                           dst32, stride, stats, row_n2, row_d2);                                                              \
        return hipGetLastError();                                                                                              \
    }
    VOD_ING(0, 0) VOD_ING(1, 0) VOD_ING(2, 0) VOD_ING(0, 1) VOD_ING(1, 1) VOD_ING(2, 1)
#undef VOD_ING
    return hipErrorInvalidValue;
}

// ---- the maxima of the bound over the ORDINARY rows, and the outliers ----------------------------------------------------------
// words of the statistics buffer (ExactStatsWords in vodhip_internal.h): [0] max n2 bits, [1] max d2 bits (atomicMax at ingest),
// [2] ordinary max n2, [3] ordinary max d2, [4] number of outliers, [5] cut n2, [6] cut d2 (float bits), [8 + j] / [24 + j]: rows of
// level j of n2 / d2, [64 ...] the outliers' rows.
namespace {

constexpr int LEVELS = 15;  // cuts M / 4^1 .. M / 4^15 (a factor 2^15 in the norm)

// the level of a value: the smallest j >= 1 with v > M / 4^j (LEVELS + 1: below every cut).  Power-of-two scaling: exact.
__device__ __forceinline__ int level_of(float v, float m) {
    float t = m;
#pragma unroll 1
    for (int j = 1; j <= LEVELS; ++j) {
        t *= 0.25f;
        if (v > t) return j;
    }
    return LEVELS + 1;
}

__global__ __launch_bounds__(256) void exact_stats_levels_kernel(const float* __restrict__ row_n2, const float* __restrict__ row_d2,
                                                                 int64_t n, unsigned int* __restrict__ st) {
    __shared__ unsigned int hist[2][LEVELS + 2];
    if (threadIdx.x < 2 * (LEVELS + 2)) (&hist[0][0])[threadIdx.x] = 0u;
    __syncthreads();
    const float mn = __uint_as_float(st[EXS_MAX_N2]), md = __uint_as_float(st[EXS_MAX_D2]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = row_n2[i], b = row_d2[i];
        if (!(a < __builtin_inff())) continue;  // rows with a non-finite norm never moved the maxima: not rows of the bound
        const int ja = level_of(a, mn);
        if (ja <= LEVELS) atomicAdd(&hist[0][ja], 1u);
        if (b < __builtin_inff()) {
            const int jb = level_of(b, md);
            if (jb <= LEVELS) atomicAdd(&hist[1][jb], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x <= LEVELS) {
        if (hist[0][threadIdx.x]) atomicAdd(st + EXS_HIST_N2 + threadIdx.x, hist[0][threadIdx.x]);
        if (hist[1][threadIdx.x]) atomicAdd(st + EXS_HIST_D2 + threadIdx.x, hist[1][threadIdx.x]);
    }
}

// the deepest cut that leaves at most EXACT_MAX_OUTLIERS rows above it, per statistic (j = 0: no outliers, the cut is the maximum)
__global__ void exact_stats_pick_kernel(unsigned int* __restrict__ st) {
    if (threadIdx.x >= 2) return;
    const int which = threadIdx.x;
    const unsigned int* hist = st + (which ? EXS_HIST_D2 : EXS_HIST_N2);
    float cut = __uint_as_float(st[which ? EXS_MAX_D2 : EXS_MAX_N2]);
    unsigned int above = 0;
    float t = cut;
    for (int j = 1; j <= LEVELS; ++j) {
        above += hist[j];
        t *= 0.25f;
        if (above > (unsigned)EXACT_MAX_OUTLIERS) break;
        cut = t;
    }
    st[which ? EXS_CUT_D2 : EXS_CUT_N2] = __float_as_uint(cut);
}

__global__ __launch_bounds__(256) void exact_stats_collect_kernel(const float* __restrict__ row_n2, const float* __restrict__ row_d2,
                                                                  int64_t n, unsigned int* __restrict__ st) {
    const float cn = __uint_as_float(st[EXS_CUT_N2]), cd = __uint_as_float(st[EXS_CUT_D2]);
    float mn = 0.f, md = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = row_n2[i], b = row_d2[i];
        if (!(a < __builtin_inff())) continue;
        if (a > cn || (b < __builtin_inff() && b > cd)) {
            const unsigned int slot = atomicAdd(st + EXS_N_OUT, 1u);
            if (slot < 2u * EXACT_MAX_OUTLIERS) st[EXS_OUT_ROWS + slot] = (unsigned int)i;
        } else {
            mn = fmaxf(mn, a);
            if (b < __builtin_inff()) md = fmaxf(md, b);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mn = fmaxf(mn, __shfl_xor(mn, m, 64));
        md = fmaxf(md, __shfl_xor(md, m, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (__float_as_uint(mn) > __hip_atomic_load(st + EXS_ORD_N2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(st + EXS_ORD_N2, __float_as_uint(mn));
        if (__float_as_uint(md) > __hip_atomic_load(st + EXS_ORD_D2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(st + EXS_ORD_D2, __float_as_uint(md));
    }
}

}  // namespace

hipError_t launch_exact_stats(const float* row_n2, const float* row_d2, int64_t n_rows, unsigned int* stats, hipStream_t stream) {
    if (hipError_t e = hipMemsetAsync(stats + EXS_ORD_N2, 0, (EXS_WORDS - EXS_ORD_N2) * sizeof(unsigned int), stream); e != hipSuccess) return e;
    if (n_rows <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<int64_t>(1024, (n_rows + 255) / 256);
    hipLaunchKernelGGL(exact_stats_levels_kernel, dim3(blocks), dim3(256), 0, stream, row_n2, row_d2, n_rows, stats);
    hipLaunchKernelGGL(exact_stats_pick_kernel, dim3(1), dim3(64), 0, stream, stats);
    hipLaunchKernelGGL(exact_stats_collect_kernel, dim3(blocks), dim3(256), 0, stream, row_n2, row_d2, n_rows, stats);
    return hipGetLastError();
}

// ---- re-scoring -----------------------------------------------------------------------------------------------------------
// One workgroup per query.  LDS: the float32 query [dim_pad] + P keys.
//   LIST  candidates = the scan's top-kx list (list_s / list_i rows of the caller's batch); output = the k best by (s, id), the
//         per-query bound eps, and the completeness flag (2. above).
//   CAND  candidates = this stage's candidate list of the BAND pass (cand / cnt, or a dense chunk of dense_n slots); they are
//         folded into the running exact top-k kept in the caller's output rows (EXACT_FIRST: the rows are taken as empty); the
//         scan threshold of the next stages rises to (k-th exact score - eps) when that is higher; a list that lost
//         candidates flags the query (the host splits the pass).
__global__ __launch_bounds__(XT) void exact_rescore_kernel(ExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = reinterpret_cast<float*>(smem);
    key_t64* kb = reinterpret_cast<key_t64*>(smem + (size_t)a.dim_pad * sizeof(float));
    int* row_of = reinterpret_cast<int*>(smem + (size_t)a.dim_pad * sizeof(float) + (size_t)a.P * sizeof(key_t64));  // [P] rows of the chunk's candidates
    __shared__ float red[2 * (XT / 64)];
    const int r = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t qo = a.q_map ? (int64_t)a.q_map[r] : (int64_t)r;
    const bool cand_mode = (a.mode & EXACT_CAND) != 0;
    const bool first = (a.mode & EXACT_FIRST) != 0;
    const int k = a.k;

    // 1. the query, unrounded, and the two norms of the bound
    float n2 = 0.f, d2 = 0.f;
    for (int c = tid; c < a.dim_pad; c += XT) {
        float f = 0.f;
        if (c < a.dim) {
            if (a.q_dtype == 2) f = ((const float*)a.q_src)[qo * a.dim + c];
            else if (a.q_dtype == 0) f = (float)(((const _Float16*)a.q_src)[qo * a.dim + c]);
            else f = (float)(((const __bf16*)a.q_src)[qo * a.dim + c]);
        }
        const float fr = round_to_store(f, a.store_dtype);
        qs[c] = f;
        n2 = __builtin_fmaf(fr, fr, n2);
        d2 = __builtin_fmaf(f - fr, f - fr, d2);
    }
    n2 = wave_sum_fixed(n2);
    d2 = wave_sum_fixed(d2);
    if (lane == 0) {
        red[wave] = n2;
        red[XT / 64 + wave] = d2;
    }
    __syncthreads();
    n2 = d2 = 0.f;
#pragma unroll
    for (int w = 0; w < XT / 64; ++w) {
        n2 += red[w];
        d2 += red[XT / 64 + w];
    }
    // |s - s~| <= |q - q~| |x| + |q~| |x - x~|, plus the rounding of the two fp32 summations (each below dim_pad * 2^-23 * |q| |x|
    // whatever the order), everything inflated by 2^-9 for the float arithmetic of the norms themselves
    // (the maxima over the ORDINARY rows: the outliers are scored, not bounded - below)
    const float xn = __builtin_sqrtf(a.ord_n2), dxn = __builtin_sqrtf(a.ord_d2);
    const float qn = __builtin_sqrtf(n2), dqn = __builtin_sqrtf(d2);
    const float eps = (dqn * xn + qn * dxn + 2.f * (float)a.dim_pad * 1.1920929e-7f * (qn + dqn) * xn) * 1.002f;

    // 2. candidates, P - KR at a time behind the running top-k (KR slots)
    const int KR = cand_mode ? a.kr : 0;
    if (cand_mode)
        for (int c = tid; c < KR; c += XT) {
            key_t64 key = 0ull;
            if (!first && c < k) {
                const int64_t id = a.out_ids[qo * k + c];
                if (id >= 0) key = make_key(a.out_scores[qo * k + c], (unsigned)(id - a.id_base));
            }
            kb[c] = key;
        }
    int n_total;
    bool lost = false;
    if (cand_mode) {
        unsigned n = a.dense_n >= 0 ? (unsigned)a.dense_n : a.cnt[(size_t)r * CNT_STRIDE];
        if (n > (unsigned)a.cap) {
            lost = true;
            n = (unsigned)a.cap;
        }
        n_total = (int)n;
    } else {
        n_total = a.kx;
    }
    // the outliers of the bound are candidates of every query: behind its list (LIST) / behind its first stage's candidates (CAND)
    const int n_listed = n_total;
    const int n_out = (!cand_mode || first) ? a.n_out : 0;
    n_total += n_out;
    const int CH = a.P - KR;
    int done = 0;
    do {
        const int take = min(CH, n_total - done);
        // the chunk's candidate rows first, in ONE coalesced pass (a wave fetching its candidates' ids one by one pays a memory latency each)
        for (int j = tid; j < take; j += XT) {
            int row = -1;
            if (done + j >= n_listed) {  // an outlier: every query scores it (subject to the query's subset filter)
                row = (int)a.out_rows[done + j - n_listed];
                if (a.row_label != nullptr) {
                    const int lab = a.row_label[row];
                    bool any = false, ok = false;
                    for (int s = 0; s < a.n_qlab; ++s) {
                        const int ql = a.q_label[(size_t)qo * a.n_qlab + s];
                        any |= ql != -1;
                        ok |= ql == lab;
                    }
                    if (!(ok || !any)) row = -1;
                }
            } else {
                if (cand_mode) {
                    const key_t64 key = a.cand[(size_t)r * a.cap + done + j];
                    if (key != 0ull) row = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
                } else {
                    const int64_t id = a.list_i[qo * a.kx + done + j];
                    if (id >= 0) row = (int)id;
                }
                // a listed row that is an outlier leaves: its one copy is the injected one (LIST: behind the list; CAND: in the first stage)
                // (the predicate of exact_stats_collect_kernel: rows with a non-finite norm are no outliers - they keep their list entry)
                if (a.n_out > 0 && row >= 0) {
                    const float a2 = a.row_n2[row], b2 = a.row_d2[row];
                    if (a2 < __builtin_inff() && (a2 > a.cut_n2 || (b2 < __builtin_inff() && b2 > a.cut_d2))) row = -1;
                }
            }
            row_of[j] = row;
        }
        __syncthreads();
        const int ni = (a.dim_pad + 255) >> 8;
        for (int j0 = wave * NCR; j0 < take; j0 += NCR * (XT / 64)) {
            int rows[NCR];
#pragma unroll
            for (int n = 0; n < NCR; ++n) rows[n] = __builtin_amdgcn_readfirstlane(j0 + n < take ? row_of[j0 + n] : -1);
            float s[NCR];
            switch (ni) {
                case 1: exact_dot_preload<NCR, 1>(a.plane, a.stride, rows, qs, a.dim_pad, lane, s); break;
                case 2: exact_dot_preload<NCR, 2>(a.plane, a.stride, rows, qs, a.dim_pad, lane, s); break;
                case 3: exact_dot_preload<NCR, 3>(a.plane, a.stride, rows, qs, a.dim_pad, lane, s); break;
                case 4: exact_dot_preload<NCR, 4>(a.plane, a.stride, rows, qs, a.dim_pad, lane, s); break;
                default: exact_dot<NCR>(a.plane, a.stride, rows, qs, a.dim_pad, lane, s); break;
            }
            if (lane == 0) {
#pragma unroll
                for (int n = 0; n < NCR; ++n)
                    if (j0 + n < take) kb[KR + j0 + n] = (rows[n] >= 0 && s[n] == s[n]) ? make_key(s[n], (unsigned)rows[n]) : 0ull;  // NaN never enters
            }
        }
        int P_eff = 64;
        while (P_eff < KR + take) P_eff <<= 1;
        for (int c = KR + take + tid; c < P_eff; c += XT) kb[c] = 0ull;
        sort_desc_lds(kb, P_eff, tid);  // (starts and ends with a barrier)
        done += take;
    } while (done < n_total);

    // 3. results
    const key_t64 kth = kb[k - 1];  // (k <= KR in CAND mode, k <= kx <= P in LIST mode; slots behind the candidates are zero)
    const float s_k = kth ? unflip_f32((unsigned)(kth >> 32)) : -__builtin_inff();
    for (int c = tid; c < k; c += XT) {
        const key_t64 key = kb[c];
        a.out_scores[qo * k + c] = key ? unflip_f32((unsigned)(key >> 32)) : -__builtin_inff();
        a.out_ids[qo * k + c] = key ? a.id_base + (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : -1;
    }
    // LIST: how long a list this query needed - the entries whose scan score is within eps of the k-th exact score (the list is sorted
    // by scan score: its entries behind those cannot be hits); the host sizes the next searches' lists by the maximum over the queries
    __shared__ int need_s;
    if (!cand_mode) {
        if (tid == 0) need_s = 0;
        __syncthreads();
        for (int j = tid; j < a.kx; j += XT)
            if (a.list_i[qo * a.kx + j] >= 0 && a.list_s[qo * a.kx + j] + eps >= s_k) atomicMax(&need_s, j + 1);
        __syncthreads();
    }
    if (tid == 0) {
        if (cand_mode) {
            if (kth) {  // every row of the true top-k has s >= s_k, hence s~ >= s_k - eps
                const float t = s_k - eps;
                if (t > a.thr_s[r]) {
                    a.thr_s[r] = t;
                    a.thr_key[r] = (key_t64)flip_f32(t) << 32;  // low word 0: every row with s~ == t still passes `key > thr_key`
                }
            }
            a.cnt[(size_t)r * CNT_STRIDE] = 0;
            if (lost) {
                atomicOr(a.flag_word, 1u);
                if (a.flag_q) a.flag_q[r] = 1u;
            }
        } else {
            a.eps[qo] = eps;
            // rows outside the list have s~ <= the list's last s~, i.e. s <= that + eps
            const bool full = a.list_i[qo * a.kx + a.kx - 1] >= 0;
            const bool complete = !full || (a.list_s[qo * a.kx + a.kx - 1] + eps < s_k);
            a.flag_q[qo] = complete ? 0u : 1u;
            if (!complete) atomicOr(a.flag_word, 1u);
            if ((unsigned)need_s > __hip_atomic_load(a.flag_word + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.flag_word + 1, (unsigned)need_s);
        }
    }
}

hipError_t launch_exact_rescore(const ExactArgs& a, int64_t nq, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    const int bytes = (int)((size_t)a.dim_pad * sizeof(float) + (size_t)a.P * (sizeof(key_t64) + sizeof(int)));
    if (hipError_t e = allow_dynamic_lds((const void*)exact_rescore_kernel, bytes); e != hipSuccess) return e;
    hipLaunchKernelGGL(exact_rescore_kernel, dim3((unsigned)nq), dim3(XT), bytes, stream, a);
    return hipGetLastError();
}

}  // namespace vodhip

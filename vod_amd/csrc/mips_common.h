// Device helpers shared by the MIPS kernels (kernels_mips.hip): result-key packing, LDS-DMA, counted waits,
// MFMA wrappers, the subset predicate and the direct (global-atomic) survivor append.
#pragma once
#include "vodhip_internal.h"

namespace vodhip {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define VOD_AS1 __attribute__((address_space(1)))
#define VOD_AS3 __attribute__((address_space(3)))

// ---- key packing --------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int flip_f32(float s) {
    unsigned int u = __float_as_uint(s + 0.0f);  // -0.0 -> +0.0 so that equal floats get equal keys
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unflip_f32(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ key_t64 make_key(float s, unsigned int local_row) {
    return ((key_t64)flip_f32(s) << 32) | (key_t64)(0xFFFFFFFFu - local_row);
}

// ---- rounding to the store dtype ------------------------------------------------------------------
// Finite float32 values beyond the store dtype's range SATURATE to its largest finite value instead of rounding to +-inf (fp16:
// |v| >= 65520 would; bf16: |v| > 3.3895e38).  An infinite scan copy of a finite value makes every scan score of the row +-inf / NaN and
// the rounding error x - x~ infinite: exact mode's error bound (kernels_exact.hip) could not cover such a row (round 5: a true top-k hit
// with a component of 7e4 in an fp16 store was lost).  Saturated, x - x~ is finite, enters the row statistics, and the row is handled
// as an outlier of the bound.  NaN and +-inf inputs pass through unchanged.
__device__ __forceinline__ float saturate_for_store(float v, int store_dtype) {
    const float lim = store_dtype == 0 ? 65504.f : 3.3895313892515355e38f;
    if (v > lim && v < __builtin_inff()) return lim;
    if (v < -lim && v > -__builtin_inff()) return -lim;
    return v;
}
// the 16 bits the store / the rounded query holds for float32 `v`, and the value they stand for
__device__ __forceinline__ uint16_t store_bits(float v, int store_dtype, float* rounded) {
    v = saturate_for_store(v, store_dtype);
    if (store_dtype == 0) {
        const _Float16 h = (_Float16)v;
        *rounded = (float)h;
        return __builtin_bit_cast(uint16_t, h);
    }
    const __bf16 h = (__bf16)v;
    *rounded = (float)h;
    return __builtin_bit_cast(uint16_t, h);
}

template <int DT>
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (DT == 0) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}
template <int DT>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (DT == 0) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}

// 16 bytes per lane HBM/L2 -> LDS without a VGPR round trip; the LDS address is wave-uniform base + lane * 16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const VOD_AS1 void*)gsrc, (VOD_AS3 void*)lds_dst, 16, 0, 0);
}

// the same with a cache-policy immediate (diagnostic builds): 1 = sc0, 2 = nt (MI355X_MICROARCH "nt-weights")
template <int AUX>
__device__ __forceinline__ void glds16_aux(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const VOD_AS1 void*)gsrc, (VOD_AS3 void*)lds_dst, 16, 0, AUX);
}

// max of three / two floats in ONE instruction, no operand canonicalisation (a quiet NaN operand is ignored)
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float fmaxf_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // counted wait: all but the N youngest vector-memory operations of this wave are complete
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N == 0, "add the literal");
}

// First corpus row of x-tile `xt` (BM rows each; 256 % BM == 0) of a FILTER launch that starts at row `row_begin` (a multiple of 256):
// stage positions -> super-tiles through FilterExtra's low-discrepancy order, sub-tiles of a super-tile stay together.
__device__ __forceinline__ int filter_tile_row0(const FilterExtra& ex, int row_begin, int xt, int BM) {
    if (ex.perm_mod <= 0) return row_begin + xt * BM;
    const int per = 256 / BM;
    const unsigned long long pos = (unsigned long long)(row_begin / 256 + xt / per);
    const int super = (int)((pos * (unsigned long long)ex.perm_mul) % (unsigned long long)ex.perm_mod);
    return super * 256 + (xt % per) * BM;
}

// Subset filter (the `subset_ids` of the reference's SearchClient.search, honoured by its Elasticsearch / Qdrant engines,
// src/vod_search/es_search/client.py:185-191, qdrant_search/client.py:124-136, and ignored by its faiss client,
// faiss_search/client.py:67-72): a row is eligible for query q when q lists no label or lists the row's label.
__device__ __forceinline__ bool subset_allows(const FilterExtra& ex, int q, int row) {
    if (ex.row_label == nullptr) return true;
    const int lab = ex.row_label[row];
    const int qq = ex.q_map ? ex.q_map[q] : q;
    bool any = false, ok = false;
#pragma unroll 1
    for (int s = 0; s < ex.n_qlab; ++s) {
        const int ql = ex.q_label[(size_t)qq * ex.n_qlab + s];
        any |= ql != -1;  // -1 = empty slot; any other value (incl. an unknown id mapped to -2) restricts the query
        ok |= ql == lab;
    }
    return ok || !any;
}

// one survivor -> the query's global candidate list (exact-key test against the running k-th best, subset test)
template <bool SUBSET>
__device__ __forceinline__ void emit_candidate(key_t64 key, int q, const key_t64* __restrict__ thr_key,
                                               key_t64* __restrict__ cand, unsigned int* __restrict__ cnt, int cap,
                                               unsigned int* __restrict__ overflow, const FilterExtra& ex) {
    bool ok = key > thr_key[q];
    if constexpr (SUBSET) ok = ok && subset_allows(ex, q, (int)(0xFFFFFFFFu - (unsigned)key));
    if (ok) {
        const unsigned slot = atomicAdd(&cnt[(size_t)q * CNT_STRIDE], 1u);
        if (slot < (unsigned)cap)
            cand[(size_t)q * cap + slot] = key;
        else
            atomicOr(overflow, 1u);
    }
}

// Append the survivors among NV scores of ONE query held by this lane: count first, reserve the slots with ONE
// returning atomic, then write the keys (a chain of per-hit atomics costs a memory round trip each).
// val(i) / row(i) must be compile-time indexable.
template <int NV, bool SUBSET, typename ValFn, typename RowFn>
__device__ __forceinline__ void append_survivors(float thr, int q, int row_end, ValFn val, RowFn row,
                                                 const key_t64* __restrict__ thr_key, key_t64* __restrict__ cand,
                                                 unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow,
                                                 const FilterExtra& ex) {
    const key_t64 tk = thr_key[q];
    // opaque zero added to every row index: without it the compiler hoists the NV row keys and row-bound compares
    // (the same for every query block of the caller) out of this cold path into the caller's per-tile fast path
    int z = 0;
    asm volatile("" : "+s"(z));
    static_assert(NV <= 32, "survivor mask is 32 bits");
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float sc = val(i);
        const int rw = (row(i) + z);
        mask |= (sc >= thr && rw < row_end && make_key(sc, (unsigned)rw) > tk) ? (1u << i) : 0u;
    }
    if constexpr (SUBSET) {  // rolled loop over the threshold survivors (this instantiation only runs with row labels set)
        unsigned m2 = mask;
        while (m2) {
            const int i = __builtin_ctz(m2);
            m2 &= m2 - 1;
            if (!subset_allows(ex, q, (row(i) + z))) mask &= ~(1u << i);
        }
    }
    if (mask == 0) return;
    unsigned slot = atomicAdd(&cnt[(size_t)q * CNT_STRIDE], (unsigned)__builtin_popcount(mask));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (mask & (1u << i)) {
            if (slot < (unsigned)cap)
                cand[(size_t)q * cap + slot] = make_key(val(i), (unsigned)(row(i) + z));
            else
                atomicOr(overflow, 1u);
            ++slot;
        }
    }
}

}  // namespace vodhip

// ROUND-1 EXPERIMENT SNAPSHOT - compiled only with -DVODHIP_ABLATION (make ABLATION=1); never part of the production library.
//
// This is the round-1 kernel file as it stood, kept so that the measured-and-lost variants documented in
// profiles/README.md stay reproducible: tiles 2..8 (32x32x16 rings, ping-pong schedule, loader/consumer
// specialisation, non-persistent 16x16x32), the A3 (3 + 2 LDS slot) persistent layout, and every ABLATE / stamp build.
// Everything lives in namespace vodhip::experiments; the production kernels are in kernels_mips.hip.
#ifdef VODHIP_ABLATION
// Fused inner-product scoring + streaming top-k for gfx950 (MI355X, CDNA4).
//
// Replaces the arithmetic behind `faiss_index.search(query_vec, k)`
// (/root/reference/src/vod_search/faiss_search/server.py:72,84) for a Flat / inner-product index.
//
// Data layout in HBM
//   store X : [capacity][dim_pad] fp16|bf16, row-major, dim_pad % 64 == 0, rows >= ntotal are zero
//   queries : [nq_pad][dim_pad] same dtype (workspace), nq_pad % BN == 0, padding rows are zero
//
// mips_filter_kernel: one workgroup owns a BM(corpus rows) x BN(queries) score tile.  The K loop stages
// 64-deep slices of both operands into LDS with LDS-DMA (`global_load_lds_dwordx4`, 16 B per lane, the
// XOR swizzle applied on the per-lane SOURCE address so the LDS image stays lane-linear) and runs
// v_mfma_f32_32x32x16_{f16,bf16} with the corpus as the A operand and the queries as the B operand, so
// that in the accumulator a lane's column is ONE query and its 16 registers are 16 corpus rows.
// The score tile never leaves registers: each lane compares its scores with its query's running
// threshold (the score of the k-th best hit found in earlier chunks) and only survivors are appended
// (packed 64-bit keys) to the query's candidate list.  mips_select_kernel folds the candidates
// into the running sorted top-k and tightens the threshold between chunks.
//
// Roofline: 2*nq*N*D flop per batch on MFMA vs N*D*2 bytes of HBM; arithmetic intensity = nq flop/B.
#include "vodhip_internal.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include <type_traits>

namespace vodhip {
namespace experiments {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

// ------------------------------------------------------------------------------------------------
// key packing
// ------------------------------------------------------------------------------------------------
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

template <int DT>
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (DT == 0) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}

// Append the survivors among NV scores of ONE query held by this lane: count first, reserve the slots with ONE
// returning atomic, then write the keys (a chain of per-hit atomics costs a memory round trip each; early chunks,
// where ~10 % of the scores pass, spent most of their time there).  val(i) / row(i) must be compile-time indexable.
// Subset filter (the `subset_ids` of the reference's SearchClient.search, honoured by its Elasticsearch / Qdrant engines,
// src/vod_search/es_search/client.py:185-191, qdrant_search/client.py:124-136, and ignored by its faiss client,
// faiss_search/client.py:67-72): a row is eligible for query q when q lists no label or lists the row's label.
__device__ __forceinline__ bool subset_allows(const FilterExtra& ex, int q, int row) {
    if (ex.row_label == nullptr) return true;
    const int lab = ex.row_label[row];
    bool any = false, ok = false;
#pragma unroll 1
    for (int s = 0; s < ex.n_qlab; ++s) {
        const int ql = ex.q_label[(size_t)q * ex.n_qlab + s];
        any |= ql != -1;  // -1 = empty slot; any other value (incl. an unknown id mapped to -2) restricts the query
        ok |= ql == lab;
    }
    return ok || !any;
}

template <int NV, bool SUBSET, typename ValFn, typename RowFn>
__device__ __forceinline__ void append_survivors(float thr, int q, int row_end, ValFn val, RowFn row,
                                                 const key_t64* __restrict__ thr_key, key_t64* __restrict__ cand,
                                                 unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow,
                                                 const FilterExtra& ex) {
    const key_t64 tk = thr_key[q];
    // opaque zero added to every row index: without it the compiler hoists the NV row keys and row-bound compares
    // (the same for every query block of the caller) out of this cold path into the caller's per-tile fast path and
    // spills them - measured at ~1,000 cycles of every 256x256 tile
    int z = 0;
    asm volatile("" : "+s"(z));
    if constexpr (!SUBSET) {
        unsigned n_hit = 0;
        key_t64 key1 = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float sc = val(i);
            const int rw = (row(i) + z);
            const key_t64 key = make_key(sc, (unsigned)rw);
            const bool p = sc >= thr && rw < row_end && key > tk;
            n_hit += p ? 1u : 0u;
            key1 = p ? key : key1;
        }
        if (n_hit == 0) return;
        unsigned slot = atomicAdd(&cnt[q], n_hit);
        if (n_hit == 1) {  // the common case once the threshold is tight: no walk through NV predicated stores
            if (slot < (unsigned)cap)
                cand[(size_t)q * cap + slot] = key1;
            else
                atomicOr(overflow, 1u);
            return;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float sc = val(i);
            const int rw = (row(i) + z);
            if (sc >= thr && rw < row_end) {
                const key_t64 key = make_key(sc, (unsigned)rw);
                if (key > tk) {
                    if (slot < (unsigned)cap)
                        cand[(size_t)q * cap + slot] = key;
                    else
                        atomicOr(overflow, 1u);
                    ++slot;
                }
            }
        }
    } else {
        // subset filter: survivors of the threshold test are marked in a bit mask, then a ROLLED loop drops the
        // ineligible ones (keeps the hot code small; this instantiation only runs when row labels are set)
        static_assert(NV <= 32, "survivor mask is 32 bits");
        unsigned mask = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float sc = val(i);
            const int rw = (row(i) + z);
            mask |= (sc >= thr && rw < row_end && make_key(sc, (unsigned)rw) > tk) ? (1u << i) : 0u;
        }
        unsigned m2 = mask;
        while (m2) {
            const int i = __builtin_ctz(m2);
            m2 &= m2 - 1;
            if (!subset_allows(ex, q, (row(i) + z))) mask &= ~(1u << i);
        }
        if (mask == 0) return;
        unsigned slot = atomicAdd(&cnt[q], (unsigned)__builtin_popcount(mask));
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
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const AS1 void*)gsrc, (AS3 void*)lds_dst, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// filter kernel
// ------------------------------------------------------------------------------------------------
#ifdef VODHIP_ABLATION
// Diagnostic build only: in-kernel cycle stamps (s_memtime) of sampled workgroups, written to a buffer that no
// other code reads.  Layout: [sample 0..63][wave 0..7][slot 0..15][6 stamps].
__device__ unsigned long long g_stamps[64 * 8 * 16 * 6];
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // counted wait: all but the N youngest vector-memory operations of this wave are complete
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "add the literal");
}

// BK      : K elements per LDS stage (64 -> 128-B rows, 32 -> 64-B rows)
// NSTAGE  : LDS ring depth.  Slice t+NSTAGE-1 is being fetched while slice t is multiplied, so
//           (NSTAGE-2) whole slices stay in flight ACROSS the per-slice barrier (counted vmcnt, raw s_barrier).
template <int DT, int BM, int BN, int WM, int WN, int BK, int NSTAGE, bool DENSE, int ABLATE = 0, bool PINGPONG = false, bool SUBSET = false>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN * 64 * ((160 * 1024) / (NSTAGE * (BM + BN) * BK * 2))) / 256 >= 2 ? 2 : 1)
void mips_filter_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end,
    int n_xtiles, int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key,
    key_t64* __restrict__ cand, unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    const bool krot_on = (ex.flags & 1) != 0;  // FILTER_FLAG_KROT

    constexpr int NWAVES = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tile
    constexpr int MI = TM / 32, NJ = TN / 32;  // 32x32 blocks per wave
    constexpr int ROW_BYTES = BK * 2;
    constexpr int CH = ROW_BYTES / 16;         // 16-B chunks per LDS row (8 | 4)
    constexpr int RPI = 64 / CH;               // rows covered by one LDS-DMA wave-instruction (8 | 16)
    constexpr int KK = BK / 16;                // MFMA k-steps per slice (4 | 2)
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NA = BM / RPI / NWAVES;      // LDS-DMA wave-instructions per wave per slice
    constexpr int NB = BN / RPI / NWAVES;
    constexpr int G = NA + NB;
    static_assert(BM % (RPI * NWAVES) == 0 && BN % (RPI * NWAVES) == 0, "tile/wave mismatch");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the n_qtiles
    // workgroups that re-read one corpus tile are dealt to the same XCD back to back (L2 reuse only;
    // correctness does not depend on placement).
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jj = bid >> 3;
    const int qt = jj % n_qtiles;
    const int xt = (jj / n_qtiles) * 8 + xcd;
    if (xt >= n_xtiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int x0 = row_begin + xt * BM;  // first corpus row of the tile
    const int q0 = qt * BN;              // first query of the tile
#ifdef VODHIP_ABLATION
    unsigned long long t_entry = 0;
    if constexpr (ABLATE == 5) t_entry = stamp_now();
#endif

    // ---- per-lane LDS-DMA source pointers: lane -> (row = base + lane/CH, 16-B slot = lane%CH) ----
    // Slot s of row r holds logical chunk c = s ^ f(r), f(r) = (r>>1)&7 for 128-B rows, (r>>2)&3 for 64-B rows:
    // the 16 rows of a ds_read_b128 lane group then hit 16 distinct 16-B slots of the 256-B bank row.
    // The XOR is applied on the SOURCE address; the LDS image stays lane-linear as LDS-DMA requires.
    auto swz_of_row = [](int r) { return CH == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    const int st_row = lane / CH, st_slot = lane % CH;
    const char* a_src[NA];
    const char* b_src[NB];
#pragma unroll
    for (int t = 0; t < NA; ++t) {
        const int r = (wave * NA + t) * RPI + st_row;
        const int c = st_slot ^ swz_of_row(r);
        a_src[t] = (const char*)X + ((size_t)((ABLATE == 14 ? row_begin + (xt & 15) * BM : x0) + r) * dim_pad + c * 8) * 2;  // 14: timing-only, corpus tile always L2-hot
        if constexpr (ABLATE == 6)  // [8-row group][k-slice] blocks of 1 KiB: one wave-instruction = one contiguous KiB
            a_src[t] = (const char*)X + ((size_t)(x0 / RPI + wave * NA + t) * (dim_pad / BK)) * 1024 + lane * 16;
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int r = (wave * NB + t) * RPI + st_row;
        const int c = st_slot ^ swz_of_row(r);
        b_src[t] = (const char*)Q + ((size_t)(q0 + r) * dim_pad + c * 8) * 2;
        if constexpr (ABLATE == 6)
            b_src[t] = (const char*)Q + ((size_t)(q0 / RPI + wave * NB + t) * (dim_pad / BK)) * 1024 + lane * 16;
    }
    // K is a reduction, so its order is free: the n_qtiles workgroups that share a corpus tile (they run at the same
    // time on one XCD) walk the K slices in ROTATED orders.  Each slice is then fetched from HBM by exactly one of
    // them and found in L2 by the others a little later, instead of all of them queueing on the same HBM miss
    // Measured on C3: the DMA-only loop gets 17 % faster, the full kernel does not, and HBM traffic doubles (the
    // over-subscribed 4 MiB L2 evicts a slice before the last sharer reads it: FETCH_SIZE 15.4 -> 34 GB per batch),
    // so it is OFF by default and only kept as an experiment knob ("krot" parameter -> bit 0 of the `flags` argument).
    const int nk_ = dim_pad / BK;
    const int krot = krot_on ? (qt * nk_) / n_qtiles : 0;  // rotation is opt-in (flags bit 0): see note above
    // `part` of `nparts` of slice `ks`'s LDS-DMA into ring slot ks % NSTAGE
    auto stage_part = [&](int ks, int part, int nparts) {
        char* sa = smem + (ks % NSTAGE) * STAGE_BYTES;
        char* sb = sa + A_BYTES;
        const int kbyte = ABLATE == 6 ? ks * 1024 : ((ks + krot) % nk_) * ROW_BYTES;
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if ((u * nparts) / G != part) continue;
            if (u < NA) {
                if constexpr (ABLATE != 8) glds16(a_src[u] + kbyte, sa + (wave * NA + u) * RPI * ROW_BYTES);
            } else {
                if constexpr (ABLATE != 7) glds16(b_src[u - NA] + kbyte, sb + (wave * NB + (u - NA)) * RPI * ROW_BYTES);
            }
        }
    };

    // ---- fragment read addressing --------------------------------------------------------------
    // MFMA 32x32x16: lane l supplies A[row l&31][k = 8h..8h+7] and B[k = 8h..8h+7][col l&31], h = l>>5.
    // Logical chunk of k-step kk is 2*kk + h; every block starts at a multiple of 32 rows, so f(row) depends
    // on the lane only.
    const int fr = lane & 31, fh = lane >> 5;
    const int swz = swz_of_row(fr);
    const int a_row_off = (wm * TM + fr) * ROW_BYTES;
    const int b_row_off = A_BYTES + (wn * TN + fr) * ROW_BYTES;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = dim_pad / BK;
    if constexpr (ABLATE == 12) {
        if (wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(1);  // static priority for the later-dispatched half
    }
    // prologue: fill NSTAGE-1 ring slots
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage_part(s, 0, 1);

    // thresholds of this lane's queries (issued after the DMA so they do not delay it; consumed in the epilogue)
    float thr[NJ];
    if constexpr (!DENSE) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = q0 + wn * TN + j * 32 + fr;
            thr[j] = (q < nq && ABLATE != 2 && ABLATE != 4 && ABLATE != 14 && !(ABLATE >= 6 && ABLATE <= 8)) ? thr_s[q] : __builtin_inff();
        }
    }

    auto load_frags = [&](const char* base, int kk, u32x4(&af)[MI], u32x4(&bf)[NJ]) {
        if constexpr (ABLATE == 4 || (ABLATE >= 6 && ABLATE <= 8)) return;
        if constexpr (ABLATE == 2) {  // timing-only build: no LDS reads, fragments made up from registers
#pragma unroll
            for (int i = 0; i < MI; ++i) { af[i] = u32x4{(unsigned)(kk + i), (unsigned)lane, 0x3c003c00u, 0x3c003c00u}; asm volatile("" : "+v"(af[i])); }
#pragma unroll
            for (int j = 0; j < NJ; ++j) { bf[j] = u32x4{(unsigned)(kk + j), (unsigned)lane, 0x3c003c00u, 0x3c003c00u}; asm volatile("" : "+v"(bf[j])); }
            return;
        }
        const int slot_off = ((2 * kk + fh) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const u32x4*)(base + a_row_off + i * 32 * ROW_BYTES + slot_off);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[j] = *(const u32x4*)(base + b_row_off + j * 32 * ROW_BYTES + slot_off);
    };
    auto mfma_group = [&](u32x4(&af)[MI], u32x4(&bf)[NJ]) {
        if constexpr (ABLATE == 4 || (ABLATE >= 6 && ABLATE <= 8)) return;
        if constexpr (ABLATE == 13) __builtin_amdgcn_s_setprio(1);  // per-group priority flips measured -4 % here
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<DT>(af[i], bf[j], acc[i][j]);
        if constexpr (ABLATE == 13) __builtin_amdgcn_s_setprio(0);
    };
    // One K-slice.  Software pipelined inside the slice: the fragments of k-step kk+1 are read from LDS while
    // the MFMAs of k-step kk run, and the LDS-DMA of slice t+NSTAGE-1 is issued in KK parts between the MFMA
    // groups instead of as one burst in front of them.
    auto kslice = [&](int t, auto pre_tag) {
        constexpr bool PRE = decltype(pre_tag)::value;
        const char* base = smem + (t % NSTAGE) * STAGE_BYTES;
        const int ks = t + NSTAGE - 1;
        u32x4 af0[MI], bf0[NJ], af1[MI], bf1[NJ];
        load_frags(base, 0, af0, bf0);
        if constexpr (PRE && ABLATE != 1) stage_part(ks, 0, KK);
        load_frags(base, 1, af1, bf1);
        mfma_group(af0, bf0);
        if constexpr (PRE && ABLATE != 1) stage_part(ks, 1, KK);
        if constexpr (KK == 4) {
            load_frags(base, 2, af0, bf0);
            mfma_group(af1, bf1);
            if constexpr (PRE && ABLATE != 1) stage_part(ks, 2, KK);
            load_frags(base, 3, af1, bf1);
            mfma_group(af0, bf0);
            if constexpr (PRE && ABLATE != 1) stage_part(ks, 3, KK);
        }
        mfma_group(af1, bf1);
    };
    if constexpr (!PINGPONG) {
        // Slice t is complete in LDS once all but the (NSTAGE-2)*G youngest DMAs of every wave have landed and
        // every wave has passed the barrier (which also proves nobody still reads the slot being refilled).
        int t = 0;
#ifdef VODHIP_ABLATION
        if constexpr (ABLATE == 5) {
            // sampled workgroups of the large launches record, per K-slice: before the DMA wait, after it, after the
            // barrier, after the slice's MFMAs were issued (stamps have lgkmcnt(0): all LDS reads retired)
            const bool sampled = (gridDim.x > 4096) && ((bid & 1023) == 7) && ((bid >> 10) < 64);
            unsigned long long* out = g_stamps + (((size_t)(bid >> 10) * 8 + wave) * 16) * 6;
            const unsigned long long tk0 = stamp_now();
            for (; t + NSTAGE - 1 < nk; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long s0 = stamp_now();
                wait_vmcnt<(NSTAGE - 2) * G>();
                const unsigned long long s1 = stamp_now();
                __builtin_amdgcn_s_barrier();
                const unsigned long long s2 = stamp_now();
                __builtin_amdgcn_sched_barrier(0);
                kslice(t, std::true_type{});
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long s3 = stamp_now();
                if (sampled && lane == 0 && t < 16) {
                    out[t * 6 + 0] = s0; out[t * 6 + 1] = s1; out[t * 6 + 2] = s2; out[t * 6 + 3] = s3; out[t * 6 + 4] = tk0;
                }
            }
        }
#endif
        for (; t + NSTAGE - 1 < nk; ++t) {
            wait_vmcnt<(NSTAGE - 2) * G>();
            if constexpr (ABLATE != 3) __builtin_amdgcn_s_barrier();
            if constexpr (ABLATE == 15) {
                // L2 touch-prefetch of the NEXT corpus tile of this XCD label (xt + 8): the n_qtiles workgroups that
                // share tile xt each touch 1/n_qtiles of its 128-B lines, BM/n_qtiles lines per K slice, one lane per
                // line, dumped into a junk LDS area (no VGPR, counted by vmcnt like the other LDS-DMAs).
                const int x0n = x0 + 8 * BM;
                const int per_slice = BM / n_qtiles;  // lines per slice for this workgroup
                if (x0n + BM <= row_end && wave == (t % NWAVES) && lane < per_slice) {
                    const size_t line = (size_t)qt * (size_t)(per_slice * nk) + (size_t)t * per_slice + lane;
                    const char* src = (const char*)X + (size_t)x0n * dim_pad * 2 + line * 128;
                    __builtin_amdgcn_global_load_lds((const AS1 void*)src, (AS3 void*)(smem + NSTAGE * STAGE_BYTES + wave * 256), 4, 0, 0);
                }
            }
            kslice(t, std::true_type{});
        }
        for (; t < nk; ++t) {  // tail: nothing left to fetch, the ring drains
            if (NSTAGE >= 4 && nk - 1 - t >= 2) wait_vmcnt<(NSTAGE >= 4 ? 2 : 0) * G>();
            else if (NSTAGE >= 3 && nk - 1 - t >= 1) wait_vmcnt<(NSTAGE >= 3 ? 1 : 0) * G>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            kslice(t, std::false_type{});
        }
    } else {
        // ---- ping-pong schedule (BK = 32, 4-slot ring) ------------------------------------------------
        // The two waves that share a SIMD (wave w and w + NWAVES/2) run the same program one barrier apart:
        // a phase is  LOAD {6 ds_read_b128 (+ the LDS-DMA of slice t+3 in odd phases)} | s_barrier |
        // MFMA {8 x v_mfma 32x32x16} | s_barrier,  and the second half of the workgroup takes one extra
        // barrier up front, so while one wave of a SIMD sits in its MFMA section its partner is in its LOAD
        // section (LDS latency, DMA issue cost and waits hide behind the partner's matrix work).
        // Hazards (b_n = n-th barrier; first half runs LOAD_p before b_2p, MFMA_p after it; second half one later):
        //  WAR  slot (t+3)%4 was last read in LOAD_(2t-1); the second half retires those reads right after
        //       b_(4t-1), so DMA(t+3) may be issued after b_4t: it is issued in LOAD_(2t+1) by both halves.
        //  RAW  after issuing DMA(t+3) every wave waits until all but its 2*G youngest DMAs landed (=> slice
        //       t+1 complete) BEFORE its next barrier; the first read of slice t+1 (LOAD_(2t+2)) is behind it.
        static_assert(!PINGPONG || (BK == 32 && NSTAGE == 4), "ping-pong schedule is written for BK=32, 4 slots");
        const bool second_half = wave >= NWAVES / 2;
        wait_vmcnt<2 * G>();  // slice 0 landed (slices 1, 2 may still be in flight)
        __builtin_amdgcn_s_barrier();
        if (second_half) __builtin_amdgcn_s_barrier();
        u32x4 af[MI], bf[NJ];
        for (int t = 0; t < nk; ++t) {
            const char* base = smem + (t % NSTAGE) * STAGE_BYTES;
            // phase 2t
            load_frags(base, 0, af, bf);
            __builtin_amdgcn_s_barrier();
            mfma_group(af, bf);
            __builtin_amdgcn_s_barrier();
            // phase 2t+1
            load_frags(base, 1, af, bf);
            const int rem = nk - 2 - t;  // slices beyond t+1 that exist
            if (rem >= 2) {
                stage_part(t + 3, 0, 1);
                wait_vmcnt<2 * G>();
            } else if (rem == 1) {
                wait_vmcnt<G>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            mfma_group(af, bf);
            __builtin_amdgcn_s_barrier();
        }
        if (!second_half) __builtin_amdgcn_s_barrier();  // balance the barrier count
    }

#ifdef VODHIP_ABLATION
    unsigned long long t_kend = 0;
    if constexpr (ABLATE == 5) { __builtin_amdgcn_sched_barrier(0); t_kend = stamp_now(); }
#endif
    // ---- epilogue: threshold filter --------------------------------------------------------------
    // C layout of v_mfma_f32_32x32x16: col = lane&31 (query), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = q0 + wn * TN + j * 32 + fr;
        const bool q_ok = q < nq;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rbase = x0 + wm * TM + i * 32 + 4 * fh;
            if constexpr (DENSE) {
                if (q_ok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        const float s = acc[i][j][r];
                        if (row < row_end) {
                            const key_t64 key = (s == s) ? make_key(s, (unsigned)row) : 0ull;  // NaN never enters
                            cand[(size_t)q * cap + (row - row_begin)] = key;
                        }
                    }
                    if constexpr (SUBSET) {  // subset filter: blank the ineligible rows (rolled loop, dense chunk only)
#pragma unroll 1
                        for (int r = 0; r < 16; ++r) {
                            const int row = rbase + (r & 3) + 8 * (r >> 2);
                            if (row < row_end && !subset_allows(ex, q, row)) cand[(size_t)q * cap + (row - row_begin)] = 0ull;
                        }
                    }
                }
            } else {
                float m = acc[i][j][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
                const bool hit = m >= thr[j];  // false for NaN and for padded queries (thr = +inf)
                if (__any(hit)) {
                    if (hit)
                        append_survivors<16, SUBSET>(
                            thr[j], q, row_end, [&](int r) { return acc[i][j][r]; },
                            [&](int r) { return rbase + (r & 3) + 8 * (r >> 2); }, thr_key, cand, cnt, cap, overflow, ex);
                }
            }
        }
    }
#ifdef VODHIP_ABLATION
    if constexpr (ABLATE == 5) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t_end = stamp_now();
        const bool sampled = (gridDim.x > 4096) && ((bid & 1023) == 7) && ((bid >> 10) < 64);
        if (sampled && lane == 0) {
            unsigned long long* out = g_stamps + (((size_t)(bid >> 10) * 8 + wave) * 16 + 15) * 6;
            out[0] = t_entry; out[1] = t_kend; out[2] = t_end;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// specialised filter kernel: 8 MFMA ("consumer") waves + NL LDS-DMA ("loader") waves per workgroup
// ------------------------------------------------------------------------------------------------
// Measured on MI355X (profiles/r01_ablation.md): the L2 -> LDS path sustains ~48 GB/s per CU (12.3 TB/s
// chip-wide) for this access pattern, i.e. ~2400 cycles per 64 KB K-slice, about as long as the slice's
// MFMA work; when the MFMA waves issue the LDS-DMA themselves, in-order issue parks them behind a full
// memory pipeline and the two costs ADD (16 ms = 10 ms DMA + 6 ms MFMA per batch).  Here the DMA is issued
// by dedicated waves (one per SIMD) whose stalls cost nothing, the MFMA waves only read LDS and multiply.
template <int MI, int NJ, bool DENSE, bool SUBSET = false>
__device__ __forceinline__ void filter_epilogue(const f32x16 (&acc)[MI][NJ], const float (&thr)[NJ], int qbase, int rbase0,
                                                int fr, int fh, int nq, int row_begin, int row_end,
                                                const key_t64* __restrict__ thr_key, key_t64* __restrict__ cand,
                                                unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, const FilterExtra& ex) {
    // C layout of v_mfma_f32_32x32x16: col = lane&31 (query), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = qbase + j * 32 + fr;
        const bool q_ok = q < nq;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rbase = rbase0 + i * 32 + 4 * fh;
            if constexpr (DENSE) {
                if (q_ok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        const float s = acc[i][j][r];
                        if (row < row_end) {
                            const key_t64 key = (s == s) ? make_key(s, (unsigned)row) : 0ull;  // NaN never enters
                            cand[(size_t)q * cap + (row - row_begin)] = key;
                        }
                    }
                    if constexpr (SUBSET) {  // subset filter: blank the ineligible rows (rolled loop, dense chunk only)
#pragma unroll 1
                        for (int r = 0; r < 16; ++r) {
                            const int row = rbase + (r & 3) + 8 * (r >> 2);
                            if (row < row_end && !subset_allows(ex, q, row)) cand[(size_t)q * cap + (row - row_begin)] = 0ull;
                        }
                    }
                }
            } else {
                float m = acc[i][j][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
                const bool hit = m >= thr[j];  // false for NaN and for padded queries (thr = +inf)
                if (__any(hit)) {
                    if (hit)
                        append_survivors<16, SUBSET>(
                            thr[j], q, row_end, [&](int r) { return acc[i][j][r]; },
                            [&](int r) { return rbase + (r & 3) + 8 * (r >> 2); }, thr_key, cand, cnt, cap, overflow, ex);
                }
            }
        }
    }
}

template <int DT, int BK, int NSTAGE, int NL, bool DENSE, bool STAMP = false, int ABL = 0>
__global__ __launch_bounds__((8 + NL) * 64, 3) void mips_filter_spec_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end,
    int n_xtiles, int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key,
    key_t64* __restrict__ cand, unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    const bool krot_on = (ex.flags & 1) != 0;  // FILTER_FLAG_KROT

    constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NCW = 8;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NJ = TN / 32;
    constexpr int ROW_BYTES = BK * 2, CH = ROW_BYTES / 16, RPI = 64 / CH, KK = BK / 16;
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NI = (BM + BN) / RPI;  // LDS-DMA wave-instructions per slice
    constexpr int GL = NI / NL;          // ... per loader wave
    static_assert(NI % NL == 0 && (BM / RPI) % GL == 0, "loader split");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jj = bid >> 3;
    const int qt = jj % n_qtiles;
    const int xt = (jj / n_qtiles) * 8 + xcd;
    if (xt >= n_xtiles) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = row_begin + xt * BM;
    const int q0 = qt * BN;
    const int nk = dim_pad / BK;
    auto swz_of_row = [](int r) { return CH == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };

    if (wave >= NCW) {
        // ================================ loader wave ================================
        // Loader waves issue a handful of instructions per slice but each one feeds the whole workgroup: they must
        // win issue arbitration against the MFMA waves (measured: with the MFMA waves at raised priority a loader
        // needed 1,550 cycles to issue 8 LDS-DMAs, 450 without MFMA traffic).
        __builtin_amdgcn_s_setprio(3);
        const int lw = wave - NCW;
        const int st_row = lane / CH, st_slot = lane % CH;
        const int krot = krot_on ? (qt * nk) / n_qtiles : 0;  // optional rotated K order (see mips_filter_kernel)
        auto stage = [&](int ks) {
            char* sbase = smem + (ks % NSTAGE) * STAGE_BYTES;
            const int kbyte = ((ks + krot) % nk) * ROW_BYTES;
#pragma unroll
            for (int u = 0; u < GL; ++u) {
                const int g = lw * GL + u;           // wave-uniform instruction index inside the slice
                const int r = g * RPI + st_row;      // row inside [A rows | B rows]
                const bool is_a = g * RPI < BM;      // wave-uniform
                const int rr = is_a ? r : r - BM;
                const int c = st_slot ^ swz_of_row(rr);
                const uint16_t* src = is_a ? X + (size_t)(x0 + rr) * dim_pad : Q + (size_t)(q0 + rr) * dim_pad;
                glds16((const char*)src + c * 16 + kbyte, sbase + g * RPI * ROW_BYTES);
            }
        };
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nk) stage(s);
        for (int t = 0; t < nk; ++t) {
#ifdef VODHIP_ABLATION
            unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            if constexpr (STAMP) s0 = stamp_now();
#endif
            const int ahead = nk - 1 - t;  // slices after t that have been issued at most NSTAGE-2
            if (NSTAGE >= 4 && ahead >= 2) wait_vmcnt<(NSTAGE >= 4 ? 2 : 0) * GL>();
            else if (NSTAGE >= 3 && ahead >= 1) wait_vmcnt<(NSTAGE >= 3 ? 1 : 0) * GL>();
            else wait_vmcnt<0>();
#ifdef VODHIP_ABLATION
            if constexpr (STAMP) s1 = stamp_now();
#endif
            __builtin_amdgcn_s_barrier();  // B_t: slice t is complete; everyone is done with slice t-1
#ifdef VODHIP_ABLATION
            if constexpr (STAMP) s2 = stamp_now();
#endif
            if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1);
#ifdef VODHIP_ABLATION
            if constexpr (STAMP) {
                s3 = stamp_now();
                const bool sampled = (gridDim.x > 4096) && ((bid & 1023) == 7) && ((bid >> 10) < 64) && lw == 0;
                if (sampled && lane == 0 && t < 15) {
                    unsigned long long* out = g_stamps + (((size_t)(bid >> 10) * 8 + 7) * 16 + t) * 6;  // loader 0 uses wave slot 7
                    out[0] = s0; out[1] = s1; out[2] = s2; out[3] = s3;
                }
            }
#endif
        }
        return;
    }

    // ================================ MFMA wave ================================
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    const int swz = swz_of_row(fr);
    const int a_row_off = (wm * TM + fr) * ROW_BYTES;
    const int b_row_off = A_BYTES + (wn * TN + fr) * ROW_BYTES;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float thr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = q0 + wn * TN + j * 32 + fr;
        thr[j] = (!DENSE && q < nq && ABL == 0) ? thr_s[q] : __builtin_inff();
    }
    auto load_frags = [&](const char* base, int kk, u32x4(&af)[MI], u32x4(&bf)[NJ]) {
        if constexpr (ABL == 1) {  // timing-only: no LDS reads
#pragma unroll
            for (int i = 0; i < MI; ++i) { af[i] = u32x4{(unsigned)(kk + i), (unsigned)lane, 0u, 0u}; asm volatile("" : "+v"(af[i])); }
#pragma unroll
            for (int j = 0; j < NJ; ++j) { bf[j] = u32x4{(unsigned)(kk + j), (unsigned)lane, 0u, 0u}; asm volatile("" : "+v"(bf[j])); }
            return;
        }
        const int slot_off = ((2 * kk + fh) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const u32x4*)(base + a_row_off + i * 32 * ROW_BYTES + slot_off);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[j] = *(const u32x4*)(base + b_row_off + j * 32 * ROW_BYTES + slot_off);
    };
    auto mfma_group = [&](u32x4(&af)[MI], u32x4(&bf)[NJ]) {
        if constexpr (ABL == 2) {
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(bf[j]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<DT>(af[i], bf[j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    for (int t = 0; t < nk; ++t) {
#ifdef VODHIP_ABLATION
        unsigned long long s0 = 0, s2 = 0;
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); s0 = stamp_now(); }
#endif
        __builtin_amdgcn_s_barrier();  // B_t
#ifdef VODHIP_ABLATION
        if constexpr (STAMP) { s2 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
#endif
        const char* base = smem + (t % NSTAGE) * STAGE_BYTES;
        // single fragment set (the 168-VGPR budget of 3 waves/SIMD leaves no room for a second one): the LDS
        // latency of one wave's reads is covered by the MFMAs of the other MFMA wave on the same SIMD
        u32x4 af[MI], bf[NJ];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            load_frags(base, kk, af, bf);
            mfma_group(af, bf);
        }
#ifdef VODHIP_ABLATION
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long s3 = stamp_now();
            const bool sampled = (gridDim.x > 4096) && ((bid & 1023) == 7) && ((bid >> 10) < 64) && wave < 7;
            if (sampled && lane == 0 && t < 15) {
                unsigned long long* out = g_stamps + (((size_t)(bid >> 10) * 8 + wave) * 16 + t) * 6;
                out[0] = s0; out[1] = s0; out[2] = s2; out[3] = s3;
            }
        }
#endif
    }
    filter_epilogue<MI, NJ, DENSE>(acc, thr, q0 + wn * TN, x0 + wm * TM, fr, fh, nq, row_begin, row_end, thr_key, cand, cnt,
                                   cap, overflow, ex);
}

// ------------------------------------------------------------------------------------------------
// filter kernel, v_mfma_f32_16x16x32 flavour (256 x 256 tile, 64-deep slices, 2 LDS slots)
// ------------------------------------------------------------------------------------------------
// Same staging, swizzle, K rotation and threshold filter as mips_filter_kernel; the matrix work is issued as
// 16x16x32 MFMAs (two per 32x32x16's worth of flops at half the cycles each).  On gfx950 the chip sustains a
// higher clock on this shape for the same flops per cycle (MI355X_MICROARCH "DVFS give-back" item 7).
// Fragment maps: lane l supplies A[row l&15][k = 8(l>>4) .. +7] and B[k = 8(l>>4) .. +7][col l&15];
// C/D: col = l&15 (query), row = 4(l>>4) + reg.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (DT == 0) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}

template <int DT, bool DENSE, bool SUBSET = false>
__global__ __launch_bounds__(512, 2) void mips_filter16_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end,
    int n_xtiles, int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key,
    key_t64* __restrict__ cand, unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    const bool krot_on = (ex.flags & 1) != 0;  // FILTER_FLAG_KROT

    constexpr int BM = 256, BN = 256, WN = 4, NWAVES = 8, BK = 64, NSTAGE = 2;
    constexpr int TM = 128, TN = 64;
    constexpr int MB = TM / 16, NB16 = TN / 16;  // 8 x 4 blocks of 16x16 per wave
    constexpr int ROW_BYTES = BK * 2, RPI = 8;
    constexpr int A_BYTES = BM * ROW_BYTES, STAGE_BYTES = (BM + BN) * ROW_BYTES;
    constexpr int NA = BM / RPI / NWAVES, NBI = BN / RPI / NWAVES, G = NA + NBI;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jj = bid >> 3;
    const int qt = jj % n_qtiles;
    const int xt = (jj / n_qtiles) * 8 + xcd;
    if (xt >= n_xtiles) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int x0 = row_begin + xt * BM;
    const int q0 = qt * BN;
    const int nk = dim_pad / BK;
    const int krot = krot_on ? (qt * nk) / n_qtiles : 0;

    const int st_row = lane >> 3, st_slot = lane & 7;
    const char* a_src[NA];
    const char* b_src[NBI];
#pragma unroll
    for (int t = 0; t < NA; ++t) {
        const int r = (wave * NA + t) * RPI + st_row;
        a_src[t] = (const char*)X + ((size_t)(x0 + r) * dim_pad + (st_slot ^ ((r >> 1) & 7)) * 8) * 2;
    }
#pragma unroll
    for (int t = 0; t < NBI; ++t) {
        const int r = (wave * NBI + t) * RPI + st_row;
        b_src[t] = (const char*)Q + ((size_t)(q0 + r) * dim_pad + (st_slot ^ ((r >> 1) & 7)) * 8) * 2;
    }
    auto stage_part = [&](int ks, int part, int nparts) {
        char* sa = smem + (ks % NSTAGE) * STAGE_BYTES;
        char* sb = sa + A_BYTES;
        const int kbyte = ((ks + krot) % nk) * ROW_BYTES;
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if ((u * nparts) / G != part) continue;
            if (u < NA)
                glds16(a_src[u] + kbyte, sa + (wave * NA + u) * RPI * ROW_BYTES);
            else
                glds16(b_src[u - NA] + kbyte, sb + (wave * NBI + (u - NA)) * RPI * ROW_BYTES);
        }
    };

    // fragment addressing: row = block*16 + (l&15); logical chunk of k32-step ks is 4*ks + (l>>4); slot = chunk ^ f(row),
    // f(row) = (row>>1)&7 depends on the lane only because blocks start at multiples of 16 rows
    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;
    const int a_row_off = (wm * TM + fr) * ROW_BYTES;
    const int b_row_off = A_BYTES + (wn * TN + fr) * ROW_BYTES;

    f32x4 acc[MB][NB16];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_part(0, 0, 1);
    float thr[NB16];
#pragma unroll
    for (int j = 0; j < NB16; ++j) {
        const int q = q0 + wn * TN + j * 16 + fr;
        thr[j] = (!DENSE && q < nq) ? thr_s[q] : __builtin_inff();
    }

    // per k32-step: B fragments (4) stay for the step; A fragments are read in two halves of 4 row-blocks, the second
    // half while the first half's 16 MFMAs run (software pipelining along M keeps the live fragment set at 48 VGPRs)
    auto kstep = [&](const char* base, int ks, bool pre, int t, int part0) {
        const int slot_off = ((4 * ks + fq) ^ swz) << 4;
        u32x4 bf[NB16], a0[4], a1[4];
#pragma unroll
        for (int j = 0; j < NB16; ++j) bf[j] = *(const u32x4*)(base + b_row_off + j * 16 * ROW_BYTES + slot_off);
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = *(const u32x4*)(base + a_row_off + i * 16 * ROW_BYTES + slot_off);
        if (pre) stage_part(t + 1, part0, 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[i] = *(const u32x4*)(base + a_row_off + (4 + i) * 16 * ROW_BYTES + slot_off);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB16; ++j) acc[i][j] = mfma16<DT>(a0[i], bf[j], acc[i][j]);
        if (pre) stage_part(t + 1, part0 + 1, 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB16; ++j) acc[4 + i][j] = mfma16<DT>(a1[i], bf[j], acc[4 + i][j]);
    };
    for (int t = 0; t < nk; ++t) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const char* base = smem + (t % NSTAGE) * STAGE_BYTES;
        const bool pre = t + 1 < nk;
        kstep(base, 0, pre, t, 0);
        kstep(base, 1, pre, t, 2);
    }

    // ---- epilogue: one max / compare / ballot per 16-query column block, then the rare slow path ----
#pragma unroll
    for (int j = 0; j < NB16; ++j) {
        const int q = q0 + wn * TN + j * 16 + fr;
        const bool q_ok = q < nq;
        if constexpr (DENSE) {
            if (q_ok) {
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = x0 + wm * TM + i * 16 + 4 * fq + r;
                        const float sc = acc[i][j][r];
                        if (row < row_end) cand[(size_t)q * cap + (row - row_begin)] = (sc == sc) ? make_key(sc, (unsigned)row) : 0ull;
                    }
                if constexpr (SUBSET) {  // subset filter: blank the ineligible rows (rolled loop, dense chunk only)
#pragma unroll 1
                    for (int v = 0; v < MB * 4; ++v) {
                        const int row = x0 + wm * TM + (v >> 2) * 16 + 4 * fq + (v & 3);
                        if (row < row_end && !subset_allows(ex, q, row)) cand[(size_t)q * cap + (row - row_begin)] = 0ull;
                    }
                }
            }
        } else {
            float m = acc[0][j][0];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[i][j][r]);
            const bool hit = m >= thr[j];
            if (__any(hit)) {
                if (hit)
                    append_survivors<MB * 4, SUBSET>(
                        thr[j], q, row_end, [&](int v) { return acc[v >> 2][j][v & 3]; },
                        [&](int v) { return x0 + wm * TM + (v >> 2) * 16 + 4 * fq + (v & 3); }, thr_key, cand, cnt, cap, overflow, ex);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// persistent flavour of mips_filter16_kernel: one workgroup per CU walks a list of corpus tiles
// ------------------------------------------------------------------------------------------------
// Same tile, staging and MFMA shape as mips_filter16_kernel.  The grid is (a multiple of 8 * n_qtiles) <= #CUs and a
// workgroup keeps its q-tile while it steps through corpus tiles xt, xt + G/n_qtiles, ...  The K slices of
// consecutive tiles form ONE stream through the two LDS slots: the LDS-DMA of the next tile's first slice is issued
// during the current tile's last slice, so its HBM latency hides behind that slice's MFMAs and the threshold-filter
// epilogue instead of opening every tile with an idle matrix pipe (measured prologue: ~2,200 cycles of a 48,000
// cycle tile), and there is no per-tile workgroup launch / drain.
constexpr int PSTG_CAP = 1024;    // records in the LDS survivor list of the persistent kernel
constexpr int PSTG_FLUSH = 384;   // flush when at least this many are pending (checked once per tile)
constexpr int PSTG_BYTES = PSTG_CAP * 12 + 16;
template <int DT, bool A3 = false, bool SUBSET = false, bool STAMP = false, int ABL = 0>
__global__ __launch_bounds__(512, 2) void mips_filter16p_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end,
    int n_xtiles, int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key,
    key_t64* __restrict__ cand, unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    constexpr int BM = 256, BN = 256, WN = 4, NWAVES = 8, BK = 64, NSTAGE = 2;
    constexpr int TM = 128, TN = 64;
    constexpr int MB = TM / 16, NB16 = TN / 16;
    constexpr int ROW_BYTES = BK * 2, RPI = 8;
    constexpr int A_BYTES = BM * ROW_BYTES, STAGE_BYTES = (BM + BN) * ROW_BYTES;
    constexpr int NA = BM / RPI / NWAVES, NBI = BN / RPI / NWAVES, G = NA + NBI;
    // Survivor staging (two-slot layout only; the A3 layout has no LDS to spare): a lane that finds survivors appends
    // (key, query) records to a workgroup list in LDS with ONE LDS atomic - no global round trip, and no vmcnt wait
    // that would drain the LDS-DMAs in flight across the epilogue.  The list is flushed to the global candidate
    // lists (exact-key test, subset test, one global atomic per record, all records in parallel) when it holds
    // PSTG_FLUSH records and at the end of the kernel.  A lane whose reservation does not fit takes the direct path.
    constexpr bool STAGED = !A3;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    key_t64* const stg_key = (key_t64*)(smem + NSTAGE * STAGE_BYTES);
    int* const stg_q = (int*)(smem + NSTAGE * STAGE_BYTES + PSTG_CAP * 8);
    unsigned* const stg_cnt = (unsigned*)(smem + NSTAGE * STAGE_BYTES + PSTG_CAP * 12);
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jj = bid >> 3;
    const int qt = jj % n_qtiles;
    const int xt0 = (jj / n_qtiles) * 8 + xcd;
    const int xt_step = (int)gridDim.x / n_qtiles;  // gridDim.x is a multiple of 8 * n_qtiles
    if (xt0 >= n_xtiles) return;
    const int n_my = (n_xtiles - 1 - xt0) / xt_step + 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int q0 = qt * BN;
    const int nk = dim_pad / BK;
    const size_t tile_step_bytes = (size_t)xt_step * BM * dim_pad * 2;

    const int st_row = lane >> 3, st_slot = lane & 7;
    const char* a_src[NA];
    const char* b_src[NBI];
#pragma unroll
    for (int t = 0; t < NA; ++t) {
        const int r = (wave * NA + t) * RPI + st_row;
        a_src[t] = (const char*)X + ((size_t)(row_begin + xt0 * BM + r) * dim_pad + (st_slot ^ ((r >> 1) & 7)) * 8) * 2;
    }
#pragma unroll
    for (int t = 0; t < NBI; ++t) {
        const int r = (wave * NBI + t) * RPI + st_row;
        b_src[t] = (const char*)Q + ((size_t)(q0 + r) * dim_pad + (st_slot ^ ((r >> 1) & 7)) * 8) * 2;
    }
    auto stage_part = [&](int slot, int kbyte, int part, int nparts) {
        char* sa = smem + slot * STAGE_BYTES;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if ((u * nparts) / G != part) continue;
            if (u < NA)
                glds16(a_src[u] + kbyte, sa + (wave * NA + u) * RPI * ROW_BYTES);
            else
                glds16(b_src[u - NA] + kbyte, sb + (wave * NBI + (u - NA)) * RPI * ROW_BYTES);
        }
    };
    // A3 layout: three corpus slots [0, 3*A_BYTES) + two query slots behind them (all 160 KiB of LDS): the corpus
    // operand (HBM first touch, slow) is fetched TWO slices ahead, the L2-hot query operand one slice ahead.
    constexpr int B_BASE_A3 = 3 * A_BYTES;
    auto stage_a = [&](int aslot, int kbyte, int half) {  // half 0|1: first / second NA/2 wave-instructions
#pragma unroll
        for (int u = 0; u < NA; ++u)
            if (u / (NA / 2) == half) glds16(a_src[u] + kbyte, smem + aslot * A_BYTES + (wave * NA + u) * RPI * ROW_BYTES);
    };
    auto stage_b = [&](int bslot, int kbyte, int half) {
#pragma unroll
        for (int u = 0; u < NBI; ++u)
            if (u / (NBI / 2) == half)
                glds16(b_src[u] + kbyte, smem + B_BASE_A3 + bslot * A_BYTES + (wave * NBI + u) * RPI * ROW_BYTES);
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;
    const int a_row_off = (wm * TM + fr) * ROW_BYTES;
    const int b_row_off = A_BYTES + (wn * TN + fr) * ROW_BYTES;

    // thresholds first, and retired (the dummy use makes the compiler wait HERE): a later compiler-inserted
    // vmcnt wait for them would also drain the LDS-DMAs that are meant to stay in flight across the epilogue
    float thr[NB16];
#pragma unroll
    for (int j = 0; j < NB16; ++j) {
        const int q = q0 + wn * TN + j * 16 + fr;
        thr[j] = (q < nq) ? thr_s[q] : __builtin_inff();
    }
#ifdef VODHIP_ABLATION
    if constexpr (STAMP) {  // bit 1 of the flags: no survivors at all (epilogue floor)
        if (ex.flags & 2)
#pragma unroll
            for (int j = 0; j < NB16; ++j) thr[j] = __builtin_inff();
    }
#endif
    if constexpr (ABL != 0) {
#pragma unroll
        for (int j = 0; j < NB16; ++j) thr[j] = __builtin_inff();
    }
#pragma unroll
    for (int j = 0; j < NB16; ++j) asm volatile("" : "+v"(thr[j]));
    // stream state of the A3 variant: next slice to fetch for each operand (global slice index, tile-local k index)
    const int S = n_my * nk;
    int sa_next = 0, ta_next = 0, sb_next = 0, tb_next = 0;
    auto issue_a = [&](int half) {  // corpus slice sa_next into slot sa_next % 3; bookkeeping advances after half 1
        stage_a(sa_next % 3, ta_next * ROW_BYTES, half);
        if (half == 1) {
            ++sa_next;
            if (++ta_next == nk) {
                ta_next = 0;
#pragma unroll
                for (int u = 0; u < NA; ++u) a_src[u] += tile_step_bytes;
            }
        }
    };
    auto issue_b = [&](int half) {
        stage_b(sb_next & 1, tb_next * ROW_BYTES, half);
        if (half == 1) {
            ++sb_next;
            if (++tb_next == nk) tb_next = 0;
        }
    };
    if constexpr (A3) {
        issue_a(0); issue_a(1);                    // A(0)
        issue_b(0); issue_b(1);                    // B(0)
        if (S > 1) { issue_a(0); issue_a(1); }     // A(1)
    } else {
        stage_part(0, 0, 0, 1);
    }

    if constexpr (STAGED) {
        for (int e = tid; e < PSTG_CAP; e += 512) stg_key[e] = 0;  // key 0 = empty record
        if (tid == 0) *stg_cnt = 0;
        // made visible by the barrier of the first K slice (which every wave passes before its first epilogue)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    auto flush_staged = [&](unsigned c) {  // called by all 512 threads with the same c
        if (c > (unsigned)PSTG_CAP) c = PSTG_CAP;
        for (unsigned e = tid; e < c; e += 512) {
            const key_t64 key = stg_key[e];
            const int q = stg_q[e];
            stg_key[e] = 0;
            if (key == 0) continue;  // reserved by a lane that then took the direct path
            bool ok = key > thr_key[q];
            if constexpr (SUBSET) ok = ok && subset_allows(ex, q, (int)(0xFFFFFFFFu - (unsigned)key));
            if (ok) {
                const unsigned slot = atomicAdd(&cnt[q], 1u);
                if (slot < (unsigned)cap)
                    cand[(size_t)q * cap + slot] = key;
                else
                    atomicOr(overflow, 1u);
            }
        }
    };

    int g = 0;  // global slice counter of this workgroup: LDS slot = g & 1
#ifdef VODHIP_ABLATION
    unsigned long long st_k = 0, st_e = 0, st_t0 = 0, st_e1 = 0, st_e2 = 0, st_nhit = 0;
    if constexpr (STAMP) st_t0 = stamp_now();
#endif
    for (int it = 0; it < n_my; ++it) {
#ifdef VODHIP_ABLATION
        unsigned long long st_a = 0, st_b = 0;
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_a = stamp_now(); }
#endif
        const int x0 = row_begin + (xt0 + it * xt_step) * BM;
        // no zero fill (128 v_mov per tile): the first k32 step of the tile's first slice multiplies into a constant-0 C
        f32x4 acc[MB][NB16];
        auto do_slice = [&](int t, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            if constexpr (A3) {
                // everything but the 4 youngest LDS-DMAs (= the corpus slice g+1, if it exists) has landed
                if (g + 1 < S) wait_vmcnt<NA>(); else wait_vmcnt<0>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            const char* base = smem + (g & 1) * STAGE_BYTES;
            const char* base_a = A3 ? smem + (g % 3) * A_BYTES - 0 : base;
            const char* base_b = A3 ? smem + B_BASE_A3 + (g & 1) * A_BYTES - A_BYTES : base;  // b_row_off already adds A_BYTES
            // what to fetch during this slice: the next slice of this tile, or slice 0 of the next tile
            bool pre = true;
            int kbyte = (t + 1) * ROW_BYTES;
            if constexpr (!A3) {
                if (t + 1 == nk) {
                    kbyte = 0;
                    pre = it + 1 < n_my;
                    if (pre) {
#pragma unroll
                        for (int u = 0; u < NA; ++u) a_src[u] += tile_step_bytes;
                    }
                }
            }
            const bool pre_b = sb_next < S, pre_a = sa_next < S;  // A3: query slice g+1 first, then corpus slice g+2
            const int nslot = (g + 1) & 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int slot_off = ((4 * ks + fq) ^ swz) << 4;
                u32x4 bf[NB16], a0[4], a1[4];
                if constexpr ((ABL & 2) != 0) {  // timing only: fragments made up in registers, no LDS reads
#pragma unroll
                    for (int j = 0; j < NB16; ++j) bf[j] = u32x4{(unsigned)(lane + j), (unsigned)g, 0x3c003c00u, (unsigned)t};
#pragma unroll
                    for (int i = 0; i < 4; ++i) { a0[i] = u32x4{(unsigned)(lane ^ i), 0x3c003c00u, (unsigned)g, 1u}; a1[i] = a0[i]; }
                }
                if constexpr ((ABL & 2) == 0) {
#pragma unroll
                for (int j = 0; j < NB16; ++j) bf[j] = *(const u32x4*)(base_b + b_row_off + j * 16 * ROW_BYTES + slot_off);
#pragma unroll
                for (int i = 0; i < 4; ++i) a0[i] = *(const u32x4*)(base_a + a_row_off + i * 16 * ROW_BYTES + slot_off);
                }
                if constexpr (A3) {
                    if (ks == 0) { if (pre_b) issue_b(0); } else { if (pre_a) issue_a(0); }
                } else {
                    if (pre && !(ABL & 1)) stage_part(nslot, kbyte, 2 * ks, 4);
                }
                if constexpr ((ABL & 2) == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a1[i] = *(const u32x4*)(base_a + a_row_off + (4 + i) * 16 * ROW_BYTES + slot_off);
                }
                const bool zero_c = FIRST && ks == 0;  // compile-time after unrolling
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NB16; ++j) {
                        if constexpr ((ABL & 4) != 0) {  // consume the fragments (forces the waits) without any instruction
                            asm volatile("" ::"v"(a0[i]), "v"(bf[j]));
                            if (zero_c) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        } else {
                            acc[i][j] = mfma16<DT>(a0[i], bf[j], zero_c ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j]);
                        }
                    }
                if constexpr (A3) {
                    if (ks == 0) { if (pre_b) issue_b(1); } else { if (pre_a) issue_a(1); }
                } else {
                    if (pre && !(ABL & 1)) stage_part(nslot, kbyte, 2 * ks + 1, 4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NB16; ++j)
                    {
                        if constexpr ((ABL & 4) != 0) {
                            asm volatile("" ::"v"(a1[i]), "v"(bf[j]));
                            if (zero_c) acc[4 + i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        } else {
                            acc[4 + i][j] = mfma16<DT>(a1[i], bf[j], zero_c ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[4 + i][j]);
                        }
                    }
            }
        };
        do_slice(0, std::true_type{});
        ++g;
        for (int t = 1; t < nk; ++t, ++g) do_slice(t, std::false_type{});

#ifdef VODHIP_ABLATION
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_b = stamp_now(); st_k += st_b - st_a; }
#endif
        // threshold filter of this tile (the next tile's first slice is already in flight)
#pragma unroll
        for (int j = 0; j < NB16; ++j) {
            const int q = q0 + wn * TN + j * 16 + fr;
            float m = acc[0][j][0];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[i][j][r]);
            const bool hit = m >= thr[j];
            if (__any(hit)) {
#ifdef VODHIP_ABLATION
                if constexpr (STAMP) ++st_nhit;
#endif
                if (hit) {
                    // opaque copies: everything derived from the tile's row base and bound stays INSIDE this cold
                    // path (the compiler otherwise hoists the 32 row keys / bound compares, which do not depend on
                    // j, into the per-tile fast path and spills them)
                    int x0_o = x0, row_end_o = row_end;
                    asm volatile("" : "+s"(x0_o), "+s"(row_end_o));
                    auto val = [&](int v) { return acc[v >> 2][j][v & 3]; };
                    auto row = [&](int v) { return x0_o + wm * TM + (v >> 2) * 16 + 4 * fq + (v & 3); };
                    if constexpr (STAGED) {
                        // the common case - ONE score of the lane passes, and it is the lane maximum m - runs through
                        // short straight-line code into the LDS list
                        unsigned n1 = 0;
                        int off1 = 0;
#pragma unroll
                        for (int v = 0; v < MB * 4; ++v) {
                            const bool p = val(v) >= thr[j];
                            n1 += p ? 1u : 0u;
                            off1 = p ? (v >> 2) * 16 + (v & 3) : off1;
                        }
                        if (n1 == 1) {
                            const int rw = x0_o + wm * TM + 4 * fq + off1;
                            if (rw < row_end_o) {
                                const unsigned pos1 = atomicAdd(stg_cnt, 1u);
                                const key_t64 key = make_key(m, (unsigned)rw);
                                if (pos1 < (unsigned)PSTG_CAP) {
                                    stg_key[pos1] = key;
                                    stg_q[pos1] = q;
                                } else {
                                    bool ok = key > thr_key[q];
                                    if constexpr (SUBSET) ok = ok && subset_allows(ex, q, rw);
                                    if (ok) {
                                        const unsigned slot = atomicAdd(&cnt[q], 1u);
                                        if (slot < (unsigned)cap)
                                            cand[(size_t)q * cap + slot] = key;
                                        else
                                            atomicOr(overflow, 1u);
                                    }
                                }
                            }
                        } else {  // several survivors in one lane (rare once the threshold is tight): direct path
                            append_survivors<MB * 4, SUBSET>(thr[j], q, row_end_o, val, row, thr_key, cand, cnt, cap, overflow, ex);
                        }
                    } else {
                        append_survivors<MB * 4, SUBSET>(thr[j], q, row_end_o, val, row, thr_key, cand, cnt, cap, overflow, ex);
                    }
                }
            }
        }
#ifdef VODHIP_ABLATION
        unsigned long long st_c = 0, st_d = 0;
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_c = stamp_now(); st_e1 += st_c - st_b; }
#endif
        if constexpr (STAGED) {
            // every wave's records of this tile are in the list; the decision to flush is workgroup-uniform
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned c = *(volatile unsigned*)stg_cnt;
#ifdef VODHIP_ABLATION
            if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_d = stamp_now(); st_e2 += st_d - st_c; }
#endif
            if (c >= (unsigned)PSTG_FLUSH) {
                flush_staged(c);
                __syncthreads();  // everyone has read c and its records
                if (tid == 0) *stg_cnt = 0;
                // the reset is ordered before the next appends by the barrier of the next K slice
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
#ifdef VODHIP_ABLATION
        if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); st_e += stamp_now() - st_b; }
#endif
    }
    if constexpr (STAGED) {
        __syncthreads();
        flush_staged(*(volatile unsigned*)stg_cnt);
    }
#ifdef VODHIP_ABLATION
    if constexpr (STAMP) {
        if (gridDim.x == 256 && n_my > 100 && lane == 0 && bid < 64) {
            unsigned long long* out = g_stamps + ((size_t)bid * 8 + wave) * 16 * 6;
            out[0] = st_k; out[1] = st_e; out[2] = stamp_now() - st_t0; out[3] = (unsigned long long)n_my; out[4] = st_e1; out[5] = st_e2; out[6] = st_nhit;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// select kernel: fold the chunk's candidates into the running sorted top-k, tighten the threshold
// ------------------------------------------------------------------------------------------------
// Descending bitonic sort of P (power of two) keys in LDS by 256 threads.  Each thread gathers up to CE_UNROLL
// compare-exchange pairs into registers before comparing and writing back, so the LDS reads of a stage overlap
// instead of forming one dependent read-compare-write chain per pair.
__device__ __forceinline__ void bitonic_sort_desc_lds(key_t64* keys, int P, int tid) {
    constexpr int CE_UNROLL = 4;
    const int half = P >> 1;
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t0 = tid; t0 < half; t0 += 256 * CE_UNROLL) {
                key_t64 a[CE_UNROLL], b[CE_UNROLL];
                int pos[CE_UNROLL];
#pragma unroll
                for (int u = 0; u < CE_UNROLL; ++u) {
                    const int t = t0 + u * 256;
                    pos[u] = 2 * t - (t & (stride - 1));
                    if (t < half) {
                        a[u] = keys[pos[u]];
                        b[u] = keys[pos[u] + stride];
                    }
                }
#pragma unroll
                for (int u = 0; u < CE_UNROLL; ++u) {
                    const int t = t0 + u * 256;
                    if (t < half) {
                        const bool desc = (pos[u] & size) == 0;
                        if ((a[u] < b[u]) == desc) {
                            keys[pos[u]] = b[u];
                            keys[pos[u] + stride] = a[u];
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
}

// k-th largest of `total` DISTINCT-or-zero 64-bit keys in LDS (zeros = padding, `total` >= k): MSB-first radix select,
// 8 passes of one byte with a 256-bin LDS histogram.  Returns the key on every thread.  `hist` = 256 ints + 2 scratch.
__device__ key_t64 radix_select_kth_lds(const key_t64* keys, int total, int k, int* hist, int tid) {
    // hist: [0..255] bins | 256 digit / key low | 257 rank / key high | 258 (caller's compaction counter) | 259 bin count
    //       | 260..263 per-wave totals.  MSB-first byte passes; as soon as the bin that holds the k-th key holds ONE key
    //       (typically after 3-4 of the 8 bytes: distinct float scores) that key is looked up directly.
    key_t64 prefix = 0, mask = 0;
    int rank = k;  // 1-based rank from the top among the keys that match the prefix
    const int lane = tid & 63, wave = tid >> 6;
    for (int byte = 7; byte >= 0; --byte) {
        hist[tid] = 0;
        __syncthreads();
        const int sh = byte * 8;
        for (int i = tid; i < total; i += 256) {
            const key_t64 e = keys[i];
            if ((e & mask) == prefix) atomicAdd(&hist[(int)((e >> sh) & 255ull)], 1);
        }
        __syncthreads();
        // suffix sums over the 256 bins: thread t learns above(t) = #keys in bins > t; exactly one t has
        // above(t) < rank <= above(t) + hist[t].  Wave-level shuffle scan + 4 wave totals through LDS.
        const int mine = hist[tid];
        int v = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_down(v, off);
            if (lane + off < 64) v += u;
        }
        if (lane == 0) hist[260 + wave] = v;
        __syncthreads();
        int above = v - mine;
        for (int w = wave + 1; w < 4; ++w) above += hist[260 + w];
        if (above < rank && rank <= above + mine) {
            hist[256] = tid;
            hist[257] = rank - above;
            hist[259] = mine;
        }
        __syncthreads();
        const int digit = hist[256];
        rank = hist[257];
        const int in_bin = hist[259];
        prefix |= (key_t64)digit << sh;
        mask |= 255ull << sh;
        __syncthreads();
        if (in_bin == 1 && byte > 0) {  // workgroup-uniform: the k-th key is the only one with this prefix
            for (int i = tid; i < total; i += 256) {
                const key_t64 e = keys[i];
                if ((e & mask) == prefix) {
                    hist[256] = (int)(unsigned)(e & 0xFFFFFFFFull);
                    hist[257] = (int)(unsigned)(e >> 32);
                }
            }
            __syncthreads();
            const key_t64 kth = ((key_t64)(unsigned)hist[257] << 32) | (key_t64)(unsigned)hist[256];
            __syncthreads();
            return kth;
        }
    }
    return prefix;
}

// One workgroup per query.  The LDS buffer holds SB keys (SB >= 2*kp, power of two): the running top-k sits in
// front, candidates are folded in rounds of SB - kp.  Between chunks only the k-th best key (the threshold) and the
// SET of the k best are needed, so intermediate launches use a radix select + compaction (unsorted top-k);
// the last launch of a search (`final_sort`) sorts, which is what the output stage reads.
__global__ __launch_bounds__(256) void mips_select_kernel(key_t64* __restrict__ topk, int kp, int k, int sb,
                                                          const key_t64* __restrict__ cand,
                                                          unsigned int* __restrict__ cnt, int cap, int dense_n,
                                                          float* __restrict__ thr_s, key_t64* __restrict__ thr_key,
                                                          unsigned int* __restrict__ overflow, int final_sort,
                                                          int64_t id_base, float* __restrict__ out_scores,
                                                          int64_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    key_t64* keys = (key_t64*)smem;
    int* hist = (int*)(keys + sb);  // 264 ints: bins, scratch words, compaction counter at [258] (see radix_select_kth_lds)
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    unsigned n = dense_n >= 0 ? (unsigned)dense_n : cnt[q];
    if (n > (unsigned)cap) {
        if (tid == 0) atomicOr(overflow, 1u);
        n = cap;
    }
    for (int i = tid; i < kp; i += 256) keys[i] = topk[(size_t)q * kp + i];
    const int room = sb - kp;
    int done = 0;
    key_t64 kth = 0;
    do {
        const int take = min((int)n - done, room);
        const int total = kp + take;
        const bool last_round = done + take >= (int)n;
        if (final_sort && last_round) {
            int P = 64;
            while (P < total) P <<= 1;
            for (int i = kp + tid; i < P; i += 256) keys[i] = (i < total) ? cand[(size_t)q * cap + done + (i - kp)] : 0ull;
            bitonic_sort_desc_lds(keys, P, tid);
            kth = keys[k - 1];
        } else {
            for (int i = kp + tid; i < total; i += 256) keys[i] = cand[(size_t)q * cap + done + (i - kp)];
            __syncthreads();
            kth = radix_select_kth_lds(keys, total, k, hist, tid);  // total >= kp >= k (zeros pad the running top-k)
            // compaction: the keys >= kth (exactly k of them unless kth == 0) move to the front, zeros behind
            key_t64 mine[16];  // SB <= 4096 keys / 256 threads
            int n_mine = 0;
            for (int i = tid; i < total; i += 256) {
                const key_t64 e = keys[i];
                if (e >= kth && e != 0ull && n_mine < 16) mine[n_mine++] = e;
            }
            if (tid == 0) hist[258] = 0;
            __syncthreads();
            int base = n_mine ? atomicAdd(&hist[258], n_mine) : 0;
            __syncthreads();  // every thread has read its keys before anyone overwrites the front
            for (int i = tid; i < kp; i += 256) keys[i] = 0ull;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < n_mine) keys[base + u] = mine[u];
            __syncthreads();
        }
        done += take;
    } while (done < (int)n);
    for (int i = tid; i < kp; i += 256) topk[(size_t)q * kp + i] = keys[i];
    if (out_scores != nullptr) {  // last select of a search: the sorted keys leave as (float32 score, int64 id) rows
        for (int c = tid; c < k; c += 256) {
            const key_t64 key = keys[c];
            out_scores[(size_t)q * k + c] = key ? unflip_f32((unsigned)(key >> 32)) : -__builtin_inff();
            out_ids[(size_t)q * k + c] = key ? id_base + (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu)) : -1;
        }
    }
    if (tid == 0) {
        thr_key[q] = kth;
        thr_s[q] = kth ? unflip_f32((unsigned)(kth >> 32)) : -__builtin_inff();
        cnt[q] = 0;
    }
}

__global__ void mips_output_kernel(const key_t64* __restrict__ topk, int kp, int k, int64_t nq, int64_t id_base,
                                   float* __restrict__ out_scores, int64_t* __restrict__ out_ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * k) return;
    const int64_t q = i / k;
    const int c = (int)(i % k);
    const key_t64 key = topk[q * kp + c];
    if (key == 0) {
        out_scores[i] = -__builtin_inff();
        out_ids[i] = -1;
    } else {
        out_scores[i] = unflip_f32((unsigned)(key >> 32));
        out_ids[i] = id_base + (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
    }
}

// ------------------------------------------------------------------------------------------------
// row conversion (ingest + query staging): src [n, dim] of f16|bf16|f32 -> dst [n, dst_stride] f16|bf16,
// columns >= dim zero-filled.  One thread per 8 destination elements (16-B stores).
// ------------------------------------------------------------------------------------------------
template <int SRC, int DST>
__global__ void convert_rows_kernel(const void* __restrict__ src, int64_t n_rows, int64_t dim, uint16_t* __restrict__ dst,
                                    int64_t dst_stride) {
    const int64_t chunks_per_row = dst_stride / 8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * chunks_per_row) return;
    const int64_t row = i / chunks_per_row;
    const int64_t c0 = (i % chunks_per_row) * 8;
    uint16_t out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t c = c0 + e;
        uint16_t v = 0;
        if (c < dim) {
            float f;
            if constexpr (SRC == 2) {
                f = ((const float*)src)[row * dim + c];
            } else if constexpr (SRC == 0) {
                f = (float)(((const _Float16*)src)[row * dim + c]);
            } else {
                f = (float)(((const __bf16*)src)[row * dim + c]);
            }
            if constexpr (DST == 0) {
                const _Float16 h = (_Float16)f;
                v = __builtin_bit_cast(uint16_t, h);
            } else {
                const __bf16 h = (__bf16)f;
                v = __builtin_bit_cast(uint16_t, h);
            }
        }
        out[e] = v;
    }
    *(uint4*)(dst + row * dst_stride + c0) = *(const uint4*)out;
}

hipError_t launch_convert_rows(const void* src, int src_dtype, int64_t n_rows, int64_t dim, void* dst, int dst_dtype,
                               int64_t dst_stride, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    const int64_t total = n_rows * (dst_stride / 8);
    const int threads = 256;
    const unsigned blocks = (unsigned)((total + threads - 1) / threads);
    uint16_t* d = (uint16_t*)dst;
#define VOD_CONV(S, D)                                                                                         \
    if (src_dtype == S && dst_dtype == D) {                                                                    \
        hipLaunchKernelGGL((convert_rows_kernel<S, D>), dim3(blocks), dim3(threads), 0, stream, src, n_rows, dim, d, \
                           dst_stride);                                                                        \
        return hipGetLastError();                                                                              \
    }
    VOD_CONV(0, 0) VOD_CONV(1, 0) VOD_CONV(2, 0) VOD_CONV(0, 1) VOD_CONV(1, 1) VOD_CONV(2, 1)
#undef VOD_CONV
    return hipErrorInvalidValue;
}

// One launch at the head of every search pass: stage the queries (convert to the store dtype, zero the padded columns
// and rows), clear the running top-k / counters / thresholds and the overflow word.  Replaces two fills, the query
// conversion and the init kernel (4 launches and their gaps: ~20 us of a 0.8-2 ms small search).
__global__ void mips_prepare_kernel(const void* __restrict__ q_src, int q_dtype, int64_t nq, int64_t dim,
                                    uint16_t* __restrict__ q_pad, int store_dtype, int64_t nq_pad, int64_t dim_pad,
                                    key_t64* __restrict__ topk, int64_t n_topk, unsigned int* __restrict__ cnt,
                                    float* __restrict__ thr_s, key_t64* __restrict__ thr_key,
                                    unsigned int* __restrict__ overflow, int clear_overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t chunks_per_row = dim_pad / 8;
    if (i < nq_pad * chunks_per_row) {
        const int64_t row = i / chunks_per_row;
        const int64_t c0 = (i % chunks_per_row) * 8;
        uint16_t out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t c = c0 + e;
            uint16_t v = 0;
            if (row < nq && c < dim) {
                float f;
                if (q_dtype == 2) f = ((const float*)q_src)[row * dim + c];
                else if (q_dtype == 0) f = (float)(((const _Float16*)q_src)[row * dim + c]);
                else f = (float)(((const __bf16*)q_src)[row * dim + c]);
                if (store_dtype == 0) {
                    const _Float16 h = (_Float16)f;
                    v = __builtin_bit_cast(uint16_t, h);
                } else {
                    const __bf16 h = (__bf16)f;
                    v = __builtin_bit_cast(uint16_t, h);
                }
            }
            out[e] = v;
        }
        *(uint4*)(q_pad + row * dim_pad + c0) = *(const uint4*)out;
    }
    if (i < n_topk) topk[i] = 0;
    if (i < nq_pad) {
        cnt[i] = 0;
        thr_s[i] = -__builtin_inff();
        thr_key[i] = 0;
    }
    if (i == 0 && clear_overflow) *overflow = 0u;
}

hipError_t launch_search_prepare(const SearchWorkspace& ws, const void* q_src, int q_dtype, int64_t nq, int64_t dim,
                                 int store_dtype, int64_t nq_pad, int64_t dim_pad, bool clear_overflow, hipStream_t stream) {
    const int64_t n_topk = nq_pad * ws.kp;
    const int64_t n = std::max(n_topk, nq_pad * (dim_pad / 8));
    const int threads = 256;
    const unsigned blocks = (unsigned)((n + threads - 1) / threads);
    hipLaunchKernelGGL(mips_prepare_kernel, dim3(blocks), dim3(threads), 0, stream, q_src, q_dtype, nq, dim,
                       (uint16_t*)ws.q_pad, store_dtype, nq_pad, dim_pad, ws.topk, n_topk, ws.cnt, ws.thr_s, ws.thr_key,
                       ws.overflow, clear_overflow ? 1 : 0);
    return hipGetLastError();
}

#ifdef VODHIP_ABLATION
extern "C" int vodhip_debug_read_stamps(unsigned long long* host_out, long long n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

int filter_tile_rows(int tile) { return (tile == 1 || tile == 5) ? 128 : 256; }  // ablation ids 11..13 are 256
int filter_tile_cols(int tile) { return tile == 42 ? 64 : tile == 46 ? 128 : (tile == 1 || tile == 5) ? 128 : 256; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel): it is a driver call on the launch path
// (three persistent launches per batch otherwise pay it every time).
static hipError_t allow_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find({dev, kernel});
    if (it != granted.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) granted[{dev, kernel}] = bytes;
    return e;
}

template <int DT, int BM, int BN, int WM, int WN, int BK, int NSTAGE, bool DENSE, int ABLATE = 0, bool PINGPONG = false, bool SUBSET = false>
static hipError_t launch_filter_cfg(const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin,
                                    int64_t row_end, int64_t nq, int64_t nq_pad, const SearchWorkspace& ws,
                                    hipStream_t stream) {
    const int n_xtiles = (int)((row_end - row_begin + BM - 1) / BM);
    const int n_qtiles = (int)(nq_pad / BN);
    const int xgroups = (n_xtiles + 7) / 8;
    const unsigned grid = (unsigned)xgroups * 8u * (unsigned)n_qtiles;
    constexpr int threads = WM * WN * 64;
    constexpr size_t lds = (size_t)NSTAGE * (BM + BN) * BK * 2 + (ABLATE == 15 ? 2048 : 0);
    auto kern = mips_filter_kernel<DT, BM, BN, WM, WN, BK, NSTAGE, DENSE, ABLATE, PINGPONG, SUBSET>;
    if (hipError_t e = allow_dynamic_lds((const void*)kern, (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, (const uint16_t*)store, (const uint16_t*)q_pad,
                       (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key,
                       ws.cand, ws.cnt, (int)ws.cap, ws.overflow, ws.extra);
    return hipGetLastError();
}

hipError_t launch_filter(int store_dtype, int tile, bool dense, const void* store, const void* q_pad, int64_t dim_pad,
                         int64_t row_begin, int64_t row_end, int64_t nq, int64_t nq_pad, const SearchWorkspace& ws,
                         hipStream_t stream) {
    if (row_end <= row_begin) return hipSuccess;
    const bool subset = ws.extra.row_label != nullptr;
    if (subset) {
        // the subset filter is instantiated for the production variants only (1, 42, 46: small batches; 8/9/10: 256x256 16x16x32)
        if (tile != 1 && tile != 8 && tile != 9 && tile != 10 && tile != 42 && tile != 46) return hipErrorNotSupported;
#define VOD_SUBN(DT, DENSE, BN_, WM_, WN_) return launch_filter_cfg<DT, 256, BN_, WM_, WN_, 64, 3, DENSE, 0, false, true>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream)
        if (tile == 42) {
            if (store_dtype == 0 && dense) VOD_SUBN(0, true, 64, 4, 1);
            if (store_dtype == 0 && !dense) VOD_SUBN(0, false, 64, 4, 1);
            if (store_dtype == 1 && dense) VOD_SUBN(1, true, 64, 4, 1);
            if (store_dtype == 1 && !dense) VOD_SUBN(1, false, 64, 4, 1);
        }
        if (tile == 46) {
            if (store_dtype == 0 && dense) VOD_SUBN(0, true, 128, 4, 2);
            if (store_dtype == 0 && !dense) VOD_SUBN(0, false, 128, 4, 2);
            if (store_dtype == 1 && dense) VOD_SUBN(1, true, 128, 4, 2);
            if (store_dtype == 1 && !dense) VOD_SUBN(1, false, 128, 4, 2);
        }
#undef VOD_SUBN
        if (tile == 1) {
#define VOD_SUB1(DT, DENSE) return launch_filter_cfg<DT, 128, 128, 2, 2, 64, 2, DENSE, 0, false, true>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream)
            if (store_dtype == 0 && dense) VOD_SUB1(0, true);
            if (store_dtype == 0 && !dense) VOD_SUB1(0, false);
            if (store_dtype == 1 && dense) VOD_SUB1(1, true);
            if (store_dtype == 1 && !dense) VOD_SUB1(1, false);
#undef VOD_SUB1
        }
    }
#define VOD_FILTER(DT, TILE, BM, BN, WM, WN, BK, NS)                                                                   \
    if (store_dtype == DT && tile == TILE) {                                                                           \
        return dense ? launch_filter_cfg<DT, BM, BN, WM, WN, BK, NS, true>(store, q_pad, dim_pad, row_begin, row_end,  \
                                                                           nq, nq_pad, ws, stream)                     \
                     : launch_filter_cfg<DT, BM, BN, WM, WN, BK, NS, false>(store, q_pad, dim_pad, row_begin, row_end, \
                                                                            nq, nq_pad, ws, stream);                   \
    }
#define VOD_FILTER_DT(TILE, BM, BN, WM, WN, BK, NS) VOD_FILTER(0, TILE, BM, BN, WM, WN, BK, NS) VOD_FILTER(1, TILE, BM, BN, WM, WN, BK, NS)
    VOD_FILTER_DT(1, 128, 128, 2, 2, 64, 2)   // 64 KB LDS, 2 workgroups / CU
    VOD_FILTER_DT(2, 256, 256, 2, 4, 64, 2)   // 128 KB LDS, drain-to-zero double buffer
    VOD_FILTER_DT(3, 256, 256, 2, 4, 32, 4)   // 128 KB LDS, 4-slot ring, 2 slices in flight across the barrier
    VOD_FILTER_DT(5, 128, 128, 2, 2, 32, 4)   // 64 KB LDS ring, 2 workgroups / CU
    VOD_FILTER_DT(42, 256, 64, 4, 1, 64, 3)   // nq <= 64 (HBM-bound): 256 corpus rows x 64 queries, 4 waves, 3-slot ring (120 KB): a fifth of the LDS-DMA bytes are queries (half with the 128x128 tile)
    VOD_FILTER_DT(46, 256, 128, 4, 2, 64, 3)  // 65..128 queries: 256 x 128, 8 waves, 3-slot ring (144 KB)
#undef VOD_FILTER_DT
    if ((tile == 9 || tile == 10) && !dense) {  // persistent 256x256 / 16x16x32 (10: 3 corpus + 2 query LDS slots): one workgroup per CU streams its list of corpus tiles
        const int n_xtiles = (int)((row_end - row_begin + 255) / 256);
        const int n_qtiles = (int)(nq_pad / 256);
        int dev = 0, n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        const int unit = 8 * n_qtiles;
        const int total = ((n_xtiles + 7) / 8) * unit;
        int grid = (n_cu / unit) * unit;
        if (grid < unit) grid = unit;
        if (grid > total) grid = total;
        const size_t lds = tile == 10 ? 160 * 1024 : 128 * 1024 + PSTG_BYTES;
#define VOD_K16P(DT)                                                                                                   \
    {                                                                                                                  \
        auto kern = tile == 10 ? (subset ? mips_filter16p_kernel<DT, true, true> : mips_filter16p_kernel<DT, true, false>)  \
                               : (subset ? mips_filter16p_kernel<DT, false, true> : mips_filter16p_kernel<DT, false, false>); \
        (void)0;                                                                      \
        hipError_t e = allow_dynamic_lds((const void*)kern, (int)lds);   \
        if (e != hipSuccess) return e;                                                                                 \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, stream, (const uint16_t*)store,                 \
                           (const uint16_t*)q_pad, (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles,     \
                           (int)nq, ws.thr_s, ws.thr_key, ws.cand, ws.cnt, (int)ws.cap, ws.overflow, ws.extra);        \
        return hipGetLastError();                                                                                      \
    }
        if (store_dtype == 0) VOD_K16P(0)
        if (store_dtype == 1) VOD_K16P(1)
#undef VOD_K16P
    }
    if (tile == 8 || tile == 9 || tile == 10) {  // 256x256, 16x16x32 MFMA shape (9, 10: dense chunk of the persistent flavours)
        const int n_xtiles = (int)((row_end - row_begin + 255) / 256);
        const int n_qtiles = (int)(nq_pad / 256);
        const unsigned grid = (unsigned)((n_xtiles + 7) / 8) * 8u * (unsigned)n_qtiles;
        constexpr size_t lds = 128 * 1024;
#define VOD_K16(DT, DENSE)                                                                                             \
    {                                                                                                                  \
        auto kern = subset ? mips_filter16_kernel<DT, DENSE, true> : mips_filter16_kernel<DT, DENSE, false>;           \
        hipError_t e = allow_dynamic_lds((const void*)kern, (int)lds);   \
        if (e != hipSuccess) return e;                                                                                 \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, (const uint16_t*)store, (const uint16_t*)q_pad,   \
                           (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s,          \
                           ws.thr_key, ws.cand, ws.cnt, (int)ws.cap, ws.overflow, ws.extra);                                     \
        return hipGetLastError();                                                                                      \
    }
        if (store_dtype == 0 && dense) VOD_K16(0, true)
        if (store_dtype == 0 && !dense) VOD_K16(0, false)
        if (store_dtype == 1 && dense) VOD_K16(1, true)
        if (store_dtype == 1 && !dense) VOD_K16(1, false)
#undef VOD_K16
    }
    if (tile == 6 || tile == 7) {  // specialised: 8 MFMA waves + 4 loader waves
        const int n_xtiles = (int)((row_end - row_begin + 255) / 256);
        const int n_qtiles = (int)(nq_pad / 256);
        const unsigned grid = (unsigned)((n_xtiles + 7) / 8) * 8u * (unsigned)n_qtiles;
        constexpr size_t lds = 128 * 1024;
#define VOD_SPEC(DT, BKK, NS, DENSE)                                                                                   \
    {                                                                                                                  \
        auto kern = mips_filter_spec_kernel<DT, BKK, NS, 4, DENSE>;                                                    \
        hipError_t e = allow_dynamic_lds((const void*)kern, (int)lds);   \
        if (e != hipSuccess) return e;                                                                                 \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, stream, (const uint16_t*)store, (const uint16_t*)q_pad,   \
                           (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s,          \
                           ws.thr_key, ws.cand, ws.cnt, (int)ws.cap, ws.overflow, ws.extra);                                     \
        return hipGetLastError();                                                                                      \
    }
        if (tile == 6) {
            if (store_dtype == 0 && dense) VOD_SPEC(0, 64, 2, true)
            if (store_dtype == 0 && !dense) VOD_SPEC(0, 64, 2, false)
            if (store_dtype == 1 && dense) VOD_SPEC(1, 64, 2, true)
            if (store_dtype == 1 && !dense) VOD_SPEC(1, 64, 2, false)
        } else {
            if (store_dtype == 0 && dense) VOD_SPEC(0, 32, 4, true)
            if (store_dtype == 0 && !dense) VOD_SPEC(0, 32, 4, false)
            if (store_dtype == 1 && dense) VOD_SPEC(1, 32, 4, true)
            if (store_dtype == 1 && !dense) VOD_SPEC(1, 32, 4, false)
        }
#undef VOD_SPEC
    }
    if (tile == 4) {  // 256x256, BK=32, 4-slot ring, ping-pong between the two waves of each SIMD
#define VOD_PP(DT, DENSE) return launch_filter_cfg<DT, 256, 256, 2, 4, 32, 4, DENSE, 0, true>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream)
        if (store_dtype == 0 && dense) VOD_PP(0, true);
        if (store_dtype == 0 && !dense) VOD_PP(0, false);
        if (store_dtype == 1 && dense) VOD_PP(1, true);
        if (store_dtype == 1 && !dense) VOD_PP(1, false);
#undef VOD_PP
    }
#ifdef VODHIP_ABLATION  // timing-only builds (wrong results): which resource bounds the K loop?
    if (store_dtype == 0 && tile >= 11 && tile <= 13 && !dense) {
        if (tile == 11) return launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, 1>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
        if (tile == 12) return launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, 2>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
        if (tile == 13) return launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, 3>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    }
    if (store_dtype == 0 && tile == 15 && !dense)  // DMA-only, 4-slot ring of 32-deep slices
        return launch_filter_cfg<0, 256, 256, 2, 4, 32, 4, false, 4>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    if (store_dtype == 0 && tile == 16 && !dense)  // ring without DMA in the loop
        return launch_filter_cfg<0, 256, 256, 2, 4, 32, 4, false, 1>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    if (store_dtype == 0 && tile == 22)  // tile 2 WITHOUT the rotated K order (A/B reference)
        return dense ? launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, true, 10>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream)
                     : launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, 10>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
#define VOD_KNOB(T, AB) if (store_dtype == 0 && tile == T) return dense ? launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, true, AB>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream) : launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, AB>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    VOD_KNOB(25, 13)  // WITH s_setprio flips around the MFMA groups
    VOD_KNOB(26, 12)  // static priority for waves 4..7 instead
    VOD_KNOB(27, 14)  // corpus operand always L2-hot (timing only)
    VOD_KNOB(28, 15)  // L2 touch-prefetch of the next corpus tile (results stay exact)
#undef VOD_KNOB
    if (store_dtype == 0 && tile >= 32 && tile <= 38 && !dense) {  // ablated persistent kernel: tile - 31 = bit mask (1 no LDS-DMA, 2 no ds_read, 4 no MFMA)
        const int n_xtiles = (int)((row_end - row_begin + 255) / 256);
        const int n_qtiles = (int)(nq_pad / 256);
        const int unit = 8 * n_qtiles, total = ((n_xtiles + 7) / 8) * unit;
        int grid = (256 / unit) * unit;
        if (grid > total) grid = total;
        void (*kern)(const uint16_t*, const uint16_t*, int, int, int, int, int, int, const float*, const key_t64*, key_t64*,
                     unsigned int*, int, unsigned int*, FilterExtra) = nullptr;
        switch (tile - 31) {
            case 1: kern = mips_filter16p_kernel<0, false, false, false, 1>; break;
            case 2: kern = mips_filter16p_kernel<0, false, false, false, 2>; break;
            case 3: kern = mips_filter16p_kernel<0, false, false, false, 3>; break;
            case 4: kern = mips_filter16p_kernel<0, false, false, false, 4>; break;
            case 5: kern = mips_filter16p_kernel<0, false, false, false, 5>; break;
            case 6: kern = mips_filter16p_kernel<0, false, false, false, 6>; break;
            default: kern = mips_filter16p_kernel<0, false, false, false, 7>; break;
        }
        hipError_t e = allow_dynamic_lds((const void*)kern, 128 * 1024 + PSTG_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), 128 * 1024 + PSTG_BYTES, stream, (const uint16_t*)store, (const uint16_t*)q_pad,
                           (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key, ws.cand,
                           ws.cnt, (int)ws.cap, ws.overflow, ws.extra);
        return hipGetLastError();
    }
    if (store_dtype == 0 && tile == 29 && !dense) {  // stamped persistent kernel
        const int n_xtiles = (int)((row_end - row_begin + 255) / 256);
        const int n_qtiles = (int)(nq_pad / 256);
        const int unit = 8 * n_qtiles, total = ((n_xtiles + 7) / 8) * unit;
        int grid = (256 / unit) * unit;
        if (grid > total) grid = total;
        auto kern = mips_filter16p_kernel<0, false, false, true>;
        hipError_t e = allow_dynamic_lds((const void*)kern, 128 * 1024 + PSTG_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), 128 * 1024 + PSTG_BYTES, stream, (const uint16_t*)store, (const uint16_t*)q_pad,
                           (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key, ws.cand,
                           ws.cnt, (int)ws.cap, ws.overflow, ws.extra);
        return hipGetLastError();
    }
    if (store_dtype == 0 && tile == 20 && !dense)  // DMA-only, corpus operand only
        return launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, 7>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    if (store_dtype == 0 && tile == 21 && !dense)  // DMA-only, query operand only
        return launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, 8>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    if (store_dtype == 0 && tile == 19 && !dense)  // DMA-only, blocked source layout
        return launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, 6>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    if (store_dtype == 0 && (tile == 18 || tile == 23 || tile == 24) && !dense) {  // stamped builds of tile 7 (23: no ds_read, 24: no MFMA)
        const int n_xtiles = (int)((row_end - row_begin + 255) / 256);
        const int n_qtiles = (int)(nq_pad / 256);
        const unsigned grid = (unsigned)((n_xtiles + 7) / 8) * 8u * (unsigned)n_qtiles;
        auto kern = mips_filter_spec_kernel<0, 32, 4, 4, false, true>;
        if (tile == 23) kern = mips_filter_spec_kernel<0, 32, 4, 4, false, true, 1>;
        if (tile == 24) kern = mips_filter_spec_kernel<0, 32, 4, 4, false, true, 2>;
        hipError_t e = allow_dynamic_lds((const void*)kern, 128 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(768), 128 * 1024, stream, (const uint16_t*)store, (const uint16_t*)q_pad,
                           (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key,
                           ws.cand, ws.cnt, (int)ws.cap, ws.overflow, ws.extra);
        return hipGetLastError();
    }
    if (store_dtype == 0 && tile == 17 && !dense)  // stamped build of tile 2
        return launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, 5>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    if (store_dtype == 0 && tile == 14 && !dense) {
        return launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, false, 4>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    }
    if (store_dtype == 0 && ((tile >= 11 && tile <= 24) || tile == 29 || (tile >= 32 && tile <= 38)) && dense)
        return launch_filter_cfg<0, 256, 256, 2, 4, 64, 2, true>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
#endif
#undef VOD_FILTER
    return hipErrorInvalidValue;
}

hipError_t launch_select(const SearchWorkspace& ws, int64_t nq, int k, int64_t dense_n, bool final_sort, hipStream_t stream,
                         int64_t id_base, float* out_scores, int64_t* out_ids) {
    int sb = 2048;  // keys per workgroup buffer: 16 KB -> 8 workgroups per CU
    while (sb < 2 * ws.kp) sb <<= 1;
    const size_t lds = (size_t)sb * sizeof(key_t64) + 264 * sizeof(int);
    hipLaunchKernelGGL(mips_select_kernel, dim3((unsigned)nq), dim3(256), lds, stream, ws.topk, (int)ws.kp, k, sb, ws.cand,
                       ws.cnt, (int)ws.cap, (int)dense_n, ws.thr_s, ws.thr_key, ws.overflow, final_sort ? 1 : 0, id_base,
                       final_sort ? out_scores : nullptr, final_sort ? out_ids : nullptr);
    return hipGetLastError();
}

hipError_t launch_output(const SearchWorkspace& ws, int64_t nq, int k, int64_t id_base, float* out_scores,
                         int64_t* out_ids, hipStream_t stream) {
    const int64_t total = nq * k;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(mips_output_kernel, dim3(blocks), dim3(256), 0, stream, ws.topk, (int)ws.kp, k, nq, id_base,
                       out_scores, out_ids);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// merge of per-shard top-k lists: one workgroup per query, bitonic sort of (score, id) pairs
// ordered by (score desc, id asc); invalid entries (id < 0) sink to the end.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pair_before(float sa, int64_t ia, float sb, int64_t ib) {
    // true when (sa, ia) must come before (sb, ib)
    const bool va = ia >= 0, vb = ib >= 0;
    if (va != vb) return va;
    if (!va) return false;
    if (sa != sb) return sa > sb;
    return ia < ib;
}

__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ scores, const int64_t* __restrict__ ids,
                                                         int64_t stride_s, int64_t stride_i, int n_shards, int64_t nq, int k, int k_out,
                                                         float* __restrict__ out_scores, int64_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int total = n_shards * k;
    int P = 64;
    while (P < total) P <<= 1;
    int64_t* sid = (int64_t*)smem;
    float* ssc = (float*)(smem + (size_t)P * 8);
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < P; i += 256) {
        float s = -__builtin_inff();
        int64_t id = -1;
        if (i < total) {
            const int sh = i / k, c = i % k;
            s = scores[(int64_t)sh * stride_s + q * k + c];
            id = ids[(int64_t)sh * stride_i + q * k + c];
            if (id < 0 || s != s) {
                id = -1;
                s = -__builtin_inff();
            }
        }
        ssc[i] = s;
        sid[i] = id;
    }
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (P >> 1); t += 256) {
                const int pos = 2 * t - (t & (stride - 1));
                const float sa = ssc[pos], sb = ssc[pos + stride];
                const int64_t ia = sid[pos], ib = sid[pos + stride];
                const bool fwd = (pos & size) == 0;  // this run must end up best-first
                const bool swap = fwd ? pair_before(sb, ib, sa, ia) : pair_before(sa, ia, sb, ib);
                if (swap) {
                    ssc[pos] = sb;
                    ssc[pos + stride] = sa;
                    sid[pos] = ib;
                    sid[pos + stride] = ia;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < k_out; i += 256) {
        float s = -__builtin_inff();
        int64_t id = -1;
        if (i < P) {
            s = ssc[i];
            id = sid[i];
        }
        out_scores[q * k_out + i] = s;
        out_ids[q * k_out + i] = id;
    }
}

hipError_t launch_merge_topk(const float* scores, const int64_t* ids, int64_t stride_s, int64_t stride_i, int n_shards,
                             int64_t nq, int k, int k_out, float* out_scores, int64_t* out_ids, hipStream_t stream) {
    if (nq == 0) return hipSuccess;
    const int total = n_shards * k;
    size_t P = 64;
    while ((int)P < total) P <<= 1;
    const size_t lds = P * 12;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = allow_dynamic_lds((const void*)merge_topk_kernel, 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)nq), dim3(256), lds, stream, scores, ids, stride_s, stride_i, n_shards, nq, k, k_out,
                       out_scores, out_ids);
    return hipGetLastError();
}

}  // namespace experiments
}  // namespace vodhip
#endif  // VODHIP_ABLATION

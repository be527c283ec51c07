// libvodhip -- the FILTER stage on a 384 x 256 workgroup tile (tile 12).
//
// The K loop of the 256 x 256 kernel (kernels_mips.hip) is power-limited on MI355X: the chip holds 2.3 GHz on the MFMAs alone,
// 1.8 GHz once the fragment reads run beside them and 1.6 GHz with the LDS-DMA as well (tools/ubench/kloop.hip), so removing
// stall cycles lowers the clock instead of the run time.  What does help is less energy per flop, i.e. fewer LDS and LDS-DMA
// bytes per MFMA, and those are set by the tile shapes alone:
//   wave tile 128 x 64 (8 + 4 fragments per 32 MFMAs)  ->  96 x 128 (6 + 8 fragments per 48 MFMAs): 22 % fewer LDS read bytes / flop
//   workgroup tile 256 x 256                          ->  384 x 256:                               17 % fewer LDS-DMA bytes / flop
// 8 waves = 4 (corpus) x 2 (queries); 192 accumulator registers per lane, which only fits because every LDS-DMA piece is
// hand-issued in the SGPR-base form with ONE shared 32-bit lane offset (the builtin keeps a 64-bit pointer per piece) and the
// thresholds live in LDS.  Both operands travel in 32-deep k-steps through a ring of three 40 KB slots (corpus 384 rows x 64 B +
// queries 256 rows x 64 B), requested two k-steps ahead in pieces of 16 rows x 64 B; one `s_waitcnt vmcnt(5)` + `s_barrier`
// per k-step (a wave's 5 youngest vector-memory operations are the pieces of the previous k-step; everything older - what this
// k-step reads - has landed).  120 KB of ring + 15 KB of survivor lists + 1 KB of thresholds.
// Fragment maps, epilogue and survivor lists as in kernels_mips.hip; results are bit-identical (same products, same order).
// FILTER only: the bootstrap (group maxima) runs on the 256 x 256 kernel, dense chunks on the 128 x 128 one.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mips_common.h"

namespace vodhip {

namespace {
constexpr int WIDE_BM = 384, WIDE_BN = 256;
constexpr int WIDE_A_BYTES = WIDE_BM * 64;                  // corpus part of a slot: 384 rows x 64 B
constexpr int WIDE_SLOT = WIDE_A_BYTES + WIDE_BN * 64;      // 40 KB
constexpr int WIDE_LIST_BASE = 3 * WIDE_SLOT;
constexpr int WIDE_WL_CAP = 160;                            // records per wave list
constexpr int WIDE_WL_FLUSH = 96;                           // flush when at least this many are pending (checked once per tile)
constexpr int WIDE_THR_BASE = WIDE_LIST_BASE + 8 * WIDE_WL_CAP * 12;
constexpr int WIDE_LDS_BYTES = WIDE_THR_BASE + WIDE_BN * 4;
static_assert(WIDE_LDS_BYTES <= 160 * 1024, "LDS budget");

// One LDS-DMA piece (16 B per lane, 1 KiB per wave): global address = `sbase` (wave-uniform SGPR pair) + `voff` (32-bit per-lane
// offset), LDS address = `lds_addr` (wave-uniform, through M0) + 16 * lane.  NT: the nt cache policy.
template <bool NT>
__device__ __forceinline__ void wide_glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
}  // namespace

template <int DT, bool SUBSET>
__global__ __launch_bounds__(512, 2) void mips_filter16w_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end, int n_xtiles,
    int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key, key_t64* __restrict__ cand,
    unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    constexpr int BM = WIDE_BM, BN = WIDE_BN, NWAVES = 8, TM = 96, TN = 128, MB = TM / 16, NB16 = TN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    const int wm = wave >> 1, wn = wave & 1;
    const int q0 = qt * BN;
    const int nk2 = dim_pad / 32;  // k-steps per tile
    const int row_bytes = dim_pad * 2;
    const size_t tile_step_bytes = (size_t)xt_step * BM * (size_t)row_bytes;
    const unsigned smem_off = (unsigned)(size_t)(VOD_AS3 const char*)smem;

    // ---- LDS image of a k-step: 64-byte rows; the 16-byte chunk c of row r sits at physical chunk c ^ g((r >> 2) & 3),
    // g = (0, 2, 3, 1): the 16 rows x 4 lane groups of a fragment read then hit 64 distinct banks.  A piece = 16 rows x 64 B:
    // lane l loads row (l >> 2), physical chunk (l & 3); (r >> 2) & 3 = (l >> 4) & 3 for every piece (pieces start at multiples of 16).
    const unsigned voff = (unsigned)(lane >> 2) * (unsigned)row_bytes + (unsigned)(((lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3)) << 4);
    const char* c_base = (const char*)X + ((size_t)row_begin + (size_t)xt0 * BM + (size_t)wave * 48) * (size_t)row_bytes;  // this wave's 3 corpus pieces
    const char* const q_base = (const char*)Q + ((size_t)q0 + (size_t)wave * 32) * (size_t)row_bytes;                        // and 2 query pieces
    const bool corpus_nt = (ex.flags & FILTER_FLAG_CORPUS_NT) != 0;

    const int fr = lane & 15, fq = lane >> 4;
    const int chunk = (fq ^ ((0x78 >> (2 * ((fr >> 2) & 3))) & 3)) << 4;
    const int a_row_off = (wm * TM + fr) * 64 + chunk;
    const int b_row_off = WIDE_A_BYTES + (wn * TN + fr) * 64 + chunk;

    // thresholds of the workgroup's 256 queries: LDS (the epilogue reads them)
    float* const thr_lds = (float*)(smem + WIDE_THR_BASE);
    if (tid < BN) {
        const int q = q0 + tid;
        thr_lds[tid] = q < nq ? thr_s[q] : __builtin_inff();
    }
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0), visible to the waitcnt pass: nothing of this is pending later
    __builtin_amdgcn_s_barrier();

    // ---- fetch cursor: k-step f_t of tile ordinal f_it goes to ring slot f_slot; past the last tile it stays on that tile (the
    // extra pieces land in slots nobody reads: the piece COUNT per k-step never changes, which is what vmcnt(5) relies on)
    int f_it = 0, f_t = 0, f_kbyte = 0, f_slot = 0;
    auto dma = [&](int piece) {  // 0..2 corpus, 3..4 queries
        const unsigned slot = smem_off + (unsigned)(f_slot * WIDE_SLOT);
        if (piece < 3) {
            const char* src = c_base + (f_kbyte + piece * 16 * row_bytes);
            const unsigned dst = slot + (wave * 3 + piece) * 16 * 64;
            if (corpus_nt) wide_glds16<true>(src, voff, dst);
            else wide_glds16<false>(src, voff, dst);
        } else {
            wide_glds16<false>(q_base + (f_kbyte + (piece - 3) * 16 * row_bytes), voff, slot + WIDE_A_BYTES + (wave * 2 + piece - 3) * 16 * 64);
        }
    };
    auto next_fetch = [&]() {
        f_kbyte += 64;
        f_slot = f_slot == 2 ? 0 : f_slot + 1;
        if (++f_t == nk2) {
            f_t = 0;
            f_kbyte = 0;
            if (f_it + 1 < n_my) {
                ++f_it;
                c_base += tile_step_bytes;
            }
        }
    };

    // ---- per-wave survivor list (see kernels_mips.hip) -------------------------------------------------
    key_t64* const wl_key = (key_t64*)(smem + WIDE_LIST_BASE) + wave * WIDE_WL_CAP;
    int* const wl_q = (int*)(smem + WIDE_LIST_BASE + NWAVES * WIDE_WL_CAP * 8) + wave * WIDE_WL_CAP;
    int wl_n = 0;  // wave-uniform
    auto wl_flush = [&]() {
        const int n = wl_n < WIDE_WL_CAP ? wl_n : WIDE_WL_CAP;
        constexpr int PER_LANE = (WIDE_WL_CAP + 63) / 64;
        key_t64 fk[PER_LANE];
        int fq_[PER_LANE];
        bool ok[PER_LANE];
        unsigned slot[PER_LANE];
#pragma unroll
        for (int u = 0; u < PER_LANE; ++u) {
            const int e = lane + 64 * u;
            ok[u] = e < n;
            fk[u] = ok[u] ? wl_key[e] : 0ull;
            fq_[u] = ok[u] ? wl_q[e] : 0;
        }
#pragma unroll
        for (int u = 0; u < PER_LANE; ++u) {
            ok[u] = ok[u] && fk[u] > thr_key[fq_[u]];
            if constexpr (SUBSET) ok[u] = ok[u] && subset_allows(ex, fq_[u], (int)(0xFFFFFFFFu - (unsigned)fk[u]));
        }
#pragma unroll
        for (int u = 0; u < PER_LANE; ++u) slot[u] = ok[u] ? atomicAdd(&cnt[(size_t)fq_[u] * CNT_STRIDE], 1u) : 0u;
#pragma unroll
        for (int u = 0; u < PER_LANE; ++u) {
            if (ok[u]) {
                if (slot[u] < (unsigned)cap)
                    cand[(size_t)fq_[u] * cap + slot[u]] = fk[u];
                else
                    atomicOr(overflow, 1u);
            }
        }
        // a wait the compiler's waitcnt pass can SEE (the builtin, not inline asm): otherwise it carries the returning atomics
        // of this cold path to the head of the tile loop as possibly pending and drains the LDS-DMA ring there, once per tile
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        wl_n = 0;
    };
    auto wl_append = [&](bool p, key_t64 key, int q) {
        const unsigned long long bal = __ballot(p);
        if (bal == 0ull) return;
        const int pos = wl_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (p) {
            if (pos < WIDE_WL_CAP) {
                wl_key[pos] = key;
                wl_q[pos] = q;
            } else {
                emit_candidate<SUBSET>(key, q, thr_key, cand, cnt, cap, overflow, ex);
            }
        }
        if (wl_n + __builtin_popcountll(bal) > WIDE_WL_CAP) __builtin_amdgcn_s_waitcnt(0x0F70);  // (see wl_flush)
        wl_n += __builtin_popcountll(bal);
    };

    u32x4 fa[MB], fb[NB16];
    f32x4 acc[MB][NB16];
    auto mma = [&](int i0, int i1, bool zero_c) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB16; ++j)
                if (i >= i0 && i < i1) acc[i][j] = mfma16<DT>(fa[i], fb[j], zero_c ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j]);
    };

    // ---- epilogue of one tile: lane l holds, for each of its 8 queries (column block j, column l & 15), the 24 rows
    // 4 (l >> 4) + reg of row blocks i = 0..5 ------------------------------------------------------------
    auto epilogue = [&](int x0) {
#pragma unroll
        for (int j = 0; j < NB16; ++j) {
            const int q = q0 + wn * TN + j * 16 + fr;
            float m = acc[0][j][0];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[i][j][r]);
            const float th = thr_lds[wn * TN + j * 16 + fr];
            const bool hit = m >= th;  // false for NaN and for padded queries (thr = +inf)
            if (__any(hit)) {
                // cold path; opaque copies keep everything derived from the tile's row base / the query inside it
                int x0_o = x0, row_end_o = row_end, q_o = q;
                asm volatile("" : "+s"(x0_o), "+s"(row_end_o), "+v"(q_o));
                auto val = [&](int v) { return acc[v >> 2][j][v & 3]; };
                unsigned mask = 0;
                if (hit) {
#pragma unroll
                    for (int v = 0; v < MB * 4; ++v) mask |= (val(v) >= th) ? (1u << v) : 0u;
                }
                const bool multi = __any((mask & (mask - 1u)) != 0u);
                do {
                    const bool p = mask != 0u;
                    const int b = p ? __builtin_ctz(mask) : 0;  // 0..23
                    mask &= mask - 1u;
                    float sc = m;
                    if (multi) {
                        // register select by the bits of b over 24 values: 12 + 6 + 3 + 1 + 1 v_cndmask (inline asm: as C++ selects
                        // LLVM turns the tree into an indexed load from a scratch copy of the accumulator)
                        const unsigned long long s0 = __ballot(b & 1), s1 = __ballot(b & 2), s2 = __ballot(b & 4),
                                                 s3 = __ballot(b & 8), s4 = __ballot(b & 16);
                        auto sel = [](float lo, float hi, unsigned long long sm) {
                            float r;
                            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(hi), "s"(sm));
                            return r;
                        };
                        float t12[12], t6[6], t3[3];
#pragma unroll
                        for (int u = 0; u < 12; ++u) t12[u] = sel(val(2 * u), val(2 * u + 1), s0);
#pragma unroll
                        for (int u = 0; u < 6; ++u) t6[u] = sel(t12[2 * u], t12[2 * u + 1], s1);
#pragma unroll
                        for (int u = 0; u < 3; ++u) t3[u] = sel(t6[2 * u], t6[2 * u + 1], s2);
                        sc = sel(sel(t3[0], t3[1], s3), t3[2], s4);  // b < 16: bit 3 picks t3[0] / t3[1]; b >= 16: t3[2]
                    }
                    const int rw = x0_o + wm * TM + 4 * fq + (b >> 2) * 16 + (b & 3);
                    wl_append(p && rw < row_end_o, make_key(sc, (unsigned)rw), q_o);
                } while (__any(mask != 0u));
            }
        }
        if (wl_n >= WIDE_WL_FLUSH) wl_flush();
    };

    // ---- the k-step stream: k-step h reads ring slot h % 3 and requests k-step h + 2 into the slot k-step h - 1 has left ------
    for (int p = 0; p < 5; ++p) dma(p);
    next_fetch();
    for (int p = 0; p < 5; ++p) dma(p);
    next_fetch();
    int r_slot = 0;
    auto kstep = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        wait_vmcnt<5>();
        __builtin_amdgcn_s_barrier();
        const char* base = smem + r_slot * WIDE_SLOT;
        auto read_a = [&](int i) { fa[i] = *(const u32x4*)(base + a_row_off + i * 16 * 64); };
        // the corpus fragments roll through THREE register sets: row block i + 3 is read once the MFMAs of row block i are issued
        // (its registers are free then; 3 x 8 MFMAs = 384 cycles cover the LDS round trip).  192 accumulator + 32 query-fragment
        // + 12..16 corpus-fragment registers leave a dozen for everything else.
#pragma unroll
        for (int j = 0; j < NB16; ++j) fb[j] = *(const u32x4*)(base + b_row_off + j * 16 * 64);
        read_a(0);
        read_a(1);
        read_a(2);
        __builtin_amdgcn_sched_barrier(0);
        dma(0);
        dma(1);
        mma(0, 1, FIRST);
        __builtin_amdgcn_sched_barrier(0);
        read_a(3);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 2, FIRST);
        __builtin_amdgcn_sched_barrier(0);
        read_a(4);
        dma(2);
        __builtin_amdgcn_sched_barrier(0);
        mma(2, 3, FIRST);
        __builtin_amdgcn_sched_barrier(0);
        read_a(5);
        dma(3);
        dma(4);
        __builtin_amdgcn_sched_barrier(0);
        mma(3, MB, FIRST);
        __builtin_amdgcn_sched_barrier(0);
        next_fetch();
        r_slot = r_slot == 2 ? 0 : r_slot + 1;
    };
    for (int it = 0; it < n_my; ++it) {
        kstep(std::true_type{});
        for (int t = 1; t < nk2; ++t) kstep(std::false_type{});
        epilogue(row_begin + (xt0 + it * xt_step) * BM);
    }
    wait_vmcnt<0>();  // the run-ahead pieces of the last k-steps
    wl_flush();
}

// ---- launcher -----------------------------------------------------------------------------------------
namespace {
template <int DT, bool SUBSET>
hipError_t launch_wide(const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end, int64_t nq,
                       int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    const int n_qtiles = (int)(nq_pad / WIDE_BN);
    const int n_xtiles = (int)((row_end - row_begin + WIDE_BM - 1) / WIDE_BM);
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const int unit = 8 * n_qtiles;
    const int total = ((n_xtiles + 7) / 8) * unit;
    int grid = (n_cu / unit) * unit;
    if (grid < unit) grid = unit;
    if (grid > total) grid = total;
    auto kern = mips_filter16w_kernel<DT, SUBSET>;
    if (hipError_t e = allow_dynamic_lds((const void*)kern, WIDE_LDS_BYTES); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), WIDE_LDS_BYTES, stream, (const uint16_t*)store, (const uint16_t*)q_pad,
                       (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key, ws.cand,
                       ws.cnt, (int)ws.cap, ws.overflow, ws.extra);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_filter_wide(int store_dtype, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end,
                              int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    const bool subset = ws.extra.row_label != nullptr;
    if (store_dtype == 0)
        return subset ? launch_wide<0, true>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream)
                      : launch_wide<0, false>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    return subset ? launch_wide<1, true>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream)
                  : launch_wide<1, false>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
}

}  // namespace vodhip

// libvodhip -- the FILTER stage on a deeper LDS ring (tiles 10 / 11).
//
// Same tile, wave layout, fragment maps, epilogue and survivor lists as the persistent 256 x 256 kernel of
// kernels_mips.hip (8 waves = 2 x 4 of 128 x 64, v_mfma_f32_16x16x32, scores never leave registers); what differs is how
// the operands reach the LDS.  There, both operands travel in 64-deep slices through two 64 KB slots: the slice a workgroup
// multiplies next was requested ONE slice (~1.8 us) earlier, and the s_waitcnt vmcnt(0) + s_barrier in front of every slice
// waits for the slowest of its 64 LDS-DMA pieces - with the corpus operand coming from HBM (first touch, or a hit on a line
// a sibling workgroup is still missing on) that wait is exposed: tools/ubench/kloop (the K loop rebuilt from its parts)
// runs the two-slot loop in 13.2 ms per headline batch with the operands streaming and in 10.9 ms with them L2-resident.
// Here the two operands get rings of their own, sized by what each needs:
//   corpus   3 slots x (256 rows x 64 k) = 96 KB, requested TWO slices ahead (HBM latency),
//   queries  3 slots x (256 rows x 32 k) = 48 KB, half slices requested 2 (3) k-steps ahead (an L2 hit: a k-step is enough),
//   8 survivor lists x 160 records = 15 KB, the 256 thresholds 1 KB                                (all 160 KB).
// One `s_waitcnt vmcnt(6)` + `s_barrier` per 32-deep k-step: the six youngest vector-memory operations of a wave are always
// the pieces of the two most recent k-steps, everything older - in particular what this k-step reads - has landed.
// PIPE (tile 11) additionally reads the fragments of k-step h+1 while k-step h multiplies (two fragment sets in registers).
// Results are bit-identical to tile 8: same products, same summation order.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mips_common.h"

namespace vodhip {

namespace {
constexpr int RING_C_SLOT = 256 * 128;           // one 64-deep corpus slice: 256 rows x 128 B
constexpr int RING_Q_SLOT = 256 * 64;            // one 32-deep query half slice: 256 rows x 64 B
constexpr int RING_Q_BASE = 3 * RING_C_SLOT;
constexpr int RING_LIST_BASE = RING_Q_BASE + 3 * RING_Q_SLOT;
constexpr int RING_WL_CAP = 160;                 // records per wave list
constexpr int RING_WL_FLUSH = 96;                // flush when at least this many are pending (checked once per tile)
constexpr int RING_THR_BASE = RING_LIST_BASE + 8 * RING_WL_CAP * 12;  // 256 thresholds
constexpr int RING_LDS_BYTES = RING_THR_BASE + 256 * 4;
static_assert(RING_LDS_BYTES <= 160 * 1024, "LDS budget");
}  // namespace

// One LDS-DMA piece (16 B per lane, 1 KiB per wave) in the SGPR-base form: global address = `sbase` (wave-uniform, an SGPR
// pair) + `voff` (32-bit per-lane offset); LDS address = `lds_addr` (wave-uniform, via M0) + 16 * lane.  Hand-issued because
// the builtin makes LLVM keep a 64-bit VGPR pointer per piece (12 pieces: 24 registers this kernel does not have), and
// because its waitcnt pass then guards LDS reads against the pieces it knows to be in flight.  NT: the nt cache policy.
template <bool NT>
__device__ __forceinline__ void glds16_saddr(const void* sbase, unsigned voff, unsigned lds_addr) {
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_offset_of(const void* p) { return (unsigned)(size_t)(VOD_AS3 const char*)p; }

#ifndef RING_PIPE_ACROSS_TILES
#define RING_PIPE_ACROSS_TILES 1
#endif
template <int DT, bool SUBSET, bool PIPE>
__global__ __launch_bounds__(512, 2) void mips_filter16r_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end, int n_xtiles,
    int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key, key_t64* __restrict__ cand,
    unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    constexpr int BM = 256, BN = 256, WN = 4, NWAVES = 8, TM = 128, TN = 64, MB = TM / 16, NB16 = TN / 16;
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
    const int wm = wave / WN, wn = wave % WN;
    const int q0 = qt * BN;
    const int nk = dim_pad / 64;  // 64-deep slices per tile = pairs of k-steps
    const int row_bytes = dim_pad * 2;
    const size_t tile_step_bytes = (size_t)xt_step * BM * (size_t)row_bytes;

    // ---- LDS-DMA sources: wave-uniform base (SGPRs) + one 32-bit per-lane offset, so that the pieces cost 3 VGPRs, not 12
    // 64-bit pointers.  Corpus piece t of a slice: rows (wave*4 + t)*8 .. +7, 128 B each, 16-byte chunk c of row r at physical
    // chunk c ^ ((r >> 1) & 7) (the fragment reads below are then bank-conflict free); (r >> 1) & 7 = 4 (t & 1) + (lane >> 4).
    // Query piece t of a half slice: rows (wave*2 + t)*16 .. +15, 64 B each, chunk c of row r at c ^ g((r >> 2) & 3) with
    // g = (0, 2, 3, 1); (r >> 2) & 3 = (lane >> 4) & 3.
    const char* c_base = (const char*)X + ((size_t)row_begin + (size_t)xt0 * BM + (size_t)wave * 32) * (size_t)row_bytes;
    const char* const q_base = (const char*)Q + ((size_t)q0 + (size_t)wave * 32) * (size_t)row_bytes;
    unsigned a_voff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        a_voff[par] = (unsigned)(lane >> 3) * (unsigned)row_bytes + (unsigned)(((lane & 7) ^ (4 * par + (lane >> 4))) << 4);
    const unsigned b_voff = (unsigned)(lane >> 2) * (unsigned)row_bytes + (unsigned)(((lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3)) << 4);
    const bool corpus_nt = (ex.flags & FILTER_FLAG_CORPUS_NT) != 0;  // single q-tile: corpus lines are read once (nt cache policy)

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;
    const int a_row_off = (wm * TM + fr) * 128;
    const int gq = (0x78 >> (2 * ((fr >> 2) & 3))) & 3;
    const int b_row_off = RING_Q_BASE + (wn * TN + fr) * 64 + ((fq ^ gq) << 4);

    // thresholds of the workgroup's 256 queries: LDS (the epilogue reads them; 4 VGPRs less across the K loop)
    float* const thr_lds = (float*)(smem + RING_THR_BASE);
    if (tid < BN) {
        const int q = q0 + tid;
        thr_lds[tid] = q < nq ? thr_s[q] : __builtin_inff();
    }
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0), visible to the waitcnt pass: nothing of this is pending later
    __builtin_amdgcn_s_barrier();

    const unsigned smem_off = lds_offset_of(smem);
    // ---- fetch cursors ------------------------------------------------------------------------------
    // corpus: slice c_t of tile ordinal c_it goes to ring slot c_slot; past the last tile the cursor stays on it (the extra
    // pieces land in slots nobody reads: the piece COUNT per k-step never changes, which is what vmcnt(6) relies on)
    int c_it = 0, c_t = 0, c_kbyte = 0, c_slot = 0;
    auto dma_c = [&](int piece) {
        const unsigned dst = smem_off + c_slot * RING_C_SLOT + (wave * 4 + piece) * 8 * 128;
        const char* src = c_base + (c_kbyte + piece * 8 * row_bytes);
        if (corpus_nt) glds16_saddr<true>(src, a_voff[piece & 1], dst);
        else glds16_saddr<false>(src, a_voff[piece & 1], dst);
    };
    auto next_c = [&]() {
        c_kbyte += 128;
        c_slot = c_slot == 2 ? 0 : c_slot + 1;
        if (++c_t == nk) {
            c_t = 0;
            c_kbyte = 0;
            if (c_it + 1 < n_my) {
                ++c_it;
                c_base += tile_step_bytes;
            }
        }
    };
    // queries: the same 2 * nk half slices for every tile
    int q_kbyte = 0, q_slot = 0;
    auto dma_q = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            glds16_saddr<false>(q_base + (q_kbyte + t * 16 * row_bytes), b_voff, smem_off + RING_Q_BASE + q_slot * RING_Q_SLOT + (wave * 2 + t) * 16 * 64);
        q_kbyte += 64;
        if (q_kbyte == row_bytes) q_kbyte = 0;
        q_slot = q_slot == 2 ? 0 : q_slot + 1;
    };

    // ---- per-wave survivor list (see kernels_mips.hip) -------------------------------------------------
    key_t64* const wl_key = (key_t64*)(smem + RING_LIST_BASE) + wave * RING_WL_CAP;
    int* const wl_q = (int*)(smem + RING_LIST_BASE + NWAVES * RING_WL_CAP * 8) + wave * RING_WL_CAP;
    int wl_n = 0;  // wave-uniform
    auto wl_flush = [&]() {
        const int n = wl_n < RING_WL_CAP ? wl_n : RING_WL_CAP;
        constexpr int PER_LANE = (RING_WL_CAP + 63) / 64;
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
        // of this cold path to the head of the tile loop as possibly pending and drains the whole LDS-DMA ring there with a
        // vmcnt(0) of its own, once per tile
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        wl_n = 0;
    };
    auto wl_append = [&](bool p, key_t64 key, int q) {
        const unsigned long long bal = __ballot(p);
        if (bal == 0ull) return;
        const int pos = wl_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (p) {
            if (pos < RING_WL_CAP) {
                wl_key[pos] = key;
                wl_q[pos] = q;
            } else {
                emit_candidate<SUBSET>(key, q, thr_key, cand, cnt, cap, overflow, ex);
            }
        }
        if (wl_n + __builtin_popcountll(bal) > RING_WL_CAP) __builtin_amdgcn_s_waitcnt(0x0F70);  // (see wl_flush)
        wl_n += __builtin_popcountll(bal);
    };

    struct Frags {
        u32x4 b[NB16], a[MB];
    };
    f32x4 acc[MB][NB16];

    // fragment reads of one k-step: corpus slice in ring slot `cs` (k-step ks of it), query half slice in ring slot `qs`
    auto read_b = [&](int qs, Frags& f) {
        const char* qb = smem + qs * RING_Q_SLOT + b_row_off;
#pragma unroll
        for (int j = 0; j < NB16; ++j) f.b[j] = *(const u32x4*)(qb + j * 16 * 64);
    };
    auto read_a = [&](int cs, int ks, Frags& f, int i0, int i1) {
        const char* cb = smem + cs * RING_C_SLOT + a_row_off + (((4 * ks + fq) ^ swz) << 4);
#pragma unroll
        for (int i = 0; i < MB; ++i)
            if (i >= i0 && i < i1) f.a[i] = *(const u32x4*)(cb + i * 16 * 128);
    };
    auto mma = [&](const Frags& f, int i0, int i1, bool zero_c) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB16; ++j)
                if (i >= i0 && i < i1) acc[i][j] = mfma16<DT>(f.a[i], f.b[j], zero_c ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j]);
    };

    // ---- epilogue of one tile (the FILTER epilogue of kernels_mips.hip) --------------------------------
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
                // cold path; opaque copies keep everything derived from the tile's row base inside it
                int x0_o = x0, row_end_o = row_end, q_o = q;
                asm volatile("" : "+s"(x0_o), "+s"(row_end_o), "+v"(q_o));  // (q: else 12 hoisted 64-bit addresses of thr_key / cnt / cand per lane)
                auto val = [&](int v) { return acc[v >> 2][j][v & 3]; };
                unsigned mask = 0;
                if (hit) {
#pragma unroll
                    for (int v = 0; v < MB * 4; ++v) mask |= (val(v) >= th) ? (1u << v) : 0u;
                }
                const bool multi = __any((mask & (mask - 1u)) != 0u);
                do {
                    const bool p = mask != 0u;
                    const int b = p ? __builtin_ctz(mask) : 0;
                    mask &= mask - 1u;
                    float sc = m;
                    if (multi) {
                        // register select by the bits of b (inline asm: as C++ selects LLVM turns the tree into an indexed
                        // load from a SCRATCH copy of the accumulator, stored after every MFMA of the hot loop)
                        const unsigned long long s0 = __ballot(b & 1), s1 = __ballot(b & 2), s2 = __ballot(b & 4),
                                                 s3 = __ballot(b & 8), s4 = __ballot(b & 16);
                        auto sel = [](float lo, float hi, unsigned long long sm) {
                            float r;
                            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(hi), "s"(sm));
                            return r;
                        };
                        float t16[16], t8[8], t4[4], t2[2];
#pragma unroll
                        for (int u = 0; u < 16; ++u) t16[u] = sel(val(2 * u), val(2 * u + 1), s0);
#pragma unroll
                        for (int u = 0; u < 8; ++u) t8[u] = sel(t16[2 * u], t16[2 * u + 1], s1);
#pragma unroll
                        for (int u = 0; u < 4; ++u) t4[u] = sel(t8[2 * u], t8[2 * u + 1], s2);
#pragma unroll
                        for (int u = 0; u < 2; ++u) t2[u] = sel(t4[2 * u], t4[2 * u + 1], s3);
                        sc = sel(t2[0], t2[1], s4);
                    }
                    const int rw = x0_o + wm * TM + 4 * fq + (b >> 2) * 16 + (b & 3);
                    wl_append(p && rw < row_end_o, make_key(sc, (unsigned)rw), q_o);
                } while (__any(mask != 0u));
            }
        }
        if (wl_n >= RING_WL_FLUSH) wl_flush();
    };

    // ---- the k-step stream ------------------------------------------------------------------------------
    // Issue order of the LDS-DMA pieces (per wave): every k-step issues the 2 query pieces of a later half slice, then 2
    // corpus pieces; the prologue issues what the first k-steps would have found in flight, in that same order, so that at
    // every sync the 6 youngest operations are exactly the pieces of the two most recent k-steps.
    //   plain: k-step h = (slice g, ks) issues query half h+2 and corpus pieces (2ks, 2ks+1) of slice g+2; it reads its own
    //          fragments right after the sync.
    //   PIPE:  k-step h issues query half h+3 and corpus pieces of slice g+3 (pieces 0, 1 in a ks = 1 step, 2, 3 in the
    //          following ks = 0 step); it reads the fragments of k-step h+1, which the sync therefore has to cover.
    int rc = 0, rq = 0;  // ring slots of the fragments read next: corpus slice / query half slice
    auto sync = [&]() {
        wait_vmcnt<6>();
        if constexpr (PIPE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave holds k-step h's fragments: their slots may be refilled
        __builtin_amdgcn_s_barrier();
    };
    Frags f0, f1;
    if constexpr (!PIPE) {
        dma_c(0); dma_c(1); dma_c(2); dma_c(3); next_c();  // slice 0
        dma_q();                                           // half 0
        dma_c(0); dma_c(1);                                // slice 1, first pair   (as k-step -2 ... )
        dma_q();                                           // half 1                (... and k-step -1 would have)
        dma_c(2); dma_c(3); next_c();                      // slice 1, second pair
    } else {
        dma_c(0); dma_c(1); dma_c(2); dma_c(3); next_c();  // slice 0
        dma_q();                                           // half 0
        dma_c(0); dma_c(1);                                // slice 1, first pair
        dma_q();                                           // half 1
        dma_c(2); dma_c(3); next_c();                      // slice 1, second pair
        dma_q();                                           // half 2
        dma_c(0); dma_c(1);                                // slice 2, first pair (its second pair is k-step 0's)
        if constexpr (RING_PIPE_ACROSS_TILES) {
            wait_vmcnt<10>();                              // slice 0 and half 0 have landed
            __builtin_amdgcn_s_barrier();
            read_b(0, f0);
            read_a(0, 0, f0, 0, MB);
            rq = 1;
        }
    }

    // one k-step.  KS: which half of the corpus slice; FIRST: first k-step of a tile (multiplies into a constant-0 C);
    // last: last k-step of a tile.  PIPE does not read ahead across the tile boundary (the epilogue then runs with one
    // fragment set less in registers - no spills, and no reloads the waitcnt pass would guard with a vmcnt(0) at the head of
    // the tile loop): the first k-step of a tile reads its own fragments, behind a barrier of its own because the slots it
    // reads are the ones this k-step's LDS-DMAs refill.
    auto kstep = [&](auto ks_tag, auto first_tag, bool last, Frags& cur, Frags& nxt) {
        constexpr int KS = decltype(ks_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        sync();
        if constexpr (PIPE && FIRST && !RING_PIPE_ACROSS_TILES) {
            read_b(rq, cur);
            read_a(rc, 0, cur, 0, MB);
            rq = rq == 2 ? 0 : rq + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (!PIPE) {
            read_b(rq, cur);
            read_a(rc, KS, cur, 0, 4);
            __builtin_amdgcn_sched_barrier(0);
            mma(cur, 0, 1, FIRST);
            __builtin_amdgcn_sched_barrier(0);
            read_a(rc, KS, cur, 4, 8);
            dma_q();
            __builtin_amdgcn_sched_barrier(0);
            mma(cur, 1, 4, FIRST);
            dma_c(2 * KS);
            mma(cur, 4, 6, FIRST);
            dma_c(2 * KS + 1);
            mma(cur, 6, 8, FIRST);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KS == 1) {
                next_c();
                rc = rc == 2 ? 0 : rc + 1;
            }
            rq = rq == 2 ? 0 : rq + 1;
        } else {
            // fragments of the NEXT k-step: (same slice, ks 1) after a ks = 0 step, (next slice, ks 0) after a ks = 1 step
            const int rc_n = KS == 0 ? rc : (rc == 2 ? 0 : rc + 1);
            mma(cur, 0, 1, FIRST);
            __builtin_amdgcn_sched_barrier(0);
            if (!last || RING_PIPE_ACROSS_TILES) {
                read_b(rq, nxt);
                read_a(rc_n, KS ^ 1, nxt, 0, MB);
            }
            __builtin_amdgcn_sched_barrier(0);
            dma_q();
            mma(cur, 1, 4, FIRST);
            dma_c(KS == 1 ? 0 : 2);
            mma(cur, 4, 6, FIRST);
            dma_c(KS == 1 ? 1 : 3);
            mma(cur, 6, 8, FIRST);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KS == 0) next_c();
            rc = rc_n;
            if (!last || RING_PIPE_ACROSS_TILES) rq = rq == 2 ? 0 : rq + 1;
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    for (int it = 0; it < n_my; ++it) {
        kstep(K0{}, std::true_type{}, false, f0, f1);
        kstep(K1{}, std::false_type{}, nk == 1, f1, f0);
        for (int t = 1; t < nk; ++t) {
            kstep(K0{}, std::false_type{}, false, f0, f1);
            kstep(K1{}, std::false_type{}, t == nk - 1, f1, f0);
        }
        epilogue(row_begin + (xt0 + it * xt_step) * BM);
    }
    wait_vmcnt<0>();  // the run-ahead pieces of the last k-steps
    wl_flush();
}

// ---- launcher -----------------------------------------------------------------------------------------
namespace {
template <int DT, bool SUBSET, bool PIPE>
hipError_t launch_ring(const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end, int64_t nq,
                       int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    const int n_qtiles = (int)(nq_pad / 256);
    const int n_xtiles = (int)((row_end - row_begin + 255) / 256);
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const int unit = 8 * n_qtiles;
    const int total = ((n_xtiles + 7) / 8) * unit;
    int grid = (n_cu / unit) * unit;
    if (grid < unit) grid = unit;
    if (grid > total) grid = total;
    auto kern = mips_filter16r_kernel<DT, SUBSET, PIPE>;
    if (hipError_t e = allow_dynamic_lds((const void*)kern, RING_LDS_BYTES); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), RING_LDS_BYTES, stream, (const uint16_t*)store, (const uint16_t*)q_pad,
                       (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key, ws.cand,
                       ws.cnt, (int)ws.cap, ws.overflow, ws.extra);
    return hipGetLastError();
}
template <int DT>
hipError_t launch_ring_dt(bool pipe, bool subset, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin,
                          int64_t row_end, int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    if (pipe)
        return subset ? launch_ring<DT, true, true>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream)
                      : launch_ring<DT, false, true>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    return subset ? launch_ring<DT, true, false>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream)
                  : launch_ring<DT, false, false>(store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
}
}  // namespace

hipError_t launch_filter_ring(int store_dtype, bool pipe, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin,
                              int64_t row_end, int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    const bool subset = ws.extra.row_label != nullptr;
    return store_dtype == 0 ? launch_ring_dt<0>(pipe, subset, store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream)
                            : launch_ring_dt<1>(pipe, subset, store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
}

}  // namespace vodhip

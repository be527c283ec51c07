// libvodhip -- the FILTER stage on the guide's "256^2 8-phase" K loop (round 5).  Tile 14 = the production kernel of batches with two or
// more query tiles (C3: -2.4 %, C4 shard: -4.0 % against tile 8, profiles/r05_ab_8phase.txt); tile 13 = the guide's read schedule
// (B0 fragments re-read in phase 4), experiment builds only.
//
// Same tile, wave layout (8 waves = 2 x 4 of 128 x 64), fragment maps (v_mfma_f32_16x16x32), epilogue, survivor lists and stage
// order as the persistent kernel of kernels_mips.hip; the K loop is rebuilt after /opt/skills/guides/cdna_hip_programming.md
// ("The 256^2 8-phase template", lines 376-437):
//   * LDS operands: 8 half-tile buffers of 16 KB (2 K-tiles x [A0 | B1 | A1 | B0]; a half-tile = 128 rows x 64 k) = 128 KB.
//     A-half h = the rows {wm*128 + h*64 .. +63} of both wave rows, B-half h = the queries {wn*64 + h*32 .. +31} of all four wave
//     columns, so that a phase's register sub-tile is one quadrant (64 rows x 32 queries) of EVERY wave's 128 x 64 output;
//   * one K-tile (64 deep) = 4 phases, one C quadrant x K = 64 = 16 MFMAs per wave each:
//       qd 0: (m0, n0)  ds_read A0 (8) + B0 (4)      qd 1: (m0, n1)  B1 (4)      qd 2: (m1, n1)  A1 (8)      qd 3: (m1, n0)  B0 (4) again
//     every phase = [ds_read the sub-tile | 2 x global_load_lds of ONE half-tile, 7 half-tiles ahead | lgkmcnt(0)] s_barrier
//     [s_setprio 1 | 16 MFMA | s_setprio 0] s_barrier;
//   * the two wave rows run staggered by one barrier (wm = 1 takes an extra s_barrier up front): on every SIMD the read section of
//     one wave runs beside the MFMA section of the other;
//   * `s_waitcnt vmcnt(6)` once per K-tile, in phase qd 3 (never 0 in the steady state): with a lead of 7 half-tiles everything but
//     the three youngest half-tiles has landed, i.e. the whole NEXT K-tile; it is read from the next phase on (one barrier later for
//     the wave row that runs ahead, two for the other);
//   * WAR: the half-tile staged in phase g overwrites the buffer last read in phase g - 1; those reads are retired (lgkmcnt(0)) BEFORE
//     that phase's first barrier, and the stage is issued after the phase's second barrier of the wave row that runs ahead - i.e. after
//     the first barrier of the other row, which has retired its reads as well.
// The half-tile stream continues across corpus tiles (the first half-tiles of the next tile are in flight during a tile's last
// K-tile and its epilogue).  A flush of the survivor lists (global loads / atomics / stores, rare) drains vmcnt afterwards, so the
// counted waits only ever count LDS-DMA pieces.  Results are bit-identical to tile 8: same products, same summation order.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mips_common.h"

namespace vodhip {

namespace {
constexpr int P8_HALF = 128 * 128;         // one half-tile: 128 rows x 128 B
constexpr int P8_OPERANDS = 8 * P8_HALF;   // 128 KB
constexpr int P8_WL_CAP = 256;             // records per wave list
constexpr int P8_WL_FLUSH = 176;
#ifndef P8_TRANSPOSE_EMIT
#define P8_TRANSPOSE_EMIT 1  // 0: the survivor path of rounds 1-6a (per-lane hit mask + select tree), kept for A/B builds
#endif
#ifndef P8_TRANSPOSE_MAX_LANES
#define P8_TRANSPOSE_MAX_LANES 2  // query blocks with at most this many lanes holding survivors take the transpose path
#endif
constexpr int P8_XPOSE = 8 * 2 * 32 * 4;   // survivor transpose: per wave 2 slots of a lane's 32 sums of one query (2 KB)
constexpr int P8_LDS = P8_OPERANDS + 8 * P8_WL_CAP * 12 + P8_XPOSE;
static_assert(P8_LDS <= 160 * 1024, "LDS budget");
}  // namespace

// KEEPB0 (tile 14): the B0 fragments stay in registers from phase qd 0 to qd 3 (16 more VGPRs, 4 fewer ds_read_b128 per K-tile)
// LEAD = half-tiles the LDS-DMA stream runs ahead of the phase that issues it (7 = the guide's; 6 / 5: experiment builds, tiles 15 / 16)
template <int DT, bool SUBSET, bool KEEPB0, int LEAD = 7>
__global__ __launch_bounds__(512, 2) void mips_filter8ph_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end, int n_xtiles,
    int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key, key_t64* __restrict__ cand,
    unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    constexpr int BM = 256, BN = 256, WN = 4, NWAVES = 8, TM = 128, TN = 64, MB = TM / 16, NB16 = TN / 16, ROW_BYTES = 128;
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
    const int nk = dim_pad / 64;  // K-tiles per corpus tile
    const size_t row_stride = (size_t)dim_pad * 2;

    // ---- LDS-DMA sources.  Wave w stages LDS rows [16 w, 16 w + 16) of every half-tile: two wave-instructions of 8 rows x 128 B.
    // LDS row lr of A-half h is tile row (lr >> 6) * 128 + h * 64 + (lr & 63); of B-half h query (lr >> 5) * 64 + h * 32 + (lr & 31).
    // 16-byte chunk c of LDS row lr sits at slot c ^ ((lr >> 1) & 7) (applied on the source address; fragment reads conflict-free).
    const int st_row = lane >> 3, st_slot = lane & 7;
    int super_cur = ex.perm_mod > 0 ? filter_tile_row0(ex, row_begin, xt0, BM) / BM : 0;
    int super_epi = super_cur;
    const int perm_inc = ex.perm_mod > 0 ? (int)(((unsigned long long)xt_step * (unsigned long long)ex.perm_mul) % (unsigned long long)ex.perm_mod) : 0;
    const char* a_src[2];
    const char* b_src[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sw = (u * 4 + (st_row >> 1)) & 7;
        const int arow = (wave >> 2) * 128 + (wave & 3) * 16 + u * 8 + st_row;  // + h * 64
        const int brow = (wave >> 1) * 64 + (wave & 1) * 16 + u * 8 + st_row;   // + h * 32
        a_src[u] = (const char*)X + ((size_t)filter_tile_row0(ex, row_begin, xt0, BM) + arow) * row_stride + ((st_slot ^ sw) << 4);
#ifdef VODHIP_ABLATION  // diagnostic builds, kflags bit 128: every q-tile stages the rows of q-tile 0 (an XCD's query working set shrinks to 256 rows; bytes / timing only)
        const int q0_stage = (ex.flags & (128 << 8)) ? 0 : q0;
#else
        const int q0_stage = q0;
#endif
        b_src[u] = (const char*)Q + (size_t)(q0_stage + brow) * row_stride + ((st_slot ^ sw) << 4);
    }
    const size_t a_half_step = 64 * row_stride, b_half_step = 32 * row_stride;
    const size_t tile_step_bytes = (size_t)xt_step * BM * row_stride;

    // the issue stream: half-tile n = 4 * (stream K-tile) + kind, kinds in staging order [A0, B1, A1, B0]; LDS buffer = n & 7.
    // The kind is a COMPILE-TIME argument (phase qd stages kind (qd + 3) & 3): the four source pointers stay in registers (indexed
    // by a run-time kind the compiler keeps them in scratch and puts a vmcnt(0) in front of every LDS-DMA).
    const bool corpus_nt = (ex.flags & FILTER_FLAG_CORPUS_NT) != 0;
    int s_t = 0, s_it = 0, s_par = 0;  // stream K-tile: index inside its corpus tile, corpus tile, buffer parity
    int issued = 0;                     // half-tiles staged so far
    bool s_active = n_my > 0;
    auto stage = [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
        if (s_active) {
            char* dst = smem + (s_par * 4 + KIND) * P8_HALF + wave * 16 * ROW_BYTES;
            const int kbyte = s_t * ROW_BYTES;
            if constexpr (KIND == 0 || KIND == 2) {
                const size_t off = (KIND == 2 ? a_half_step : 0) + kbyte;
                if (corpus_nt) {  // ONE query tile: every corpus line is read once, by one workgroup (as in the two-slot kernel)
                    glds16_aux<2>(a_src[0] + off, dst);
                    glds16_aux<2>(a_src[1] + off, dst + 8 * ROW_BYTES);
                } else {
                    glds16(a_src[0] + off, dst);
                    glds16(a_src[1] + off, dst + 8 * ROW_BYTES);
                }
            } else if constexpr (KIND == 1) {
                glds16(b_src[0] + b_half_step + kbyte, dst);
                glds16(b_src[1] + b_half_step + kbyte, dst + 8 * ROW_BYTES);
            } else {
                glds16(b_src[0] + kbyte, dst);
                glds16(b_src[1] + kbyte, dst + 8 * ROW_BYTES);
            }
            ++issued;
            if constexpr (KIND == 3) {  // the stream's K-tile is complete
                s_par ^= 1;
                if (++s_t == nk) {  // ... and moves on to this workgroup's next corpus tile
                    s_t = 0;
                    ++s_it;
                    if (s_it < n_my) {
                        long long step = (long long)tile_step_bytes;
                        if (ex.perm_mod > 0) {
                            int nxt = super_cur + perm_inc;
                            if (nxt >= ex.perm_mod) nxt -= ex.perm_mod;
                            step = (long long)(nxt - super_cur) * (long long)BM * (long long)row_stride;
                            super_cur = nxt;
                        }
                        a_src[0] += step;
                        a_src[1] += step;
                    } else {
                        s_active = false;
                    }
                }
            }
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;

    // thresholds first, and retired: a later compiler-inserted vmcnt wait for them would also drain the LDS-DMAs in flight
    float thr[NB16];
#pragma unroll
    for (int j = 0; j < NB16; ++j) {
        const int q = q0 + wn * TN + j * 16 + fr;
        thr[j] = q < nq ? thr_s[q] : __builtin_inff();
    }
#pragma unroll
    for (int j = 0; j < NB16; ++j) asm volatile("" : "+v"(thr[j]));
    wait_vmcnt<0>();

    // ---- per-wave survivor list (as in the persistent kernel) -------------------------------------
    key_t64* const wl_key = (key_t64*)(smem + P8_OPERANDS) + wave * P8_WL_CAP;
    int* const wl_q = (int*)(smem + P8_OPERANDS + NWAVES * P8_WL_CAP * 8) + wave * P8_WL_CAP;
    int wl_n = 0;  // wave-uniform
    [[maybe_unused]] float* const xpose = (float*)(smem + P8_OPERANDS + NWAVES * P8_WL_CAP * 12) + wave * 64;
    auto wl_flush = [&]() {
        const int n = wl_n < P8_WL_CAP ? wl_n : P8_WL_CAP;
        constexpr int PER_LANE = P8_WL_CAP / 64;
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
        wl_n = 0;
        wait_vmcnt<0>();  // the counted waits of the K loop must only ever see LDS-DMA pieces (loads and stores retire independently)
    };
    auto wl_append = [&](bool p, key_t64 key, int q) {
        const unsigned long long bal = __ballot(p);
        if (bal == 0ull) return;
        const int pos = wl_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        bool direct = false;
        if (p) {
            if (pos < P8_WL_CAP) {
                wl_key[pos] = key;
                wl_q[pos] = q;
            } else {
                emit_candidate<SUBSET>(key, q, thr_key, cand, cnt, cap, overflow, ex);
                direct = true;
            }
        }
        wl_n += __builtin_popcountll(bal);
        if (__any(direct)) wait_vmcnt<0>();
    };

    f32x4 acc[MB][NB16];
    u32x4 fa[4], fb[2], fa1[4], fb1[2];  // the quadrant's fragments: k-step 0 (fa, fb) and k-step 1 (fa1, fb1)
    u32x4 fc[2], fc1[2];                 // KEEPB0: the B1 fragments get registers of their own
    // timing knobs.  Compile-time (-DP8_KNOBS: read from "kflags"): as run-time branches around s_setprio / the stagger barrier they
    // changed the compiler's schedule of the whole loop - the SAME kernel went from -1.2 % to +2.3 % against tile 8 on C3
    // (profiles/r05_ab_8phase.txt).  Measured with the knobs: no stagger +10 %, no s_setprio +2 %.
#ifdef P8_KNOBS
    const bool knob_no_stagger = (ex.flags & (1 << 8)) != 0, knob_no_prio = (ex.flags & (2 << 8)) != 0;
#else
    constexpr bool knob_no_stagger = false, knob_no_prio = false;
#endif

    // fragment reads of one half-tile buffer: A quadrant rows wm*64 + i'*16 + fr, B quadrant rows wn*32 + j'*16 + fr
    auto read_a = [&](const char* half) {
        const char* base = half + (wm * 64 + fr) * ROW_BYTES;
        const int s0 = ((0 + fq) ^ swz) << 4, s1 = ((4 + fq) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = *(const u32x4*)(base + i * 16 * ROW_BYTES + s0);
            fa1[i] = *(const u32x4*)(base + i * 16 * ROW_BYTES + s1);
        }
    };
    auto read_b = [&](const char* half) {
        const char* base = half + (wn * 32 + fr) * ROW_BYTES;
        const int s0 = ((0 + fq) ^ swz) << 4, s1 = ((4 + fq) ^ swz) << 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j] = *(const u32x4*)(base + j * 16 * ROW_BYTES + s0);
            fb1[j] = *(const u32x4*)(base + j * 16 * ROW_BYTES + s1);
        }
    };
    auto read_b1 = [&](const char* half) {  // B1 into fc / fc1 (KEEPB0) or over B0 (fb / fb1)
        const char* base = half + (wn * 32 + fr) * ROW_BYTES;
        const int s0 = ((0 + fq) ^ swz) << 4, s1 = ((4 + fq) ^ swz) << 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (KEEPB0) {
                fc[j] = *(const u32x4*)(base + j * 16 * ROW_BYTES + s0);
                fc1[j] = *(const u32x4*)(base + j * 16 * ROW_BYTES + s1);
            } else {
                fb[j] = *(const u32x4*)(base + j * 16 * ROW_BYTES + s0);
                fb1[j] = *(const u32x4*)(base + j * 16 * ROW_BYTES + s1);
            }
        }
    };
    auto mma_quadrant = [&](int mh, int nh, bool zero_c) {
        if (!knob_no_prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int mhh = 0; mhh < 2; ++mhh)
#pragma unroll
                    for (int nhh = 0; nhh < 2; ++nhh)
                        if (mhh == mh && nhh == nh) {
                            f32x4& c = acc[mhh * 4 + i][nhh * 2 + j];
                            c = mfma16<DT>(fa[i], (KEEPB0 && nhh == 1) ? fc[j] : fb[j], zero_c ? f32x4{0.f, 0.f, 0.f, 0.f} : c);
                        }
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int mhh = 0; mhh < 2; ++mhh)
#pragma unroll
                    for (int nhh = 0; nhh < 2; ++nhh)
                        if (mhh == mh && nhh == nh) {
                            f32x4& c = acc[mhh * 4 + i][nhh * 2 + j];
                            c = mfma16<DT>(fa1[i], (KEEPB0 && nhh == 1) ? fc1[j] : fb1[j], c);
                        }
            }
        if (!knob_no_prio) __builtin_amdgcn_s_setprio(0);
    };

    // ---- epilogue of one tile (the persistent kernel's, FILTER mode) ------------------------------
    // What it costs (round 6: compile-time part removal, experiments/tools/build_p8_parts.sh with -DP8_ABL_NO_EPILOGUE / -DP8_ABL_NO_EMIT;
    // profiles/r06_ab_epilogue.txt): on C3 the block maxima + compares 0.24 ms of a 13.2 ms batch, the SURVIVOR path below 0.85-1.28 ms
    // (11 % of C2, 12 % of a 1.25 M-row shard).  It is not a cold path: a stage lets ~k * growth rows per query pass, 1.4-2.4 per wave
    // tile, so ~45 % of the (wave, 16-query block) pairs take it, and the 8 waves meet at the next barrier - every tile pays its slowest
    // wave.  Hence the planner's small stage growth (vodhip_api.hip); moving the maxima into the K loop's phases and narrowing the hit mask
    // by fragment were built and measured: slower / equal (same profile).
    auto epilogue = [&](int x0) {
#ifdef P8_ABL_NO_EPILOGUE  // timing only (experiments/tools/build_p8_parts.sh): the accumulators stay live, nothing is filtered
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB16; ++j) asm volatile("" ::"v"(acc[i][j]));
        (void)x0;
        return;
#else
        // the block maxima of this lane, all four query blocks first: four independent v_max3 chains (16 deep each) the scheduler can
        // interleave - behind the branches of the survivor path they would run one after the other.  (v_max3, not fmaxf: hipcc quiets
        // every operand of fmaxf first, two more v_max per value; a quiet NaN operand of v_max3 is ignored, which is what the per-value
        // compares below do with it too.)
        float mj[NB16];
#pragma unroll
        for (int j = 0; j < NB16; ++j) {
            mj[j] = vmax3(acc[0][j][0], acc[0][j][1], acc[0][j][2]);
            mj[j] = fmaxf_raw(mj[j], acc[0][j][3]);
        }
#pragma unroll
        for (int i = 1; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB16; ++j) {
                mj[j] = vmax3(mj[j], acc[i][j][0], acc[i][j][1]);
                mj[j] = vmax3(mj[j], acc[i][j][2], acc[i][j][3]);
            }
#ifdef P8_ABL_NO_EMIT  // timing only: the tests run, their survivors are dropped
#pragma unroll
        for (int j = 0; j < NB16; ++j) {
            const unsigned long long any_hit = __ballot(mj[j] >= thr[j]);
            asm volatile("" ::"s"(any_hit));
        }
        (void)x0;
        return;
#else
#if P8_TRANSPOSE_EMIT
        // ... and the four tests: most tiles of a search's late stages hold no survivor for any of the wave's 64 queries - one branch, not four
        unsigned long long hl[NB16];
#pragma unroll
        for (int j = 0; j < NB16; ++j) hl[j] = __ballot(mj[j] >= thr[j]);  // (false for NaN and for padded queries: thr = +inf)
        if ((hl[0] | hl[1] | hl[2] | hl[3]) == 0ull) {
            if (wl_n >= P8_WL_FLUSH) wl_flush();
            return;
        }
        static_assert(NB16 == 4, "the early-out above");
#endif
#pragma unroll
        for (int j = 0; j < NB16; ++j) {
            const float m = mj[j];
#if !P8_TRANSPOSE_EMIT
            const int q = q0 + wn * TN + j * 16 + fr;
#endif
            const bool hit = m >= thr[j];  // false for NaN and for padded queries (thr = +inf)
#if P8_TRANSPOSE_EMIT
            // The survivor path (7-10 % of a C3 batch, profiles/r06_ab_epilogue.txt), two forms:
            //  * TRANSPOSE: a lane that holds a survivor writes its 32 sums of this query to LDS (8 ds_write_b128) and lane t of a 32-lane half
            //    reads sum t back: one compare, one key and one append per half handle all 32 in parallel, with row offsets that depend on the
            //    lane number alone - no mask, no select, no per-survivor loop.  Two source lanes per pass; the source lane's threshold and
            //    coordinates travel as scalars (v_readlane, s_ff1).  Cheapest when few lanes hold survivors (the late, large stages of a
            //    search: one survivor in one lane is the common case) and when a lane holds several (rows sorted by topic: -4.7 %).
            //  * MASK (rounds 1-6a): every lane compares its 32 sums into a hit mask and appends its own survivors, all lanes at once; the
            //    score of a single survivor is the lane's maximum.  Cheapest when many lanes hold one survivor each (the early, small stages;
            //    a 1.25 M-row shard is mostly those: all-transpose was +5.8 % there).
            // Same records either way, in a different order (the select kernel's result does not depend on it).
            const unsigned long long hit_lanes = hl[j];
            if (hit_lanes != 0ull) {
                // (opaque: or hipcc computes the lane-dependent offsets of all four query blocks once, outside the K loop, and holds them in
                // vector registers the loop does not have - it spilled 18)
                int x0_o = x0, row_end_o = row_end, lane_o = lane;
                asm volatile("" : "+s"(x0_o), "+s"(row_end_o));
                asm volatile("" : "+v"(lane_o));
                // (`src_lanes`: the lanes whose sums travel; `mine`: this lane is one of them)
                auto transpose_emit = [&](unsigned long long src_lanes, bool mine) __attribute__((always_inline)) {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(src_lanes >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)src_lanes, 0u));
                    const int half = lane_o >> 5, t = lane_o & 31;
                    unsigned long long rest = src_lanes;
                    int done = 0;  // source lanes handled by earlier passes
                    do {
                        const int slot = rank - done;
                        if (mine && (unsigned)slot < 2u) {
                            float* dst = xpose + slot * 32;
#pragma unroll
                            for (int i = 0; i < MB; ++i) *(f32x4*)(dst + 4 * i) = acc[i][j];
                        }
                        const int l0 = __builtin_ctzll(rest);
                        rest &= rest - 1ull;
                        const bool two = rest != 0ull;
                        const int l1 = two ? __builtin_ctzll(rest) : l0;
                        rest &= rest - 1ull;  // (0 stays 0)
                        const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thr[j]), l0));
                        const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thr[j]), l1));
                        const float sc = xpose[lane_o];  // (LDS operations of one wave execute in order: the stores above have landed)
                        const int src = half ? l1 : l0;
                        const float thr_src = half ? t1 : t0;
                        const int q_src = q0 + wn * TN + j * 16 + (src & 15);
                        const int rw = x0_o + wm * TM + 4 * (src >> 4) + (t >> 2) * 16 + (t & 3);
                        wl_append((half == 0 || two) && sc >= thr_src && rw < row_end_o, make_key(sc, (unsigned)rw), q_src);
                        done += 2;
                    } while (rest != 0ull);
                };
                if (__builtin_popcountll(hit_lanes) <= P8_TRANSPOSE_MAX_LANES) {
                    transpose_emit(hit_lanes, hit);
                } else {
                    // many lanes: every lane compares its 32 sums into a hit mask; a lane with ONE survivor appends it from its maximum, all
                    // such lanes at once; the lanes with several (few) send theirs through the transpose
                    unsigned mask = 0;
                    if (hit) {
#pragma unroll
                        for (int v = 0; v < MB * 4; ++v) mask |= (acc[v >> 2][j][v & 3] >= thr[j]) ? (1u << v) : 0u;
                    }
                    const bool several = (mask & (mask - 1u)) != 0u;
                    const int q = q0 + wn * TN + j * 16 + (lane_o & 15);
                    const int b = hit ? __builtin_ctz(mask) : 0;
                    const int rw = x0_o + wm * TM + 4 * (lane_o >> 4) + (b >> 2) * 16 + (b & 3);
                    wl_append(hit && !several && rw < row_end_o, make_key(m, (unsigned)rw), q);
                    const unsigned long long several_lanes = __ballot(several);
                    if (several_lanes != 0ull) transpose_emit(several_lanes, several);
                }
            }
        }
#else
            if (__any(hit)) {
                int x0_o = x0, row_end_o = row_end;
                asm volatile("" : "+s"(x0_o), "+s"(row_end_o));
                auto val = [&](int v) { return acc[v >> 2][j][v & 3]; };
                unsigned mask = 0;
                if (hit) {
#pragma unroll
                    for (int v = 0; v < MB * 4; ++v) mask |= (val(v) >= thr[j]) ? (1u << v) : 0u;
                }
                const bool multi = __any((mask & (mask - 1u)) != 0u);
                do {
                    const bool p = mask != 0u;
                    const int b = p ? __builtin_ctz(mask) : 0;
                    mask &= mask - 1u;
                    float sc = m;
                    if (multi) {
                        const unsigned long long s0 = __ballot(b & 1), s1 = __ballot(b & 2), s2 = __ballot(b & 4), s3 = __ballot(b & 8),
                                                 s4 = __ballot(b & 16);
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
                    wl_append(p && rw < row_end_o, make_key(sc, (unsigned)rw), q);
                } while (__any(mask != 0u));
            }
        }
#endif
        if (wl_n >= P8_WL_FLUSH) wl_flush();
#endif  // P8_ABL_NO_EMIT
#endif  // P8_ABL_NO_EPILOGUE
    };

    // ---- prologue: 7 half-tiles in flight, the first K-tile landed and visible -----------------------
    stage(K0{}); stage(K1{}); stage(K2{}); stage(K3{});
    stage(K0{});
    if constexpr (LEAD >= 6) stage(K1{});
    if constexpr (LEAD >= 7) stage(K2{});
    using S0 = std::integral_constant<int, (0 + LEAD) & 3>;  // the kind phase qd stages: (qd + LEAD) & 3
    using S1 = std::integral_constant<int, (1 + LEAD) & 3>;
    using S2 = std::integral_constant<int, (2 + LEAD) & 3>;
    using S3 = std::integral_constant<int, (3 + LEAD) & 3>;
    auto wait_next_ktile = [&](int done_ktiles) {
        // everything but the half-tiles staged beyond K-tile `done_ktiles` (0-based: that one must be complete) may stay in flight
        const int ahead = issued - 4 * (done_ktiles + 1);
        if (ahead >= 3) wait_vmcnt<6>();
        else if (ahead == 2) wait_vmcnt<4>();
        else if (ahead == 1) wait_vmcnt<2>();
        else wait_vmcnt<0>();
    };
    wait_next_ktile(0);
    __builtin_amdgcn_s_barrier();
    if (wm == 1 && !knob_no_stagger) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind

    int kt = 0;  // global K-tile counter of this workgroup: operand buffers (kt & 1) * 4 ..
    for (int it = 0; it < n_my; ++it) {
        const int x0 = ex.perm_mod > 0 ? super_epi * BM : row_begin + (xt0 + it * xt_step) * BM;
        for (int t = 0; t < nk; ++t, ++kt) {
            const char* kb = smem + (kt & 1) * 4 * P8_HALF;  // [A0 | B1 | A1 | B0] of this K-tile
            const bool first = t == 0;
            // qd 0: (m0, n0)
            read_a(kb + 0 * P8_HALF);
            read_b(kb + 3 * P8_HALF);
            stage(S0{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (first) mma_quadrant(0, 0, true); else mma_quadrant(0, 0, false);
            __builtin_amdgcn_s_barrier();
            // qd 1: (m0, n1)
            read_b1(kb + 1 * P8_HALF);
            stage(S1{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (first) mma_quadrant(0, 1, true); else mma_quadrant(0, 1, false);
            __builtin_amdgcn_s_barrier();
            // qd 2: (m1, n1)
            read_a(kb + 2 * P8_HALF);
            stage(S2{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (first) mma_quadrant(1, 1, true); else mma_quadrant(1, 1, false);
            __builtin_amdgcn_s_barrier();
            // qd 3: (m1, n0); the next K-tile's half-tiles must have landed before this phase's first barrier
            if constexpr (!KEEPB0) read_b(kb + 3 * P8_HALF);
            stage(S3{});
            wait_next_ktile(kt + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (first) mma_quadrant(1, 0, true); else mma_quadrant(1, 0, false);
            __builtin_amdgcn_s_barrier();
        }
        epilogue(x0);
        // the super-tile of the next corpus tile: the issue stream reached it (at the latest) two K-tiles ago, track it separately
        if (ex.perm_mod > 0) {
            int nxt = super_epi + perm_inc;
            if (nxt >= ex.perm_mod) nxt -= ex.perm_mod;
            super_epi = nxt;
        }
    }
    if (wm == 0 && !knob_no_stagger) __builtin_amdgcn_s_barrier();  // balances the stagger
    wl_flush();
}

template <int DT, bool KEEPB0, int LEAD = 7>
static hipError_t launch_8phase_dt(const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end, int n_xtiles,
                                   int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    const bool subset = ws.extra.row_label != nullptr;
    const int n_qtiles = (int)(nq_pad / 256);
    const int n_cu = ws.n_cu > 0 ? ws.n_cu : 256;
    const int unit = 8 * n_qtiles;
    const int total = ((n_xtiles + 7) / 8) * unit;
    int grid = (n_cu / unit) * unit;
    if (grid < unit) grid = unit;
    if (grid > total) grid = total;
    auto launch = [&](auto kern) -> hipError_t {
        if (hipError_t e = allow_dynamic_lds((const void*)kern, P8_LDS); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), P8_LDS, stream, (const uint16_t*)store, (const uint16_t*)q_pad, (int)dim_pad,
                           (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key, ws.cand, ws.cnt, (int)ws.cap,
                           ws.overflow, ws.extra);
        return hipGetLastError();
    };
    if constexpr (LEAD != 7) return launch(mips_filter8ph_kernel<DT, false, KEEPB0, LEAD>);  // (experiment builds: no subset instantiation)
    else return subset ? launch(mips_filter8ph_kernel<DT, true, KEEPB0>) : launch(mips_filter8ph_kernel<DT, false, KEEPB0>);
}

hipError_t launch_filter_8phase(int store_dtype, int variant, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end,
                                int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    int n_xtiles = (int)((row_end - row_begin + 255) / 256);
    if (ws.extra.perm_mod > 0) row_end = ws.extra.row_bound;  // permuted stage order: whole positions, rows masked at ntotal
    const bool keep_b0 = variant != 13;
#ifdef VODHIP_EXPERIMENTS
    if (variant == 15)
        return store_dtype == 0 ? launch_8phase_dt<0, true, 6>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream)
                                : launch_8phase_dt<1, true, 6>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream);
    if (variant == 16)
        return store_dtype == 0 ? launch_8phase_dt<0, true, 5>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream)
                                : launch_8phase_dt<1, true, 5>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream);
    if (!keep_b0)
        return store_dtype == 0 ? launch_8phase_dt<0, false>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream)
                                : launch_8phase_dt<1, false>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream);
#endif
    (void)keep_b0;
    return store_dtype == 0 ? launch_8phase_dt<0, true>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream)
                            : launch_8phase_dt<1, true>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream);
}

}  // namespace vodhip

// Fused inner-product scoring + streaming top-k for gfx950 (MI355X, CDNA4).
//
// Replaces the arithmetic behind `faiss_index.search(query_vec, k)`
// (/root/reference/src/vod_search/faiss_search/server.py:72,84) for a Flat / inner-product index.
//
// Data layout in HBM
//   store X : [capacity][dim_pad] fp16|bf16, row-major, dim_pad % 64 == 0, rows >= ntotal are zero or stale
//   queries : [nq_pad][dim_pad] same dtype (workspace), nq_pad % BN == 0, padding rows are zero
//
// A filter workgroup owns BM (corpus rows) x BN (queries) score tiles.  The K loop stages 64-deep slices of both
// operands into LDS with LDS-DMA (`global_load_lds_dwordx4`, 16 B per lane, the XOR swizzle applied on the per-lane
// SOURCE address so the LDS image stays lane-linear) and runs MFMA with the corpus as the A operand and the queries
// as the B operand, so that in the accumulator a lane's column is ONE query and its registers are corpus rows.
// The score tile never leaves registers.  Three epilogues (MODE):
//   FILTER  each lane compares its scores with its query's running threshold (a lower bound of the k-th best score)
//           and only survivors are appended (packed 64-bit keys) to the query's candidate list;
//   DENSE   every score of a short chunk becomes a candidate (tiny indexes, the exhaustive fallback);
//   GMAX    threshold bootstrap: the lane writes the MAXIMUM of its rows (a group of 16 / 32 sampled rows) per query;
//           the k-th largest of any set of group maxima is a valid lower bound of the k-th best score.
// mips_select_kernel folds candidates into the running top-k and tightens the threshold between stages.
//
// Roofline: 2*nq*N*D flop per batch on MFMA vs N*D*2 bytes of HBM; arithmetic intensity = nq flop/B.
#include "mips_common.h"
#include "wg_sort.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <type_traits>

namespace vodhip {

enum : int { MODE_FILTER = 0, MODE_DENSE = 1, MODE_GMAX = 2 };

// ------------------------------------------------------------------------------------------------
// generic filter kernel (v_mfma_f32_32x32x16): small batches (nq <= 128), short chunks, dense chunks
// ------------------------------------------------------------------------------------------------
// NSTAGE : LDS ring depth.  Slice t+NSTAGE-1 is being fetched while slice t is multiplied, so (NSTAGE-2) whole
//          slices stay in flight ACROSS the per-slice barrier (counted vmcnt, raw s_barrier).
// Sampled rows (GMAX): S = n_groups * 16 rows at stride `rstride`, STRATIFIED: member m of lane group g is sampled row
//          m * n_groups + g, so every group (= the 16 rows whose maximum one lane reports) holds one row of each
//          sixteenth of the store and the bound is as tight for topic-sorted stores as for shuffled ones.
template <int DT, int BM, int BN, int WM, int WN, int NSTAGE, int MODE, bool SUBSET = false>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN * 64 * ((160 * 1024) / (NSTAGE * (BM + BN) * 128))) / 256 >= 2 ? 2 : 1)
void mips_filter_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end,
    int n_xtiles, int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key,
    key_t64* __restrict__ cand, unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    constexpr int BK = 64;
    constexpr int NWAVES = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tile
    constexpr int MI = TM / 32, NJ = TN / 32;  // 32x32 blocks per wave
    constexpr int ROW_BYTES = BK * 2;
    constexpr int RPI = 8;                     // rows covered by one LDS-DMA wave-instruction
    constexpr int KK = BK / 16;                // MFMA k-steps per slice
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NA = BM / RPI / NWAVES;      // LDS-DMA wave-instructions per wave per slice
    constexpr int NB = BN / RPI / NWAVES;
    constexpr int G = NA + NB;
    static_assert(BM % (RPI * NWAVES) == 0 && BN % (RPI * NWAVES) == 0, "tile/wave mismatch");
    static_assert(NSTAGE >= 2 && NSTAGE <= 3, "ring depth");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the n_qtiles workgroups that
    // re-read one corpus tile are dealt to the same XCD back to back (L2 reuse only; correctness does not depend on it)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jj = bid >> 3;
    const int qt = jj % n_qtiles;
    const int xt = (jj / n_qtiles) * 8 + xcd;
    if (xt >= n_xtiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // first corpus row of the tile (FILTER: through the stage order of FilterExtra; DENSE: contiguous)
    const int x0 = MODE == MODE_FILTER ? filter_tile_row0(ex, row_begin, xt, BM) : row_begin + xt * BM;
    const int q0 = qt * BN;              // first query of the tile

    // per-lane LDS-DMA source pointers: lane -> (row = base + lane/8, 16-B slot = lane%8).  Slot s of row r holds
    // logical chunk c = s ^ ((r>>1)&7): the 16 rows of a ds_read_b128 lane group then hit 16 distinct 16-B slots of
    // the 256-B bank row.  The XOR is applied on the SOURCE address; the LDS image stays lane-linear.
    const int st_row = lane >> 3, st_slot = lane & 7;
    const char* a_src[NA];
    const char* b_src[NB];
#pragma unroll
    for (int t = 0; t < NA; ++t) {
        const int r = (wave * NA + t) * RPI + st_row;
        const int c = st_slot ^ ((r >> 1) & 7);
        size_t grow = (size_t)(x0 + r);
        if constexpr (MODE == MODE_GMAX) {
            // tile row r belongs to lane group (wm', i, fh) and is its member `idx` (C layout below): sample index =
            // idx * n_groups + group id, i.e. every group holds ONE row of each of 16 strata of the store
            const int bb = r & 31;
            const int grp_in_tile = ((r / TM) * MI + (r % TM) / 32) * 2 + ((bb >> 2) & 1);
            const int idx = (bb >> 3) * 4 + (bb & 3);
            grow = ((size_t)idx * (size_t)ex.sample_groups + (size_t)xt * (BM / 16) + grp_in_tile) * (size_t)ex.sample_rstride + (size_t)ex.sample_offset;
        }
        a_src[t] = (const char*)X + (grow * dim_pad + c * 8) * 2;
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int r = (wave * NB + t) * RPI + st_row;
        const int c = st_slot ^ ((r >> 1) & 7);
        b_src[t] = (const char*)Q + ((size_t)(q0 + r) * dim_pad + c * 8) * 2;
    }
    // `part` of `nparts` of slice `ks`'s LDS-DMA into ring slot ks % NSTAGE
    auto stage_part = [&](int ks, int part, int nparts) {
        char* sa = smem + (ks % NSTAGE) * STAGE_BYTES;
        char* sb = sa + A_BYTES;
        const int kbyte = ks * ROW_BYTES;
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if ((u * nparts) / G != part) continue;
            if (u < NA)
                glds16(a_src[u] + kbyte, sa + (wave * NA + u) * RPI * ROW_BYTES);
            else
                glds16(b_src[u - NA] + kbyte, sb + (wave * NB + (u - NA)) * RPI * ROW_BYTES);
        }
    };

    // fragment read addressing.  MFMA 32x32x16: lane l supplies A[row l&31][k = 8h..8h+7] and B[k = 8h..8h+7][col l&31],
    // h = l>>5.  Logical chunk of k-step kk is 2*kk + h; every block starts at a multiple of 32 rows, so the swizzle
    // term depends on the lane only.
    const int fr = lane & 31, fh = lane >> 5;
    const int swz = (fr >> 1) & 7;
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
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage_part(s, 0, 1);

    // thresholds of this lane's queries (issued after the DMA so they do not delay it; consumed in the epilogue)
    float thr[NJ];
    if constexpr (MODE == MODE_FILTER) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = q0 + wn * TN + j * 32 + fr;
            thr[j] = (q < nq) ? thr_s[q] : __builtin_inff();
        }
    }

    auto load_frags = [&](const char* base, int kk, u32x4(&af)[MI], u32x4(&bf)[NJ]) {
        const int slot_off = ((2 * kk + fh) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const u32x4*)(base + a_row_off + i * 32 * ROW_BYTES + slot_off);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[j] = *(const u32x4*)(base + b_row_off + j * 32 * ROW_BYTES + slot_off);
    };
    auto mfma_group = [&](u32x4(&af)[MI], u32x4(&bf)[NJ]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<DT>(af[i], bf[j], acc[i][j]);
    };
    // One K-slice, software pipelined: the fragments of k-step kk+1 are read from LDS while the MFMAs of k-step kk
    // run, and the LDS-DMA of slice t+NSTAGE-1 is issued in KK parts between the MFMA groups.
    auto kslice = [&](int t, auto pre_tag) {
        constexpr bool PRE = decltype(pre_tag)::value;
        const char* base = smem + (t % NSTAGE) * STAGE_BYTES;
        const int ks = t + NSTAGE - 1;
        u32x4 af0[MI], bf0[NJ], af1[MI], bf1[NJ];
        load_frags(base, 0, af0, bf0);
        if constexpr (PRE) stage_part(ks, 0, KK);
        load_frags(base, 1, af1, bf1);
        mfma_group(af0, bf0);
        if constexpr (PRE) stage_part(ks, 1, KK);
        load_frags(base, 2, af0, bf0);
        mfma_group(af1, bf1);
        if constexpr (PRE) stage_part(ks, 2, KK);
        load_frags(base, 3, af1, bf1);
        mfma_group(af0, bf0);
        if constexpr (PRE) stage_part(ks, 3, KK);
        mfma_group(af1, bf1);
    };
    // Slice t is complete in LDS once all but the (NSTAGE-2)*G youngest DMAs of every wave have landed and every wave
    // has passed the barrier (which also proves nobody still reads the slot being refilled).
    int t = 0;
    for (; t + NSTAGE - 1 < nk; ++t) {
        wait_vmcnt<(NSTAGE - 2) * G>();
        __builtin_amdgcn_s_barrier();
        kslice(t, std::true_type{});
    }
    for (; t < nk; ++t) {  // tail: nothing left to fetch, the ring drains
        if (NSTAGE >= 3 && nk - 1 - t >= 1) wait_vmcnt<(NSTAGE >= 3 ? 1 : 0) * G>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        kslice(t, std::false_type{});
    }

    // ---- epilogue.  C layout of v_mfma_f32_32x32x16: col = lane&31 (query), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = q0 + wn * TN + j * 32 + fr;
        const bool q_ok = q < nq;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rbase = x0 + wm * TM + i * 32 + 4 * fh;
            if constexpr (MODE == MODE_DENSE) {
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
                if constexpr (MODE == MODE_GMAX) {
                    // group = the 16 sampled rows this lane holds of block (wm, i): slot in [0, n_xtiles * BM/16)
                    const int slot = xt * (BM / 16) + (wm * MI + i) * 2 + fh;
                    if (q_ok) cand[(size_t)q * cap + slot] = (m == m) ? make_key(m, (unsigned)slot) : 0ull;
                } else {
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
}

// ------------------------------------------------------------------------------------------------
// persistent filter kernel (v_mfma_f32_16x16x32, 256 x 256 tile): one workgroup per CU walks a list of corpus tiles
// ------------------------------------------------------------------------------------------------
// 8 waves (2 x 4, 128 x 64 each), 64-deep slices, 2 LDS slots.  The grid is (a multiple of 8 * n_qtiles) <= #CUs and a
// workgroup keeps its q-tile while it steps through corpus tiles xt, xt + G/n_qtiles, ...  The K slices of consecutive
// tiles form ONE stream through the two LDS slots: the LDS-DMA of the next tile's first slice is issued during the
// current tile's last slice, so its latency hides behind that slice's MFMAs and the epilogue.
// Fragment maps of v_mfma_f32_16x16x32: lane l supplies A[row l&15][k = 8(l>>4) .. +7] and B[k = 8(l>>4) .. +7][col l&15];
// C/D: col = l&15 (query), row = 4(l>>4) + reg.
//
// STAGGER (MI355X_MICROARCH "Two waves per SIMD" item 9): the two waves that share a SIMD (w and w + 4) run the same
// per-slice program and would reach their LDS read bursts, their matrix work and the barrier together, so the reads
// of one never hide behind the MFMAs of the other.  With STAGGER waves 4..7 run one k32-step behind: they keep the
// fragments of a slice's second k-step in registers across the barrier and multiply them while waves 0..3 read the
// next slice (interval = M R M R for waves 4..7, R M R M for waves 0..3), and their tile epilogue moves into the next
// tile's first interval, beside the partner's MFMAs.  Results are bit-identical (same products, same summation order).
//
// Survivors go to a per-wave list in LDS (positions from a ballot prefix: no atomic, no workgroup barrier) that the wave
// itself flushes to the global candidate lists (exact-key test, subset test, one global atomic per record; all records of a
// flush share the two memory round trips) when it holds WSTG_FLUSH records and at kernel end; a record that does not fit
// is emitted directly.
constexpr int WSTG_CAP = 256;    // records per wave list (8 lists x 3 KB next to the two 64 KB operand slots)
constexpr int WSTG_FLUSH = 176;  // flush when at least this many are pending (checked once per tile)
constexpr int WSTG_BYTES = 8 * WSTG_CAP * 12;
template <int DT, int MODE, bool SUBSET, bool STAGGER>
__global__ __launch_bounds__(512, 2) void mips_filter16p_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end,
    int n_xtiles, int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key,
    key_t64* __restrict__ cand, unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    static_assert(MODE == MODE_FILTER || MODE == MODE_GMAX, "dense chunks run on mips_filter_kernel");
    constexpr int BM = 256, BN = 256, WN = 4, NWAVES = 8, BK = 64, NSTAGE = 2;
    constexpr int TM = 128, TN = 64;
    constexpr int MB = TM / 16, NB16 = TN / 16;
    constexpr int ROW_BYTES = BK * 2, RPI = 8;
    constexpr int A_BYTES = BM * ROW_BYTES, STAGE_BYTES = (BM + BN) * ROW_BYTES;
    constexpr int NA = BM / RPI / NWAVES, NBI = BN / RPI / NWAVES, G = NA + NBI;

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
    const int nk = dim_pad / BK;
    // FILTER: tile xt = rows [row_begin + xt*256, +256).  GMAX: tile xt = the 8 lane groups xt*8 .. xt*8+7 of a stratified
    // row sample (see the a_src set-up; every sampled row lies below ntotal by construction of the schedule).
    const size_t tile_rows_stride = MODE == MODE_GMAX ? (size_t)8 * (size_t)ex.sample_rstride : (size_t)BM;
    size_t tile_step_bytes = (size_t)xt_step * tile_rows_stride * dim_pad * 2;
#ifdef VODHIP_ABLATION  // timing-only knobs of diagnostic builds (`make ABLATION=1`, "kflags" parameter); results are wrong with bit 0 / 3
    const bool abl_l2hot = (ex.flags & (1 << 8)) != 0;      // every workgroup re-reads the same 16 corpus tiles: no HBM first touch
    const bool abl_dma_early = (ex.flags & (2 << 8)) != 0;  // all LDS-DMA of the next slice right after the barrier
    const bool abl_prio = (ex.flags & (4 << 8)) != 0;       // static s_setprio 1 for waves 4..7
    const bool abl_nosurv = (ex.flags & (8 << 8)) != 0;     // thresholds +inf: epilogue floor
    const bool abl_corpus_nt = (ex.flags & (16 << 8)) != 0;   // corpus LDS-DMA with the nt cache policy
    const bool abl_corpus_sc0 = (ex.flags & (32 << 8)) != 0;  // corpus LDS-DMA with sc0
    const bool abl_blocked = (ex.flags & (64 << 8)) != 0;     // corpus read AS IF stored [tile][k-slice][256 rows][128 B]: one tile = 384 contiguous KB (timing only)
    const bool abl_q0 = (ex.flags & (128 << 8)) != 0;         // every q-tile reads the rows of q-tile 0: the query working set of an XCD shrinks from nq to 256 rows (bytes / timing only)
    if (abl_l2hot) tile_step_bytes = 0;
    if (abl_prio && wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(1);
#else
    constexpr bool abl_l2hot = false, abl_dma_early = false, abl_nosurv = false, abl_blocked = false, abl_q0 = false;
#endif

    const int st_row = lane >> 3, st_slot = lane & 7;
    const char* a_src[NA];
    const char* b_src[NBI];
#pragma unroll
    for (int t = 0; t < NA; ++t) {
        const int r = (wave * NA + t) * RPI + st_row;
        size_t grow = (size_t)filter_tile_row0(ex, row_begin, abl_l2hot ? (xt0 & 15) : xt0, BM) + r;
        if constexpr (MODE == MODE_GMAX) {
            // tile row r = wm'*128 + i*16 + 4*fq' + rr is member (i, rr) of lane group (wm', fq'): sample index =
            // member * n_groups + group id, i.e. every group holds ONE row of each of 32 strata of the store
            const int grp_in_tile = (r >> 7) * 4 + ((r >> 2) & 3);
            const int idx = ((r >> 4) & 7) * 4 + (r & 3);
            grow = ((size_t)idx * (size_t)ex.sample_groups + (size_t)xt0 * 8 + grp_in_tile) * (size_t)ex.sample_rstride + (size_t)ex.sample_offset;
        }
        a_src[t] = (const char*)X + (grow * dim_pad + (st_slot ^ ((r >> 1) & 7)) * 8) * 2;
        if (abl_blocked)
            a_src[t] = (const char*)X + ((size_t)row_begin + (size_t)xt0 * BM) * dim_pad * 2 + (size_t)r * ROW_BYTES + (st_slot ^ ((r >> 1) & 7)) * 16;
    }
#pragma unroll
    for (int t = 0; t < NBI; ++t) {
        const int r = (wave * NBI + t) * RPI + st_row;
        b_src[t] = (const char*)Q + ((size_t)((abl_q0 ? 0 : q0) + r) * dim_pad + (st_slot ^ ((r >> 1) & 7)) * 8) * 2;
    }
    const bool corpus_nt = (ex.flags & FILTER_FLAG_CORPUS_NT) != 0;  // single q-tile: corpus lines are read once (nt cache policy)
    auto stage_part = [&](int slot, int kbyte, int part, int nparts) {
        char* sa = smem + slot * STAGE_BYTES;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if ((u * nparts) / G != part) continue;
            if (u < NA) {
                const int kb = abl_blocked ? kbyte * BM : kbyte;
                if (corpus_nt) { glds16_aux<2>(a_src[u] + kb, sa + (wave * NA + u) * RPI * ROW_BYTES); continue; }
#ifdef VODHIP_ABLATION
                if (abl_corpus_nt) { glds16_aux<2>(a_src[u] + kb, sa + (wave * NA + u) * RPI * ROW_BYTES); continue; }
                if (abl_corpus_sc0) { glds16_aux<1>(a_src[u] + kb, sa + (wave * NA + u) * RPI * ROW_BYTES); continue; }
#endif
                glds16(a_src[u] + kb, sa + (wave * NA + u) * RPI * ROW_BYTES);
            } else {
                glds16(b_src[u - NA] + kbyte, sb + (wave * NBI + (u - NA)) * RPI * ROW_BYTES);
            }
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;
    const int a_row_off = (wm * TM + fr) * ROW_BYTES;
    const int b_row_off = A_BYTES + (wn * TN + fr) * ROW_BYTES;

    // thresholds first, and retired (the dummy use makes the compiler wait HERE): a later compiler-inserted vmcnt wait
    // for them would also drain the LDS-DMAs that are meant to stay in flight across the epilogue
    float thr[NB16];
#pragma unroll
    for (int j = 0; j < NB16; ++j) {
        const int q = q0 + wn * TN + j * 16 + fr;
        thr[j] = (MODE == MODE_FILTER && q < nq && !abl_nosurv) ? thr_s[q] : __builtin_inff();
    }
#pragma unroll
    for (int j = 0; j < NB16; ++j) asm volatile("" : "+v"(thr[j]));

    stage_part(0, 0, 0, 1);

    // ---- per-wave survivor list ------------------------------------------------------------------
    key_t64* const wl_key = (key_t64*)(smem + NSTAGE * STAGE_BYTES) + wave * WSTG_CAP;
    int* const wl_q = (int*)(smem + NSTAGE * STAGE_BYTES + NWAVES * WSTG_CAP * 8) + wave * WSTG_CAP;
    int wl_n = 0;  // wave-uniform
    // A flush costs the wave (and, at the next barrier, its workgroup) two dependent global round trips - the exact-key
    // test against thr_key, then the returning atomic that reserves the slot - whatever the number of records, so all
    // (up to 4 per lane) go through each phase together and the list is sized to make flushes rare.
    auto wl_flush = [&]() {
        const int n = wl_n < WSTG_CAP ? wl_n : WSTG_CAP;
        constexpr int PER_LANE = WSTG_CAP / 64;
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
    };
    // every lane with `p` appends (key, q): list position = wl_n + rank of the lane among the appending lanes
    auto wl_append = [&](bool p, key_t64 key, int q) {
        const unsigned long long bal = __ballot(p);
        if (bal == 0ull) return;
        const int pos = wl_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (p) {
            if (pos < WSTG_CAP) {
                wl_key[pos] = key;
                wl_q[pos] = q;
            } else {
                emit_candidate<SUBSET>(key, q, thr_key, cand, cnt, cap, overflow, ex);
            }
        }
        wl_n += __builtin_popcountll(bal);
    };

    struct Frags {
        u32x4 b[NB16], a[MB];
    };
    f32x4 acc[MB][NB16];

    auto read_b = [&](const char* base, int ks, Frags& f) {
        const int slot_off = ((4 * ks + fq) ^ swz) << 4;
#pragma unroll
        for (int j = 0; j < NB16; ++j) f.b[j] = *(const u32x4*)(base + b_row_off + j * 16 * ROW_BYTES + slot_off);
    };
    auto read_a = [&](const char* base, int ks, Frags& f, int i0, int i1) {
        const int slot_off = ((4 * ks + fq) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < MB; ++i)
            if (i >= i0 && i < i1) f.a[i] = *(const u32x4*)(base + a_row_off + i * 16 * ROW_BYTES + slot_off);
    };
    auto mma = [&](const Frags& f, int i0, int i1, bool zero_c) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB16; ++j)
                if (i >= i0 && i < i1) acc[i][j] = mfma16<DT>(f.a[i], f.b[j], zero_c ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j]);
    };

    // ---- epilogue of one tile (x_tile = tile index, x0 = its first row) -------------------------------
    auto epilogue = [&](int x_tile, int x0) {
#pragma unroll
        for (int j = 0; j < NB16; ++j) {
            const int q = q0 + wn * TN + j * 16 + fr;
            float m = acc[0][j][0];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[i][j][r]);
            if constexpr (MODE == MODE_GMAX) {
                // group = the 32 sampled rows this lane holds: slot in [0, n_xtiles * 8)
                const int slot = x_tile * 8 + wm * 4 + fq;
                if (q < nq) cand[(size_t)q * cap + slot] = (m == m) ? make_key(m, (unsigned)slot) : 0ull;
            } else {
                const bool hit = m >= thr[j];  // false for NaN and for padded queries (thr = +inf)
                if (__any(hit)) {
                    // cold path.  Opaque copies: everything derived from the tile's row base and bound stays INSIDE
                    // (the compiler otherwise hoists row keys / bound compares, which do not depend on j, into the
                    // per-tile fast path and spills them)
                    int x0_o = x0, row_end_o = row_end;
                    asm volatile("" : "+s"(x0_o), "+s"(row_end_o));
                    auto val = [&](int v) { return acc[v >> 2][j][v & 3]; };
                    // per-lane survivor mask over its 32 rows, then a ROLLED loop that appends one survivor per lane and
                    // pass to the wave's LDS list (short code: this path is cold, and instruction-cache misses of a long
                    // unrolled version cost more than the loop).  The common case - every hit lane has ONE survivor, which
                    // is its maximum m - needs no register select.
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
                            // register select by the bits of b: 16 + 8 + 4 + 2 + 1 v_cndmask.  Inline asm on purpose: written
                            // as C++ selects LLVM rewrites the tree into an indexed load from a SCRATCH copy of the whole
                            // accumulator, stored after every MFMA of the hot loop
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
                        wl_append(p && rw < row_end_o, make_key(sc, (unsigned)rw), q);
                    } while (__any(mask != 0u));
                }
            }
        }
        if constexpr (MODE == MODE_FILTER) {
            if (wl_n >= WSTG_FLUSH) wl_flush();
        }
    };

    // what slice g+1 is (fetched during slice g): the next slice of this tile, or slice 0 of the next tile
    int super_cur = 0, super_epi = 0, perm_inc = 0;  // permuted stage order: super-tile being fetched / multiplied, position step mod T
    if (MODE == MODE_FILTER && ex.perm_mod > 0) {
        super_cur = filter_tile_row0(ex, row_begin, xt0, BM) / BM;
        perm_inc = (int)(((unsigned long long)xt_step * (unsigned long long)ex.perm_mul) % (unsigned long long)ex.perm_mod);
    }
    auto next_fetch = [&](int it, int t, bool& pre, int& kbyte) {
        pre = true;
        kbyte = (t + 1) * ROW_BYTES;
        if (t + 1 == nk) {
            kbyte = 0;
            super_epi = super_cur;
            pre = it + 1 < n_my;
            if (pre) {
                long long step = (long long)tile_step_bytes;
                if (MODE == MODE_FILTER && ex.perm_mod > 0 && !abl_l2hot) {
                    // the next position's super-tile lies anywhere in the store: (p + xt_step) * P mod T = current + perm_inc (mod T)
                    int nxt = super_cur + perm_inc;
                    if (nxt >= ex.perm_mod) nxt -= ex.perm_mod;
                    step = (long long)(nxt - super_cur) * (long long)BM * (long long)dim_pad * 2;
                    super_cur = nxt;
                }
#pragma unroll
                for (int u = 0; u < NA; ++u) a_src[u] += step;
            }
        }
    };
    // first row of tile `it` of this workgroup.  With the permuted stage order the non-staggered loop keeps it incrementally (the epilogue
    // of tile `it` runs right after the fetch stream moved on to tile it + 1: `super_epi` is the value before that step); the staggered
    // variant, whose epilogue is deferred by a slice, recomputes it
    auto tile_row0 = [&](int it) {
        if (MODE == MODE_FILTER && ex.perm_mod > 0) return STAGGER ? filter_tile_row0(ex, row_begin, xt0 + it * xt_step, BM) : super_epi * BM;
        return row_begin + (xt0 + it * xt_step) * BM;
    };
    int g = 0;  // global slice counter of this workgroup: LDS slot = g & 1
    // waves 0..3 (all waves without STAGGER), per slice:  R(ks0) M(ks0) R(ks1) M(ks1)
    // no zero fill (128 v_mov per tile): the first k32 step of the tile's first slice multiplies into a constant-0 C
    auto slice_a = [&](int it, int t, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const char* base = smem + (g & 1) * STAGE_BYTES;
        bool pre;
        int kbyte;
        next_fetch(it, t, pre, kbyte);
        const int nslot = (g + 1) & 1;
        if (abl_dma_early && pre) stage_part(nslot, kbyte, 0, 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frags f;
            read_b(base, ks, f);
            read_a(base, ks, f, 0, 4);
            if (pre && !abl_dma_early) stage_part(nslot, kbyte, 2 * ks, 4);
            read_a(base, ks, f, 4, 8);
            mma(f, 0, 4, FIRST && ks == 0);
            if (pre && !abl_dma_early) stage_part(nslot, kbyte, 2 * ks + 1, 4);
            mma(f, 4, 8, FIRST && ks == 0);
        }
        ++g;
    };
    // waves 4..7 with STAGGER, per slice:  M(previous slice's ks1) R(ks0) M(ks0) R(ks1).  `fp` carries the fragments of
    // a slice's second k-step across the barrier.  The slice is written in two halves because the deferred epilogue of
    // the previous tile sits between the head and the body of a tile's first slice.
    Frags fp;
    bool pre_b = false;
    int kbyte_b = 0;
    auto slice_b_head = [&](int it, int t) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        next_fetch(it, t, pre_b, kbyte_b);
        // the LDS-DMA of the next slice first: its slot was last read before this barrier by every wave
        if (pre_b) stage_part((g + 1) & 1, kbyte_b, 0, 2);
    };
    auto slice_b_body = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const char* base = smem + (g & 1) * STAGE_BYTES;
        __builtin_amdgcn_sched_barrier(0);
        Frags f;
        read_b(base, 0, f);
        read_a(base, 0, f, 0, 8);
        if (pre_b) stage_part((g + 1) & 1, kbyte_b, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(f, 0, 8, FIRST);
        __builtin_amdgcn_sched_barrier(0);
        read_b(base, 1, fp);
        read_a(base, 1, fp, 0, 8);
        // these reads must be complete before this wave passes the next barrier: the slot is refilled behind it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ++g;
    };
    if (!STAGGER || wave < NWAVES / 2) {
        for (int it = 0; it < n_my; ++it) {
            slice_a(it, 0, std::true_type{});
            for (int t = 1; t < nk; ++t) slice_a(it, t, std::false_type{});
            epilogue(xt0 + it * xt_step, tile_row0(it));
        }
    } else {
        // iteration `it` = head of tile it's first slice, the last k-step and the epilogue of tile it-1 (ONE epilogue
        // site: its survivor path is long, cold code), then the rest of tile it; one extra iteration drains the last tile
        for (int it = 0; it <= n_my; ++it) {
            if (it < n_my) slice_b_head(it, 0);
            if (it > 0) {
                mma(fp, 0, 8, false);
                __builtin_amdgcn_sched_barrier(0);
                epilogue(xt0 + (it - 1) * xt_step, tile_row0(it - 1));
            }
            if (it < n_my) {
                slice_b_body(std::true_type{});
                for (int t = 1; t < nk; ++t) {
                    slice_b_head(it, t);
                    mma(fp, 0, 8, false);
                    slice_b_body(std::false_type{});
                }
            }
        }
    }
    if constexpr (MODE == MODE_FILTER) wl_flush();
}

// ------------------------------------------------------------------------------------------------
// select kernel: fold the stage's candidates into the running sorted top-k, tighten the threshold
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

// k-th largest of the workgroup's 64-bit keys, KPT per thread in REGISTERS (zeros = padding): MSB-first radix select, up to 8 bits
// per pass with a 256-bin LDS histogram.  Returns the key on every thread (0 when fewer than k keys are real).
//   * The keys never sit in LDS: every pass (and the caller's compaction) runs over statically indexed registers, so its LDS
//     atomics issue back to back.  (Round 2 re-read the keys from LDS in a rolled loop each pass: ~70 ns of read latency per key.)
//   * The score bits every real key shares are skipped up front (AND / OR of the high words: one wave reduction): they are the
//     expensive passes - the sign / exponent byte of a batch of candidate scores is ONE bin, i.e. serialised LDS atomics - and the
//     first digit starts AT the highest differing bit, so it spreads the keys over its bins.
//   * Two barriers per pass: the bins alternate between two buffers (the idle one is cleared while the other is filled), and the
//     suffix scan over the 256 bins is done by ONE wavefront (4 bins per lane, shuffle scan) which publishes digit / rank / bin size.
//   * As soon as the bin that holds the k-th key holds ONE key (typically after 2 passes: distinct float scores) its holder
//     publishes it.
// hist (SEL_HIST_INTS ints): [0..255] | [256..511] bins | 512 digit | 513 rank | 514 bin size | 515 / 516 key low / high |
//                            517 compaction counter (the caller's) | 520..531 per-wave reduction words.
constexpr int SEL_HIST_INTS = 544;

// Wave-wide inclusive scan / reduction with DPP row shifts and row broadcasts (GFX9 encodings: row_shr:n = 0x110 + n, row_bcast:15 =
// 0x142, row_bcast:31 = 0x143): six VALU instructions instead of six ds_bpermute round trips (~100 cycles each) - the select kernel
// is a chain of such steps.  After the sequence lane l holds op(values of lanes 0..l); lane 63 the reduction.
template <typename Op>
__device__ __forceinline__ int wave_scan_dpp(int v, int identity, Op op) {
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false));  // lane 15 / 47 -> rows 1 / 3
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false));  // lane 31 -> rows 2 and 3
    return v;
}
#ifdef VODHIP_ABLATION
// phase stamps of workgroup 0: [0..7] the final select, [8..15] the threshold-only one, [16..23] a middle one; [24 + 10 * kind ..]
// inside the radix select of that launch: after the AND / OR reduction, then (bins filled, digit published) per pass
__device__ long long g_probe_select[256];
#endif
#define RSEL_PROBE(i) VODHIP_PROBE(g_probe_select, ppb + (i))
template <int KPT>
__device__ __forceinline__ key_t64 radix_select_kth_regs(const key_t64 (&mine)[KPT], int k, int* hist, int tid, [[maybe_unused]] int ppb) {
    const int lane = tid & 63, wave = tid >> 6;
    unsigned a_hi = ~0u, o_hi = 0u;
    int nz = 0;  // real keys of this WAVE (ballots: wave-uniform, no shuffle)
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const key_t64 e = mine[u];
        const bool real = e != 0ull;
        if (real) {
            a_hi &= (unsigned)(e >> 32);
            o_hi |= (unsigned)(e >> 32);
        }
        nz += __builtin_popcountll(__ballot(real));
    }
    a_hi = (unsigned)wave_scan_dpp((int)a_hi, -1, [](int x, int y) { return x & y; });
    o_hi = (unsigned)wave_scan_dpp((int)o_hi, 0, [](int x, int y) { return x | y; });
    hist[tid] = 0;  // the bins of the first pass
    if (lane == 63) {  // holds the reductions
        int* w = hist + 520 + wave * 3;
        w[0] = (int)a_hi;
        w[1] = (int)o_hi;
        w[2] = nz;
    }
    __syncthreads();
    a_hi = ~0u, o_hi = 0u, nz = 0;
#pragma unroll
    for (int w_ = 0; w_ < 4; ++w_) {
        const int* w = hist + 520 + w_ * 3;
        a_hi &= (unsigned)w[0];
        o_hi |= (unsigned)w[1];
        nz += w[2];
    }
    RSEL_PROBE(0);
    if (nz < k) return 0ull;  // the k-th best is padding
    // digits of up to 8 bits from the highest score bit in which two real keys differ; equal scores everywhere: the ids decide
    const unsigned diff_hi = a_hi ^ o_hi;
    int hi_bit = diff_hi ? 63 - __builtin_clz(diff_hi) : 31;
    key_t64 mask = hi_bit == 63 ? 0ull : ~0ull << (hi_bit + 1);
    key_t64 prefix = ((key_t64)a_hi << 32) & mask;
    int rank = k;  // 1-based rank from the top among the real keys that match the prefix
    for (int pass = 0; hi_bit >= 0; ++pass) {
        int* bins = hist + (pass & 1) * 256;
        const int w = min(8, hi_bit + 1), sh = hi_bit + 1 - w;
        const key_t64 dmask = (1ull << w) - 1ull;
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
            const key_t64 e = mine[u];
            if (e != 0ull && (e & mask) == prefix) atomicAdd(&bins[(int)((e >> sh) & dmask)], 1);
        }
        hist[(((pass + 1) & 1) << 8) + tid] = 0;  // the other buffer: its last reader finished before the previous pass's second barrier
        __syncthreads();
        if (pass < 4) RSEL_PROBE(1 + 2 * pass);
        if (wave == 0) {
            // lane l owns bins 252 - 4l .. 255 - 4l (the HIGH bins sit in the low lanes: a prefix scan over the lanes counts the keys
            // above); suffix sums inside the lane, prefix scan over the lanes
            const int4 h = *reinterpret_cast<const int4*>(bins + 252 - 4 * lane);
            const int s3 = h.w, s2 = s3 + h.z, s1 = s2 + h.y, s0 = s1 + h.x;
            const int above_lane = wave_scan_dpp(s0, 0, [](int x, int y) { return x + y; }) - s0;  // keys in the bins of LOWER lanes
            const int ab[4] = {above_lane + s1, above_lane + s2, above_lane + s3, above_lane};
            const int hh[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ab[j] < rank && rank <= ab[j] + hh[j]) {  // exactly one (lane, j) satisfies this
                    hist[512] = 252 - 4 * lane + j;
                    hist[513] = rank - ab[j];
                    hist[514] = hh[j];
                }
        }
        __syncthreads();
        if (pass < 4) RSEL_PROBE(2 + 2 * pass);
        const int digit = hist[512];
        rank = hist[513];
        const int in_bin = hist[514];
        prefix |= (key_t64)digit << sh;
        mask |= dmask << sh;
        hi_bit = sh - 1;
        if (in_bin == 1 && hi_bit >= 0) {  // workgroup-uniform: the k-th key is the only real key with this prefix
#pragma unroll
            for (int u = 0; u < KPT; ++u) {
                const key_t64 e = mine[u];
                if (e != 0ull && (e & mask) == prefix) {
                    hist[515] = (int)(unsigned)(e & 0xFFFFFFFFull);
                    hist[516] = (int)(unsigned)(e >> 32);
                }
            }
            __syncthreads();
            return ((key_t64)(unsigned)hist[516] << 32) | (key_t64)(unsigned)hist[515];
        }
    }
    return prefix;
}

// The kp keys of `keys` (the k best, unordered, zeros behind) sorted descending.  Up to 512 keys ONE wavefront sorts them in its
// registers (1-8 keys per lane, shuffles only: the 28 barrier-separated LDS stages this replaces were more than half of the last
// select of a search; ranking by counting - kp * kp / 256 broadcast LDS reads per thread, no dependent chain - measured 3.0 us
// against 1.8); above that the 256-thread register network of wg_sort.h.
__device__ __forceinline__ void sort_front_desc(key_t64* keys, int kp, int tid) {
    __syncthreads();
    if (kp <= 512) {
        if (tid < 64) {
            switch (kp) {
                case 64: wave_sort_regs<1, true>(keys, tid); break;
                case 128: wave_sort_regs<2, true>(keys, tid); break;
                case 256: wave_sort_regs<4, true>(keys, tid); break;
                default: wave_sort_regs<8, true>(keys, tid); break;
            }
        }
        __syncthreads();
    } else if (!wg_sort_lds_256<true>(keys, kp, tid)) {
        bitonic_sort_desc_lds(keys, kp, tid);
    }
}

// One workgroup per query; every thread holds KPT keys in registers (256 * KPT >= 2 * kp): the running top-k (kp keys) plus up to
// 256 * KPT - kp candidates per round.  Between stages only the k-th best key (the threshold) and the SET of the k best are needed, so
// intermediate launches use a radix select + compaction (unsorted top-k); the last launch of a search (SELECT_FINAL) then sorts
// those k and writes the result rows.
// SELECT_THRESHOLD_ONLY (after a GMAX stage): the candidates are group maxima, not rows - only the threshold leaves.
// The threshold never decreases: a stage whose candidates do not fill the top-k keeps the bound it was given.
// LDS: the histogram words + `front`, kp keys: where a round's survivors are gathered (and sorted, and read by the next round).
enum : int { SELECT_FINAL = 1, SELECT_THRESHOLD_ONLY = 2 };
#define SEL_PROBE(i) VODHIP_PROBE(g_probe_select, pb + (i))
template <int KPT>
__global__ __launch_bounds__(256) void mips_select_kernel(key_t64* __restrict__ topk, int kp, int k,
                                                          const key_t64* __restrict__ cand,
                                                          unsigned int* __restrict__ cnt, int cap, int dense_n,
                                                          float* __restrict__ thr_s, key_t64* __restrict__ thr_key,
                                                          unsigned int* __restrict__ overflow, unsigned int* __restrict__ ovf_q, int flags,
                                                          int64_t id_base, float* __restrict__ out_scores,
                                                          int64_t* __restrict__ out_ids, const int* __restrict__ q_map) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* hist = (int*)smem;  // SEL_HIST_INTS ints: two sets of bins, scratch words, compaction counter at [517] (see radix_select_kth_regs)
    key_t64* front = (key_t64*)(hist + SEL_HIST_INTS);
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const bool final_sort = (flags & SELECT_FINAL) != 0;
    const bool thr_only = (flags & SELECT_THRESHOLD_ONLY) != 0;
    [[maybe_unused]] const int pb = final_sort ? 0 : thr_only ? 8 : 16;
    SEL_PROBE(0);
    // The first round's keys are requested BEFORE the candidate count is known (the candidate buffer holds `cap` slots per query
    // whatever the count): one global-memory latency instead of two in a kernel that lasts a few microseconds.
    key_t64 mine[KPT];
    const key_t64* cq = cand + (size_t)q * cap;
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = tid + u * 256;
        if (i < kp) mine[u] = thr_only ? 0ull : topk[(size_t)q * kp + i];
        else mine[u] = (i - kp) < cap ? cq[i - kp] : 0ull;
    }
    unsigned n = dense_n >= 0 ? (unsigned)dense_n : cnt[(size_t)q * CNT_STRIDE];
    if (n > (unsigned)cap) {  // this query lost candidates in this stage: the host recovers it (and only it)
        if (tid == 0) {
            atomicOr(overflow, 1u);
            if (ovf_q) ovf_q[q] = 1u;
        }
        n = cap;
    }
    const int room = KPT * 256 - kp;
    int done = 0;
    key_t64 kth = 0;
    do {
        const int take = min((int)n - done, room);
        const int total = kp + take;
        if (done > 0) {  // a further round (rare): the survivors so far + the next candidates
#pragma unroll
            for (int u = 0; u < KPT; ++u) {
                const int i = tid + u * 256;
                mine[u] = i < kp ? front[i] : (i < total ? cq[done + (i - kp)] : 0ull);
            }
        } else {
#pragma unroll
            for (int u = 0; u < KPT; ++u)
                if (tid + u * 256 >= total) mine[u] = 0ull;  // slots behind the count hold stale keys of an earlier stage
        }
        if (tid == 0) hist[517] = 0;  // compaction counter (ordered by the barriers inside the select)
        SEL_PROBE(1);
        kth = radix_select_kth_regs<KPT>(mine, k, hist, tid, 24 + 10 * (pb >> 3));
        SEL_PROBE(2);
        done += take;
        // compaction: the keys >= kth (exactly k of them unless kth == 0) are gathered in `front`, zeros behind.  Not needed when
        // only the threshold leaves and no further round follows.
        if (!thr_only || done < (int)n) {
            // slots by ballot: a key's position = the wave's base + the survivors of this wave before it (earlier register slots,
            // then lower lanes); ONE LDS atomic per wave (256 same-address atomics with return cost ~1 us)
            unsigned long long keep[KPT];
            int n_wave = 0;
#pragma unroll
            for (int u = 0; u < KPT; ++u) {
                keep[u] = __ballot(mine[u] >= kth && mine[u] != 0ull);
                n_wave += __builtin_popcountll(keep[u]);
            }
            int base = 0;
            if ((tid & 63) == 0 && n_wave) base = atomicAdd(&hist[517], n_wave);
            base = __builtin_amdgcn_readfirstlane(base);
            const unsigned long long below = (1ull << (tid & 63)) - 1ull;
#pragma unroll
            for (int u = 0; u < KPT; ++u) {
                const int at = base + __builtin_popcountll(keep[u] & below);
                if (((keep[u] >> (tid & 63)) & 1ull) && at < kp) front[at] = mine[u];
                base += __builtin_popcountll(keep[u]);
            }
            __syncthreads();
            const int n_kept = min(hist[517], kp);
            for (int i = n_kept + tid; i < kp; i += 256) front[i] = 0ull;
            __syncthreads();
        }
    } while (done < (int)n);
    SEL_PROBE(3);
    if (final_sort) {  // the k best are in `front`, unordered
        sort_front_desc(front, kp, tid);
        kth = front[k - 1];
    }
    SEL_PROBE(4);
    if (!thr_only)
        for (int i = tid; i < kp; i += 256) topk[(size_t)q * kp + i] = front[i];
    if (out_scores != nullptr) {  // last select of a search: the sorted keys leave as (float32 score, int64 id) rows
        const size_t qo = q_map ? (size_t)q_map[q] : (size_t)q;
        for (int c = tid; c < k; c += 256) {
            const key_t64 key = front[c];
            out_scores[qo * k + c] = key ? unflip_f32((unsigned)(key >> 32)) : -__builtin_inff();
            out_ids[qo * k + c] = key ? id_base + (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu)) : -1;
        }
    }
    if (tid == 0) {
        // a group maximum stands for "some row with this score": every row with the score must pass `key > thr_key`
        if (thr_only) kth &= 0xFFFFFFFF00000000ull;
        const key_t64 old = thr_key[q];
        if (kth > old) {
            thr_key[q] = kth;
            thr_s[q] = unflip_f32((unsigned)(kth >> 32));
        }
        cnt[(size_t)q * CNT_STRIDE] = 0;
    }
    SEL_PROBE(5);
    if (tid == 0 && q == 0) {
        [[maybe_unused]] const long long nn = (long long)n;
#ifdef VODHIP_ABLATION
        g_probe_select[2 * (pb + 6)] = nn;  // candidates this launch folded for query 0
        g_probe_select[2 * (pb + 6) + 1] = 1;
#endif
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
            float rounded;
            v = store_bits(f, DST, &rounded);  // (finite float32 values beyond the store dtype's range saturate: mips_common.h)
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
// and rows), clear the running top-k / counters and the overflow word, and initialise the thresholds: -inf, or - for
// the recovery pass after a candidate-list overflow - the k-th score of the previous (incomplete but valid) result
// `seed_scores / seed_ids [nq, k]`, which is a lower bound of the true k-th best.
__global__ void mips_prepare_kernel(const void* __restrict__ q_src, int q_dtype, int64_t nq, int64_t dim,
                                    uint16_t* __restrict__ q_pad, int store_dtype, int64_t nq_pad, int64_t dim_pad,
                                    key_t64* __restrict__ topk, int64_t n_topk, unsigned int* __restrict__ cnt,
                                    float* __restrict__ thr_s, key_t64* __restrict__ thr_key,
                                    unsigned int* __restrict__ overflow, unsigned int* __restrict__ ovf_q, int clear_overflow,
                                    const float* __restrict__ seed_scores, const int64_t* __restrict__ seed_ids, int k,
                                    const int* __restrict__ q_map, const float* __restrict__ seed_margin) {
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
                const int64_t srow = q_map ? (int64_t)q_map[row] : row;
                float f;
                if (q_dtype == 2) f = ((const float*)q_src)[srow * dim + c];
                else if (q_dtype == 0) f = (float)(((const _Float16*)q_src)[srow * dim + c]);
                else f = (float)(((const __bf16*)q_src)[srow * dim + c]);
                float rounded;
                v = store_bits(f, store_dtype, &rounded);  // (finite values beyond the store dtype's range saturate: mips_common.h)
            }
            out[e] = v;
        }
        *(uint4*)(q_pad + row * dim_pad + c0) = *(const uint4*)out;
    }
    if (i < n_topk) topk[i] = 0;
    if (i < nq_pad) {
        cnt[i * CNT_STRIDE] = 0;
        float ts = -__builtin_inff();
        key_t64 tk = 0;
        const int64_t srow = (q_map && i < nq) ? (int64_t)q_map[i] : i;
        if (ovf_q != nullptr && i < nq) ovf_q[i] = 0u;
        if (seed_scores != nullptr && i < nq && seed_ids[srow * k + (k - 1)] >= 0) {
            float s = seed_scores[srow * k + (k - 1)];
            if (seed_margin != nullptr) s -= seed_margin[srow];
            if (s == s) {
                ts = s;
                tk = (key_t64)flip_f32(s) << 32;  // low word 0: every row with this score still passes `key > thr_key`
            }
        }
        thr_s[i] = ts;
        thr_key[i] = tk;
    }
    if (i == 0 && clear_overflow) {
        overflow[0] = 0u;
        overflow[1] = 0u;  // exact mode's "some list did not prove complete" word (kernels_exact.hip)
        overflow[2] = 0u;  // exact mode's "list entries within eps of the k-th exact score, maximum over the queries" word
    }
}

hipError_t launch_search_prepare(const SearchWorkspace& ws, const void* q_src, int q_dtype, int64_t nq, int64_t dim,
                                 int store_dtype, int64_t nq_pad, int64_t dim_pad, bool clear_overflow,
                                 const float* seed_scores, const int64_t* seed_ids, int k, const int* q_map, hipStream_t stream,
                                 const float* seed_margin) {
    const int64_t n_topk = nq_pad * ws.kp;
    const int64_t n = std::max(n_topk, nq_pad * (dim_pad / 8));
    const int threads = 256;
    const unsigned blocks = (unsigned)((n + threads - 1) / threads);
    hipLaunchKernelGGL(mips_prepare_kernel, dim3(blocks), dim3(threads), 0, stream, q_src, q_dtype, nq, dim,
                       (uint16_t*)ws.q_pad, store_dtype, nq_pad, dim_pad, ws.topk, n_topk, ws.cnt, ws.thr_s, ws.thr_key,
                       ws.overflow, ws.ovf_q, clear_overflow ? 1 : 0, seed_scores, seed_ids, k, q_map, seed_margin);
    return hipGetLastError();
}

int filter_tile_rows(int tile) { return tile == 1 ? 128 : 256; }
int filter_tile_cols(int tile) { return tile == 42 ? 64 : (tile == 46 || tile == 1) ? 128 : 256; }
int filter_group_rows(int tile) { return filter_tile_is_persistent(tile) ? 32 : 16; }  // rows per GMAX group (tiles 10 / 11 bootstrap on tile 8's kernel)

hipError_t allow_dynamic_lds(const void* kernel, int bytes) {
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

namespace {

struct FilterLaunch {
    const void* store;
    const void* q_pad;
    int64_t dim_pad, row_begin, row_end, nq, nq_pad;
    int n_xtiles;  // FILTER / DENSE: ceil(rows / BM); GMAX: number of sampled tiles
    const SearchWorkspace* ws;
    hipStream_t stream;
};

template <int DT, int BM, int BN, int WM, int WN, int NSTAGE, int MODE, bool SUBSET>
hipError_t launch_generic(const FilterLaunch& L) {
    const int n_qtiles = (int)(L.nq_pad / BN);
    const unsigned grid = (unsigned)((L.n_xtiles + 7) / 8) * 8u * (unsigned)n_qtiles;
    constexpr int threads = WM * WN * 64;
    constexpr size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    auto kern = mips_filter_kernel<DT, BM, BN, WM, WN, NSTAGE, MODE, SUBSET>;
    if (hipError_t e = allow_dynamic_lds((const void*)kern, (int)lds); e != hipSuccess) return e;
    const SearchWorkspace& ws = *L.ws;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, L.stream, (const uint16_t*)L.store, (const uint16_t*)L.q_pad,
                       (int)L.dim_pad, (int)L.row_begin, (int)L.row_end, L.n_xtiles, n_qtiles, (int)L.nq, ws.thr_s,
                       ws.thr_key, ws.cand, ws.cnt, (int)ws.cap, ws.overflow, ws.extra);
    return hipGetLastError();
}

template <int DT, int BM, int BN, int WM, int WN, int NSTAGE>
hipError_t launch_generic_mode(const FilterLaunch& L, int mode, bool subset) {
    if (mode == MODE_GMAX) return launch_generic<DT, BM, BN, WM, WN, NSTAGE, MODE_GMAX, false>(L);
    if (mode == MODE_DENSE)
        return subset ? launch_generic<DT, BM, BN, WM, WN, NSTAGE, MODE_DENSE, true>(L)
                      : launch_generic<DT, BM, BN, WM, WN, NSTAGE, MODE_DENSE, false>(L);
    return subset ? launch_generic<DT, BM, BN, WM, WN, NSTAGE, MODE_FILTER, true>(L)
                  : launch_generic<DT, BM, BN, WM, WN, NSTAGE, MODE_FILTER, false>(L);
}

template <int DT, int MODE, bool SUBSET, bool STAGGER>
hipError_t launch_persistent(const FilterLaunch& L) {
    const int n_qtiles = (int)(L.nq_pad / 256);
    const int n_cu = L.ws->n_cu > 0 ? L.ws->n_cu : 256;  // read once at index create: no driver call on the launch path
    const int unit = 8 * n_qtiles;
    const int total = ((L.n_xtiles + 7) / 8) * unit;
    int grid = (n_cu / unit) * unit;
    if (grid < unit) grid = unit;
    if (grid > total) grid = total;
    constexpr size_t lds = 128 * 1024 + WSTG_BYTES;
    auto kern = mips_filter16p_kernel<DT, MODE, SUBSET, STAGGER>;
    if (hipError_t e = allow_dynamic_lds((const void*)kern, (int)lds); e != hipSuccess) return e;
    const SearchWorkspace& ws = *L.ws;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, L.stream, (const uint16_t*)L.store, (const uint16_t*)L.q_pad,
                       (int)L.dim_pad, (int)L.row_begin, (int)L.row_end, L.n_xtiles, n_qtiles, (int)L.nq, ws.thr_s, ws.thr_key,
                       ws.cand, ws.cnt, (int)ws.cap, ws.overflow, ws.extra);
    return hipGetLastError();
}

template <int DT, bool STAGGER>
hipError_t launch_persistent_mode(const FilterLaunch& L, int mode, bool subset) {
    if (mode == MODE_GMAX) return launch_persistent<DT, MODE_GMAX, false, STAGGER>(L);
    return subset ? launch_persistent<DT, MODE_FILTER, true, STAGGER>(L) : launch_persistent<DT, MODE_FILTER, false, STAGGER>(L);
}

template <int DT>
hipError_t launch_filter_dt(int tile, int mode, bool subset, const FilterLaunch& L) {
    switch (tile) {
        case 1: return launch_generic_mode<DT, 128, 128, 2, 2, 2>(L, mode, subset);   // 64 KB LDS, 2 workgroups / CU: short chunks, dense chunks
        case 42: return launch_generic_mode<DT, 256, 64, 4, 1, 3>(L, mode, subset);   // nq <= 64 (HBM-bound): 256 rows x 64 queries, 4 waves, 3-slot ring (120 KB)
        case 46: return launch_generic_mode<DT, 256, 128, 4, 2, 3>(L, mode, subset);  // 65..128 queries: 256 x 128, 8 waves, 3-slot ring (144 KB)
        case 8: return launch_persistent_mode<DT, false>(L, mode, subset);             // persistent 256 x 256, both waves of a SIMD in lockstep
        case 9: return launch_persistent_mode<DT, true>(L, mode, subset);              // persistent 256 x 256, waves 4..7 staggered by one k-step
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

hipError_t launch_filter(int store_dtype, int tile, int mode, const void* store, const void* q_pad, int64_t dim_pad,
                         int64_t row_begin, int64_t row_end, int64_t n_sample_tiles, int64_t nq, int64_t nq_pad,
                         const SearchWorkspace& ws, hipStream_t stream) {
    if (mode != MODE_GMAX && row_end <= row_begin) return hipSuccess;
    if (mode == MODE_GMAX && n_sample_tiles <= 0) return hipSuccess;
    const bool subset = ws.extra.row_label != nullptr;
    if (filter_tile_is_persistent(tile) && mode == MODE_DENSE) tile = 1;  // nq_pad is a multiple of 256, which the 128-wide tile divides
    if (tile == 14 && mode == MODE_FILTER)  // the 8-phase K loop (kernels_mips_8phase.hip): FILTER stages of batches with >= 2 query tiles
        return launch_filter_8phase(store_dtype, 14, store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
    if (tile == 17 || tile == 18) {  // query tile resident in registers (experiment builds; shapes they do not take run the production 8-phase kernel)
#ifdef VODHIP_EXPERIMENTS
        if (tile == 17 && mode == MODE_FILTER && !subset && filter_qres_supports(dim_pad))
            return launch_filter_qres(store_dtype, store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
        if (tile == 18 && mode == MODE_FILTER && !subset && filter_ksplit_supports(dim_pad))
            return launch_filter_ksplit(store_dtype, store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
#endif
        if (mode == MODE_FILTER) return launch_filter_8phase(store_dtype, 14, store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
        tile = 8;
    }
    if (tile >= 10 && tile <= 16) {  // 10 - 13: experiment FILTER kernels, `make ABLATION=1 EXPERIMENTS=1` builds only (vodhip_index_set_param refuses the ids otherwise)
#ifdef VODHIP_EXPERIMENTS
        if (mode == MODE_FILTER && (tile == 13 || tile >= 15))  // 8-phase K loop variants: B0 re-read in phase 4 (13), LDS-DMA lead 6 / 5 (15 / 16)
            return launch_filter_8phase(store_dtype, tile, store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
        if (mode == MODE_FILTER && tile == 12)  // 384 x 256 workgroup tile
            return launch_filter_wide(store_dtype, store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
        if (mode == MODE_FILTER)  // deep ring (10), with fragments read a k-step ahead (11)
            return launch_filter_ring(store_dtype, tile == 11, store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, ws, stream);
#endif
        tile = 8;  // the bootstrap (group maxima of a row sample) and dense chunks run on the two-slot kernel's family
    }
    const int bm = filter_tile_rows(tile);
    int n_xtiles = mode == MODE_GMAX ? (int)n_sample_tiles : (int)((row_end - row_begin + bm - 1) / bm);
    if (mode == MODE_FILTER && ws.extra.perm_mod > 0) {
        // permuted stage order: whole 256-row positions (the partly filled super-tile may sit at any of them), rows masked at ntotal
        n_xtiles = (int)((row_end - row_begin + 255) / 256) * (256 / bm);
        row_end = ws.extra.row_bound;
    }
    FilterLaunch L{store, q_pad, dim_pad, row_begin, row_end, nq, nq_pad, n_xtiles, &ws, stream};
    return store_dtype == 0 ? launch_filter_dt<0>(tile, mode, subset, L) : launch_filter_dt<1>(tile, mode, subset, L);
}

template <int KPT>
static hipError_t launch_select_kpt(const SearchWorkspace& ws, int64_t nq, int k, int64_t dense_n, int flags, hipStream_t stream,
                                    int64_t id_base, float* out_scores, int64_t* out_ids, const int* q_map) {
    const size_t lds = SEL_HIST_INTS * sizeof(int) + (size_t)ws.kp * sizeof(key_t64);
    const bool final_sort = (flags & SELECT_FINAL) != 0;
    hipLaunchKernelGGL(mips_select_kernel<KPT>, dim3((unsigned)nq), dim3(256), lds, stream, ws.topk, (int)ws.kp, k, ws.cand,
                       ws.cnt, (int)ws.cap, (int)dense_n, ws.thr_s, ws.thr_key, ws.overflow, ws.ovf_q, flags, id_base,
                       final_sort ? out_scores : nullptr, final_sort ? out_ids : nullptr, q_map);
    return hipGetLastError();
}

hipError_t launch_select(const SearchWorkspace& ws, int64_t nq, int k, int64_t dense_n, int flags, hipStream_t stream,
                         int64_t id_base, float* out_scores, int64_t* out_ids, const int* q_map) {
    // keys per round: 2048 (8 per thread) unless the running top-k alone needs more; a stage with a KNOWN large candidate count per
    // query (the bootstrap's group maxima) gets a round that takes them at once
    int sb = 2048;
    while (sb < 2 * ws.kp) sb <<= 1;
    while (dense_n > sb - ws.kp && sb < 8192) sb <<= 1;
    switch (sb) {
        case 2048: return launch_select_kpt<8>(ws, nq, k, dense_n, flags, stream, id_base, out_scores, out_ids, q_map);
        case 4096: return launch_select_kpt<16>(ws, nq, k, dense_n, flags, stream, id_base, out_scores, out_ids, q_map);
        default: return launch_select_kpt<32>(ws, nq, k, dense_n, flags, stream, id_base, out_scores, out_ids, q_map);
    }
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

// Every shard's list arrives sorted best-first with its pads at the tail (what the search writes), so the merge is a tree of
// pairwise TOP-L merges instead of a sort: list A against the mirrored list B through one half-cleaner stage leaves the L best of
// both in A as a bitonic sequence, log2(L) more compare-exchange stages sort it; log2(n_shards) levels.  8 x 100 entries: 24
// stages on shrinking data against the 55 full-width stages of a bitonic sort of 1024 pairs - 14 us per launch at nq = 1024
// (the host-side floor of tools/bench_merge.py) against 55 us; 8 x 200 at nq = 512: 18 against 81.  (A rank-by-binary-search
// variant was no faster than the full sort at 8 shards: ~40 k search steps of ~20 instructions per query.)  Lists are padded to L = pow2 >=
// max(k, min(k_out, n_shards * k)) entries, shards to a power of two.  The kernel verifies the precondition while loading and
// sorts everything (same network, all stages) when a list is not sorted.
__device__ __forceinline__ void merge_ce(float* ssc, int64_t* sid, int a, int b) {  // best of (a, b) to a
    const float sa = ssc[a], sb = ssc[b];
    const int64_t ia = sid[a], ib = sid[b];
    if (pair_before(sb, ib, sa, ia)) {
        ssc[a] = sb;
        ssc[b] = sa;
        sid[a] = ib;
        sid[b] = ia;
    }
}

__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ scores, const int64_t* __restrict__ ids,
                                                         int64_t stride_s, int64_t stride_i, int n_shards, int64_t nq, int k, int k_out,
                                                         int L, int S2, int flat, float* __restrict__ out_scores,
                                                         int64_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // entries in LDS: list t at [t * L, t * L + L); `flat` (padded lists would not fit): one list of all n_shards * k entries
    const int P = S2 * L;
    int64_t* sid = (int64_t*)smem;
    float* ssc = (float*)(smem + (size_t)P * 8);
    int* s_flag = (int*)(smem + (size_t)P * 12);  // all LDS in the one dynamic array
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lmask = L - 1, lshift = 31 - __builtin_clz(L);
    if (tid == 0) *s_flag = 0;
    for (int i = tid; i < P; i += 256) {
        const int t = flat ? i / k : i >> lshift, c = flat ? i % k : i & lmask;
        float s = -__builtin_inff();
        int64_t id = -1;
        if (t < n_shards && c < k) {
            s = scores[(int64_t)t * stride_s + q * k + c];
            id = ids[(int64_t)t * stride_i + q * k + c];
            if (id < 0 || s != s) {
                id = -1;
                s = -__builtin_inff();
            }
        }
        ssc[i] = s;
        sid[i] = id;
    }
    __syncthreads();
    // precondition check: within a list, no entry may come strictly before its predecessor (pads sink to the tail)
    int unsorted = 0;
    for (int i = tid; i < P; i += 256)
        if ((i & lmask) + 1 < L) unsorted |= pair_before(ssc[i + 1], sid[i + 1], ssc[i], sid[i]) ? 1 : 0;
    if (unsorted) *s_flag = 1;
    __syncthreads();
    if (*s_flag) {
        // general case: bitonic sort of all P entries, best first
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                for (int t = tid; t < (P >> 1); t += 256) {
                    const int pos = 2 * t - (t & (stride - 1));
                    if ((pos & size) == 0) merge_ce(ssc, sid, pos, pos + stride); else merge_ce(ssc, sid, pos + stride, pos);
                }
            }
        }
    } else {
        for (int w = 1; w < S2; w <<= 1) {  // level: lists j and j + w (j a multiple of 2w) -> the L best of both, sorted, in list j
            const int n_pairs = S2 / (2 * w);
            __syncthreads();
            for (int e = tid; e < n_pairs * L; e += 256) {  // half-cleaner against the mirrored partner
                const int pr = e >> lshift, c = e & lmask;
                const int a = (pr * 2 * w) * L + c, b = (pr * 2 * w + w) * L + (L - 1 - c);
                merge_ce(ssc, sid, a, b);
            }
            for (int stride = L >> 1; stride > 0; stride >>= 1) {  // list j is bitonic now: sort it
                __syncthreads();
                for (int e = tid; e < n_pairs * (L >> 1); e += 256) {
                    const int pr = e >> (lshift - 1), c = e & ((L >> 1) - 1);
                    const int pos = (pr * 2 * w) * L + 2 * c - (c & (stride - 1));
                    merge_ce(ssc, sid, pos, pos + stride);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < k_out; i += 256) {
        float s = -__builtin_inff();
        int64_t id = -1;
        if (i < ((*s_flag) ? P : L)) {
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
    int L = 1, S2 = 1;
    while (L < std::max(k, std::min(k_out, total))) L <<= 1;
    while (S2 < n_shards) S2 <<= 1;
    int flat = 0;
    if ((size_t)S2 * L * 12 > 128 * 1024) {  // padded lists too large (odd shard counts with k in the thousands): sort everything
        flat = 1;
        S2 = 1;
        L = 1;
        while (L < total) L <<= 1;
    }
    const size_t lds = (size_t)S2 * L * 12 + 16;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = allow_dynamic_lds((const void*)merge_topk_kernel, 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)nq), dim3(256), lds, stream, scores, ids, stride_s, stride_i, n_shards, nq, k,
                       k_out, L, S2, flat, out_scores, out_ids);
    return hipGetLastError();
}

hipError_t read_probe_select(long long* out) {
#ifdef VODHIP_ABLATION
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe_select), sizeof(long long) * 256);
#else
    (void)out;
    return hipErrorNotSupported;
#endif
}

}  // namespace vodhip

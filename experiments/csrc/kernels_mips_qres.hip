// libvodhip experiment (round 6) -- the FILTER stage with the QUERY TILE RESIDENT IN REGISTERS (tile 17).
//
// Round-5 verdict, item 3: in the 129-256-query regime (C2; C3 at nq 256) every 256 x 256 workgroup tile of the 8-phase kernel re-stages
// the SAME 256 x dim query tile through LDS-DMA + ds_read for every corpus tile - half of all staged bytes, half of the LDS-DMA stream's
// depth (3.5 of the 7 half-tiles in flight are query bytes from L2, not corpus bytes from HBM).  Here a workgroup is 4 waves, ONE per SIMD
// (up to 512 registers per lane); wave w keeps the B fragments (v_mfma_f32_16x16x32: B = queries) of ITS 64 queries for the whole
// contraction in registers - 64 q x dim x 2 B / 64 lanes = NK x 4 fragments of 4 registers (dim 768: 384 registers) - loaded once per
// launch.  Only corpus rows move: a ring of 32 units of [32 rows x 64 k] (4 KB, one LDS-DMA wave-instruction per wave per unit) streams
// 30 units (120 KB per CU) ahead; every wave reads EVERY unit's A fragments (2 row blocks x 2 k-steps = 4 ds_read_b128) and issues
// 16 MFMAs on them (2 row blocks x 4 query blocks x 2 k-steps); loop order rows-outer: a 32-row group accumulates its full contraction in
// 8 accumulators (32 registers), then takes the FILTER epilogue of the persistent kernels (per-lane query column, threshold in a
// register, survivors -> per-wave LDS lists).  One s_barrier per unit.  Same products in the same order as tiles 8 / 14: bit-identical.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mips_common.h"

namespace vodhip {

namespace {
constexpr int QR_UNIT = 32 * 128;          // one unit: 32 rows x 128 B (64 k)
constexpr int QR_RING = 32;                // units in the LDS ring (128 KB)
constexpr int QR_LEAD = 30;                // units the LDS-DMA stream runs ahead of the unit being multiplied
constexpr int QR_OPERANDS = QR_RING * QR_UNIT;
constexpr int QR_NA = 14;                  // k-steps whose query fragments live in the accumulator file (14 x 4 fragments x 4 = 224 AGPRs; hipcc keeps the 32 accumulators there too)
constexpr int QR_WL_CAP = 256;             // records per wave list
constexpr int QR_WL_FLUSH = 176;
constexpr int QR_BTAIL = 256 * 64;         // the LAST k-step's query fragments: [256 queries][32 k] = 16 KB, read from LDS by every row group (the register files are full)
constexpr int QR_LISTS = QR_OPERANDS + QR_BTAIL;
constexpr int QR_LDS = QR_LISTS + 4 * QR_WL_CAP * 12;
static_assert(QR_LDS <= 160 * 1024, "LDS budget");

template <int N>
__device__ __forceinline__ void qr_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace

template <int DT, int NK, bool NT>
__global__ __launch_bounds__(256, 1) void mips_filter_qres_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end, int n_xtiles,
    int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key, key_t64* __restrict__ cand,
    unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    constexpr int BM = 256, BN = 256, NWAVES = 4, TN = 64, NB16 = TN / 16, ROW_BYTES = 128, NS = NK / 2, GROUPS = BM / 32;
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
    const int q0 = qt * BN;
    const size_t row_stride = (size_t)dim_pad * 2;

    // ---- the LDS-DMA stream.  Wave w stages LDS rows [8 w, 8 w + 8) of every unit: ONE wave-instruction of 8 rows x 128 B.
    // 16-byte chunk c of LDS row lr sits at slot c ^ ((lr >> 1) & 7) (applied on the source address; fragment reads conflict-free).
    const int st_row = lane >> 3, st_slot = lane & 7;
    int super_cur = ex.perm_mod > 0 ? filter_tile_row0(ex, row_begin, xt0, BM) / BM : 0;
    int super_epi = super_cur;
    const int perm_inc = ex.perm_mod > 0 ? (int)(((unsigned long long)xt_step * (unsigned long long)ex.perm_mul) % (unsigned long long)ex.perm_mod) : 0;
    const int lr = wave * 8 + st_row;
    const char* a_src = (const char*)X + ((size_t)filter_tile_row0(ex, row_begin, xt0, BM) + lr) * row_stride + ((st_slot ^ ((lr >> 1) & 7)) << 4);
    const size_t group_step = 32 * row_stride;
    const size_t tile_step_bytes = (size_t)xt_step * BM * row_stride;
    // The stream is BRANCH-FREE inside a unit (a branch per unit splits the MFMA stream into basic blocks the compiler cannot schedule
    // across): the slice of the unit staged beside unit s of a row group is the compile-time (s + LEAD) % NS; the row group's source
    // pointer advances once per group; past the workgroup's last tile the stream keeps re-reading its last row group (harmless: the ring
    // slots it overwrites are consumed, and the vmcnt accounting stays regular: every unit issues exactly one piece).
    const char* sg_ptr = a_src;   // per-lane source of slice 0 of the row group being staged
    int sg_group = 0, sg_it = 0;  // ... its index inside the tile, the tile
    auto stage_slice = [&](int slice, int unit) {
        char* dst = smem + (unit & (QR_RING - 1)) * QR_UNIT + wave * 8 * ROW_BYTES;
        if constexpr (NT) glds16_aux<2>(sg_ptr + slice * ROW_BYTES, dst);  // ONE query tile: every corpus line is read once, by one workgroup
        else glds16(sg_ptr + slice * ROW_BYTES, dst);
    };
    auto stage_next_group = [&]() {
        if (sg_group < GROUPS - 1) {
            ++sg_group;
            sg_ptr += group_step;
        } else if (sg_it + 1 < n_my) {
            ++sg_it;
            sg_group = 0;
            long long step = (long long)tile_step_bytes;
            if (ex.perm_mod > 0) {
                int nxt = super_cur + perm_inc;
                if (nxt >= ex.perm_mod) nxt -= ex.perm_mod;
                step = (long long)(nxt - super_cur) * (long long)BM * (long long)row_stride;
                super_cur = nxt;
            }
            sg_ptr += step - (long long)(GROUPS - 1) * (long long)group_step;
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;

    // prologue of the stream first: LEAD units in flight while the query fragments are fetched
    static_assert(QR_LEAD < QR_RING && NS * 2 == NK, "ring / slice geometry");
    {
        int sl = 0;
#pragma unroll 1
        for (int u = 0; u < QR_LEAD; ++u) {
            stage_slice(sl, u);
            if (++sl == NS) {
                sl = 0;
                stage_next_group();
            }
        }
    }

    // ---- the wave's queries: thresholds and ALL B fragments, resident for the whole launch
    float thr[NB16];
#pragma unroll
    for (int j = 0; j < NB16; ++j) {
        const int q = q0 + wave * TN + j * 16 + fr;
        thr[j] = q < nq ? thr_s[q] : __builtin_inff();
    }
    // (the last k-step's fragments go to LDS, lane-linear: lane l of query block j holds query row j * 16 + l / 4, 16-byte piece l % 4 -
    // exactly the fragment a lane with fr = l / 4 ... no: the FRAGMENT lane (fr, fq) reads its 16 bytes at (j * 16 + fr) * 64 + fq * 16)
    {
        char* bt = smem + QR_OPERANDS + wave * TN * 64;
#pragma unroll
        for (int j = 0; j < NB16; ++j)
            glds16((const char*)Q + (size_t)(q0 + wave * TN + j * 16 + (lane >> 2)) * row_stride + (NK - 1) * 64 + (lane & 3) * 16, bt + j * 1024);
    }
    u32x4 fb[NK - 1][NB16];
#pragma unroll
    for (int s = 0; s < NK - 1; ++s)
#pragma unroll
        for (int j = 0; j < NB16; ++j)
            fb[s][j] = *(const u32x4*)((const char*)Q + (size_t)(q0 + wave * TN + j * 16 + fr) * row_stride + s * 64 + fq * 16);
#pragma unroll
    for (int j = 0; j < NB16; ++j) asm volatile("" : "+v"(thr[j]));
    qr_wait_vmcnt<0>();  // (also the prologue units: everything issued so far has landed)
    // where the fragments live: the accumulator file takes the first QR_NA k-steps (256 registers), the vector file the rest - the MFMA
    // reads either as its B operand
#pragma unroll
    for (int s = 0; s < NK - 1; ++s)
#pragma unroll
        for (int j = 0; j < NB16; ++j) {
            if (s < QR_NA) asm volatile("" : "+a"(fb[s][j]));
            else asm volatile("" : "+v"(fb[s][j]));
        }

    // ---- per-wave survivor list (as in the persistent kernels) -------------------------------------
    key_t64* const wl_key = (key_t64*)(smem + QR_LISTS) + wave * QR_WL_CAP;
    int* const wl_q = (int*)(smem + QR_LISTS + NWAVES * QR_WL_CAP * 8) + wave * QR_WL_CAP;
    int wl_n = 0;  // wave-uniform
    auto wl_flush = [&]() {
        const int n = wl_n < QR_WL_CAP ? wl_n : QR_WL_CAP;
        constexpr int PER_LANE = QR_WL_CAP / 64;
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
        for (int u = 0; u < PER_LANE; ++u) ok[u] = ok[u] && fk[u] > thr_key[fq_[u]];
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
        qr_wait_vmcnt<0>();  // the counted waits of the unit loop must only ever see LDS-DMA pieces
    };
    auto wl_append = [&](bool p, key_t64 key, int q) {
        const unsigned long long bal = __ballot(p);
        if (bal == 0ull) return;
        const int pos = wl_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        bool direct = false;
        if (p) {
            if (pos < QR_WL_CAP) {
                wl_key[pos] = key;
                wl_q[pos] = q;
            } else {
                emit_candidate<false>(key, q, thr_key, cand, cnt, cap, overflow, ex);
                direct = true;
            }
        }
        wl_n += __builtin_popcountll(bal);
        if (__any(direct)) qr_wait_vmcnt<0>();
    };

    f32x4 acc[2][NB16];
    u32x4 fa[2][2][2];  // [buffer][k-step][row block]: the A fragments of the unit being multiplied and of the next one

    auto read_unit = [&](int unit, u32x4 (&f)[2][2]) {
        const char* base = smem + (unit & (QR_RING - 1)) * QR_UNIT + fr * ROW_BYTES;
        const int s0 = ((0 + fq) ^ swz) << 4, s1 = ((4 + fq) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f[0][i] = *(const u32x4*)(base + i * 16 * ROW_BYTES + s0);
            f[1][i] = *(const u32x4*)(base + i * 16 * ROW_BYTES + s1);
        }
    };

    // ---- epilogue of one 32-row group (the persistent kernels', FILTER mode) ----------------------
    auto epilogue = [&](int x0) {
#pragma unroll
        for (int j = 0; j < NB16; ++j) {
            __builtin_amdgcn_sched_barrier(0);  // one query block at a time: its 8 accumulator registers leave the accumulator file, are reduced, and are dead
            const int q = q0 + wave * TN + j * 16 + fr;
            float m = acc[0][j][0];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[i][j][r]);
            const bool hit = m >= thr[j];  // false for NaN and for padded queries (thr = +inf)
            if (__any(hit)) {
                int x0_o = x0, row_end_o = row_end, q_o = q;
                asm volatile("" : "+s"(x0_o), "+s"(row_end_o));
                asm volatile("" : "+v"(q_o));  // (opaque: or the per-query addresses of the cold path are hoisted into registers held across the unit loop)
                unsigned mask = 0;
                if (hit) {
#pragma unroll
                    for (int v = 0; v < 8; ++v) mask |= (acc[v >> 2][j][v & 3] >= thr[j]) ? (1u << v) : 0u;
                }
                do {
                    const bool p = mask != 0u;
                    const int b = p ? __builtin_ctz(mask) : 0;
                    mask &= mask - 1u;
                    float sc = acc[0][j][0];
#pragma unroll
                    for (int v = 1; v < 8; ++v) sc = (b == v) ? acc[v >> 2][j][v & 3] : sc;
                    const int rw = x0_o + 4 * fq + (b >> 2) * 16 + (b & 3);
                    wl_append(p && rw < row_end_o, make_key(sc, (unsigned)rw), q_o);
                } while (__any(mask != 0u));
            }
        }
        if (wl_n >= QR_WL_FLUSH) wl_flush();
    };

    __builtin_amdgcn_s_barrier();  // every wave's pieces of the prologue units have landed
    read_unit(0, fa[0]);
#ifdef QR_NO_READS
    read_unit(1, fa[1]);
#endif

    int cu = 0;                                   // the unit being multiplied (stream order)
    for (int it = 0; it < n_my; ++it) {
        const int x0 = ex.perm_mod > 0 ? super_epi * BM : row_begin + (xt0 + it * xt_step) * BM;
#pragma unroll 1
        for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
            for (int s = 0; s < NS; ++s, ++cu) {
                // k-step 0 of unit s
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NB16; ++j)
                        acc[i][j] = mfma16<DT>(fa[s & 1][0][i], fb[2 * s][j], s == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j]);
                // the next unit's fragments (landed and visible since the previous unit's barrier), the stream's next unit
#ifndef QR_NO_READS   // (QR_NO_*: timing-only ablations of qres_bench builds - results are wrong)
                read_unit(cu + 1, fa[(s + 1) & 1]);
#endif
#ifndef QR_NO_GLDS
                stage_slice((s + QR_LEAD) % NS, cu + QR_LEAD);
#endif
                if ((s + QR_LEAD + 1) % NS == 0) stage_next_group();  // (compile-time position: once per row group)
                // k-step 1 (the group's last one takes its query fragments from LDS)
                if (2 * s + 1 == NK - 1) {  // (compile-time after unrolling)
                    u32x4 ft[NB16];
                    const char* bt = smem + QR_OPERANDS + (wave * TN + fr) * 64 + fq * 16;
#pragma unroll
                    for (int j = 0; j < NB16; ++j) ft[j] = *(const u32x4*)(bt + j * 1024);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NB16; ++j) acc[i][j] = mfma16<DT>(fa[s & 1][1][i], ft[j], acc[i][j]);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NB16; ++j) acc[i][j] = mfma16<DT>(fa[s & 1][1][i], fb[2 * s + 1 < NK - 1 ? 2 * s + 1 : 0][j], acc[i][j]);
                }
                // unit cu + 2 must have landed before the barrier; the reads of unit cu + 1 retire before it too (WAR on the ring)
#ifndef QR_NO_VMWAIT
                qr_wait_vmcnt<QR_LEAD - 2>();
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef QR_NO_BARRIER
                __builtin_amdgcn_s_barrier();
#endif
            }
#ifndef QR_NO_EPI
            epilogue(x0 + g * 32);
#else
            {  // (keeps every accumulator alive at the price of 32 adds)
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NB16; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) t += acc[i][j][r];
                if (t == 12345.678f) wl_append(true, 1ull, 0);
            }
#endif
        }
        if (ex.perm_mod > 0) {
            int nxt = super_epi + perm_inc;
            if (nxt >= ex.perm_mod) nxt -= ex.perm_mod;
            super_epi = nxt;
        }
    }
    wl_flush();
}

template <int DT, int NK>
static hipError_t launch_qres_nk(const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end, int n_xtiles,
                                 int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    const int n_qtiles = (int)(nq_pad / 256);
    const int n_cu = ws.n_cu > 0 ? ws.n_cu : 256;
    const int unit = 8 * n_qtiles;
    const int total = ((n_xtiles + 7) / 8) * unit;
    int grid = (n_cu / unit) * unit;
    if (grid < unit) grid = unit;
    if (grid > total) grid = total;
    auto launch = [&](auto kern) -> hipError_t {
        if (hipError_t e = allow_dynamic_lds((const void*)kern, QR_LDS); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), QR_LDS, stream, (const uint16_t*)store, (const uint16_t*)q_pad, (int)dim_pad,
                           (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key, ws.cand, ws.cnt, (int)ws.cap,
                           ws.overflow, ws.extra);
        return hipGetLastError();
    };
    return (ws.extra.flags & FILTER_FLAG_CORPUS_NT) ? launch(mips_filter_qres_kernel<DT, NK, true>) : launch(mips_filter_qres_kernel<DT, NK, false>);
}

bool filter_qres_supports(int64_t dim_pad) { return dim_pad == 768 || dim_pad == 384; }

hipError_t launch_filter_qres(int store_dtype, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end,
                              int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    int n_xtiles = (int)((row_end - row_begin + 255) / 256);
    if (ws.extra.perm_mod > 0) row_end = ws.extra.row_bound;  // permuted stage order: whole positions, rows masked at ntotal
#define VOD_QR(D, N) \
    if (store_dtype == D && dim_pad == N * 32) return launch_qres_nk<D, N>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream);
    VOD_QR(0, 24) VOD_QR(1, 24) VOD_QR(0, 12) VOD_QR(1, 12)
#undef VOD_QR
    return hipErrorInvalidValue;
}

}  // namespace vodhip

// libvodhip experiment (round 6) -- the FILTER stage with the query tile resident in registers, TWO waves per SIMD: a K-SPLIT wave pair (tile 18).
//
// Tile 17 (kernels_mips_qres.hip) kept a wave's 64 queries x 768 k of B fragments in 384 registers, which only fits with ONE wave per SIMD -
// and one wave cannot cover its own LDS-DMA issue and barrier waits (+5..16 %, profiles/r06_ab_query_resident.txt).  Here the two waves of a pair
// share 64 queries and split the CONTRACTION: wave (p, 0) keeps the fragments of k-steps 0..NK/2-1, wave (p, 1) those of NK/2..NK-1 (192
// registers each at dim 768: fits the 256 of a two-waves-per-SIMD kernel).  A 16-row block is multiplied by (p, 0) over the first half of k; its 4
// accumulator fragments then travel through LDS (4 ds_write_b128 + 4 ds_read_b128 per 96 MFMAs) to (p, 1), which CONTINUES THE SAME accumulation
// chain over the second half and runs the FILTER epilogue - the MFMAs of a chain are the production kernels', in the same order: bit-identical.
// While (p, 1) finishes row block t - 1, (p, 0) starts row block t: both waves of a SIMD always have MFMAs to issue.  No query byte is staged or
// read from LDS: 8 instead of 12 ds_read_b128-equivalents and 2 instead of 4 LDS-DMA pieces per k-step of the production tile; the idealised loop
// (experiments/ubench/kloop, modes 63 vs 49215, N(0, 1) fp16 operands) prices that at 12.92 -> 11.96-12.02 ms per C3 batch (-7 %).
// Corpus: ring of 8 half-units of [16 rows x 384 k] (12 KB; a row block's first half is needed in step t, its second half in step t + 1), staged
// 3 steps (72 KB) ahead with 3 LDS-DMA pieces per wave and step, counted vmcnt(6), ONE s_barrier per step of 48 MFMAs per wave.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mips_common.h"

namespace vodhip {

namespace {
// timing ablations (compile-time, experiments/tools/ab_ksplit_parts.sh builds one library per bit; results are WRONG with any bit set):
//   1 no FILTER epilogue   2 no accumulator hand-off   4 no LDS-DMA in the step loop   8 no s_barrier in the step loop   16 fragments read once
#ifndef KS_ABL
#define KS_ABL 0
#endif
// k-step groups (of the NG = 6 of a half at dim 768) a wave multiplies BEFORE the barrier in the middle of its step (see first_part below)
#ifndef KS_NF
#define KS_NF 4
#endif
constexpr int KS_RING = 9;                  // half-units in the LDS ring (9: a half's staging at ITS step start never meets a slot the other half still reads)
constexpr int KS_HAND = 4 * 2 * 4096;       // accumulator hand-off: 4 pairs x 2 parities x (4 fragments x 64 lanes x 16 B)
constexpr int KS_WL_CAP = 256;              // records per wave list (the 4 second-half waves)
constexpr int KS_WL_FLUSH = 176;

// LDS-DMA hidden from hipcc (cdna_hip_programming.md 5.7): with the builtin, the compiler put an s_waitcnt vmcnt(0) in front of the first
// fragment read of every step (it cannot prove that the ds_read does not alias the pieces in flight) - the stream was drained once per step.
// M0 = the wave-uniform LDS byte address of the piece; saved and restored inside the statement (the compiler reserves M0).
// (scalar base + 32-bit per-lane offset: the lane part of a piece's address is ONE constant VGPR, everything that moves is scalar)
template <bool NT>
__device__ __forceinline__ void ks_glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// Counted waits (N > 0) only ever count LDS-DMA pieces, which hipcc does not see (inline asm): an asm statement.  The DRAINS (N = 0) close code
// hipcc does see - fragment loads, the atomics and stores of a list flush - and use the builtin (0x0F70 = vmcnt(0), expcnt / lgkmcnt untouched):
// hipcc must know those have retired, or it protects their destination registers with a vmcnt(0) of its own at the next reuse - in the step loop,
// once per step, draining the LDS-DMA stream (seen in the ISA: before the hand-off reads and before the second MFMA of a step).
template <int N>
__device__ __forceinline__ void ks_wait_vmcnt() {
    if constexpr (N == 0)
        __builtin_amdgcn_s_waitcnt(0x0F70);
    else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace

template <int DT, int NK, bool NT>
__global__ __launch_bounds__(512, 2) void mips_filter_ksplit_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end, int n_xtiles,
    int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key, key_t64* __restrict__ cand,
    unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow, FilterExtra ex) {
    constexpr int BM = 256, BN = 256, TN = 64, NB16 = TN / 16, ROW_BYTES = 128, NKH = NK / 2, NSL = NKH / 2, RB_PER_TILE = BM / 16;
    constexpr int HALF_BYTES = NSL * 16 * ROW_BYTES;     // one half-unit: NSL slices of [16 rows x 128 B]
    constexpr int PIECES = NSL * 2;                       // LDS-DMA pieces (8 rows x 128 B) per half-unit
    constexpr int PPW = PIECES / 4;                       // pieces per wave and step (4 waves stage one half-unit)
    constexpr int GS = 2, NG = NKH / GS;  // (NG = 4 at dim 768: two groups before the other half's barrier, two after)                  // A fragments are read GS k-steps at a time, one group ahead of the MFMAs
    // The step is cut UNEVENLY by its middle barrier: NF groups before, NG - NF after, NF > NG / 2.  Each half ends one of its two segments with
    // work that issues no MFMA (second half: the FILTER epilogue; first half: the hand-off stores and the counted waits); with an even cut the
    // SIMD's other wave had finished ITS MFMAs of that segment by then and sat at the barrier: the pipe idled (measured: epilogue 9 %, hand-off
    // 9 % of the batch).  With NF groups first, the other wave still has (2 NF - NG) groups of MFMAs to issue while this one does that work.
    constexpr int NF = KS_NF < NG ? KS_NF : NG;
    static_assert(NKH % GS == 0 && PIECES % 4 == 0 && NKH % 2 == 0 && NF >= 1, "geometry");
    constexpr int OPERANDS = KS_RING * HALF_BYTES, LISTS = OPERANDS + KS_HAND;
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
    const int pair = wave & 3, half = wave >> 2;  // (waves w and w + 4 sit on one SIMD: the pair)
    const int q0 = qt * BN;
    const size_t row_stride = (size_t)dim_pad * 2;
    const int n_rb = n_my * RB_PER_TILE;          // row blocks of this workgroup; steps = n_rb + 1

    // ---- the LDS-DMA stream: step t stages the half-units of step t + 3 = the SECOND half of row block t + 2 (waves 0..3) and the FIRST half of
    // row block t + 3 (waves 4..7); wave w stages pieces PPW (w & 3) .. + PPW - 1 of its half-unit (piece q = slice q / 2, rows 8 (q & 1) .. + 7).
    // 16-byte chunk c of LDS row lr sits at slot c ^ ((lr >> 1) & 7) (applied on the source address; fragment reads conflict-free).
    const int st_row = lane >> 3, st_slot = lane & 7;
    const int which = half;                       // 0: this wave stages second halves (H2), 1: first halves (H1)
    int sg_rb = which ? 0 : -1;                   // the row block whose half this wave stages next (H1 stream starts at 0, H2 at -1 = a dummy)
    int super_cur = ex.perm_mod > 0 ? filter_tile_row0(ex, row_begin, xt0, BM) / BM : 0;
    int super_epi = super_cur;
    const int perm_inc = ex.perm_mod > 0 ? (int)(((unsigned long long)xt_step * (unsigned long long)ex.perm_mul) % (unsigned long long)ex.perm_mod) : 0;
    // a piece's source = scalar base (row block, slice, row half: wave-uniform) + ONE per-lane offset: row st_row of the 8, swizzled 16-byte chunk
    // ((lr >> 1) & 7 with lr = 8 (q & 1) + st_row: the row half adds 4 to the swizzle)
    const char* sg_base = (const char*)X + (size_t)filter_tile_row0(ex, row_begin, xt0, BM) * row_stride + (size_t)(which ? 0 : NSL) * ROW_BYTES;  // scalar
    // (rows 8..15 of the block swizzle with (lr >> 1) & 7 = ((st_row >> 1) + 4) & 7: chunk ^ 4, i.e. byte offset ^ 64 - one register, not an array:
    // hipcc puts a dynamically indexed two-element array into scratch, and a scratch load in the loop drains the LDS-DMA stream)
    const unsigned sg_voff = (unsigned)(st_row * row_stride) + (unsigned)((st_slot ^ ((st_row >> 1) & 7)) << 4);
    const size_t rb_step = 16 * row_stride;
    const size_t tile_step_bytes = (size_t)xt_step * BM * row_stride;
    int issued_steps = 0;                         // steps whose half-units this wave has staged
    const unsigned lds_base = (unsigned)(uintptr_t)(VOD_AS3 char*)smem;  // LDS byte address of the dynamic segment
    auto stage_step = [&]() {
        // unit n = 2 * step + which lands in ring slot n & 7
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(((2 * issued_steps + which) % KS_RING) * HALF_BYTES));
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            const int qpc = PPW * (wave & 3) + u;  // wave-uniform
            const char* sb = sg_base + (size_t)(qpc >> 1) * ROW_BYTES + (size_t)(qpc & 1) * 8 * row_stride;
            const unsigned vo = sg_voff ^ ((unsigned)(qpc & 1) << 6);
            if ((KS_ABL & 4) == 0 || issued_steps < 3)
                ks_glds16<NT>(sb, vo, dst + (unsigned)((qpc >> 1) * (16 * ROW_BYTES) + (qpc & 1) * (8 * ROW_BYTES)));
            else
                asm volatile("" ::"v"(vo), "s"(sb));
        }
        ++issued_steps;
        // advance to the next row block (the dummy block -1 of the H2 stream and everything past the last block re-read a valid block)
        if (sg_rb >= 0 && sg_rb + 1 < n_rb) {
            long long step = (long long)rb_step;
            if ((sg_rb + 1) % RB_PER_TILE == 0) {  // next corpus tile
                step = (long long)tile_step_bytes;
                if (ex.perm_mod > 0) {
                    int nxt = super_cur + perm_inc;
                    if (nxt >= ex.perm_mod) nxt -= ex.perm_mod;
                    step = (long long)(nxt - super_cur) * (long long)BM * (long long)row_stride;
                    super_cur = nxt;
                }
                step -= (long long)(RB_PER_TILE - 1) * (long long)rb_step;
            }
            sg_base += step;
        }
        ++sg_rb;
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;

    // prologue of the stream: the half-units of steps 0, 1, 2
    stage_step();
    stage_step();
    stage_step();

    // ---- this wave's half of the query fragments, resident for the whole launch; the thresholds (second-half waves filter)
    // (thresholds live in LDS - 1 KB behind the lists - and are read per row block: the register file is full)
    float* const thr_lds = (float*)(smem + LISTS + 4 * KS_WL_CAP * 12);
    if (tid < BN) thr_lds[tid] = (q0 + tid) < nq ? thr_s[q0 + tid] : __builtin_inff();
    u32x4 fb[NKH][NB16];
#pragma unroll
    for (int s = 0; s < NKH; ++s)
#pragma unroll
        for (int j = 0; j < NB16; ++j)
            fb[s][j] = *(const u32x4*)((const char*)Q + (size_t)(q0 + pair * TN + j * 16 + fr) * row_stride + (size_t)(half * NKH + s) * 64 + fq * 16);
#pragma unroll
    for (int s = 0; s < NKH; ++s)
#pragma unroll
        for (int j = 0; j < NB16; ++j) asm volatile("" : "+v"(fb[s][j]));
    ks_wait_vmcnt<0>();  // (also the prologue half-units: everything issued so far has landed)

    // ---- per-wave survivor list of the second-half waves (as in the persistent kernels) --------------
    key_t64* const wl_key = (key_t64*)(smem + LISTS) + pair * KS_WL_CAP;
    int* const wl_q = (int*)(smem + LISTS + 4 * KS_WL_CAP * 8) + pair * KS_WL_CAP;
    int wl_n = 0;  // wave-uniform
    auto wl_flush = [&]() {
        const int n = wl_n < KS_WL_CAP ? wl_n : KS_WL_CAP;
        constexpr int PER_LANE = KS_WL_CAP / 64;
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
        ks_wait_vmcnt<0>();  // the counted waits of the step loop must only ever see LDS-DMA pieces
    };
    auto wl_append = [&](bool p, key_t64 key, int q) {
        const unsigned long long bal = __ballot(p);
        if (bal == 0ull) return;
        const int pos = wl_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        bool direct = false;
        if (p) {
            if (pos < KS_WL_CAP) {
                wl_key[pos] = key;
                wl_q[pos] = q;
            } else {
                emit_candidate<false>(key, q, thr_key, cand, cnt, cap, overflow, ex);
                direct = true;
            }
        }
        wl_n += __builtin_popcountll(bal);
        if (__any(direct)) ks_wait_vmcnt<0>();
    };

    f32x4 acc[NB16];
    u32x4 fa[2][GS];  // the A fragments of the k-step group being multiplied and of the next one

    auto read_group = [&](const char* unit, int g, u32x4 (&f)[GS]) {
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            const int s = g * GS + u;  // k-step inside the half: slice s / 2, 32-deep half s & 1
            if constexpr ((KS_ABL & 16) != 0)
                asm volatile("" : "+v"(f[u]));
            else
                f[u] = *(const u32x4*)(unit + (s >> 1) * (16 * ROW_BYTES) + fr * ROW_BYTES + ((((s & 1) * 4 + fq) ^ swz) << 4));
        }
    };
    if constexpr ((KS_ABL & 16) != 0) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int u = 0; u < GS; ++u) fa[b][u] = *(const u32x4*)(smem + fr * ROW_BYTES + ((fq ^ swz) << 4));
    }
    auto ks_step_barrier = [&]() {
        if constexpr ((KS_ABL & 8) == 0) __builtin_amdgcn_s_barrier();
    };

    // ---- epilogue of one 16-row block (second-half waves; the persistent kernels', FILTER mode) ------
    auto epilogue = [&](int x0) {
        if constexpr ((KS_ABL & 1) != 0) {  // keep the accumulators alive, filter nothing
            float keep = 0.f;
#pragma unroll
            for (int j = 0; j < NB16; ++j) keep += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
            if (keep == 12345.678f) wl_append(true, make_key(keep, 0u), q0);
            return;
        }
#pragma unroll
        for (int j = 0; j < NB16; ++j) {
            const int q = q0 + pair * TN + j * 16 + fr;
            const float m = fmaxf(fmaxf(acc[j][0], acc[j][1]), fmaxf(acc[j][2], acc[j][3]));
            const float thr_j = thr_lds[pair * TN + j * 16 + fr];
            const bool hit = m >= thr_j;  // false for NaN and for padded queries (thr = +inf)
            if (__any(hit)) {
                int x0_o = x0, row_end_o = row_end, q_o = q;
                asm volatile("" : "+s"(x0_o), "+s"(row_end_o));
                asm volatile("" : "+v"(q_o));  // (opaque: or the per-query addresses of the cold path are hoisted into registers held across the loop)
                unsigned mask = 0;
                if (hit) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) mask |= (acc[j][v] >= thr_j) ? (1u << v) : 0u;
                }
                do {
                    const bool p = mask != 0u;
                    const int b = p ? __builtin_ctz(mask) : 0;
                    mask &= mask - 1u;
                    float sc = acc[j][0];
#pragma unroll
                    for (int v = 1; v < 4; ++v) sc = (b == v) ? acc[j][v] : sc;
                    const int rw = x0_o + 4 * fq + b;
                    wl_append(p && rw < row_end_o, make_key(sc, (unsigned)rw), q_o);
                } while (__any(mask != 0u));
            }
        }
        if (wl_n >= KS_WL_FLUSH) wl_flush();
    };

    __builtin_amdgcn_s_barrier();  // every wave's pieces of the prologue half-units have landed

    // hand-off buffers of this pair: [parity][fragment j][lane] 16 B
    char* const hand = smem + OPERANDS + pair * (2 * 4096) + lane * 16;
    int x0_tile = ex.perm_mod > 0 ? super_epi * BM : row_begin + xt0 * BM;  // first row of the corpus tile the second-half wave is filtering
    int epi_rb = 0;                                                          // ... and its row block inside that tile
    int epi_it = 0;
    // The two halves run HALF A STEP APART (the 8-phase template's stagger: on every SIMD the step-boundary work of one wave - hand-off, first
    // fragment reads, LDS-DMA issue, epilogue, waits - sits beside the MFMAs of the other).  One loop iteration = two barriers b1, b2:
    //   first-half wave:   [first part of step i]            b1   [second part of step i, hand-off out]        b2 = ITS boundary
    //   second-half wave:  [second part of step i - 1, epilogue]  b1 = ITS boundary   [first part of step i]   b2
    auto first_part = [&](int t) {  // step start: the accumulators, the first fragment groups, this step's LDS-DMA pieces, NG / 2 groups of MFMAs
        const char* unit = smem + ((2 * t + (half ? 0 : 1)) % KS_RING) * HALF_BYTES;  // first half reads H1(t) = unit 2 t + 1, second half H2(t - 1) = unit 2 t
        if (half) {
            // the partial sums of row block t - 1 from the pair's first-half wave (written before the barrier that ended ITS step t - 1)
            const char* hb = hand + ((t - 1) & 1) * 4096;
            if constexpr ((KS_ABL & 2) != 0) {
#pragma unroll
                for (int j = 0; j < NB16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int j = 0; j < NB16; ++j) acc[j] = *(const f32x4*)(hb + j * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        read_group(unit, 0, fa[0]);
        stage_step();
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            if (g + 1 < NG) read_group(unit, g + 1, fa[(g + 1) & 1]);
#pragma unroll
            for (int v = 0; v < GS; ++v)
#pragma unroll
                for (int j = 0; j < NB16; ++j) acc[j] = mfma16<DT>(fa[g & 1][v], fb[g * GS + v][j], acc[j]);
        }
    };
    auto second_part = [&](int t) {
        const char* unit = smem + ((2 * t + (half ? 0 : 1)) % KS_RING) * HALF_BYTES;
#pragma unroll
        for (int g = NF; g < NG; ++g) {
            if (g + 1 < NG) read_group(unit, g + 1, fa[(g + 1) & 1]);
#pragma unroll
            for (int v = 0; v < GS; ++v)
#pragma unroll
                for (int j = 0; j < NB16; ++j) acc[j] = mfma16<DT>(fa[g & 1][v], fb[g * GS + v][j], acc[j]);
        }
    };
    // THIS half's step boundary.  First-half waves stage the second halves of row blocks whose reader's step starts half a step AFTER the
    // barrier two steps on: the pieces of the last two steps may stay in flight (vmcnt(2 PPW)); second-half waves stage first halves whose
    // reader starts half a step BEFORE the next boundary: only the last step's pieces may stay in flight.  The step's fragment reads and
    // hand-off accesses retire here too (WAR on the ring and on the hand-off buffers).
    if (half == 0) {
        for (int i = 0; i <= n_rb; ++i) {
            if (i < n_rb) first_part(i);
            ks_step_barrier();  // b1: the second half's boundary
            if (i < n_rb) {
                second_part(i);
                char* hb = hand + (i & 1) * 4096;
                if constexpr ((KS_ABL & 2) != 0) {
#pragma unroll
                    for (int j = 0; j < NB16; ++j) asm volatile("" ::"v"(acc[j]));
                } else {
#pragma unroll
                    for (int j = 0; j < NB16; ++j) *(f32x4*)(hb + j * 1024) = acc[j];
                }
            }
            ks_wait_vmcnt<2 * PPW>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ks_step_barrier();  // b2
        }
    } else {
        for (int i = 0; i <= n_rb; ++i) {
            if (i >= 1) {  // (step i - 1 of this half works on row block i - 2: nothing to filter in its very first step)
                second_part(i - 1);
                if (i >= 2) {
                    epilogue(x0_tile + epi_rb * 16);
                    if (++epi_rb == RB_PER_TILE) {  // the next row block opens the workgroup's next corpus tile
                        epi_rb = 0;
                        ++epi_it;
                        if (ex.perm_mod > 0) {
                            int nxt = super_epi + perm_inc;
                            if (nxt >= ex.perm_mod) nxt -= ex.perm_mod;
                            super_epi = nxt;
                            x0_tile = super_epi * BM;
                        } else {
                            x0_tile = row_begin + (xt0 + epi_it * xt_step) * BM;
                        }
                    }
                }
            }
            ks_wait_vmcnt<PPW>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ks_step_barrier();  // b1
            first_part(i);                 // (step n_rb of this half works on the last row block; at i = 0 on the dummy block -1: nothing is filtered)
            ks_step_barrier();  // b2: the first half's boundary
        }
        // the last row block: second part of step n_rb + its epilogue (no barrier follows: the first-half waves are done)
        second_part(n_rb);
        epilogue(x0_tile + epi_rb * 16);
    }
    if (half) wl_flush();
}

template <int DT, int NK>
static hipError_t launch_ksplit_nk(const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end, int n_xtiles,
                                   int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    const int n_qtiles = (int)(nq_pad / 256);
    const int n_cu = ws.n_cu > 0 ? ws.n_cu : 256;
    const int unit = 8 * n_qtiles;
    const int total = ((n_xtiles + 7) / 8) * unit;
    int grid = (n_cu / unit) * unit;
    if (grid < unit) grid = unit;
    if (grid > total) grid = total;
    constexpr int lds = KS_RING * (NK / 4) * 16 * 128 + KS_HAND + 4 * KS_WL_CAP * 12 + 1024;  // 9 x 12 KB + 32 KB + 12 KB = 152 KB at dim 768
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto launch = [&](auto kern) -> hipError_t {
        if (hipError_t e = allow_dynamic_lds((const void*)kern, lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, stream, (const uint16_t*)store, (const uint16_t*)q_pad, (int)dim_pad,
                           (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key, ws.cand, ws.cnt, (int)ws.cap,
                           ws.overflow, ws.extra);
        return hipGetLastError();
    };
    return (ws.extra.flags & FILTER_FLAG_CORPUS_NT) ? launch(mips_filter_ksplit_kernel<DT, NK, true>) : launch(mips_filter_ksplit_kernel<DT, NK, false>);
}

bool filter_ksplit_supports(int64_t dim_pad) { return dim_pad == 768; }

hipError_t launch_filter_ksplit(int store_dtype, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end,
                                int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream) {
    int n_xtiles = (int)((row_end - row_begin + 255) / 256);
    if (ws.extra.perm_mod > 0) row_end = ws.extra.row_bound;  // permuted stage order: whole positions, rows masked at ntotal
    if (dim_pad != 768) return hipErrorInvalidValue;
    return store_dtype == 0 ? launch_ksplit_nk<0, 24>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream)
                            : launch_ksplit_nk<1, 24>(store, q_pad, dim_pad, row_begin, row_end, n_xtiles, nq, nq_pad, ws, stream);
}

}  // namespace vodhip

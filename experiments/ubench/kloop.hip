// Micro-benchmark, not product code: the K loop of the persistent 256x256 search kernel rebuilt from its parts, to see what
// each part costs in the best schedule the hardware allows (no epilogue, results meaningless).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/ubench/kloop tools/ubench/kloop.hip
//   tools/ubench/kloop <mode bits> [k-steps per workgroup] [repeats]
// One workgroup of 8 waves per CU (2 per SIMD), wave tile 128 x 64 of v_mfma_f32_16x16x32_f16: per k-step (32 deep) a wave
// issues 12 ds_read_b128, 32 MFMAs and 4 LDS-DMA pieces of 1 KiB.  Mode bits:
//   1 MFMA   2 ds_read   4 reads prefetched one k-step ahead (else read -> wait -> multiply)   8 LDS-DMA
//   16 corpus pieces stream through HBM (else: from 16 L2-hot tiles)   32 vmcnt(0) + s_barrier every 2 k-steps
//   64 the two waves of a SIMD run half a k-step apart (waves 4..7 start with 16 extra MFMAs)
//   128 "deep" ring (implies prefetched reads): corpus in 3 slots of 64-deep slices fetched ~2 slices ahead, queries in 3 slots
//       of 32-deep half slices fetched 3 k-steps ahead, one vmcnt(6) + s_barrier per k-step (mode bit 32), 144 KB of LDS;
//       without bit 4 the fragments of a k-step are read right after its barrier
//   512 (deep ring) the barrier only in front of every other k-step (TIMING ONLY: the ring is not safe like that)
//   1024 (two-slot loop) the production kernel's order inside a k-step: reads B + A[0..3], 2 pieces, reads A[4..7], 16 MFMAs,
//        2 pieces, 16 MFMAs
//   8192 a pause at every tile end (24 k-steps) that differs between workgroups: ~2,000 + (hash % 2,048) cycles, what the epilogue
//        and its survivors do to the four workgroups that share a corpus tile through L2
//   256 ONE query tile: every workgroup streams its own corpus tiles (nothing shared through L2), corpus pieces with nt
//   16384 (two-slot loop, round 6) NO query bytes on-chip: the B fragments are made up once, no query pieces are staged and none are read -
//        what ANY query-resident design could save at full (two waves per SIMD) MFMA issue: 8 instead of 12 ds_read_b128 and 2 instead of 4
//        LDS-DMA pieces per k-step
//   32768 (with 16384) + the accumulator hand-off of a K-SPLIT wave pair (each wave of a SIMD keeps the fragments of HALF the contraction and the
//        partial sums of a 16-row block pass from one to the other through LDS: 8 b128 accesses per 96 MFMAs): 3 ds_write_b128 + 3 ds_read_b128
//        every 2 k-steps
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

constexpr int ROW_BYTES = 128, A_BYTES = 256 * ROW_BYTES, STAGE_BYTES = 2 * A_BYTES;
constexpr int DIM_BYTES = 1536, NK = 12;  // 768 fp16 per row, 12 slices of 64


__device__ __forceinline__ void tile_end_pause(int bid, int tile_no) {
    unsigned h = (unsigned)(bid * 2654435761u) ^ (unsigned)(tile_no * 40503u);
    h ^= h >> 13;
    const unsigned long long until = __builtin_amdgcn_s_memtime() + 2000ull + (h & 2047u);
    while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(2);
}

struct Frags {
    u32x4 b[4], a[8];
};

template <int MODE>
__global__ __launch_bounds__(512, 2) void kloop(const char* __restrict__ X, const char* __restrict__ Q, float* __restrict__ out,
                                                int ksteps, int n_xtiles, unsigned long long* __restrict__ clk) {
    constexpr bool MFMA = MODE & 1, READ = (MODE & 2) != 0, PIPE = (MODE & 4) != 0, DMA = (MODE & 8) != 0, STREAM = (MODE & 16) != 0,
                   BAR = (MODE & 32) != 0, SKEW = (MODE & 64) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    constexpr bool ONEQ = (MODE & 256) != 0, NOQ = (MODE & 16384) != 0, HAND = (MODE & 32768) != 0;
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3, qt = ONEQ ? 0 : (jj & 3);
    const int xt0 = ONEQ ? bid : (jj >> 2) * 8 + xcd, xt_step = ONEQ ? (int)gridDim.x : (int)gridDim.x / 4;
    // fill the LDS with finite fp16 values (0.125 .. 0.25, random sign)
    for (int e = tid; e < 2 * STAGE_BYTES / 4; e += 512) {
        unsigned h = (unsigned)e * 2654435761u;
        ((unsigned*)smem)[e] = 0x30003000u | (h & 0x8fff8fffu);
    }
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4, swz = (fr >> 1) & 7;
    const int a_off = (wm * 128 + fr) * ROW_BYTES, b_off = A_BYTES + (wn * 64 + fr) * ROW_BYTES;
    const int st_row = lane >> 3, st_slot = lane & 7;
    const char* a_src[4];
    const char* b_src[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = (wave * 4 + t) * 8 + st_row;
        const int xt = STREAM ? xt0 : (xt0 & 15);
        a_src[t] = X + ((size_t)xt * 256 + r) * DIM_BYTES + (st_slot ^ ((r >> 1) & 7)) * 16;
        b_src[t] = Q + ((size_t)qt * 256 + r) * DIM_BYTES + (st_slot ^ ((r >> 1) & 7)) * 16;
    }
    const size_t tile_step = STREAM ? (size_t)xt_step * 256 * DIM_BYTES : 0;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto read_frags = [&](Frags& f, int slot, int ks) {
        const char* base = smem + slot * STAGE_BYTES;
        const int so = ((4 * ks + fq) ^ swz) << 4;
        if constexpr (!NOQ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) f.b[j] = *(const u32x4*)(base + b_off + j * 16 * ROW_BYTES + so);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) f.a[i] = *(const u32x4*)(base + a_off + i * 16 * ROW_BYTES + so);
    };
    auto mma = [&](const Frags& f, int i0, int i1) {
#ifdef KLOOP_JMAJOR  // consecutive MFMAs share the B (query) fragment instead of the A (corpus) fragment
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i >= i0 && i < i1) {
                    if constexpr (MFMA)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f.a[i]), __builtin_bit_cast(f16x8, f.b[j]), acc[i][j], 0, 0, 0);
                    else
                        asm volatile("" ::"v"(f.a[i]), "v"(f.b[j]));
                }
        __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i >= i0 && i < i1) {
                    if constexpr (MFMA)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f.a[i]), __builtin_bit_cast(f16x8, f.b[j]), acc[i][j], 0, 0, 0);
                    else
                        asm volatile("" ::"v"(f.a[i]), "v"(f.b[j]));
                }
#endif
    };
    int kbyte = 0, t_in_tile = 0;
    auto dma = [&](int slot, int piece) {  // piece 0..7 of the slice that lands in `slot`: 4 corpus + 4 query
        if constexpr (DMA) {
            char* sa = smem + slot * STAGE_BYTES;
            if (piece < 4)
                __builtin_amdgcn_global_load_lds((const AS1 void*)(a_src[piece] + kbyte), (AS3 void*)(sa + (wave * 4 + piece) * 8 * ROW_BYTES), 16, 0, ONEQ ? 2 : 0);
            else if constexpr (!NOQ)
                __builtin_amdgcn_global_load_lds((const AS1 void*)(b_src[piece - 4] + kbyte), (AS3 void*)(sa + A_BYTES + (wave * 4 + piece - 4) * 8 * ROW_BYTES), 16, 0, 0);
        }
    };
    auto next_slice = [&]() {  // advance the DMA source to the next 64-deep slice (next tile after 12)
        kbyte += ROW_BYTES;
        if (++t_in_tile == NK) {
            t_in_tile = 0;
            kbyte = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) a_src[t] += tile_step;
        }
    };
    auto sync = [&]() {
        if constexpr (BAR) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    };
    Frags f0, f1;
    if constexpr (NOQ) {  // the "resident" query fragments: random-looking fp16 bit patterns, fixed for the launch
#pragma unroll
        for (int j = 0; j < 4; ++j) f0.b[j] = u32x4{0x30003000u + lane * 0x01030507u, 0xb1003100u ^ (lane * 0x00110013u), 0x32003200u + j * 0x00770031u, 0xb3003300u ^ (lane << 3)};
#pragma unroll
        for (int j = 0; j < 4; ++j) f1.b[j] = f0.b[j];
    }
    if constexpr (!READ) {  // fragments made up once
#pragma unroll
        for (int j = 0; j < 4; ++j) f0.b[j] = u32x4{0x30003000u + lane, 0x31003100u, 0x32003200u, 0x33003300u + j};
#pragma unroll
        for (int i = 0; i < 8; ++i) f0.a[i] = u32x4{0x30003000u + i, 0x31003100u + lane, 0xb200b200u, 0x33003300u};
        f1 = f0;
    }
    if constexpr (SKEW) {
        if (wave >= 4) {
            if constexpr (READ) read_frags(f1, 0, 0);
            mma(f1, 0, 4);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int g = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (READ && PIPE) read_frags(f0, 0, 0);
    for (int it = 0; it < ksteps; it += 2, ++g) {
        const int slot = g & 1, nslot = slot ^ 1;
        if constexpr (PIPE) {
            // k-step 0: multiply f0 while f1 (k-step 1) is read
            mma(f0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (READ) read_frags(f1, slot, 1);
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the reads to the end of the k-step)
            dma(nslot, 0);
            mma(f0, 1, 3);
            dma(nslot, 1);
            mma(f0, 3, 5);
            dma(nslot, 2);
            mma(f0, 5, 7);
            dma(nslot, 3);
            mma(f0, 7, 8);
            __builtin_amdgcn_sched_barrier(0);
            // k-step 1: multiply f1 while f0 (k-step 0 of the next slice) is read; the barrier sits before the reads
            mma(f1, 0, 1);
            dma(nslot, 4);
            mma(f1, 1, 3);
            dma(nslot, 5);
            mma(f1, 3, 4);
            __builtin_amdgcn_sched_barrier(0);
            sync();
            if constexpr (READ) read_frags(f0, nslot, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(f1, 4, 6);
            dma(slot, 6);  // (the last two pieces of the following slice go to the slot just released)
            mma(f1, 6, 7);
            dma(slot, 7);
            mma(f1, 7, 8);
            if constexpr (HAND) {  // (the accumulators really travel: written, then read back into the accumulation chain)
                char* hb = smem + 2 * STAGE_BYTES + (wave * 64 + lane) * 16 * 3;
#pragma unroll
                for (int u = 0; u < 3; ++u) *(f32x4*)(hb + u * 16) = acc[u][0];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 3; ++u) acc[u + 3][1] += *(const f32x4*)(hb + u * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            sync();
            if constexpr ((MODE & 1024) != 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const char* base = smem + slot * STAGE_BYTES;
                    const int so = ((4 * ks + fq) ^ swz) << 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) f0.b[j] = *(const u32x4*)(base + b_off + j * 16 * ROW_BYTES + so);
#pragma unroll
                    for (int i = 0; i < 4; ++i) f0.a[i] = *(const u32x4*)(base + a_off + i * 16 * ROW_BYTES + so);
                    dma(nslot, 4 * ks + 0);
                    dma(nslot, 4 * ks + 1);
#pragma unroll
                    for (int i = 4; i < 8; ++i) f0.a[i] = *(const u32x4*)(base + a_off + i * 16 * ROW_BYTES + so);
                    mma(f0, 0, 4);
                    dma(nslot, 4 * ks + 2);
                    dma(nslot, 4 * ks + 3);
                    mma(f0, 4, 8);
                    __builtin_amdgcn_sched_barrier(0);
                }
                next_slice();
                if constexpr ((MODE & 8192) != 0) { if (t_in_tile == 0) tile_end_pause(bid, g); }
                continue;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (READ) read_frags(f0, slot, ks);
                dma(nslot, 4 * ks + 0);
                dma(nslot, 4 * ks + 1);
                mma(f0, 0, 4);
                dma(nslot, 4 * ks + 2);
                dma(nslot, 4 * ks + 3);
                mma(f0, 4, 8);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        next_slice();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[2 * bid] = t1 - t0; clk[2 * bid + 1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if constexpr (!MFMA) s += __builtin_bit_cast(float, f0.a[0][0]) + __builtin_bit_cast(float, f1.b[0][0]);
    out[(size_t)bid * 512 + tid] = s;
}


constexpr int C_SLOT = 256 * 128, Q_SLOT = 256 * 64, Q_BASE = 3 * C_SLOT;
template <int MODE>
__global__ __launch_bounds__(512, 2) void kdeep(const char* __restrict__ X, const char* __restrict__ Q, float* __restrict__ out,
                                                int ksteps, int n_xtiles, unsigned long long* __restrict__ clk) {
    constexpr bool MFMA = MODE & 1, PIPE = (MODE & 4) != 0, DMA = (MODE & 8) != 0, STREAM = (MODE & 16) != 0, BAR = (MODE & 32) != 0,
                   ONEQ = (MODE & 256) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3, qt = ONEQ ? 0 : (jj & 3);
    const int xt0 = ONEQ ? bid : (jj >> 2) * 8 + xcd, xt_step = ONEQ ? (int)gridDim.x : (int)gridDim.x / 4;
    for (int e = tid; e < (3 * C_SLOT + 3 * Q_SLOT) / 4; e += 512) {
        unsigned h = (unsigned)e * 2654435761u;
        ((unsigned*)smem)[e] = 0x30003000u | (h & 0x8fff8fffu);
    }
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4, swz = (fr >> 1) & 7;
    const int a_off = (wm * 128 + fr) * 128;
    // query half slices: 64-byte rows; 16-byte chunk c of row r sits at physical chunk c ^ g((r >> 2) & 3), g = (0, 2, 3, 1)
    const int gq = (0x78 >> (2 * ((fr >> 2) & 3))) & 3;
    const int b_off = Q_BASE + (wn * 64 + fr) * 64 + ((fq ^ gq) << 4);
    const char* a_src[4];
    const char* b_src[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = (wave * 4 + t) * 8 + (lane >> 3);
        const int xt = STREAM ? xt0 : (xt0 & 15);
        a_src[t] = X + ((size_t)xt * 256 + r) * DIM_BYTES + ((lane & 7) ^ ((r >> 1) & 7)) * 16;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = (wave * 2 + t) * 16 + (lane >> 2);
        const int g = (0x78 >> (2 * ((r >> 2) & 3))) & 3;
        b_src[t] = Q + ((size_t)qt * 256 + r) * DIM_BYTES + ((lane & 3) ^ g) * 16;
    }
    const size_t tile_step = STREAM ? (size_t)xt_step * 256 * DIM_BYTES : 0;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto read_frags = [&](Frags& f, int cslot, int qslot, int ks) {
        const char* cb = smem + cslot * C_SLOT;
        const char* qb = smem + qslot * Q_SLOT;
        const int so = ((4 * ks + fq) ^ swz) << 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = *(const u32x4*)(qb + b_off + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) f.a[i] = *(const u32x4*)(cb + a_off + i * 16 * 128 + so);
    };
    auto mma = [&](const Frags& f, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i >= i0 && i < i1) {
                    if constexpr (MFMA)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f.a[i]), __builtin_bit_cast(f16x8, f.b[j]), acc[i][j], 0, 0, 0);
                    else
                        asm volatile("" ::"v"(f.a[i]), "v"(f.b[j]));
                }
    };
    // corpus fetch cursor (slice s_c, 4 pieces over two k-steps) and query fetch cursor (half slice)
    int c_kbyte = 0, c_t = 0, c_slot = 0, q_kbyte = 0, q_slot = 0;
    auto dma_c = [&](int piece) {
        if constexpr (DMA)
            __builtin_amdgcn_global_load_lds((const AS1 void*)(a_src[piece] + c_kbyte), (AS3 void*)(smem + c_slot * C_SLOT + (wave * 4 + piece) * 8 * 128), 16, 0, ONEQ ? 2 : 0);
    };
    auto next_c = [&]() {
        c_kbyte += 128;
        c_slot = c_slot == 2 ? 0 : c_slot + 1;
        if (++c_t == NK) {
            c_t = 0;
            c_kbyte = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) a_src[t] += tile_step;
        }
    };
    auto dma_q = [&]() {
        if constexpr (DMA) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                __builtin_amdgcn_global_load_lds((const AS1 void*)(b_src[t] + q_kbyte), (AS3 void*)(smem + Q_BASE + q_slot * Q_SLOT + (wave * 2 + t) * 16 * 64), 16, 0, 0);
        }
        q_kbyte += 64;
        if (q_kbyte == DIM_BYTES) q_kbyte = 0;
        q_slot = q_slot == 2 ? 0 : q_slot + 1;
    };
    // prologue: corpus slices 0, 1 and query half slices 0, 1, 2 in flight; then the steady state issues corpus slice g+2 over
    // k-steps (2g-1, 2g) and query half h+3 in k-step h
    dma_c(0); dma_c(1); dma_c(2); dma_c(3); next_c();
    dma_c(0); dma_c(1); dma_c(2); dma_c(3); next_c();
    dma_q(); dma_q(); dma_q();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    Frags f0, f1;
    read_frags(f0, 0, 0, 0);
    int rc = 0, rq = 1;  // slots of the NEXT fragments to read: corpus slot of k-step h+1, query slot of k-step h+1
    auto kstep = [&](Frags& cur, Frags& nxt, int ks_next, bool corpus_pair_first) {
        // sync S_h: everything k-step h+1 reads has landed in every wave, and every wave holds k-step h's fragments
        if (BAR && (!(MODE & 512) || corpus_pair_first)) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (PIPE) {
            mma(cur, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nxt, rc, rq, ks_next);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            read_frags(cur, rc, rq, ks_next);  // (timing only: same slots one k-step later)
            __builtin_amdgcn_sched_barrier(0);
            mma(cur, 0, 1);
        }
        dma_q();
        mma(cur, 1, 4);
        dma_c(corpus_pair_first ? 0 : 2);
        mma(cur, 4, 6);
        dma_c(corpus_pair_first ? 1 : 3);
        mma(cur, 6, 8);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int h = 0; h < ksteps; h += 2) {
        // k-step h (ks = 0 of slice g): next fragments = (g, ks 1): same corpus slot, next query slot; corpus pieces 2, 3 of slice g+2
        kstep(f0, f1, 1, false);
        next_c();
        rq = rq == 2 ? 0 : rq + 1;
        // k-step h+1 (ks = 1): next fragments = (g+1, ks 0): next corpus slot; corpus pieces 0, 1 of slice g+3
        rc = rc == 2 ? 0 : rc + 1;
        kstep(f1, f0, 0, true);
        rq = rq == 2 ? 0 : rq + 1;
        if constexpr ((MODE & 8192) != 0) { if (((h >> 1) + 1) % NK == 0) tile_end_pause(bid, h); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[2 * bid] = t1 - t0; clk[2 * bid + 1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if constexpr (!MFMA) s += __builtin_bit_cast(float, f0.a[0][0]) + __builtin_bit_cast(float, f1.b[0][0]);
    out[(size_t)bid * 512 + tid] = s;
}


// ---- mode bit 2048: a 384 (corpus) x 256 (queries) workgroup tile, 8 waves = 4 x 2 of 96 x 128 (192 accumulator registers):
// 22 % fewer LDS read bytes and 17 % fewer LDS-DMA bytes per flop than 128 x 64 wave tiles.  Two 80 KB slots, the production
// order inside a k-step (reads B + half of A, pieces, rest of A, MFMAs, pieces, MFMAs), hand-issued SGPR-base LDS-DMA.
template <bool NT>
__device__ __forceinline__ void glds16_saddr(const void* sbase, unsigned voff, unsigned lds_addr) {
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
constexpr int W_A_BYTES = 384 * 128, W_STAGE = W_A_BYTES + 256 * 128;
template <int MODE>
__global__ __launch_bounds__(512, 2) void kwide(const char* __restrict__ X, const char* __restrict__ Q, float* __restrict__ out,
                                                int ksteps, int n_xtiles, unsigned long long* __restrict__ clk) {
    constexpr bool BAR = (MODE & 32) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3, qt = jj & 3;
    const int xt0 = (jj >> 2) * 8 + xcd, xt_step = (int)gridDim.x / 4;
    for (int e = tid; e < 2 * W_STAGE / 4; e += 512) {
        unsigned h = (unsigned)e * 2654435761u;
        ((unsigned*)smem)[e] = 0x30003000u | (h & 0x8fff8fffu);
    }
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4, swz = (fr >> 1) & 7;
    const int a_off = (wm * 96 + fr) * 128, b_off = W_A_BYTES + (wn * 128 + fr) * 128;
    // pieces: corpus 48 per slice (6 per wave: rows (wave*6 + t)*8 ..), queries 32 (4 per wave)
    const char* c_base = X + ((size_t)xt0 * 384 + (size_t)wave * 48) * DIM_BYTES;
    const char* q_base = Q + ((size_t)qt * 256 + (size_t)wave * 32) * DIM_BYTES;
    unsigned a_voff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) a_voff[par] = (unsigned)(lane >> 3) * DIM_BYTES + (unsigned)(((lane & 7) ^ ((4 * par + (lane >> 4)) & 7)) << 4);
    const size_t tile_step = (size_t)xt_step * 384 * DIM_BYTES;
    f32x4 acc[6][8];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kbyte = 0, t_in_tile = 0;
    auto dma = [&](int slot, int piece) {  // 0..5 corpus, 6..9 queries
        const unsigned sa = (unsigned)(slot * W_STAGE);
        if (piece < 6) {
            // (wave*6 + piece)*8 rows: ((r >> 1) & 7) = ((wave*6 + piece)*4 + (lane >> 4)) & 7 -> parity of (wave*6 + piece) selects the offset
            glds16_saddr<false>(c_base + kbyte + piece * 8 * DIM_BYTES, a_voff[(wave * 6 + piece) & 1], sa + (wave * 6 + piece) * 8 * 128);
        } else {
            const int t = piece - 6;
            glds16_saddr<false>(q_base + kbyte + t * 8 * DIM_BYTES, a_voff[(wave * 4 + t) & 1], sa + W_A_BYTES + (wave * 4 + t) * 8 * 128);
        }
    };
    u32x4 fa[6], fb[8];
    auto mma = [&](int i0, int i1) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (i >= i0 && i < i1)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[j]), acc[i][j], 0, 0, 0);
    };
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int g = 0;
    for (int it = 0; it < ksteps; it += 2, ++g) {
        const int slot = g & 1, nslot = slot ^ 1;
        if constexpr (BAR) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* base = smem + slot * W_STAGE;
            const int so = ((4 * ks + fq) ^ swz) << 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[j] = *(const u32x4*)(base + b_off + j * 16 * 128 + so);
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] = *(const u32x4*)(base + a_off + i * 16 * 128 + so);
            dma(nslot, 5 * ks + 0);
            dma(nslot, 5 * ks + 1);
#pragma unroll
            for (int i = 3; i < 6; ++i) fa[i] = *(const u32x4*)(base + a_off + i * 16 * 128 + so);
            mma(0, 3);
            dma(nslot, 5 * ks + 2);
            dma(nslot, 5 * ks + 3);
            dma(nslot, 5 * ks + 4);
            mma(3, 6);
            __builtin_amdgcn_sched_barrier(0);
        }
        kbyte += 128;
        if (++t_in_tile == NK) {
            t_in_tile = 0;
            kbyte = 0;
            c_base += tile_step;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[2 * bid] = t1 - t0; clk[2 * bid + 1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)bid * 512 + tid] = s;
}


// ---- mode bit 4096: the same 384 x 256 tile on a ring of THREE 32-deep slots (corpus 384 x 64 B + queries 256 x 64 B = 40 KB
// each: 120 KB, which leaves room for the survivor lists), pieces of 16 rows x 64 B requested two k-steps ahead, one
// vmcnt(5) + s_barrier per k-step.
constexpr int W3_A = 384 * 64, W3_SLOT = W3_A + 256 * 64;
template <int MODE>
__global__ __launch_bounds__(512, 2) void kwide3(const char* __restrict__ X, const char* __restrict__ Q, float* __restrict__ out,
                                                 int ksteps, int n_xtiles, unsigned long long* __restrict__ clk) {
    constexpr bool BAR = (MODE & 32) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3, qt = jj & 3;
    const int xt0 = (jj >> 2) * 8 + xcd, xt_step = (int)gridDim.x / 4;
    for (int e = tid; e < 3 * W3_SLOT / 4; e += 512) {
        unsigned h = (unsigned)e * 2654435761u;
        ((unsigned*)smem)[e] = 0x30003000u | (h & 0x8fff8fffu);
    }
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    const int gq = (0x78 >> (2 * ((fr >> 2) & 3))) & 3;
    const int a_off = (wm * 96 + fr) * 64 + ((fq ^ gq) << 4), b_off = W3_A + (wn * 128 + fr) * 64 + ((fq ^ gq) << 4);
    const char* c_base = X + ((size_t)xt0 * 384 + (size_t)wave * 48) * DIM_BYTES;
    const char* q_base = Q + ((size_t)qt * 256 + (size_t)wave * 32) * DIM_BYTES;
    const unsigned voff = (unsigned)(lane >> 2) * DIM_BYTES + (unsigned)(((lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3)) << 4);
    const size_t tile_step = (size_t)xt_step * 384 * DIM_BYTES;
    f32x4 acc[6][8];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kbyte = 0, t_in_tile = 0, f_slot = 0;  // fetch cursor
    auto dma = [&](int piece) {  // 0..2 corpus, 3..4 queries, of the k-step under the fetch cursor
        const unsigned sa = (unsigned)(f_slot * W3_SLOT);
        if (piece < 3) glds16_saddr<false>(c_base + kbyte + piece * 16 * DIM_BYTES, voff, sa + (wave * 3 + piece) * 16 * 64);
        else glds16_saddr<false>(q_base + kbyte + (piece - 3) * 16 * DIM_BYTES, voff, sa + W3_A + (wave * 2 + piece - 3) * 16 * 64);
    };
    auto next_fetch = [&]() {
        kbyte += 64;
        f_slot = f_slot == 2 ? 0 : f_slot + 1;
        if (++t_in_tile == 2 * NK) {
            t_in_tile = 0;
            kbyte = 0;
            c_base += tile_step;
        }
    };
    u32x4 fa[6], fb[8];
    auto mma = [&](int i0, int i1) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (i >= i0 && i < i1)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[j]), acc[i][j], 0, 0, 0);
    };
    for (int p = 0; p < 5; ++p) dma(p);
    next_fetch();
    for (int p = 0; p < 5; ++p) dma(p);
    next_fetch();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int r_slot = 0;
    for (int h = 0; h < ksteps; ++h) {
        if constexpr (BAR) {
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const char* base = smem + r_slot * W3_SLOT;
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[j] = *(const u32x4*)(base + b_off + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < 3; ++i) fa[i] = *(const u32x4*)(base + a_off + i * 16 * 64);
        dma(0);
        dma(1);
#pragma unroll
        for (int i = 3; i < 6; ++i) fa[i] = *(const u32x4*)(base + a_off + i * 16 * 64);
        mma(0, 3);
        dma(2);
        dma(3);
        dma(4);
        mma(3, 6);
        __builtin_amdgcn_sched_barrier(0);
        next_fetch();
        r_slot = r_slot == 2 ? 0 : r_slot + 1;
        if constexpr ((MODE & 8192) != 0) { if ((h + 1) % (2 * NK) == 0) tile_end_pause(bid, h); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[2 * bid] = t1 - t0; clk[2 * bid + 1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)bid * 512 + tid] = s;
}

static unsigned long long* g_clk = nullptr;
template <int MODE>
float run(const char* X, const char* Q, float* out, int ksteps, int n_xtiles, int reps) {
    constexpr bool DEEP = (MODE & 128) != 0, WIDE = (MODE & 2048) != 0, WIDE3 = (MODE & 4096) != 0;
    const int lds = WIDE3 ? 3 * W3_SLOT : WIDE ? 2 * W_STAGE : DEEP ? 3 * C_SLOT + 3 * Q_SLOT : 2 * STAGE_BYTES + ((MODE & 32768) ? 8 * 64 * 48 : 0);
    auto launch = [&]() {
        if constexpr (WIDE3) kwide3<MODE><<<256, 512, lds>>>(X, Q, out, ksteps, n_xtiles, g_clk);
        else if constexpr (WIDE) kwide<MODE><<<256, 512, lds>>>(X, Q, out, ksteps, n_xtiles, g_clk);
        else if constexpr (DEEP) kdeep<MODE><<<256, 512, lds>>>(X, Q, out, ksteps, n_xtiles, g_clk);
        else kloop<MODE><<<256, 512, lds>>>(X, Q, out, ksteps, n_xtiles, g_clk);
    };
    if constexpr (WIDE3) hipFuncSetAttribute((const void*)kwide3<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else if constexpr (WIDE) hipFuncSetAttribute((const void*)kwide<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else if constexpr (DEEP) hipFuncSetAttribute((const void*)kdeep<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else hipFuncSetAttribute((const void*)kloop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(1); }
    return ms / reps;
}

#define CASE(M) case M: ms = run<M>(X, Q, out, ksteps, n_xtiles, reps); break;

int main(int argc, char** argv) {
    const int ksteps = argc > 2 ? atoi(argv[2]) : 24 * 64;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const int tiles_per_wg = (ksteps / 24) + 2;
    const int n_xtiles = tiles_per_wg * 256 + 256;  // enough for the one-query-tile mapping (tile = bid + it * 256)
    const size_t xbytes = (size_t)n_xtiles * 256 * DIM_BYTES, qbytes = (size_t)1024 * DIM_BYTES;
    char *X, *Q;
    float* out;
    hipMalloc(&X, xbytes);
    hipMalloc(&Q, qbytes);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&g_clk, 256 * 2 * 8);
    hipMemset(g_clk, 0, 256 * 2 * 8);
    const bool normal_data = getenv("KLOOP_NORMAL") != nullptr;  // fp16 N(0, 1) operands (what the bench uses) instead of +-[0.125, 0.25)
    {  // finite random fp16 everywhere
        std::vector<unsigned> h((64u << 20) / 4);
        unsigned s = 12345u;
        auto next = [&]() { s = s * 1664525u + 1013904223u; return s; };
        for (auto& v : h) {
            if (!normal_data) {
                v = 0x30003000u | (next() & 0x8fff8fffu);
            } else {
                unsigned short hh[2];
                for (int e = 0; e < 2; ++e) {  // sum of 12 uniforms - 6 ~ N(0, 1)
                    float acc = -6.f;
                    for (int u = 0; u < 12; ++u) acc += (float)(next() >> 8) * (1.0f / 16777216.0f);
                    _Float16 f = (_Float16)acc;
                    __builtin_memcpy(&hh[e], &f, 2);
                }
                v = (unsigned)hh[0] | ((unsigned)hh[1] << 16);
            }
        }
        for (size_t o = 0; o < xbytes; o += h.size() * 4) hipMemcpy(X + o, h.data(), std::min(h.size() * 4, xbytes - o), hipMemcpyHostToDevice);
        hipMemcpy(Q, h.data(), qbytes, hipMemcpyHostToDevice);
    }
    printf("operands: %s\n", normal_data ? "fp16 N(0,1)" : "+-[0.125, 0.25) fp16");
    const double c3_ksteps = 39063.0 * 4 / 256 * 24;  // k-steps per workgroup of the headline batch
    for (int a = 1; a < 2; ++a) {
        std::vector<int> modes;
        if (argc > 1) {
            char* p = argv[1];
            while (*p) { modes.push_back((int)strtol(p, &p, 10)); if (*p == ',') ++p; }
        }
        for (int m : modes) {
            float ms = -1;
            switch (m) {
                CASE(1) CASE(2) CASE(3) CASE(7) CASE(6) CASE(8) CASE(24) CASE(9) CASE(25) CASE(11) CASE(15) CASE(27) CASE(31)
                CASE(1083) CASE(699) CASE(703) CASE(2107) CASE(2075) CASE(4155) CASE(4123) CASE(9275) CASE(8383) CASE(12347)
                CASE(135) CASE(143) CASE(159) CASE(175) CASE(191) CASE(190) CASE(134) CASE(187) CASE(315) CASE(319) CASE(447) CASE(443) CASE(287) CASE(415)
                CASE(16415) CASE(16447) CASE(16671) CASE(16703) CASE(49215) CASE(49471) CASE(43) CASE(47) CASE(59) CASE(63) CASE(35) CASE(39) CASE(67) CASE(71) CASE(127) CASE(123) CASE(95) CASE(91) CASE(79) CASE(75)
                default: printf("mode %d not instantiated\n", m); continue;
            }
            double ghz = 0;
            {
                std::vector<unsigned long long> hc(512);
                hipMemcpy(hc.data(), g_clk, 512 * 8, hipMemcpyDeviceToHost);
                std::vector<double> v;
                for (int b = 0; b < 256; ++b) if (hc[2 * b + 1]) v.push_back((double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1);
                std::sort(v.begin(), v.end());
                if (!v.empty()) ghz = v[v.size() / 2];
            }
            const bool wide = (m & (2048 | 4096)) != 0;
            const double tf = 256.0 * 8 * ksteps * (wide ? 48 : 32) * 16384.0 / (ms * 1e-3) / 1e12;
            const double c3k = wide ? 26042.0 * 4 / 256 * 24 : c3_ksteps;
            printf("mode %3d  %s%s%s%s%s%s%s  %.3f ms  = %.2f ms per C3 batch  (%.0f TFLOP/s-equivalent)\n", m, (m & 1) ? "mfma " : "", (m & 2) ? "read " : "",
                   (m & 4) ? "prefetch " : "", (m & 8) ? "dma " : "", (m & 16) ? "stream " : "", (m & 32) ? "barrier " : "", (m & 64) ? "skew " : "",
                   ms, ms * c3k / ksteps, tf);
            if (wide) printf("          384 x 256 workgroup tile, 96 x 128 wave tiles%s\n", (m & 4096) ? ", ring of three 32-deep slots (120 KB)" : "");
            if (m & 8192) printf("          tile-end pause (2,000 + hash %% 2,048 cycles, different per workgroup)\n");
            if (m & (512 | 1024)) printf("          %s%s\n", (m & 512) ? "barrier every other k-step (timing only); " : "", (m & 1024) ? "production order inside a k-step" : "");
            printf("          %s%sin-kernel clock %.3f GHz (median over workgroups)\n", (m & 128) ? "deep ring; " : "", (m & 256) ? "one query tile (per nq=256 batch: ms / 4); " : "", ghz);
        }
    }
    return 0;
}

// Workgroup-wide bitonic sort of THREADS * E keys held in REGISTERS (E keys per thread, blocked layout).
//
// The collate-side kernels (priority sampling, in-batch flattening) sort a few hundred to a few thousand 64-bit keys per
// row.  Round 2 ran every compare-exchange stage through LDS with a workgroup barrier per stage (45 stages for 512 keys,
// 78 for 4096: ~0.3 us each, the dominant cost of those kernels).  Here a thread keeps its E consecutive keys in VGPRs:
//   stride <  E        partner is another register of the same thread          (no communication),
//   stride < 64 * E    partner lives in another lane of the same wavefront      (one 64-bit wave shuffle, no barrier),
//   stride >= 64 * E   partner lives in another wavefront                       (LDS round trip + 2 barriers):
// at most 3 levels of the network touch LDS (512 keys: 3 of 45 stages; 4096 keys: 3 of 78).
#pragma once
#include <hip/hip_runtime.h>

namespace vodhip {

// (DPP lane permutations for partners up to 8 lanes away - quad_perm, row_half_mirror / row_mirror compositions - were tried in place of
// the ds_bpermute pair below: priority sampling 18.1 -> 32.9 us, in-batch flattening 26.7 -> 44.9 us.  With 4-16 waves per workgroup and
// E independent keys per lane the bpermutes are throughput-, not latency-bound, and the per-stage switch over the mask costs more.)
template <typename K>
__device__ __forceinline__ K wg_shfl_xor64(K v, int lane_mask) {
    const unsigned long long u = (unsigned long long)v;
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)u, lane_mask);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), lane_mask);
    return (K)(((unsigned long long)hi << 32) | lo);
}

// buf: LDS, THREADS * E keys; sorted in place (ascending, or descending with DESC).  All THREADS threads call it.
template <int THREADS, int E, bool DESC, typename K>
__device__ void wg_sort_regs(K* buf, int tid) {
    constexpr int P = THREADS * E;
    K v[E];
#pragma unroll
    for (int j = 0; j < E; ++j) v[j] = buf[tid * E + j];
    for (int size = 2; size <= P; size <<= 1) {
        int stride = size >> 1;
        for (; stride >= 64 * E; stride >>= 1) {  // partner in another wavefront
            __syncthreads();
#pragma unroll
            for (int j = 0; j < E; ++j) buf[tid * E + j] = v[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int i = tid * E + j;
                const K o = buf[i ^ stride];
                const bool keep_min = ((i & stride) == 0) == (((i & size) == 0) != DESC);
                v[j] = ((o < v[j]) == keep_min) ? o : v[j];  // one compare: equal keys are interchangeable
            }
        }
        for (; stride >= E; stride >>= 1) {  // partner in another lane of this wavefront
            const int lane_mask = stride / E;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int i = tid * E + j;
                const K o = wg_shfl_xor64(v[j], lane_mask);
                const bool keep_min = ((i & stride) == 0) == (((i & size) == 0) != DESC);
                v[j] = ((o < v[j]) == keep_min) ? o : v[j];  // one compare: equal keys are interchangeable
            }
        }
#pragma unroll
        for (int st = E / 2; st > 0; st >>= 1) {  // partner in another register of this thread
            if (st < size) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    if ((j & st) == 0) {
                        const int i = tid * E + j;
                        const bool up = ((i & size) == 0) != DESC;
                        const K x = v[j], y = v[j + st];
                        const bool sw = (y < x) == up;
                        v[j] = sw ? y : x;
                        v[j + st] = sw ? x : y;
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < E; ++j) buf[tid * E + j] = v[j];
    __syncthreads();
}

// ONE wavefront sorts the 64 * E keys of `buf` (LDS) in its registers: no barrier inside - the caller orders the LDS accesses of
// the other wavefronts around it.  Only lanes of one wavefront call it (`lane` = 0..63).
template <int E, bool DESC, typename K>
__device__ void wave_sort_regs(K* buf, int lane) {
    constexpr int P = 64 * E;
    K v[E];
#pragma unroll
    for (int j = 0; j < E; ++j) v[j] = buf[lane * E + j];
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride >= E; stride >>= 1) {  // partner in another lane
            const int lane_mask = stride / E;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int i = lane * E + j;
                const K o = wg_shfl_xor64(v[j], lane_mask);
                const bool keep_min = ((i & stride) == 0) == (((i & size) == 0) != DESC);
                v[j] = ((o < v[j]) == keep_min) ? o : v[j];
            }
        }
#pragma unroll
        for (int st = E / 2; st > 0; st >>= 1) {  // partner in another register of this lane
            if (st < size) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    if ((j & st) == 0) {
                        const int i = lane * E + j;
                        const bool up = ((i & size) == 0) != DESC;
                        const K x = v[j], y = v[j + st];
                        const bool sw = (y < x) == up;
                        v[j] = sw ? y : x;
                        v[j + st] = sw ? x : y;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) buf[lane * E + j] = v[j];
}

// Sort `P` keys in LDS with a 256-thread workgroup (P = 256, 512, ..., 4096; a power of two).  Returns false otherwise.
// One out-of-line copy per size: the sampling kernel sorts at two sites.
template <int E, bool DESC, typename K>
__device__ __noinline__ void wg_sort_regs_256(K* buf, int tid) {
    wg_sort_regs<256, E, DESC, K>(buf, tid);
}
template <bool DESC, typename K>
__device__ __forceinline__ bool wg_sort_lds_256(K* buf, int P, int tid) {
    switch (P) {
        case 256: wg_sort_regs_256<1, DESC, K>(buf, tid); return true;
        case 512: wg_sort_regs_256<2, DESC, K>(buf, tid); return true;
        case 1024: wg_sort_regs_256<4, DESC, K>(buf, tid); return true;
        case 2048: wg_sort_regs_256<8, DESC, K>(buf, tid); return true;
        case 4096: wg_sort_regs_256<16, DESC, K>(buf, tid); return true;
        default: return false;
    }
}

// Sort `P` keys in LDS with a 1024-thread workgroup (P = 1024, 2048, 4096, 8192).  Inlined (one call site).
template <bool DESC, typename K>
__device__ __forceinline__ bool wg_sort_lds_1024(K* buf, int P, int tid) {
    switch (P) {
        case 1024: wg_sort_regs<1024, 1, DESC, K>(buf, tid); return true;
        case 2048: wg_sort_regs<1024, 2, DESC, K>(buf, tid); return true;
        case 4096: wg_sort_regs<1024, 4, DESC, K>(buf, tid); return true;
        case 8192: wg_sort_regs<1024, 8, DESC, K>(buf, tid); return true;
        default: return false;
    }
}

}  // namespace vodhip

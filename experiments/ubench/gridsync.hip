// What a grid-wide rendezvous costs inside a persistent kernel shaped like the FILTER kernel (one 512-thread workgroup per CU, 152 KB of
// LDS each): VERDICT r4 item 3 proposes to run the select-between-stages inside ONE cooperative launch of the persistent grid
// (hipLaunchCooperativeKernel guarantees co-residency or fails).  Measures, per box:
//   * the launch + completion time of an empty cooperative launch against an ordinary one,
//   * microseconds per grid barrier (sense-reversing counter in device memory, one arriving thread per workgroup),
//   * the same barrier followed by "one workgroup per query does 6 us of select-like work and publishes a threshold" and a second barrier -
//     the sequence a fused stage boundary needs (barrier, select, barrier).
// Build: hipcc -O3 --offload-arch=gfx950 gridsync.hip -o gridsync        Run: ./gridsync [iterations]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                           \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned* sense_word, unsigned n_wg, unsigned& local_sense, long long budget) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        local_sense ^= 1u;
        __threadfence();
        if (atomicAdd(counter, 1u) == n_wg - 1u) {
            atomicExch(counter, 0u);
            __threadfence();
            atomicExch(sense_word, local_sense);
        } else {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(sense_word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != local_sense) {
#ifndef NOSLEEP
                __builtin_amdgcn_s_sleep(2);
#endif
                if (wall_clock64() - t0 > budget) {  // never hang the device: ~2 s at 100 MHz
                    ok = false;
                    break;
                }
            }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(512) void empty_kernel(unsigned* out) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned)(size_t)smem & 1u;
}

// mode 0: barriers only; mode 1: barrier, ~6 us of dependent work in one workgroup per "query" (a latency chain of LDS atomics, like the select), barrier
__global__ __launch_bounds__(512) void sync_kernel(unsigned* counter, unsigned* sense_word, int iters, int mode, unsigned* fail, float* thr) {
    extern __shared__ char smem[];
    unsigned local_sense = 0;
    int* h = (int*)smem;
    for (int it = 0; it < iters; ++it) {
        if (!grid_barrier(counter, sense_word, gridDim.x, local_sense, 200000000ll)) { if (threadIdx.x == 0) atomicOr(fail, 1u); return; }
        if (mode == 1) {
            if (threadIdx.x < 256) {
                int v = threadIdx.x;
                for (int pass = 0; pass < 24; ++pass) {  // ~0.25 us per pass: LDS atomic + barrier-free dependent read
                    atomicAdd(&h[(v + pass) & 255], 1);
                    v = h[(v * 7 + pass) & 255] + v;
                }
                if (threadIdx.x == 0) thr[blockIdx.x] = (float)v;
            }
            if (!grid_barrier(counter, sense_word, gridDim.x, local_sense, 200000000ll)) { if (threadIdx.x == 0) atomicOr(fail, 1u); return; }
        }
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    int dev = 0, n_cu = 0, coop = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    const size_t lds = 152 * 1024;
    CK(hipFuncSetAttribute((const void*)sync_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sync_kernel, 512, lds));
    printf("CUs %d, cooperative launch supported %d, workgroups of 512 threads + 152 KB LDS per CU %d\n", n_cu, coop, per_cu);
    unsigned *counter, *sense_word, *fail, *out;
    float* thr;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&sense_word, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&thr, 4096));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](auto&& fn, int reps) {
        fn();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) fn();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / reps;
    };
    const unsigned grid = (unsigned)n_cu;
    const double t_plain = timed([&] { hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(512), lds, st, out); }, 200);
    void* eargs[] = {(void*)&out};
    const double t_coop = timed([&] { CK(hipLaunchCooperativeKernel((const void*)empty_kernel, dim3(grid), dim3(512), eargs, (unsigned)lds, st)); }, 200);
    printf("empty launch of %u workgroups, back to back on one stream: ordinary %.2f us, cooperative %.2f us per launch\n", grid, t_plain, t_coop);
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(sense_word, 0, 4)); CK(hipMemset(fail, 0, 4));
        int it1 = 1, itn = iters;
        void* a1[] = {(void*)&counter, (void*)&sense_word, (void*)&it1, (void*)&mode, (void*)&fail, (void*)&thr};
        void* an[] = {(void*)&counter, (void*)&sense_word, (void*)&itn, (void*)&mode, (void*)&fail, (void*)&thr};
        // (the sense word alternates per barrier and every launch leaves it where an even number of barriers put it: use even counts)
        it1 = 2; itn = iters & ~1;
        const double t1 = timed([&] { CK(hipMemsetAsync(sense_word, 0, 4, st)); CK(hipLaunchCooperativeKernel((const void*)sync_kernel, dim3(grid), dim3(512), a1, (unsigned)lds, st)); }, 20);
        const double tn = timed([&] { CK(hipMemsetAsync(sense_word, 0, 4, st)); CK(hipLaunchCooperativeKernel((const void*)sync_kernel, dim3(grid), dim3(512), an, (unsigned)lds, st)); }, 5);
        unsigned f = 0;
        CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %.2f us per iteration over %d iterations (2 iterations: %.1f us per launch)%s\n", mode,
               mode == 0 ? "grid barrier" : "barrier + ~6 us select-like chain in every workgroup + barrier", (tn - t1) / (itn - it1), itn, t1,
               f ? "  [A BARRIER TIMED OUT]" : "");
    }
    return 0;
}

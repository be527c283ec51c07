// How fast does ONE wave per SIMD issue v_mfma_f32_16x16x32_f16 when its B operands come from a 384-register resident set (accumulator
// file + vector file), as in the query-resident FILTER kernel?  (experiment, not product)
//   hipcc -O3 --offload-arch=gfx950 -o agpr_mfma agpr_mfma.hip && ./agpr_mfma
// Variants: 0 = 8 B fragments (32 registers) reused for every k-step; 1 = 96 resident B fragments, 56 pinned to the accumulator file;
// 2 = as 1 with zero operands (clock at no switching activity).  Reported: shader cycles per MFMA (s_memtime), the clock held, TFLOP/s.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NB /* resident k-steps */, int NA /* of them in the accumulator file */>
__global__ __launch_bounds__(256, 1) void k(const unsigned* __restrict__ seed, float* __restrict__ out, int iters, unsigned long long* clk, unsigned mask) {
    const int tid = threadIdx.x;
    u32x4 fb[NB][4], fa[2];
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) fb[s][j][e] = (0x30003000u | (seed[(tid * 7 + s * 16 + j * 4 + e) & 4095] & 0x8fff8fffu)) & mask;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) fa[i][e] = (0x30003000u | (seed[(tid * 3 + i * 4 + e) & 4095] & 0x8fff8fffu)) & mask;
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (s < NA) asm volatile("" : "+a"(fb[s][j]));
            else asm volatile("" : "+v"(fb[s][j]));
        }
    f32x4 acc[2][4];
    unsigned long long c0 = 0, r0 = 0;
    if (tid == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[s][j]),
                                                                       (s == 0) ? f32x4{0, 0, 0, 0} : acc[i][j], 0, 0, 0);
        }
        asm volatile("" : "+v"(fa[0]), "+v"(fa[1]));
        float t = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0];
        if (t == 1234.5f) out[0] = t;
    }
    if (tid == 0) {
        clk[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

int main() {
    unsigned* seed;
    float* out;
    unsigned long long* clk;
    hipMalloc(&seed, 4096 * 4);
    hipMalloc(&out, 1024);
    hipMalloc(&clk, 256 * 16);
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned)(i * 2654435761u) ^ 0x9e3779b9u;
    hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int nb, unsigned mask) {
        const int iters = 20000;
        for (int warm = 0; warm < 2; ++warm) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, seed, out, iters, clk, mask);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(512);
        hipMemcpy(c.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int b = 0; b < 256; ++b) { cyc.push_back((double)c[2 * b]); ghz.push_back((double)c[2 * b] / (double)c[2 * b + 1] * 0.1); }
        std::sort(cyc.begin(), cyc.end());
        std::sort(ghz.begin(), ghz.end());
        const double mfmas = (double)iters * nb * 8;
        printf("%-44s %7.2f ms  %6.2f cycles / MFMA  clock %.2f GHz  %6.0f TFLOP/s\n", name, ms, cyc[128] / mfmas, ghz[128],
               mfmas * 16384.0 * 1024 / (ms * 1e-3) / 1e12);
    };
    run("8 B fragments reused (32 registers)", k<2, 0>, 2, 0xffffffffu);
    run("96 resident B fragments, all vector file?", k<12, 0>, 12, 0xffffffffu);
    run("96 resident, 56 in the accumulator file", k<24, 14>, 24, 0xffffffffu);
    run("96 resident, 56 in the acc file, ZERO operands", k<24, 14>, 24, 0u);
    run("8 B fragments reused, ZERO operands", k<2, 0>, 2, 0u);
    return 0;
}

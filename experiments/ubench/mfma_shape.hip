// Which MFMA shape is cheapest per flop on a power-capped MI355X?  (experiment, not product)
//
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shape mfma_shape.hip && ./mfma_shape [iterations per launch] [launches]
//
// 256 workgroups x 8 waves (2 per SIMD), 128 accumulator registers per lane in every variant - the register shape of the
// search kernel's 128 x 64 wave tile.  Variants:
//   0  v_mfma_f32_16x16x32_f16 : 32 accumulators x 4 registers, 12 operand fragments per k-step of 32   (the product's loop)
//   1  v_mfma_f32_32x32x16_f16 :  8 accumulators x 16 registers, 6 operand fragments per k-step of 16
// each "mfma only" (operands fixed in registers) and "mfma + fragment reads" (operands re-read from LDS every k-step with
// ds_read_b128, conflict-free layout, the same bytes per flop in both variants).  Reported: time per launch, TFLOP/s, the clock the
// chip holds (s_memtime cycles per s_memrealtime 100 MHz tick, median over workgroups).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static unsigned long long* g_clk;

template <int SHAPE, bool READ, int ORDER = 0>
__global__ __launch_bounds__(512) void shape_kernel(const unsigned* __restrict__ seed, float* __restrict__ out, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 128 KB of operand bytes in LDS: 4 rotating k-steps x 32 fragments of 1 KB (a 256 x 256 tile's 32-deep slice: the 2 x 4 waves
    // share them exactly as the search kernel's 128 x 64 wave tiles do - 8 row fragments per wave row, 4 column fragments per wave column)
    u32x4* frag = reinterpret_cast<u32x4*>(lds);
    const int wm = wave >> 2, wn = wave & 3;
    for (int i = tid; i < 128 * 1024 / 16; i += 512) {
        u32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = 0x30003000u | (seed[(i * 4 + e) & 4095] & 0x8fff8fffu);
        frag[i] = v;
    }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (tid == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }

    if constexpr (SHAPE == 0) {
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        u32x4 a[8], b[4];
        for (int i = 0; i < 8; ++i) a[i] = frag[(wm * 8 + i) * 64 + lane];
        for (int j = 0; j < 4; ++j) b[j] = frag[(16 + wn * 4 + j) * 64 + lane];
        for (int it = 0; it < iters; ++it) {
            if constexpr (READ) {
                const int base = (it & 3) * 32 * 64 + lane;
                for (int i = 0; i < 8; ++i) a[i] = frag[base + (wm * 8 + i) * 64];
                for (int j = 0; j < 4; ++j) b[j] = frag[base + (16 + wn * 4 + j) * 64];
            }
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                // ORDER 0: row fragment held for 4 MFMAs; 1: column fragment held for 8; 2: snake over (i, j): one operand changes per MFMA
                const int i = ORDER == 1 ? (t & 7) : (t >> 2);
                const int j = ORDER == 1 ? (t >> 3) : (ORDER == 2 && ((t >> 2) & 1)) ? 3 - (t & 3) : (t & 3);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
            }
            if constexpr (!READ) asm volatile("" : "+v"(a[0]), "+v"(b[0]));  // keep the loop a loop
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
        out[blockIdx.x * 512 + tid] = s;
    } else {
        // one iteration = TWO k-steps of 16 = the same 32-deep slice: 2 x (4 A + 2 B fragments), 2 x 8 MFMAs of 32768 flop
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
        u32x4 a[2][4], b[2][2];
        for (int h = 0; h < 2; ++h) {
            for (int i = 0; i < 4; ++i) a[h][i] = frag[(wm * 8 + h * 4 + i) * 64 + lane];
            for (int j = 0; j < 2; ++j) b[h][j] = frag[(16 + wn * 4 + h * 2 + j) * 64 + lane];
        }
        for (int it = 0; it < iters; ++it) {
            if constexpr (READ) {
                const int base = (it & 3) * 32 * 64 + lane;
                for (int h = 0; h < 2; ++h) {
                    for (int i = 0; i < 4; ++i) a[h][i] = frag[base + (wm * 8 + h * 4 + i) * 64];
                    for (int j = 0; j < 2; ++j) b[h][j] = frag[base + (16 + wn * 4 + h * 2 + j) * 64];
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[h][i]), __builtin_bit_cast(f16x8, b[h][j]), acc[i][j], 0, 0, 0);
            if constexpr (!READ) asm volatile("" : "+v"(a[0][0]), "+v"(b[0][0]));
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        out[blockIdx.x * 512 + tid] = s;
    }
    if (tid == 0) {
        clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int SHAPE, bool READ, int ORDER = 0>
static void run(const unsigned* seed, float* out, int iters, int launches) {
    const size_t lds = 128 * 1024;
    hipFuncSetAttribute((const void*)shape_kernel<SHAPE, READ, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int w = 0; w < 3; ++w) shape_kernel<SHAPE, READ, ORDER><<<256, 512, lds>>>(seed, out, iters, g_clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < launches; ++r) shape_kernel<SHAPE, READ, ORDER><<<256, 512, lds>>>(seed, out, iters, g_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= launches;
    std::vector<unsigned long long> hc(512);
    hipMemcpy(hc.data(), g_clk, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < 256; ++b) if (hc[2 * b + 1]) v.push_back((double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1);
    std::sort(v.begin(), v.end());
    const double flop = 256.0 * 8 * iters * 32 * 16384.0;
    if (ORDER) printf("order %d: ", ORDER);
    printf("%-24s %-22s %8.3f ms/launch  %7.1f TFLOP/s  clock %.3f GHz  (%.1f %% of the 2.5 PF peak)\n",
           SHAPE == 0 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_32x32x16_f16", READ ? "mfma + fragment reads" : "mfma only", ms,
           flop / (ms * 1e-3) / 1e12, v.empty() ? 0.0 : v[v.size() / 2], flop / (ms * 1e-3) / 2.5e15 * 100);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    const int launches = argc > 2 ? atoi(argv[2]) : 10;
    unsigned* seed;
    float* out;
    hipMalloc(&seed, 4096 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&g_clk, 512 * 8);
    std::vector<unsigned> h(4096);
    unsigned s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {  // interleaved twice: the second pass is on a warm package
        run<0, false>(seed, out, iters, launches);
        run<1, false>(seed, out, iters, launches);
        run<0, true>(seed, out, iters, launches);
        run<1, true>(seed, out, iters, launches);
        run<0, true, 1>(seed, out, iters, launches);
        run<0, true, 2>(seed, out, iters, launches);
    }
    return 0;
}

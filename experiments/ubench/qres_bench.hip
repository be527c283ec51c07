// Stand-alone A/B of ONE FILTER launch: the production 8-phase kernel (tile 14, from libvodhip_ablation.so) against the query-resident
// kernel (tile 17, compiled into this program from experiments/csrc/kernels_mips_qres.hip so that variants are one hipcc away).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vod_amd/csrc -I include [-DQR_...] experiments/ubench/qres_bench.hip experiments/csrc/kernels_mips_qres.hip \
//         -L vod_amd/csrc -lvodhip_ablation -Wl,-rpath,'$ORIGIN/../../vod_amd/csrc' -o experiments/ubench/qres_bench
//   qres_bench [rows = 1000000] [nq = 256] [reps = 20] [thr = 1e30] [perm = 1]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "vodhip_internal.h"

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

namespace vodhip {
extern long long* g_qr_stamps;  // (QR_STAMPS builds)
}

__global__ void fill_f16(uint16_t* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        // roughly N(0,1): sum of 4 uniforms, centred
        const float u = ((x & 255) + ((x >> 8) & 255) + ((x >> 16) & 255) + (x >> 24)) * (1.f / 256.f) - 2.f;
        p[i] = __builtin_bit_cast(uint16_t, (_Float16)(u * 1.7f));
    }
}

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 1000000;
    const int64_t nq = argc > 2 ? atoll(argv[2]) : 256;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const float thr = argc > 4 ? (float)atof(argv[4]) : 1e30f;
    const int perm = argc > 5 ? atoi(argv[5]) : 1;
    const int64_t dim = 768, nq_pad = (nq + 255) / 256 * 256, rows_pad = (rows + 255) / 256 * 256 + 256;
    uint16_t *X, *Q;
    CK(hipMalloc(&X, rows_pad * dim * 2));
    CK(hipMalloc(&Q, nq_pad * dim * 2));
    fill_f16<<<4096, 256>>>(X, rows_pad * dim, 1u);
    fill_f16<<<256, 256>>>(Q, nq_pad * dim, 77u);
    vodhip::SearchWorkspace ws;
    ws.cap = 16384;
    ws.n_cu = 256;
    CK(hipMalloc(&ws.thr_s, nq_pad * 4));
    CK(hipMalloc(&ws.thr_key, nq_pad * 8));
    CK(hipMalloc(&ws.cand, nq_pad * ws.cap * 8));
    CK(hipMalloc(&ws.cnt, nq_pad * vodhip::CNT_STRIDE * 4));
    CK(hipMalloc(&ws.overflow, 16));
    std::vector<float> t(nq_pad, thr);
    CK(hipMemcpy(ws.thr_s, t.data(), nq_pad * 4, hipMemcpyHostToDevice));
    CK(hipMemset(ws.thr_key, 0, nq_pad * 8));
    CK(hipMemset(ws.cnt, 0, nq_pad * vodhip::CNT_STRIDE * 4));
    CK(hipMemset(ws.overflow, 0, 16));
    if (nq_pad == 256) ws.extra.flags |= vodhip::FILTER_FLAG_CORPUS_NT;
    if (perm) {
        const int64_t T = rows_pad / 256 - 1;
        int64_t P = (int64_t)(T * 0.6180339887);
        auto gcd = [](int64_t a, int64_t b) { while (b) { int64_t r = a % b; a = b; b = r; } return a; };
        while (gcd(P, T) != 1) ++P;
        ws.extra.perm_mul = (int)P;
        ws.extra.perm_mod = (int)T;
    }
    ws.extra.row_bound = (int)rows;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](int tile) {
        CK(hipMemsetAsync(ws.cnt, 0, nq_pad * vodhip::CNT_STRIDE * 4, s));
        if (tile == 17) CK(vodhip::launch_filter_qres(0, X, Q, dim, 0, rows, nq, nq_pad, ws, s));
        else CK(vodhip::launch_filter_8phase(0, 14, X, Q, dim, 0, rows, nq, nq_pad, ws, s));
    };
    double ms[2] = {0, 0};
    unsigned cnt_sum[2] = {0, 0};
    for (int arm = 0; arm < 2; ++arm) {
        run(arm ? 17 : 14);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned> c(nq_pad * vodhip::CNT_STRIDE);
        CK(hipMemcpy(c.data(), ws.cnt, c.size() * 4, hipMemcpyDeviceToHost));
        for (int64_t q = 0; q < nq; ++q) cnt_sum[arm] += c[q * vodhip::CNT_STRIDE];
    }
    for (int r = 0; r < reps; ++r)
        for (int arm = 0; arm < 2; ++arm) {
            CK(hipEventRecord(e0, s));
            run(arm ? 17 : 14);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float m;
            CK(hipEventElapsedTime(&m, e0, e1));
            ms[arm] += m;
        }
    const double flops = 2.0 * nq_pad * rows * dim, bytes = (double)rows * dim * 2;
    for (int arm = 0; arm < 2; ++arm) {
        const double t_ms = ms[arm] / reps;
        printf("tile %d: %.1f us  %.2f TB/s  %.0f TFLOP/s  survivors %u\n", arm ? 17 : 14, t_ms * 1e3, bytes / t_ms / 1e9, flops / t_ms / 1e9, cnt_sum[arm]);
    }
    printf("tile 17 / tile 14 = %.3f\n", ms[1] / ms[0]);
#ifdef QR_STAMPS
    long long st[64];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(vodhip::g_qr_stamps_buf), sizeof(st)));
    printf("stamps (cycles per unit, wave 0 of workgroup 0): units %lld\n", st[15]);
    const char* names[] = {"mfma k0 + reads + glds issue", "mfma k1 issue", "vmcnt wait", "lgkm wait", "barrier", "epilogue (per unit)"};
    for (int i = 0; i < 6; ++i) printf("  %-32s %8.1f\n", names[i], (double)st[i] / (double)st[15]);
#endif
    return 0;
}

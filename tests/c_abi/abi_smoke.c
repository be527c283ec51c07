/* A plain C99 consumer of include/vodhip.h: no Python, no torch, no C++.
 * Builds an index from host float32 rows, searches it with device buffers from the HIP runtime's C API, and checks the
 * result against a brute-force loop (integer-valued data: every dot product is exact in fp32; ties -> smaller id).
 * Built and run by tests/test_c_abi.py:  gcc -std=c99 abi_smoke.c -I include -I /opt/rocm/include -lvodhip -lamdhip64 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "vodhip.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        if ((call) != 0) {                                                           \
            fprintf(stderr, "FAIL %s: %s\n", #call, vodhip_last_error());            \
            return 1;                                                                \
        }                                                                            \
    } while (0)
#define HIPCHECK(call)                                                               \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "FAIL %s: %s\n", #call, hipGetErrorString(e_));          \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(void) {
    const int64_t n = 30000, d = 96, nq = 37;
    const int k = 12;
    float* x = (float*)malloc(sizeof(float) * n * d);
    float* q = (float*)malloc(sizeof(float) * nq * d);
    unsigned s = 12345u;
    for (int64_t i = 0; i < n * d; ++i) { s = s * 1664525u + 1013904223u; x[i] = (float)((int)((s >> 16) % 17) - 8); }
    for (int64_t i = 0; i < nq * d; ++i) { s = s * 1664525u + 1013904223u; q[i] = (float)((int)((s >> 16) % 17) - 8); }

    if (vodhip_version() != VODHIP_VERSION) { fprintf(stderr, "version mismatch\n"); return 1; }
    vodhip_index_t* ix = NULL;
    CHECK(vodhip_index_create(0, d, VODHIP_F16, n, &ix));
    CHECK(vodhip_index_add(ix, x, n / 2, VODHIP_F32, VODHIP_HOST, NULL));
    CHECK(vodhip_index_add(ix, x + (n / 2) * d, n - n / 2, VODHIP_F32, VODHIP_HOST, NULL));
    int64_t ntotal = 0;
    CHECK(vodhip_index_ntotal(ix, &ntotal));
    if (ntotal != n) { fprintf(stderr, "ntotal %lld\n", (long long)ntotal); return 1; }

    void *dq = NULL, *ds = NULL, *di = NULL;
    HIPCHECK(hipMalloc(&dq, sizeof(float) * nq * d));
    HIPCHECK(hipMalloc(&ds, sizeof(float) * nq * k));
    HIPCHECK(hipMalloc(&di, sizeof(int64_t) * nq * k));
    HIPCHECK(hipMemcpy(dq, q, sizeof(float) * nq * d, hipMemcpyHostToDevice));
    CHECK(vodhip_index_search(ix, dq, VODHIP_F32, nq, k, 1000, (float*)ds, (int64_t*)di, NULL));
    float* hs = (float*)malloc(sizeof(float) * nq * k);
    int64_t* hi = (int64_t*)malloc(sizeof(int64_t) * nq * k);
    HIPCHECK(hipMemcpy(hs, ds, sizeof(float) * nq * k, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(hi, di, sizeof(int64_t) * nq * k, hipMemcpyDeviceToHost));

    /* brute force: repeated selection of the best remaining (score desc, id asc) */
    float* sc = (float*)malloc(sizeof(float) * n);
    for (int64_t a = 0; a < nq; ++a) {
        for (int64_t r = 0; r < n; ++r) {
            float acc = 0.f;
            for (int64_t c = 0; c < d; ++c) acc += q[a * d + c] * x[r * d + c];
            sc[r] = acc;
        }
        for (int j = 0; j < k; ++j) {
            int64_t best = -1;
            for (int64_t r = 0; r < n; ++r)
                if (!isnan(sc[r]) && (best < 0 || sc[r] > sc[best])) best = r;
            if (hi[a * k + j] != best + 1000 || hs[a * k + j] != sc[best]) {
                fprintf(stderr, "query %lld rank %d: got (%g, %lld) want (%g, %lld)\n", (long long)a, j, hs[a * k + j],
                        (long long)hi[a * k + j], sc[best], (long long)(best + 1000));
                return 1;
            }
            sc[best] = NAN;
        }
    }
    /* error reporting through the ABI: out-of-range k, message retrievable in C */
    if (vodhip_index_search(ix, dq, VODHIP_F32, nq, 0, 0, (float*)ds, (int64_t*)di, NULL) == 0 || vodhip_last_error()[0] == 0) {
        fprintf(stderr, "k = 0 was not rejected\n");
        return 1;
    }
    CHECK(vodhip_index_destroy(ix));
    (void)hipFree(dq); (void)hipFree(ds); (void)hipFree(di);
    free(x); free(q); free(hs); free(hi); free(sc);
    printf("C ABI smoke ok\n");
    return 0;
}

/* A plain C99 consumer of include/vodhip.h: no Python, no torch, no C++.
 * Builds an index from host float32 rows, searches it with device buffers from the HIP runtime's C API, and checks the
 * result against a brute-force loop (integer-valued data: every dot product is exact in fp32; ties -> smaller id); repeats the
 * search through the one-process node index (three shards); serves it - request fusion (`vodhip_batcher`) and the native HTTP front
 * (`vodhip_http`: POST /raw-search over a plain socket, the fallback callback for GET /) - with no interpreter in the process; then runs
 * the collate-side chain (merge -> sampling) through `vodhip_collate` on the SURVEY's hand-written three-engine case.
 * Built and run by tests/test_c_abi.py:  gcc -std=c99 abi_smoke.c -I include -I /opt/rocm/include -lvodhip -lamdhip64 */
#define __HIP_PLATFORM_AMD__ 1
#define _POSIX_C_SOURCE 200809L
#include <arpa/inet.h>
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

/* The collate-side chain through `vodhip_collate`: one struct of device pointers, no Python.  The SURVEY's hand-written case
 * (section 9, Q3): lookup [4, -1] (labels 1, 0), dense ids [2, 4] scores [.8, .4], sparse ids [8, 2] scores [3, 1], weights 1 / 1
 *   -> merged ids [4, 2, 8, -1], scores [0, .4, 2, -inf], labels [1, -1, -1, 0]  (min-subtracted, first-seen order, one pad column);
 * deterministic sampling (temperature 0), 1 positive of 3: the positive 4, then the negatives by score: 8, 2. */
static int collate_case(void) {
    const int64_t h_lidx[2] = {4, -1}, h_llbl[2] = {1, 0}, h_didx[2] = {2, 4}, h_sidx[2] = {8, 2};
    const float h_dscr[2] = {0.8f, 0.4f}, h_sscr[2] = {3.0f, 1.0f}, h_noise[7] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    enum { STRIDE = 7, KT = 3 };  /* stride = k_lookup + sum(engine_k) + 1 */
    char* base = NULL;
    HIPCHECK(hipMalloc((void**)&base, 16384));
    HIPCHECK(hipMemset(base, 0, 16384));
    size_t off = 0;
#define CARVE(type, count) (type*)(base + (off += 256) - 256 + 0 * (count))
    int64_t *lidx = CARVE(int64_t, 2), *llbl = CARVE(int64_t, 2), *didx = CARVE(int64_t, 2), *sidx = CARVE(int64_t, 2);
    float *dscr = CARVE(float, 2), *sscr = CARVE(float, 2), *noise = CARVE(float, 7);
    int64_t *m_idx = CARVE(int64_t, STRIDE), *m_lbl = CARVE(int64_t, STRIDE);
    float *m_scr = CARVE(float, STRIDE), *m_rd = CARVE(float, STRIDE), *m_rs = CARVE(float, STRIDE);
    int32_t* cursor = CARVE(int32_t, 4);
    int64_t *o_local = CARVE(int64_t, KT), *o_ids = CARVE(int64_t, KT);
    float *o_scr = CARVE(float, KT), *o_logw = CARVE(float, KT), *o_rd = CARVE(float, KT), *o_rs = CARVE(float, KT), *o_row = CARVE(float, 3);
    uint8_t* o_lab = CARVE(uint8_t, KT);
    HIPCHECK(hipMemcpy(lidx, h_lidx, sizeof h_lidx, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(llbl, h_llbl, sizeof h_llbl, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(didx, h_didx, sizeof h_didx, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(sidx, h_sidx, sizeof h_sidx, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dscr, h_dscr, sizeof h_dscr, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(sscr, h_sscr, sizeof h_sscr, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(noise, h_noise, sizeof h_noise, hipMemcpyHostToDevice));
    vodhip_collate_args_t a;
    memset(&a, 0, sizeof a);
    a.lookup_idx = lidx; a.lookup_lbl = llbl; a.k_lookup = 2; a.n_engines = 2; a.nq = 1;
    a.engine_idx[0] = didx; a.engine_scr[0] = dscr; a.engine_k[0] = 2; a.engine_weight[0] = 1.0f;
    a.engine_idx[1] = sidx; a.engine_scr[1] = sscr; a.engine_k[1] = 2; a.engine_weight[1] = 1.0f;
    a.noise = noise; a.noise_stride = STRIDE;
    a.k_positive = 1; a.k_total = KT; a.max_support_size = -1; a.temperature = 0.0f;
    a.merged_idx = m_idx; a.merged_lbl = m_lbl; a.merged_scr = m_scr; a.merged_raw[0] = m_rd; a.merged_raw[1] = m_rs; a.row_cursor = cursor;
    a.out_local = o_local; a.out_ids = o_ids; a.out_scores = o_scr; a.out_log_weights = o_logw; a.out_labels = o_lab;
    a.out_raw[0] = o_rd; a.out_raw[1] = o_rs; a.out_lse_pos = o_row; a.out_lse_neg = o_row + 1; a.out_max_sampling_id = o_row + 2;
    CHECK(vodhip_collate(&a, NULL));
    HIPCHECK(hipDeviceSynchronize());
    int64_t g_m[STRIDE], g_ids[KT];
    float g_ms[STRIDE], g_s[KT];
    uint8_t g_l[KT];
    HIPCHECK(hipMemcpy(g_m, m_idx, sizeof g_m, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(g_ms, m_scr, sizeof g_ms, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(g_ids, o_ids, sizeof g_ids, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(g_s, o_scr, sizeof g_s, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(g_l, o_lab, sizeof g_l, hipMemcpyDeviceToHost));
    const int64_t want_m[STRIDE] = {4, 2, 8, -1, -1, -1, -1}, want_ids[KT] = {4, 8, 2};
    const float want_ms[3] = {0.0f, 0.8f - 0.4f, 3.0f - 1.0f}, want_s[KT] = {0.0f, 3.0f - 1.0f, 0.8f - 0.4f};
    for (int c = 0; c < STRIDE; ++c)
        if (g_m[c] != want_m[c] || (c < 3 && g_ms[c] != want_ms[c]) || (c >= 3 && !(isinf(g_ms[c]) && g_ms[c] < 0))) {
            fprintf(stderr, "collate: merged column %d: (%lld, %g)\n", c, (long long)g_m[c], g_ms[c]);
            return 1;
        }
    for (int j = 0; j < KT; ++j)
        if (g_ids[j] != want_ids[j] || g_s[j] != want_s[j] || g_l[j] != (j == 0)) {
            fprintf(stderr, "collate: sample %d: (%lld, %g, %d)\n", j, (long long)g_ids[j], g_s[j], (int)g_l[j]);
            return 1;
        }
    a.n_engines = 9;  /* errors come back through the ABI */
    if (vodhip_collate(&a, NULL) == 0) { fprintf(stderr, "n_engines = 9 was not rejected\n"); return 1; }
    (void)hipFree(base);
    return 0;
}

/* ---- the serving layer from C: what a non-Python host (C, cgo, JNI) runs instead of the reference's uvicorn process ---- */
static void fallback(void* user, const char* method, const char* target, const uint8_t* body, int64_t n_body, uint64_t client,
                     vodhip_http_reply_t* reply) {
    (void)user; (void)body; (void)n_body; (void)client;
    if (!strcmp(method, "GET") && !strcmp(target, "/")) {
        vodhip_http_reply_set(reply, 200, "application/json", (const uint8_t*)"\"OK\"", 4, NULL);
    } else {
        const char* msg = "{\"detail\":\"Not Found\"}";
        vodhip_http_reply_set(reply, 404, "application/json", (const uint8_t*)msg, (int64_t)strlen(msg), NULL);
    }
}

static int http_exchange(int port, const char* head, const void* body, size_t n_body, char* out, size_t cap, size_t* n_out) {
    int fd = socket(AF_INET, SOCK_STREAM, 0);
    struct sockaddr_in sa;
    memset(&sa, 0, sizeof sa);
    sa.sin_family = AF_INET;
    sa.sin_port = htons((unsigned short)port);
    sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    if (fd < 0 || connect(fd, (struct sockaddr*)&sa, sizeof sa) != 0) return -1;
    if (send(fd, head, strlen(head), 0) < 0 || (n_body && send(fd, body, n_body, 0) < 0)) return -1;
    size_t got = 0;
    for (;;) {  /* "connection: close" requests: read until the server closes */
        ssize_t r = recv(fd, out + got, cap - got, 0);
        if (r <= 0) break;
        got += (size_t)r;
    }
    close(fd);
    *n_out = got;
    return 0;
}

static int serving_case(vodhip_index_t* ix, const float* q, int64_t nq, int64_t d, int k, const float* want_s, const int64_t* want_i,
                        int64_t id_base) {
    vodhip_batcher_t* b = NULL;
    vodhip_http_t* h = NULL;
    CHECK(vodhip_batcher_create(ix, NULL, NULL, NULL, d, id_base, &b));
    float* s = (float*)malloc(sizeof(float) * nq * k);
    int64_t* id = (int64_t*)malloc(sizeof(int64_t) * nq * k);
    CHECK(vodhip_batcher_search(b, q, VODHIP_F32, nq, k, NULL, 0, 7u, s, id));  /* host pointers in and out */
    for (int64_t e = 0; e < nq * k; ++e)
        if (id[e] != want_i[e] || s[e] != want_s[e]) { fprintf(stderr, "batcher entry %lld differs\n", (long long)e); return 1; }
    if (vodhip_batcher_search(b, q, VODHIP_F32, nq, 0, NULL, 0, 7u, s, id) == 0) { fprintf(stderr, "batcher accepted k = 0\n"); return 1; }
    CHECK(vodhip_http_create(b, d, fallback, NULL, 64 << 20, &h));
    const int port = vodhip_http_listen_tcp(h, "127.0.0.1", 0);
    if (port <= 0) { fprintf(stderr, "listen: %s\n", vodhip_last_error()); return 1; }
    CHECK(vodhip_http_start(h));
    /* POST /raw-search?top_k=K with the .npy bytes of the queries (header from the library's own writer) */
    uint8_t npy_head[192];
    const int64_t n_head = vodhip_wire_npy_header(VODHIP_F32, nq, d, npy_head, sizeof npy_head);
    const size_t n_body = (size_t)n_head + sizeof(float) * (size_t)(nq * d);
    uint8_t* body = (uint8_t*)malloc(n_body);
    memcpy(body, npy_head, (size_t)n_head);
    memcpy(body + n_head, q, sizeof(float) * (size_t)(nq * d));
    char head[256];
    snprintf(head, sizeof head, "POST /raw-search?top_k=%d HTTP/1.1\r\nHost: x\r\nConnection: close\r\nContent-Length: %zu\r\n\r\n", k, n_body);
    const size_t cap = 4096 + (size_t)(nq * k) * 12;
    char* reply = (char*)malloc(cap);
    size_t n_reply = 0;
    if (http_exchange(port, head, body, n_body, reply, cap, &n_reply) != 0) { fprintf(stderr, "http exchange failed\n"); return 1; }
    if (n_reply < 16 || strncmp(reply, "HTTP/1.1 200", 12) != 0) { fprintf(stderr, "raw-search: %.80s\n", reply); return 1; }
    const char* payload = NULL;
    for (size_t i = 0; i + 3 < n_reply; ++i)
        if (!memcmp(reply + i, "\r\n\r\n", 4)) { payload = reply + i + 4; break; }
    if (!payload || (size_t)(reply + n_reply - payload) != (size_t)(nq * k) * 12) { fprintf(stderr, "raw-search payload size\n"); return 1; }
    if (memcmp(payload, want_s, sizeof(float) * (size_t)(nq * k)) != 0 ||
        memcmp(payload + sizeof(float) * (size_t)(nq * k), want_i, sizeof(int64_t) * (size_t)(nq * k)) != 0) {
        fprintf(stderr, "raw-search reply differs from the direct search\n");
        return 1;
    }
    if (http_exchange(port, "GET / HTTP/1.1\r\nHost: x\r\nConnection: close\r\n\r\n", NULL, 0, reply, cap, &n_reply) != 0 ||
        !strstr(reply, "\"OK\"")) { fprintf(stderr, "GET / through the fallback failed\n"); return 1; }
    int64_t n_native = 0;
    CHECK(vodhip_http_get_stat(h, "requests_native", &n_native));
    if (n_native != 1) { fprintf(stderr, "requests_native = %lld\n", (long long)n_native); return 1; }
    /* the same searches through the library's own client (what a C / cgo / JNI trainer links): both routes, one kept-alive connection */
    vodhip_client_t* cli = NULL;
    CHECK(vodhip_client_create("127.0.0.1", port, NULL, &cli));
    for (int route = 0; route < 2; ++route) {
        memset(s, 0, sizeof(float) * (size_t)(nq * k));
        memset(id, 0, sizeof(int64_t) * (size_t)(nq * k));
        CHECK(vodhip_client_search(cli, q, VODHIP_F32, nq, d, k, route, 30.0, s, id));
        if (memcmp(s, want_s, sizeof(float) * (size_t)(nq * k)) != 0 || memcmp(id, want_i, sizeof(int64_t) * (size_t)(nq * k)) != 0) {
            fprintf(stderr, "client route %d differs from the direct search\n", route);
            return 1;
        }
    }
    /* top_k outside [1, VODHIP_MAX_K] never leaves the client (a reply of that width would not fit the caller's [nq, k] buffers) ... */
    if (vodhip_client_search(cli, q, VODHIP_F32, nq, d, 0, 0, 30.0, s, id) != -1 || !strstr(vodhip_last_error(), "out of range")) {
        fprintf(stderr, "client: top_k = 0 was not refused\n");
        return 1;
    }
    /* ... and a request the SERVER refuses (a dimension the store does not have) comes back as its HTTP status + {"detail": ...} */
    if (vodhip_client_search(cli, q, VODHIP_F32, nq - 1, d + 1, k, 0, 30.0, s, id) < 400 || !strstr(vodhip_client_last_body(cli), "detail")) {
        fprintf(stderr, "client: a wrong dimension did not come back as an error reply\n");
        return 1;
    }
    CHECK(vodhip_client_destroy(cli));
    CHECK(vodhip_http_get_stat(h, "requests_native", &n_native));
    if (n_native != 3) { fprintf(stderr, "requests_native = %lld after the client's searches\n", (long long)n_native); return 1; }
    CHECK(vodhip_http_destroy(h));
    CHECK(vodhip_batcher_destroy(b));
    free(s); free(id); free(body); free(reply);
    return 0;
}

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
    if (serving_case(ix, q, nq, d, k, hs, hi, 1000) != 0) return 1;
    CHECK(vodhip_index_destroy(ix));
    /* the same rows behind the one-process node index, three shards (all on device 0 here), host buffers in and out: must give
     * the single index's answer (ids without the +1000 offset) */
    {
        const int devs[3] = {0, 0, 0};
        vodhip_node_index_t* nx = NULL;
        CHECK(vodhip_node_index_create(3, devs, d, VODHIP_F16, n, &nx));
        CHECK(vodhip_node_index_add(nx, x, 12345, VODHIP_F32));
        CHECK(vodhip_node_index_add(nx, x + 12345 * d, n - 12345, VODHIP_F32));
        if (vodhip_node_index_n_shards(nx) != 3) { fprintf(stderr, "n_shards\n"); return 1; }
        float* ns = (float*)malloc(sizeof(float) * nq * k);
        int64_t* ni = (int64_t*)malloc(sizeof(int64_t) * nq * k);
        CHECK(vodhip_node_index_search(nx, q, VODHIP_F32, nq, k, VODHIP_HOST, ns, ni, NULL));
        for (int64_t e = 0; e < nq * k; ++e)
            if (ni[e] != hi[e] - 1000 || ns[e] != hs[e]) {
                fprintf(stderr, "node index entry %lld: got (%g, %lld) want (%g, %lld)\n", (long long)e, ns[e], (long long)ni[e], hs[e],
                        (long long)(hi[e] - 1000));
                return 1;
            }
        /* round 6: two searches in flight (the same batch twice into two result buffers), finished in order; a third is refused */
        {
            float* ns2 = (float*)malloc(sizeof(float) * nq * k);
            int64_t* ni2 = (int64_t*)malloc(sizeof(int64_t) * nq * k);
            int64_t merge_ns = -1;
            memset(ns, 0, sizeof(float) * nq * k);
            memset(ni, 0, sizeof(int64_t) * nq * k);
            CHECK(vodhip_node_index_set_param(nx, "profile", 1));
            CHECK(vodhip_node_index_search_async(nx, q, VODHIP_F32, nq, k, VODHIP_HOST, ns, ni, NULL));
            CHECK(vodhip_node_index_search_async(nx, q, VODHIP_F32, nq, k, VODHIP_HOST, ns2, ni2, NULL));
            if (vodhip_node_index_search_async(nx, q, VODHIP_F32, nq, k, VODHIP_HOST, ns2, ni2, NULL) == 0) { fprintf(stderr, "third pending search accepted\n"); return 1; }
            CHECK(vodhip_node_index_search_finish(nx));
            CHECK(vodhip_node_index_search_finish(nx));
            if (vodhip_node_index_search_finish(nx) == 0) { fprintf(stderr, "finish without a pending search\n"); return 1; }
            for (int64_t e = 0; e < nq * k; ++e)
                if (ni[e] != hi[e] - 1000 || ns[e] != hs[e] || ni2[e] != ni[e] || ns2[e] != ns[e]) { fprintf(stderr, "pipelined node search entry %lld differs\n", (long long)e); return 1; }
            CHECK(vodhip_node_index_get_stat(nx, "last_merge_ns", &merge_ns));
            if (merge_ns <= 0) { fprintf(stderr, "last_merge_ns = %lld\n", (long long)merge_ns); return 1; }
            free(ns2); free(ni2);
        }
        int64_t base1 = -1;
        CHECK(vodhip_node_index_shard(nx, 1, NULL, &base1, NULL));
        if (base1 != 10000) { fprintf(stderr, "shard 1 starts at %lld\n", (long long)base1); return 1; }
        CHECK(vodhip_node_index_destroy(nx));
        free(ns); free(ni);
    }
    if (collate_case() != 0) return 1;
    (void)hipFree(dq); (void)hipFree(ds); (void)hipFree(di);
    free(x); free(q); free(hs); free(hi); free(sc);
    printf("C ABI smoke ok\n");
    return 0;
}

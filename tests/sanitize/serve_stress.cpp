// Sanitizer harness for the native serving layer (vod_amd/csrc/vodhip_serve.hip + vodhip_http.hip compiled as HOST C++ with -fsanitize=thread / address;
// GPU sanitizers are not available on the pool, so the threading and the parsers are checked on the CPU build).  A callback engine (exact
// brute force on the host) stands in for the GPU: the batcher's scheduler / completion logic, the caller hand-off, the HTTP front's
// connection threads, the wire parsers and the shutdown paths are exactly the product's.  Built and run by tests/test_sanitizers_cpu.py.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "vodhip.h"

namespace {
constexpr int64_t N = 2000, D = 16;
std::vector<float> g_rows;
std::atomic<int> g_calls{0}, g_errors{0}, g_mismatch{0};

void brute(const float* q, int64_t nq, int k, float* s, int64_t* id) {
    std::vector<std::pair<float, int64_t>> sc((size_t)N);
    for (int64_t a = 0; a < nq; ++a) {
        for (int64_t r = 0; r < N; ++r) {
            float acc = 0.f;
            for (int64_t c = 0; c < D; ++c) acc += q[a * D + c] * g_rows[(size_t)(r * D + c)];
            sc[(size_t)r] = {acc, r};
        }
        std::sort(sc.begin(), sc.end(), [](const auto& x, const auto& y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
        for (int j = 0; j < k; ++j) {
            s[a * k + j] = j < N ? sc[(size_t)j].first : -INFINITY;
            id[a * k + j] = j < N ? sc[(size_t)j].second : -1;
        }
    }
}

int engine(void*, const float* q, int64_t nq, int k, const int32_t*, int, float* s, int64_t* id) {
    ++g_calls;
    if (k == 77) return 3;  // an engine failure: the batch's callers get the error, the batcher keeps serving
    std::this_thread::sleep_for(std::chrono::microseconds(300));
    brute(q, nq, k, s, id);
    return 0;
}

void fallback(void*, const char* method, const char* target, const uint8_t*, int64_t, uint64_t, vodhip_http_reply_t* reply) {
    const std::string msg = std::string("{\"route\": \"") + method + " " + target + "\"}";
    vodhip_http_reply_set(reply, 200, "application/json", (const uint8_t*)msg.data(), (int64_t)msg.size(), "x-from: fallback\r\n");
}

bool http_raw_search(int port, const std::vector<float>& q, int64_t nq, int k, std::vector<float>& s, std::vector<int64_t>& id, bool garbage) {
    int fd = socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_port = htons((uint16_t)port);
    sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    if (fd < 0 || connect(fd, (sockaddr*)&sa, sizeof sa) != 0) {
        if (fd >= 0) close(fd);
        return false;
    }
    uint8_t head[192];
    const int64_t nh = vodhip_wire_npy_header(VODHIP_F32, nq, D, head, sizeof head);
    std::string body((const char*)head, (size_t)nh);
    body.append((const char*)q.data(), sizeof(float) * (size_t)(nq * D));
    if (garbage) body[5] = 'X';  // not an .npy any more: goes to the fallback
    char line[256];
    snprintf(line, sizeof line, "POST /raw-search?top_k=%d HTTP/1.1\r\nHost: x\r\nConnection: close\r\nContent-Length: %zu\r\n\r\n", k, body.size());
    std::string req = std::string(line) + body;
    size_t off = 0;
    while (off < req.size()) {
        ssize_t w = send(fd, req.data() + off, req.size() - off, MSG_NOSIGNAL);
        if (w <= 0) break;
        off += (size_t)w;
    }
    std::string rep;
    char buf[65536];
    for (;;) {
        ssize_t r = recv(fd, buf, sizeof buf, 0);
        if (r <= 0) break;
        rep.append(buf, (size_t)r);
    }
    close(fd);
    const size_t p = rep.find("\r\n\r\n");
    if (p == std::string::npos) return false;
    if (garbage) return rep.find("x-from: fallback") != std::string::npos;
    if (rep.compare(0, 12, "HTTP/1.1 200") != 0 || rep.size() - p - 4 != (size_t)(nq * k) * 12) return false;
    s.resize((size_t)(nq * k));
    id.resize((size_t)(nq * k));
    memcpy(s.data(), rep.data() + p + 4, sizeof(float) * s.size());
    memcpy(id.data(), rep.data() + p + 4 + sizeof(float) * s.size(), sizeof(int64_t) * id.size());
    return true;
}
}  // namespace

int main() {
    std::mt19937 rng(7);
    g_rows.resize((size_t)(N * D));
    for (float& v : g_rows) v = (float)((int)(rng() % 9) - 4);
    vodhip_batcher_t* b = nullptr;
    if (vodhip_batcher_create(nullptr, nullptr, engine, nullptr, D, 0, &b)) { fprintf(stderr, "create: %s\n", vodhip_last_error()); return 1; }
    vodhip_batcher_set_param(b, "grace_us", 500);
    vodhip_http_t* h = nullptr;
    if (vodhip_http_create(b, D, fallback, nullptr, 1 << 20, &h)) return 1;
    const int port = vodhip_http_listen_tcp(h, "127.0.0.1", 0);
    if (port <= 0 || vodhip_http_start(h)) { fprintf(stderr, "listen: %s\n", vodhip_last_error()); return 1; }

    auto worker = [&](int t) {
        std::mt19937 r((unsigned)(100 + t));
        vodhip_client_t* cli = nullptr;
        if (vodhip_client_create("127.0.0.1", port, nullptr, &cli)) { ++g_errors; return; }
        for (int it = 0; it < 40; ++it) {
            const int64_t nq = 1 + (int64_t)(r() % 9);
            const int k = (it % 13 == 12) ? 77 : 1 + (int)(r() % 20);
            std::vector<float> q((size_t)(nq * D));
            for (float& v : q) v = (float)((int)(r() % 9) - 4);
            std::vector<float> s((size_t)(nq * k)), rs((size_t)(nq * k));
            std::vector<int64_t> id((size_t)(nq * k)), rid((size_t)(nq * k));
            bool ok = false;
            if (t % 2 == 0) {
                if (k == 77) {  // the engine refuses this batch: the call must fail (and whoever was fused with it fails too - they retry)
                    if (vodhip_batcher_search(b, q.data(), VODHIP_F32, nq, k, nullptr, 0, (uint64_t)(t + 1), s.data(), id.data()) == 0) ++g_errors;
                    continue;
                }
                for (int attempt = 0; attempt < 50 && !ok; ++attempt)  // a batch fails as a whole: collateral of a k = 77 neighbour is retried
                    ok = vodhip_batcher_search(b, q.data(), VODHIP_F32, nq, k, nullptr, 0, (uint64_t)(t + 1), s.data(), id.data()) == 0;
            } else {
                if (k == 77) {  // a non-.npy body: the fallback answers
                    if (!http_raw_search(port, q, nq, 5, s, id, true)) ++g_errors;
                    continue;
                }
                if (it % 2 == 0) {
                    for (int attempt = 0; attempt < 50 && !ok; ++attempt) ok = http_raw_search(port, q, nq, k, s, id, false);
                } else {  // the library's own client: kept-alive connection, both routes (a batch failed by a k = 77 neighbour comes back as 500: retried)
                    for (int attempt = 0; attempt < 50 && !ok; ++attempt)
                        ok = vodhip_client_search(cli, q.data(), VODHIP_F32, nq, D, k, it % 4 == 1 ? 0 : 1, 10.0, s.data(), id.data()) == 0;
                }
            }
            if (!ok) { ++g_errors; continue; }
            brute(q.data(), nq, k, rs.data(), rid.data());
            if (memcmp(s.data(), rs.data(), sizeof(float) * s.size()) || memcmp(id.data(), rid.data(), sizeof(int64_t) * id.size())) { ++g_errors; ++g_mismatch; }
        }
        vodhip_client_destroy(cli);
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < 12; ++t) ts.emplace_back(worker, t);
    for (auto& t : ts) t.join();
    int64_t batches = 0, requests = 0, native = 0;
    vodhip_batcher_get_stat(b, "batches", &batches);
    vodhip_batcher_get_stat(b, "requests", &requests);
    vodhip_http_get_stat(h, "requests_native", &native);
    // shutdown with clients still connected and mid-request
    std::thread late([&] {
        std::vector<float> q((size_t)(3 * D), 1.f), s;
        std::vector<int64_t> id;
        for (int i = 0; i < 20; ++i) (void)http_raw_search(port, q, 3, 4, s, id, false);
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(3));
    const int rc_stop = vodhip_http_destroy(h);
    late.join();
    const int rc_b = vodhip_batcher_destroy(b);
    // destroy with callers INSIDE the batcher (one batch in the engine, the rest queued behind it): everyone returns - its rows or the
    // shut-down error - and the handle is freed only after the last caller has left (ASan / TSan see a use-after-free otherwise)
    int inside_bad = 0;
    for (int round = 0; round < 8; ++round) {
        vodhip_batcher_t* b2 = nullptr;
        if (vodhip_batcher_create(nullptr, nullptr, engine, nullptr, D, 0, &b2)) return 1;
        vodhip_batcher_set_param(b2, "grace_us", 0);
        std::atomic<int> returned{0}, wrong{0};
        std::vector<std::thread> in;
        for (int t = 0; t < 6; ++t)
            in.emplace_back([&, t] {
                std::vector<float> q((size_t)(2 * D), (float)(t + 1)), s2(2 * 3), rs(2 * 3);
                std::vector<int64_t> id(2 * 3), rid(2 * 3);
                std::this_thread::sleep_for(std::chrono::microseconds(40 * t));
                const int rc = vodhip_batcher_search(b2, q.data(), VODHIP_F32, 2, 3, nullptr, 0, (uint64_t)(t + 1), s2.data(), id.data());
                if (rc == 0) {
                    brute(q.data(), 2, 3, rs.data(), rid.data());
                    if (memcmp(id.data(), rid.data(), sizeof(int64_t) * id.size())) ++wrong;
                } else if (!strstr(vodhip_last_error(), "shut")) {
                    ++wrong;
                }
                ++returned;
            });
        for (;;) {  // until all six are inside (assembled into a batch, or queued): a call on a freed handle would be the harness's bug
            int64_t assembled = 0, queued = 0;
            vodhip_batcher_get_stat(b2, "requests", &assembled);
            vodhip_batcher_get_stat(b2, "pending", &queued);
            if (assembled + queued == 6) break;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        if (vodhip_batcher_destroy(b2)) ++inside_bad;
        for (auto& t : in) t.join();
        if (returned.load() != 6 || wrong.load()) ++inside_bad;
    }
    if (inside_bad) { fprintf(stderr, "destroy with callers inside: %d bad rounds\n", inside_bad); return 1; }
    printf("serve_stress: %d engine calls, %lld batches for %lld requests, %lld native http requests, %d errors (%d result mismatches), stop %d / %d\n", g_calls.load(),
           (long long)batches, (long long)requests, (long long)native, g_errors.load(), g_mismatch.load(), rc_stop, rc_b);
    return (g_errors.load() == 0 && rc_stop == 0 && rc_b == 0 && batches < requests) ? 0 : 1;
}

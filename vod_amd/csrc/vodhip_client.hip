// Serving layer of libvodhip.so, part 3 (host code only): the CLIENT side of the search service - one kept-alive HTTP/1.1 connection that
// speaks POST /fast-search (the reference's base64-.npy-in-JSON documents: /root/reference/src/vod_search/faiss_search/client.py:64-110,
// io.py:17-32) and POST /raw-search.  For consumers that are not Python (a cgo / JNI trainer reaches the server through the same C-ABI as
// the in-process index) and for Python DataLoader workers, whose turn-around between two searches is otherwise spent in the interpreter
// (request document, http framing, JSON scan, base64, two .npy headers: ~150 us) while the GPU idles.  Declared in include/vodhip.h, H6.
#include "../../include/vodhip.h"

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vodhip_internal.h"

struct vodhip_client {
    std::string host, unix_path, host_header;
    int port = 0;
    int fd = -1;
    bool used = false;          // a reply has been read on the current connection (a failure on a USED one may be a stale keep-alive)
    std::vector<char> out;      // request (head + body)
    std::vector<char> in;       // reply (head + body)
    std::vector<uint8_t> raw;   // decoded .npy bytes of one reply field
    std::string last_body;      // body of the last error reply
};

namespace {

int cfail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    vodhip::set_last_error(buf);
    return -1;
}

void drop(vodhip_client* c) {
    if (c->fd >= 0) close(c->fd);
    c->fd = -1;
    c->used = false;
}

void set_timeouts(int fd, double timeout_s) {
    if (!(timeout_s > 0.0)) timeout_s = 120.0;
    struct timeval tv;
    tv.tv_sec = (time_t)timeout_s;
    tv.tv_usec = (suseconds_t)((timeout_s - std::floor(timeout_s)) * 1e6);
    (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    (void)setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}

int connect_now(vodhip_client* c, double timeout_s) {
    drop(c);
    if (!c->unix_path.empty()) {
        const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
        if (fd < 0) return cfail("socket(AF_UNIX): %s", strerror(errno));
        struct sockaddr_un sa;
        memset(&sa, 0, sizeof(sa));
        sa.sun_family = AF_UNIX;
        strncpy(sa.sun_path, c->unix_path.c_str(), sizeof(sa.sun_path) - 1);
        set_timeouts(fd, timeout_s);
        if (connect(fd, (struct sockaddr*)&sa, sizeof(sa))) {
            const int e = errno;
            close(fd);
            return cfail("connect(%s): %s", c->unix_path.c_str(), strerror(e));
        }
        c->fd = fd;
        return 0;
    }
    struct addrinfo hints;
    memset(&hints, 0, sizeof(hints));
    hints.ai_family = AF_UNSPEC;
    hints.ai_socktype = SOCK_STREAM;
    struct addrinfo* res = nullptr;
    char port_txt[16];
    snprintf(port_txt, sizeof(port_txt), "%d", c->port);
    const int gai = getaddrinfo(c->host.c_str(), port_txt, &hints, &res);
    if (gai) return cfail("getaddrinfo(%s): %s", c->host.c_str(), gai_strerror(gai));
    std::string last = "no address";
    for (struct addrinfo* ai = res; ai; ai = ai->ai_next) {
        const int fd = socket(ai->ai_family, ai->ai_socktype, ai->ai_protocol);
        if (fd < 0) continue;
        set_timeouts(fd, timeout_s);
        if (connect(fd, ai->ai_addr, ai->ai_addrlen) == 0) {
            int one = 1;
            (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            c->fd = fd;
            break;
        }
        last = strerror(errno);
        close(fd);
    }
    freeaddrinfo(res);
    if (c->fd < 0) return cfail("connect(%s:%d): %s", c->host.c_str(), c->port, last.c_str());
    return 0;
}

// 0 = sent; -1 = failed (errno kept)
int send_all(int fd, const char* p, size_t n) {
    while (n) {
        const ssize_t w = send(fd, p, n, MSG_NOSIGNAL);
        if (w < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        p += w;
        n -= (size_t)w;
    }
    return 0;
}

struct Reply {
    int status = 0;
    size_t body_off = 0, body_len = 0;
    int64_t x_nq = -1, x_k = -1;
    bool close_after = false;
};

bool header_is(const char* line, size_t n, const char* name, const char** val, size_t* val_n) {
    const size_t ln = strlen(name);
    if (n <= ln || line[ln] != ':') return false;
    for (size_t i = 0; i < ln; ++i) {
        char ch = line[i];
        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch - 'A' + 'a');
        if (ch != name[i]) return false;
    }
    size_t b = ln + 1, e = n;
    while (b < e && (line[b] == ' ' || line[b] == '\t')) ++b;
    while (e > b && (line[e - 1] == ' ' || line[e - 1] == '\t')) --e;
    *val = line + b;
    *val_n = e - b;
    return true;
}

// reads one reply into c->in.  0 = ok; -1 = transport failure (`*got_bytes` tells whether any reply byte had arrived); -2 = a reply this
// client cannot frame (no Content-Length)
int read_reply(vodhip_client* c, Reply* r, bool* got_bytes) {
    std::vector<char>& in = c->in;
    if (in.size() < (1u << 16)) in.resize(1u << 16);
    size_t have = 0, end = std::string::npos;
    *got_bytes = false;
    for (;;) {
        for (size_t i = 0; i + 3 < have; ++i)
            if (in[i] == '\r' && in[i + 1] == '\n' && in[i + 2] == '\r' && in[i + 3] == '\n') {
                end = i;
                break;
            }
        if (end != std::string::npos) break;
        if (have == (1u << 16)) return cfail("reply headers too large");
        const ssize_t n = recv(c->fd, in.data() + have, (1u << 16) - have, 0);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) return cfail(n == 0 ? "the server closed the connection" : ((errno == EAGAIN || errno == EWOULDBLOCK) ? "timed out waiting for the reply" : "recv: %s"), strerror(errno));
        have += (size_t)n;
        *got_bytes = true;
    }
    if (end < 12 || memcmp(in.data(), "HTTP/1.", 7)) return cfail("malformed status line");
    r->status = atoi(in.data() + 9);
    int64_t content_length = -1;
    bool chunked = false;
    size_t p = 0;
    while (p < end && in[p] != '\n') ++p;  // past the status line
    ++p;
    while (p < end) {
        size_t q = p;
        while (q < end && in[q] != '\r') ++q;
        const char* v;
        size_t vn;
        if (header_is(in.data() + p, q - p, "content-length", &v, &vn)) content_length = strtoll(std::string(v, vn).c_str(), nullptr, 10);
        else if (header_is(in.data() + p, q - p, "x-nq", &v, &vn)) r->x_nq = strtoll(std::string(v, vn).c_str(), nullptr, 10);
        else if (header_is(in.data() + p, q - p, "x-k", &v, &vn)) r->x_k = strtoll(std::string(v, vn).c_str(), nullptr, 10);
        else if (header_is(in.data() + p, q - p, "connection", &v, &vn)) r->close_after = vn == 5 && !strncasecmp(v, "close", 5);
        else if (header_is(in.data() + p, q - p, "transfer-encoding", &v, &vn)) chunked = true;
        p = q + 2;
    }
    if (content_length < 0 || chunked) return -2;
    if (content_length > (int64_t)1 << 40) return cfail("absurd Content-Length");
    const size_t body_off = end + 4, n_body = (size_t)content_length;
    if (in.size() < body_off + n_body + 1) in.resize(body_off + n_body + 1);
    size_t got = std::min(have - body_off, n_body);
    while (got < n_body) {
        const ssize_t n = recv(c->fd, in.data() + body_off + got, n_body - got, 0);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) return cfail(n == 0 ? "the server closed the connection inside a reply" : ((errno == EAGAIN || errno == EWOULDBLOCK) ? "timed out inside a reply" : "recv: %s"), strerror(errno));
        got += (size_t)n;
    }
    r->body_off = body_off;
    r->body_len = n_body;
    return 0;
}

inline int64_t b64_len(int64_t n) { return 4 * ((n + 2) / 3); }

// the span of `"key": "<text>"` in a flat JSON object of string fields (what the service replies with); escapes do not occur in base64 text
bool field_span(const char* s, size_t n, const char* key, size_t* b, size_t* e) {
    const size_t kn = strlen(key);
    for (size_t i = 0; i + kn + 4 < n; ++i) {
        if (s[i] != '"' || memcmp(s + i + 1, key, kn) || s[i + 1 + kn] != '"') continue;
        size_t j = i + 2 + kn;
        while (j < n && (s[j] == ' ' || s[j] == '\t')) ++j;
        if (j >= n || s[j] != ':') continue;
        ++j;
        while (j < n && (s[j] == ' ' || s[j] == '\t')) ++j;
        if (j >= n || s[j] != '"') return false;
        const char* q = (const char*)memchr(s + j + 1, '"', n - j - 1);
        if (!q) return false;
        *b = j + 1;
        *e = (size_t)(q - s);
        return true;
    }
    return false;
}

int take_field(vodhip_client* c, const char* body, size_t n, const char* key, int want_dtype, int64_t nq, int k, void* dst) {
    size_t b = 0, e = 0;
    if (!field_span(body, n, key, &b, &e)) return cfail("the reply has no \"%s\" field", key);
    c->raw.resize((size_t)(3 * (e - b) / 4 + 8));
    const int64_t n_raw = vodhip_b64url_decode(body + b, (int64_t)(e - b), c->raw.data());
    if (n_raw < 0) return cfail("the reply's \"%s\" is not base64", key);
    int dtype = 0;
    int64_t rows = 0, cols = 0, off = 0;
    if (vodhip_wire_parse_npy(c->raw.data(), n_raw, &dtype, &rows, &cols, &off)) return cfail("the reply's \"%s\" is not a 2-D .npy array", key);
    if (dtype != want_dtype || rows != nq || cols != k)
        return cfail("the reply's \"%s\" is [%lld, %lld] of dtype code %d, expected [%lld, %d]", key, (long long)rows, (long long)cols, dtype, (long long)nq, k);
    memcpy(dst, c->raw.data() + off, (size_t)(nq * k) * (want_dtype == 3 ? 8 : 4));
    return 0;
}

}  // namespace

extern "C" {

int vodhip_client_create(const char* host, int port, const char* unix_path, vodhip_client_t** out) {
    if (!out || (!unix_path && (!host || port < 1 || port > 65535))) return cfail("invalid arguments");
    vodhip_client* c = new vodhip_client();
    c->host = host ? host : "localhost";
    c->port = port;
    c->unix_path = unix_path ? unix_path : "";
    char hh[300];
    snprintf(hh, sizeof(hh), "%s:%d", c->host.c_str(), port);
    c->host_header = unix_path ? "localhost" : hh;
    *out = c;
    return 0;
}

int vodhip_client_destroy(vodhip_client_t* c) {
    if (!c) return 0;
    drop(c);
    delete c;
    return 0;
}

const char* vodhip_client_last_body(vodhip_client_t* c) { return c ? c->last_body.c_str() : ""; }

int vodhip_client_search(vodhip_client_t* c, const void* queries, int q_dtype, int64_t nq, int64_t dim, int k, int route, double timeout_s,
                         float* out_scores, int64_t* out_ids) {
    if (!c) return cfail("client is NULL");
    if (nq < 1 || dim < 1 || !queries || !out_scores || !out_ids) return cfail("invalid query / output arguments");
    if (q_dtype != VODHIP_F32 && q_dtype != VODHIP_F16) return cfail("q_dtype must be VODHIP_F32 or VODHIP_F16");
    if (route != 0 && route != 1) return cfail("route must be 0 (/fast-search) or 1 (/raw-search)");
    if (k < 1 || k > VODHIP_MAX_K) return cfail("k=%d out of range [1, %d]", k, VODHIP_MAX_K);
    if (nq > (int64_t)1 << 31 || dim > (int64_t)1 << 24 || nq * dim > (int64_t)1 << 40) return cfail("query batch too large");
    // ---- request: head + body in one buffer ----
    uint8_t npy_head[192];
    const int64_t n_head = vodhip_wire_npy_header(q_dtype, nq, dim, npy_head, sizeof(npy_head));
    if (n_head < 0) return -1;
    const int64_t n_data = nq * dim * (q_dtype == VODHIP_F32 ? 4 : 2);
    char tail[64];
    const int n_tail = snprintf(tail, sizeof(tail), "\", \"top_k\": %d}", k);
    const int64_t n_body = route == 1 ? n_head + n_data : 13 + b64_len(n_head + n_data) + n_tail;
    char head[512];
    const int hn = snprintf(head, sizeof(head), "POST %s%s HTTP/1.1\r\nHost: %s\r\nContent-Type: %s\r\nContent-Length: %lld\r\nConnection: keep-alive\r\n\r\n",
                            route == 1 ? "/raw-search?top_k=" : "/fast-search", route == 1 ? std::to_string(k).c_str() : "", c->host_header.c_str(),
                            route == 1 ? "application/octet-stream" : "application/json", (long long)n_body);
    if (hn <= 0 || hn >= (int)sizeof(head)) return cfail("request head too long");
    c->out.resize((size_t)hn + (size_t)n_body);
    char* w = c->out.data();
    memcpy(w, head, (size_t)hn);
    w += hn;
    if (route == 1) {
        memcpy(w, npy_head, (size_t)n_head);
        memcpy(w + n_head, queries, (size_t)n_data);
    } else {
        memcpy(w, "{\"vectors\": \"", 13);  // byte for byte what `json_body_with_arrays({"vectors": v}, {"top_k": k})` writes (vod_amd/io.py)
        w += 13;
        w += vodhip_b64url_encode(npy_head, n_head, (const uint8_t*)queries, n_data, w);
        memcpy(w, tail, (size_t)n_tail);
    }
    // ---- exchange; a kept-alive connection the server closed while it was idle is re-opened once ----
    Reply r;
    for (int attempt = 0;; ++attempt) {
        if (c->fd < 0 && connect_now(c, timeout_s)) return -1;
        if (attempt == 0 && c->used) set_timeouts(c->fd, timeout_s);
        const bool was_used = c->used;
        bool got = false;
        int rc = send_all(c->fd, c->out.data(), c->out.size());
        if (rc) (void)cfail("send: %s", strerror(errno));
        if (!rc) rc = read_reply(c, &r, &got);
        if (rc == 0) break;
        drop(c);
        if (rc == -2) return cfail("the reply carries no Content-Length");
        if (attempt == 1 || !was_used || got) return -1;  // (a fresh connection failing, or a reply cut short: not a stale keep-alive)
    }
    c->used = true;
    if (r.close_after) drop(c);
    struct Trim {  // a handle that once carried an unusually large batch does not keep its buffers (runs on every way out below)
        vodhip_client* c;
        ~Trim() {
            constexpr size_t KEEP = 64u << 20;
            if (c->out.capacity() > KEEP) std::vector<char>().swap(c->out);
            if (c->in.capacity() > KEEP) std::vector<char>().swap(c->in);
            if (c->raw.capacity() > KEEP) std::vector<uint8_t>().swap(c->raw);
        }
    } trim{c};
    const char* body = c->in.data() + r.body_off;
    if (r.status != 200) {
        c->last_body.assign(body, r.body_len);
        vodhip::set_last_error(("HTTP " + std::to_string(r.status)).c_str());
        return r.status > 0 ? r.status : -1;
    }
    if (route == 1) {
        if (r.x_nq != nq || r.x_k != k || r.body_len != (size_t)(nq * k) * 12)
            return cfail("unexpected raw reply: x-nq %lld, x-k %lld, %zu bytes for [%lld, %d]", (long long)r.x_nq, (long long)r.x_k, r.body_len, (long long)nq, k);
        memcpy(out_scores, body, (size_t)(nq * k) * 4);
        memcpy(out_ids, body + (size_t)(nq * k) * 4, (size_t)(nq * k) * 8);
        return 0;
    }
    if (take_field(c, body, r.body_len, "scores", VODHIP_F32, nq, k, out_scores)) return -1;
    if (take_field(c, body, r.body_len, "indices", 3, nq, k, out_ids)) return -1;
    return 0;
}

}  // extern "C"

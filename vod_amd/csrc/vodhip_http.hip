// Serving layer of libvodhip.so, part 2 (host code only): the HTTP/1.1 front of the search service - one native thread per kept-alive
// connection, POST /fast-search (the reference's base64-.npy-in-JSON format: /root/reference/src/vod_search/faiss_search/server.py:76-91,
// io.py:17-32) and POST /raw-search parsed, decoded, searched (through a vodhip_batcher: vodhip_serve.hip) and encoded here; every other
// request - and every request that is not the plain hot case - goes to the host's fallback callback (Python: `server.Endpoints.handle`),
// so validation and error mapping stay in ONE place.  Also the wire pieces (the .npy header NumPy writes, the /fast-search document, the
// reply body), exposed for byte-for-byte checks without a socket.  Declared in include/vodhip.h, section "H6 serving".
#include "../../include/vodhip.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "vodhip_internal.h"

using clock_t_ = std::chrono::steady_clock;
using tp_t = clock_t_::time_point;

namespace {

int sfail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    vodhip::set_last_error(buf);
    return -1;
}


}  // namespace

// ================================================================================================================================
// wire helpers: the .npy header NumPy writes / reads, the /fast-search JSON document
// ================================================================================================================================
namespace {

// np.lib.format.write_array_header_1_0 for a C-ordered 2-D array, byte for byte (NumPy >= 1.24 pads the dict so that the first
// axis can grow in place: GROWTH_AXIS_MAX_DIGITS = 21; the total header length is a multiple of 64, and a full 64 bytes of padding
// are added when it already is one).  `descr`: "<f4", "<f2", "<i8".
int64_t npy_header_2d(const char* descr, int64_t rows, int64_t cols, uint8_t* out /* >= 192 bytes */) {
    char dict[160];
    char rows_txt[32];
    snprintf(rows_txt, sizeof(rows_txt), "%lld", (long long)rows);
    int n = snprintf(dict, sizeof(dict), "{'descr': '%s', 'fortran_order': False, 'shape': (%s, %lld), }", descr, rows_txt, (long long)cols);
    const int grow = 21 - (int)strlen(rows_txt);
    for (int i = 0; i < grow; ++i) dict[n++] = ' ';
    const int hlen0 = n + 1;  // + '\n'
    const int pad = 64 - ((10 + hlen0) % 64);
    const int hlen = hlen0 + pad;
    memcpy(out, "\x93NUMPY\x01\x00", 8);
    out[8] = (uint8_t)(hlen & 0xff);
    out[9] = (uint8_t)(hlen >> 8);
    memcpy(out + 10, dict, (size_t)n);
    memset(out + 10 + n, ' ', (size_t)pad);
    out[10 + hlen - 1] = '\n';
    return 10 + hlen;
}

bool find_field(const char* h, size_t n, const char* key, const char** val, size_t* val_n) {
    // "'key': value, " inside the header dict
    const size_t kn = strlen(key);
    for (size_t i = 0; i + kn + 3 < n; ++i) {
        if (h[i] == '\'' && !memcmp(h + i + 1, key, kn) && h[i + 1 + kn] == '\'') {
            size_t j = i + 2 + kn;
            while (j < n && h[j] == ' ') ++j;
            if (j >= n || h[j] != ':') return false;
            ++j;
            while (j < n && h[j] == ' ') ++j;
            *val = h + j;
            *val_n = n - j;
            return true;
        }
    }
    return false;
}

// version 1.0, little-endian float32 / float16 (queries, scores) or int64 (indices: dtype code 3), C order, 2-D: the layouts this service
// sends and receives.  Anything else -> -1
// (the caller falls back to the host's NumPy reader, which also produces the reference's error for a malformed payload).
int parse_npy_2d(const uint8_t* data, int64_t n, int* dtype, int64_t* rows, int64_t* cols, int64_t* offset) {
    if (n < 10 || memcmp(data, "\x93NUMPY\x01\x00", 8)) return -1;
    const int64_t hlen = data[8] | (data[9] << 8);
    if (10 + hlen > n) return -1;
    const char* h = (const char*)data + 10;
    const char* v;
    size_t vn;
    if (!find_field(h, (size_t)hlen, "descr", &v, &vn) || vn < 5 || v[0] != '\'') return -1;
    if (!memcmp(v, "'<f4'", 5)) *dtype = VODHIP_F32;
    else if (!memcmp(v, "'<f2'", 5)) *dtype = VODHIP_F16;
    else if (!memcmp(v, "'<i8'", 5)) *dtype = 3;  // (the reply's indices: read by the host's client through vodhip_wire_parse_npy)
    else return -1;
    if (!find_field(h, (size_t)hlen, "fortran_order", &v, &vn) || vn < 5 || memcmp(v, "False", 5)) return -1;
    if (!find_field(h, (size_t)hlen, "shape", &v, &vn) || vn < 2 || v[0] != '(') return -1;
    size_t i = 1;
    int64_t dims[2] = {0, 0};
    for (int d = 0; d < 2; ++d) {
        while (i < vn && v[i] == ' ') ++i;
        if (i >= vn || v[i] < '0' || v[i] > '9') return -1;
        int64_t x = 0;
        int digits = 0;
        while (i < vn && v[i] >= '0' && v[i] <= '9') {
            x = x * 10 + (v[i] - '0');
            ++i;
            if (++digits > 12) return -1;
        }
        dims[d] = x;
        while (i < vn && v[i] == ' ') ++i;
        if (d == 0) {
            if (i >= vn || v[i] != ',') return -1;
            ++i;
        }
    }
    if (i < vn && v[i] == ',') {  // "(a, b,)" is legal Python
        ++i;
        while (i < vn && v[i] == ' ') ++i;
    }
    if (i >= vn || v[i] != ')') return -1;
    const int64_t es = *dtype == 3 ? 8 : (*dtype == VODHIP_F32 ? 4 : 2);
    int64_t payload = 0;
    if (__builtin_mul_overflow(dims[0], dims[1], &payload) || __builtin_mul_overflow(payload, es, &payload)) return -1;
    if (payload > n - 10 - hlen) return -1;
    *rows = dims[0];
    *cols = dims[1];
    *offset = 10 + hlen;
    return 0;
}

inline size_t skip_ws(const char* s, size_t i, size_t n) {
    while (i < n && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i;
    return i;
}

// `{"vectors": "<payload>", "top_k": K[, "subset_ids": null]}` in any key order and spacing.  0 = the plain hot document (payload
// span and top_k returned; top_k defaults to 3 like the model, models.py:43-50); 1 = anything else - unknown keys, escapes,
// non-integer top_k, a subset filter, duplicate keys, trailing bytes: the host's pydantic model decides (422 / the filtered path).
int parse_fast_search_doc(const char* s, size_t n, int64_t* vb, int64_t* ve, int64_t* top_k) {
    size_t i = skip_ws(s, 0, n);
    if (i >= n || s[i] != '{') return 1;
    ++i;
    bool have_vec = false, have_k = false, have_sub = false;
    *top_k = 3;
    for (;;) {
        i = skip_ws(s, i, n);
        if (i >= n) return 1;
        if (s[i] == '}') {
            ++i;
            break;
        }
        if (s[i] != '"') return 1;
        const size_t k0 = ++i;
        while (i < n && s[i] != '"' && s[i] != '\\') ++i;
        if (i >= n || s[i] != '"') return 1;
        const size_t kn = i - k0;
        ++i;
        i = skip_ws(s, i, n);
        if (i >= n || s[i] != ':') return 1;
        i = skip_ws(s, i + 1, n);
        if (i >= n) return 1;
        if (kn == 7 && !memcmp(s + k0, "vectors", 7)) {
            if (have_vec || s[i] != '"') return 1;
            const size_t b = ++i;
            const char* q = (const char*)memchr(s + b, '"', n - b);
            if (!q) return 1;
            const size_t e = (size_t)(q - s);
            if (memchr(s + b, '\\', e - b)) return 1;
            *vb = (int64_t)b;
            *ve = (int64_t)e;
            have_vec = true;
            i = e + 1;
        } else if (kn == 5 && !memcmp(s + k0, "top_k", 5)) {
            if (have_k) return 1;
            bool neg = false;
            if (s[i] == '-') {
                neg = true;
                ++i;
            }
            if (i >= n || s[i] < '0' || s[i] > '9') return 1;
            int64_t x = 0;
            int digits = 0;
            while (i < n && s[i] >= '0' && s[i] <= '9') {
                x = x * 10 + (s[i] - '0');
                ++i;
                if (++digits > 9) return 1;
            }
            if (i < n && (s[i] == '.' || s[i] == 'e' || s[i] == 'E')) return 1;
            *top_k = neg ? -x : x;
            have_k = true;
        } else if (kn == 10 && !memcmp(s + k0, "subset_ids", 10)) {
            if (have_sub || i + 4 > n || memcmp(s + i, "null", 4)) return 1;
            have_sub = true;
            i += 4;
        } else {
            return 1;
        }
        i = skip_ws(s, i, n);
        if (i >= n) return 1;
        if (s[i] == ',') {
            ++i;
            continue;
        }
        if (s[i] == '}') {
            ++i;
            break;
        }
        return 1;
    }
    i = skip_ws(s, i, n);
    if (i != n || !have_vec) return 1;
    return 0;
}

inline int64_t b64_len(int64_t n) { return 4 * ((n + 2) / 3); }

// `{"scores": "<b64(npy f32 [nq, k])>", "indices": "<b64(npy i64 [nq, k])>"}` - byte for byte what the host's
// `io.json_body_with_arrays({"scores": ..., "indices": ...})` writes (and so what the reference's FastSearchResponse serialises to)
int64_t fast_search_reply(const float* scores, const int64_t* ids, int64_t nq, int k, char* out, int64_t cap) {
    uint8_t hs[192], hi[192];
    const int64_t nhs = npy_header_2d("<f4", nq, k, hs), nhi = npy_header_2d("<i8", nq, k, hi);
    const int64_t total = 12 + b64_len(nhs + nq * k * 4) + 15 + b64_len(nhi + nq * k * 8) + 2;
    if (!out) return total;
    if (cap < total) return -1;
    int64_t pos = 0;
    memcpy(out + pos, "{\"scores\": \"", 12);
    pos += 12;
    pos += vodhip_b64url_encode(hs, nhs, (const uint8_t*)scores, nq * k * 4, out + pos);
    memcpy(out + pos, "\", \"indices\": \"", 15);
    pos += 15;
    pos += vodhip_b64url_encode(hi, nhi, (const uint8_t*)ids, nq * k * 8, out + pos);
    memcpy(out + pos, "\"}", 2);
    pos += 2;
    return pos == total ? total : -1;
}

}  // namespace

extern "C" {

int64_t vodhip_wire_npy_header(int dtype, int64_t rows, int64_t cols, uint8_t* out, int64_t cap) {
    if (!out || cap < 192 || rows < 0 || cols < 0) return sfail("invalid arguments");
    const char* descr = dtype == VODHIP_F32 ? "<f4" : (dtype == VODHIP_F16 ? "<f2" : (dtype == 3 ? "<i8" : nullptr));
    if (!descr) return sfail("dtype must be VODHIP_F32, VODHIP_F16 or 3 (int64)");
    return npy_header_2d(descr, rows, cols, out);
}

int vodhip_wire_parse_npy(const uint8_t* data, int64_t n, int* dtype, int64_t* rows, int64_t* cols, int64_t* data_offset) {
    if (!data || !dtype || !rows || !cols || !data_offset) return sfail("NULL argument");
    return parse_npy_2d(data, n, dtype, rows, cols, data_offset);
}

int vodhip_wire_parse_fast_search(const char* body, int64_t n, int64_t* vec_begin, int64_t* vec_end, int64_t* top_k) {
    if (!body || !vec_begin || !vec_end || !top_k || n < 0) return sfail("NULL argument");
    return parse_fast_search_doc(body, (size_t)n, vec_begin, vec_end, top_k);
}

int64_t vodhip_wire_fast_search_reply(const float* scores, const int64_t* ids, int64_t nq, int k, char* out, int64_t cap) {
    if (nq < 0 || k < 0 || (out && nq * k > 0 && (!scores || !ids))) return sfail("invalid arguments");
    return fast_search_reply(scores, ids, nq, k, out, cap);
}

}  // extern "C"

// ================================================================================================================================
// vodhip_http: HTTP/1.1 front, one thread per connection
// ================================================================================================================================
struct vodhip_http_reply {
    int status = 0;
    std::string ctype;
    std::string extra;  // "name: value\r\n" lines
    std::vector<uint8_t> payload;
};

struct vodhip_http {
    vodhip_batcher_t* batcher = nullptr;
    vodhip_http_fallback_fn fallback = nullptr;
    void* user = nullptr;
    int64_t dim = 0;
    int64_t max_body = 512ll << 20;
    int idle_timeout_s = 900;  // SO_RCVTIMEO of accepted sockets (0 = none; VODHIP_HTTP_IDLE_TIMEOUT_S at create)
    std::vector<int> listen_fds;
    std::vector<std::string> unix_paths;
    std::thread th_accept;
    std::atomic<bool> stop{false};
    int wake_pipe[2] = {-1, -1};
    std::mutex mu;
    std::condition_variable cv;
    std::set<int> conn_fds;  // open connections (shutdown() on stop unblocks their threads)
    int n_conn_threads = 0;
    std::atomic<uint64_t> next_client{1};
    std::atomic<int64_t> n_native{0}, n_fallback{0}, n_connections{0};
    uint64_t client_base = 0;  // distinguishes this front's connection tags from tags other callers of the batcher use
};

namespace {

struct Buf {  // grow-only byte buffer without value initialisation
    uint8_t* p = nullptr;
    size_t cap = 0;
    ~Buf() { free(p); }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        size_t want = std::max(n, cap * 2);
        want = (want + 4095) / 4096 * 4096;
        void* q = nullptr;
        if (posix_memalign(&q, 64, want)) return false;
        free(p);
        p = (uint8_t*)q;
        cap = want;
        return true;
    }
};

const char* reason_of(int status) {
    switch (status) {
        case 200: return "OK";
        case 400: return "Bad Request";
        case 404: return "Not Found";
        case 405: return "Method Not Allowed";
        case 411: return "Length Required";
        case 413: return "Content Too Large";
        case 422: return "Unprocessable Entity";
        case 431: return "Request Header Fields Too Large";
        case 500: return "Internal Server Error";
        case 501: return "Not Implemented";
        default: return "Status";
    }
}

bool send_all(int fd, const void* a, size_t na, const void* b2, size_t nb) {
    struct iovec iov[2];
    iov[0].iov_base = const_cast<void*>(a);
    iov[0].iov_len = na;
    iov[1].iov_base = const_cast<void*>(b2);
    iov[1].iov_len = nb;
    int first = 0;
    while (first < 2) {
        if (iov[first].iov_len == 0) {
            ++first;
            continue;
        }
        struct msghdr msg;
        memset(&msg, 0, sizeof(msg));
        msg.msg_iov = iov + first;
        msg.msg_iovlen = (size_t)(2 - first);
        ssize_t w = sendmsg(fd, &msg, MSG_NOSIGNAL);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        size_t left = (size_t)w;
        for (int i = first; i < 2 && left > 0; ++i) {
            const size_t take = std::min(left, iov[i].iov_len);
            iov[i].iov_base = (char*)iov[i].iov_base + take;
            iov[i].iov_len -= take;
            left -= take;
        }
    }
    return true;
}

bool send_reply(int fd, int status, const char* ctype, const std::string& extra, const void* payload, size_t n, bool keep) {
    char head[160];
    int hn = snprintf(head, sizeof(head), "HTTP/1.1 %d %s\r\ncontent-type: ", status, reason_of(status));
    std::string h(head, (size_t)hn);
    h += ctype;  // (the host's fallback chooses it: any length)
    hn = snprintf(head, sizeof(head), "\r\ncontent-length: %zu\r\nconnection: %s\r\n", n, keep ? "keep-alive" : "close");
    h.append(head, (size_t)hn);
    h += extra;
    h += "\r\n";
    return send_all(fd, h.data(), h.size(), payload, n);
}

std::string json_detail(const std::string& msg) {
    std::string out = "{\"detail\": \"";
    for (unsigned char c : msg) {
        if (c == '"' || c == '\\') {
            out += '\\';
            out += (char)c;
        } else if (c == '\n') {
            out += "\\n";
        } else if (c < 0x20) {
            char u[8];
            snprintf(u, sizeof(u), "\\u%04x", c);
            out += u;
        } else {
            out += (char)c;
        }
    }
    out += "\"}";
    return out;
}

struct ConnState {
    Buf body, decoded, reply;
    std::vector<float> scores;
    std::vector<int64_t> ids;
};

// the hot case of POST /fast-search and POST /raw-search.  Returns true when the request was answered here (reply sent, `ok` tells
// whether the socket is still good); false = not the plain hot case: nothing was sent, the fallback decides.
bool try_native(vodhip_http* h, int fd, uint64_t client, bool raw, int64_t raw_top_k, const uint8_t* body, size_t n, ConnState& cs, bool keep, bool* ok) {
    int64_t top_k = raw_top_k;
    const uint8_t* npy = body;
    int64_t npy_n = (int64_t)n;
    if (!raw) {
        int64_t vb = 0, ve = 0;
        if (parse_fast_search_doc((const char*)body, n, &vb, &ve, &top_k)) return false;
        if (!cs.decoded.reserve((size_t)(3 * (ve - vb) / 4 + 8))) return false;
        npy_n = vodhip_b64url_decode((const char*)body + vb, ve - vb, cs.decoded.p);
        if (npy_n < 0) return false;  // characters outside the alphabets: the host's lenient decoder decides
        npy = cs.decoded.p;
    }
    int dtype = 0;
    int64_t rows = 0, cols = 0, off = 0;
    if (parse_npy_2d(npy, npy_n, &dtype, &rows, &cols, &off)) return false;
    if (dtype == 3 || cols != h->dim || rows < 1 || rows > 65536 || top_k < 1 || top_k > VODHIP_MAX_K) return false;  // the host raises the reference's errors (and takes the oversized batches)
    const int k = (int)top_k;
    cs.scores.resize((size_t)rows * k);
    cs.ids.resize((size_t)rows * k);
    const int rc = vodhip_batcher_search(h->batcher, npy + off, dtype, rows, k, nullptr, 0, client, cs.scores.data(), cs.ids.data());
    h->n_native.fetch_add(1, std::memory_order_relaxed);
    if (rc) {
        const std::string msg = json_detail(std::string("search failed: ") + vodhip_last_error());
        *ok = send_reply(fd, 500, "application/json", "", msg.data(), msg.size(), keep);
        return true;
    }
    if (raw) {
        char extra[96];
        snprintf(extra, sizeof(extra), "x-nq: %lld\r\nx-k: %d\r\n", (long long)rows, k);
        const size_t ns = (size_t)rows * k * 4, ni = (size_t)rows * k * 8;
        if (!cs.reply.reserve(ns + ni)) return false;
        memcpy(cs.reply.p, cs.scores.data(), ns);
        memcpy(cs.reply.p + ns, cs.ids.data(), ni);
        *ok = send_reply(fd, 200, "application/octet-stream", extra, cs.reply.p, ns + ni, keep);
        return true;
    }
    const int64_t total = fast_search_reply(nullptr, nullptr, rows, k, nullptr, 0);
    if (!cs.reply.reserve((size_t)total)) return false;
    const int64_t wrote = fast_search_reply(cs.scores.data(), cs.ids.data(), rows, k, (char*)cs.reply.p, total);
    if (wrote != total) {
        const std::string msg = json_detail("internal error: reply encoder");
        *ok = send_reply(fd, 500, "application/json", "", msg.data(), msg.size(), keep);
        return true;
    }
    *ok = send_reply(fd, 200, "application/json", "", cs.reply.p, (size_t)total, keep);
    return true;
}

std::string stats_json(vodhip_http* h) {
    static const char* keys[] = {"batches", "requests", "queries", "fused_requests_max", "grace_waits", "grace_expired", "idle_ns", "busy_ns",
                                 "last_batch_queries", "last_batch_requests", "flat_scan_ns", "in_flight", "pending", "active_clients", "tiles_ns_1",
                                 "tiles_ns_2", "tiles_ns_4", "tiles_ns_8"};
    std::string out = "{";
    char tmp[320];
    for (const char* k : keys) {
        int64_t v = 0;
        (void)vodhip_batcher_get_stat(h->batcher, k, &v);
        snprintf(tmp, sizeof(tmp), "\"%s\": %lld, ", k, (long long)v);
        out += tmp;
    }
    long rss_pages = 0;  // resident set of the server process (a soak run reads it before and after: tests/fuzz/fuzz_server.py)
    if (FILE* f = fopen("/proc/self/statm", "r")) {
        long size = 0;
        if (fscanf(f, "%ld %ld", &size, &rss_pages) != 2) rss_pages = 0;
        fclose(f);
    }
    int open_now = 0;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        open_now = h->n_conn_threads;
    }
    snprintf(tmp, sizeof(tmp), "\"requests_native\": %lld, \"requests_fallback\": %lld, \"connections\": %lld, \"open_connections\": %d, \"rss_kb\": %ld}",
             (long long)h->n_native.load(), (long long)h->n_fallback.load(), (long long)h->n_connections.load(), open_now,
             rss_pages * (sysconf(_SC_PAGESIZE) / 1024));
    out += tmp;
    return out;
}

inline bool ieq(const char* a, size_t n, const char* lit) {
    if (strlen(lit) != n) return false;
    for (size_t i = 0; i < n; ++i) {
        char c = a[i];
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
        if (c != lit[i]) return false;
    }
    return true;
}

bool icontains(const std::string& s, const char* lit) {
    std::string l = s;
    for (char& c : l)
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    return l.find(lit) != std::string::npos;
}

void conn_main(vodhip_http* h, int fd, uint64_t client) {
    constexpr size_t MAX_HEAD = 64 * 1024;
    std::vector<char> head(MAX_HEAD);
    size_t head_len = 0;
    ConnState cs;
    auto fail = [&](int status, const char* detail) {
        const std::string msg = json_detail(detail);
        (void)send_reply(fd, status, "application/json", "", msg.data(), msg.size(), false);
    };
    for (;;) {
        // ---- request line + headers ----
        size_t end = std::string::npos;
        for (;;) {
            if (head_len >= 4) {
                for (size_t i = 0; i + 3 < head_len; ++i)
                    if (head[i] == '\r' && head[i + 1] == '\n' && head[i + 2] == '\r' && head[i + 3] == '\n') {
                        end = i;
                        break;
                    }
            }
            if (end != std::string::npos) break;
            if (head_len >= MAX_HEAD) {
                fail(431, "request headers too large");
                goto done;
            }
            const ssize_t r = recv(fd, head.data() + head_len, MAX_HEAD - head_len, 0);
            if (r <= 0) {
                if (r < 0 && errno == EINTR) continue;
                goto done;
            }
            head_len += (size_t)r;
        }
        {
            const char* p = head.data();
            const char* line_end = (const char*)memchr(p, '\r', end);
            if (!line_end) line_end = p + end;
            const char* sp1 = (const char*)memchr(p, ' ', (size_t)(line_end - p));
            const char* sp2 = sp1 ? (const char*)memchr(sp1 + 1, ' ', (size_t)(line_end - sp1 - 1)) : nullptr;
            if (!sp1 || !sp2) {
                fail(400, "malformed request line");
                goto done;
            }
            const std::string method(p, (size_t)(sp1 - p)), target(sp1 + 1, (size_t)(sp2 - sp1 - 1)), version(sp2 + 1, (size_t)(line_end - sp2 - 1));
            int64_t content_length = -1;
            std::string connection, expect, transfer;
            const char* q = line_end;
            const char* hend = p + end;
            bool bad_len = false;
            while (q < hend) {
                if (*q == '\r' || *q == '\n') {
                    ++q;
                    continue;
                }
                const char* le = (const char*)memchr(q, '\r', (size_t)(hend - q));
                if (!le) le = hend;
                const char* colon = (const char*)memchr(q, ':', (size_t)(le - q));
                if (colon) {
                    const char* nb = q;
                    const char* ne = colon;
                    while (ne > nb && (ne[-1] == ' ' || ne[-1] == '\t')) --ne;
                    const char* vb = colon + 1;
                    const char* ve = le;
                    while (vb < ve && (*vb == ' ' || *vb == '\t')) ++vb;
                    while (ve > vb && (ve[-1] == ' ' || ve[-1] == '\t')) --ve;
                    const size_t nn = (size_t)(ne - nb);
                    if (ieq(nb, nn, "content-length")) {
                        const long long previous = content_length;  // a repeated header must agree (RFC 9110 8.6: conflicting values -> 400)
                        content_length = 0;
                        if (vb == ve) bad_len = true;
                        for (const char* c = vb; c < ve; ++c) {
                            if (*c < '0' || *c > '9' || content_length > (1ll << 50)) {
                                bad_len = true;
                                break;
                            }
                            content_length = content_length * 10 + (*c - '0');
                        }
                        if (previous >= 0 && previous != content_length) bad_len = true;
                    } else if (ieq(nb, nn, "connection")) {
                        connection.assign(vb, (size_t)(ve - vb));
                    } else if (ieq(nb, nn, "expect")) {
                        expect.assign(vb, (size_t)(ve - vb));
                    } else if (ieq(nb, nn, "transfer-encoding")) {
                        transfer.assign(vb, (size_t)(ve - vb));
                    }
                }
                q = le;
            }
            if (icontains(transfer, "chunked")) {
                fail(501, "chunked request bodies are not supported: send Content-Length");
                goto done;
            }
            if (bad_len) {
                fail(400, "malformed Content-Length");
                goto done;
            }
            if (content_length > h->max_body) {
                fail(413, "request body too large");
                goto done;
            }
            if ((method == "POST" || method == "PUT") && content_length < 0) {
                fail(411, "Content-Length required");
                goto done;
            }
            const size_t n_body = content_length < 0 ? 0 : (size_t)content_length;
            const bool keep = (version == "HTTP/1.1" && !icontains(connection, "close")) || icontains(connection, "keep-alive");
            if (icontains(expect, "100-continue")) {
                static const char cont[] = "HTTP/1.1 100 Continue\r\n\r\n";
                if (!send_all(fd, cont, sizeof(cont) - 1, nullptr, 0)) goto done;
            }
            // ---- body: what arrived behind the headers, then straight from the socket into the body buffer ----
            if (!cs.body.reserve(n_body + 1)) {
                fail(500, "out of memory");
                goto done;
            }
            const size_t start = end + 4;
            const size_t have = head_len - start;
            const size_t take = std::min(have, n_body);
            memcpy(cs.body.p, head.data() + start, take);
            const size_t rest = have - take;
            memmove(head.data(), head.data() + start + take, rest);  // a pipelined request stays in `head`
            head_len = rest;
            size_t filled = take;
            while (filled < n_body) {
                const ssize_t r = recv(fd, cs.body.p + filled, n_body - filled, 0);
                if (r <= 0) {
                    if (r < 0 && errno == EINTR) continue;
                    goto done;
                }
                filled += (size_t)r;
            }
            // ---- route ----
            const size_t qm = target.find('?');
            const std::string path = target.substr(0, qm);
            bool ok = true;
            bool handled = false;
            if (method == "POST" && path == "/fast-search") {
                handled = try_native(h, fd, client, false, 0, cs.body.p, n_body, cs, keep, &ok);
            } else if (method == "POST" && path == "/raw-search") {
                // top_k from the query string; anything but a plain `top_k=<digits>` pair goes to the fallback
                int64_t top_k = 3;
                bool plain = true;
                if (qm != std::string::npos) {
                    const std::string qs = target.substr(qm + 1);
                    if (qs.compare(0, 6, "top_k=") == 0 && qs.size() > 6 && qs.size() < 16 && qs.find_first_not_of("0123456789", 6) == std::string::npos)
                        top_k = atoll(qs.c_str() + 6);
                    else
                        plain = false;
                }
                if (plain) handled = try_native(h, fd, client, true, top_k, cs.body.p, n_body, cs, keep, &ok);
            } else if (method == "GET" && path == "/stats") {
                const std::string js = stats_json(h);
                ok = send_reply(fd, 200, "application/json", "", js.data(), js.size(), keep);
                handled = true;
            }
            if (!handled) {
                h->n_fallback.fetch_add(1, std::memory_order_relaxed);
                vodhip_http_reply reply;
                if (h->fallback) h->fallback(h->user, method.c_str(), target.c_str(), cs.body.p, (int64_t)n_body, client, &reply);
                if (reply.status == 0) {
                    reply.status = h->fallback ? 500 : 404;
                    reply.ctype = "application/json";
                    const std::string msg = h->fallback ? "{\"detail\": \"internal error: the handler produced no reply\"}" : "{\"detail\":\"Not Found\"}";
                    reply.payload.assign(msg.begin(), msg.end());
                }
                ok = send_reply(fd, reply.status, reply.ctype.c_str(), reply.extra, reply.payload.data(), reply.payload.size(), keep);
            }
            if (!ok || !keep) goto done;
            // a kept-alive connection does not hold on to the buffers of one unusually large request
            constexpr size_t KEEP_BYTES = 64u << 20;
            for (Buf* bf : {&cs.body, &cs.decoded, &cs.reply})
                if (bf->cap > KEEP_BYTES) {
                    free(bf->p);
                    bf->p = nullptr;
                    bf->cap = 0;
                }
            if (cs.ids.capacity() * sizeof(int64_t) > KEEP_BYTES) {
                std::vector<float>().swap(cs.scores);
                std::vector<int64_t>().swap(cs.ids);
            }
        }
    }
done:
    (void)vodhip_batcher_forget_client(h->batcher, client);
    {
        std::lock_guard<std::mutex> lk(h->mu);
        h->conn_fds.erase(fd);
        close(fd);
        --h->n_conn_threads;
        h->cv.notify_all();
    }
}

void accept_main(vodhip_http* h) {
    std::vector<struct pollfd> pfds;
    for (int fd : h->listen_fds) {
        (void)fcntl(fd, F_SETFL, fcntl(fd, F_GETFL, 0) | O_NONBLOCK);  // (accepted sockets do not inherit it on Linux)
        pfds.push_back({fd, POLLIN, 0});
    }
    pfds.push_back({h->wake_pipe[0], POLLIN, 0});
    while (!h->stop.load()) {
        const int r = poll(pfds.data(), pfds.size(), -1);
        if (r < 0) {
            if (errno == EINTR) continue;
            break;
        }
        if (h->stop.load()) break;
        for (size_t i = 0; i + 1 < pfds.size(); ++i) {
            if (!(pfds[i].revents & POLLIN)) continue;
            const int fd = accept(pfds[i].fd, nullptr, nullptr);  // (the listeners are non-blocking: a connection reset between poll and accept)
            if (fd < 0) {
                if (errno == EMFILE || errno == ENFILE || errno == ENOBUFS || errno == ENOMEM)  // out of descriptors: the backlog stays
                    std::this_thread::sleep_for(std::chrono::milliseconds(20));                 // readable - do not spin on it
                continue;
            }
            int one = 1;
            (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));  // (fails harmlessly on a Unix-domain socket)
            // idle / slow-sender bound: a kept-alive connection that stays silent this long is closed (its client re-opens on the next
            // search: vodhip_client retries once, the Python client's pool likewise) - a thread and its buffers are not held for ever
            struct timeval idle = {h->idle_timeout_s, 0};
            if (h->idle_timeout_s > 0) (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &idle, sizeof(idle));
            const uint64_t client = h->client_base + h->next_client.fetch_add(1);
            {
                std::lock_guard<std::mutex> lk(h->mu);
                if (h->stop.load() || h->n_conn_threads >= 4096) {
                    close(fd);
                    continue;
                }
                h->conn_fds.insert(fd);
                ++h->n_conn_threads;
            }
            h->n_connections.fetch_add(1, std::memory_order_relaxed);
            try {
                std::thread(conn_main, h, fd, client).detach();
            } catch (const std::exception&) {  // no thread to be had (process limits): refuse this connection, keep serving the others
                std::lock_guard<std::mutex> lk(h->mu);
                h->conn_fds.erase(fd);
                --h->n_conn_threads;
                close(fd);
            }
        }
    }
}

}  // namespace

extern "C" {

int vodhip_http_reply_set(vodhip_http_reply_t* reply, int status, const char* content_type, const uint8_t* payload, int64_t n, const char* extra_headers) {
    if (!reply || status < 100 || status > 599 || n < 0 || (n > 0 && !payload)) return sfail("invalid arguments");
    reply->status = status;
    reply->ctype = content_type ? content_type : "application/json";
    reply->extra = extra_headers ? extra_headers : "";
    reply->payload.assign(payload, payload + n);
    return 0;
}

int vodhip_http_create(vodhip_batcher_t* batcher, int64_t dim, vodhip_http_fallback_fn fallback, void* user, int64_t max_body_bytes,
                       vodhip_http_t** out) {
    if (!batcher || !out || dim <= 0) return sfail("invalid arguments");
    vodhip_http* h = new vodhip_http();
    h->batcher = batcher;
    h->dim = dim;
    h->fallback = fallback;
    h->user = user;
    if (max_body_bytes > 0) h->max_body = max_body_bytes;
    if (const char* ev = getenv("VODHIP_HTTP_IDLE_TIMEOUT_S")) h->idle_timeout_s = std::max(0, atoi(ev));
    h->client_base = ((uint64_t)(uintptr_t)h) << 20;  // tags unique across fronts that share a batcher
    if (pipe(h->wake_pipe)) {
        delete h;
        return sfail("pipe() failed: %s", strerror(errno));
    }
    *out = h;
    return 0;
}

int vodhip_http_listen_tcp(vodhip_http_t* h, const char* host, int port) {
    if (!h || !host || port < 0 || port > 65535) return sfail("invalid arguments");
    struct addrinfo hints;
    memset(&hints, 0, sizeof(hints));
    hints.ai_family = AF_UNSPEC;
    hints.ai_socktype = SOCK_STREAM;
    hints.ai_flags = AI_PASSIVE;
    struct addrinfo* res = nullptr;
    char port_txt[16];
    snprintf(port_txt, sizeof(port_txt), "%d", port);
    const int gai = getaddrinfo(host[0] ? host : nullptr, port_txt, &hints, &res);
    if (gai) return sfail("getaddrinfo(%s): %s", host, gai_strerror(gai));
    int bound = 0, bound_port = port;
    std::string last_err;
    for (struct addrinfo* ai = res; ai; ai = ai->ai_next) {  // every address the name resolves to ("localhost": ::1 and 127.0.0.1)
        const int fd = socket(ai->ai_family, ai->ai_socktype, ai->ai_protocol);
        if (fd < 0) continue;
        int one = 1;
        (void)setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        if (ai->ai_family == AF_INET6) (void)setsockopt(fd, IPPROTO_IPV6, IPV6_V6ONLY, &one, sizeof(one));
        if (bound_port != port) {  // port 0: the later addresses take the port the first one got
            if (ai->ai_family == AF_INET) ((struct sockaddr_in*)ai->ai_addr)->sin_port = htons((uint16_t)bound_port);
            if (ai->ai_family == AF_INET6) ((struct sockaddr_in6*)ai->ai_addr)->sin6_port = htons((uint16_t)bound_port);
        }
        if (bind(fd, ai->ai_addr, ai->ai_addrlen) || listen(fd, 256)) {
            last_err = strerror(errno);
            close(fd);
            continue;
        }
        if (bound_port == 0) {
            struct sockaddr_storage ss;
            socklen_t sl = sizeof(ss);
            if (getsockname(fd, (struct sockaddr*)&ss, &sl) == 0)
                bound_port = ntohs(ss.ss_family == AF_INET6 ? ((struct sockaddr_in6*)&ss)->sin6_port : ((struct sockaddr_in*)&ss)->sin_port);
        }
        h->listen_fds.push_back(fd);
        ++bound;
    }
    freeaddrinfo(res);
    if (!bound) return sfail("cannot listen on %s:%d: %s", host, port, last_err.c_str());
    return bound_port;
}

int vodhip_http_listen_unix(vodhip_http_t* h, const char* path) {
    if (!h || !path || strlen(path) >= sizeof(((struct sockaddr_un*)nullptr)->sun_path)) return sfail("invalid socket path");
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return sfail("socket(AF_UNIX): %s", strerror(errno));
    struct sockaddr_un sa;
    memset(&sa, 0, sizeof(sa));
    sa.sun_family = AF_UNIX;
    strncpy(sa.sun_path, path, sizeof(sa.sun_path) - 1);
    {   // a stale socket file of a dead server is removed; one a LIVE listener still answers on is somebody's server: refuse
        const int probe = socket(AF_UNIX, SOCK_STREAM, 0);
        const bool live = probe >= 0 && connect(probe, (struct sockaddr*)&sa, sizeof(sa)) == 0;
        if (probe >= 0) close(probe);
        if (live) {
            close(fd);
            return sfail("%s: a live server is listening on this socket", path);
        }
    }
    (void)unlink(path);
    if (bind(fd, (struct sockaddr*)&sa, sizeof(sa)) || listen(fd, 256)) {
        const int e = errno;
        close(fd);
        return sfail("cannot listen on %s: %s", path, strerror(e));
    }
    h->listen_fds.push_back(fd);
    h->unix_paths.push_back(path);
    return 0;
}

int vodhip_http_start(vodhip_http_t* h) {
    if (!h || h->listen_fds.empty()) return sfail("nothing to listen on");
    if (h->th_accept.joinable()) return sfail("already started");
    h->th_accept = std::thread(accept_main, h);
    return 0;
}

int vodhip_http_stop(vodhip_http_t* h) {
    if (!h) return 0;
    h->stop.store(true);
    if (h->wake_pipe[1] >= 0) {
        const char c = 1;
        (void)!write(h->wake_pipe[1], &c, 1);
    }
    if (h->th_accept.joinable()) h->th_accept.join();
    for (int fd : h->listen_fds) close(fd);
    h->listen_fds.clear();
    for (const std::string& p : h->unix_paths) (void)unlink(p.c_str());
    h->unix_paths.clear();
    std::unique_lock<std::mutex> lk(h->mu);
    for (int fd : h->conn_fds) (void)shutdown(fd, SHUT_RDWR);  // unblocks the connection threads' recv()
    // A thread inside a search (or inside the host's fallback) finishes its request first.  No bound on this wait: the threads hold
    // `h` and its batcher, and the caller destroys both right after - returning early left them running on freed memory (round-4
    // advisor).  Every wait they can be in ends: recv() fails on the shut-down socket, a search completes, the fallback returns.
    while (h->n_conn_threads != 0) {
        h->cv.wait_for(lk, std::chrono::seconds(1), [&] { return h->n_conn_threads == 0; });
        for (int fd : h->conn_fds) (void)shutdown(fd, SHUT_RDWR);
    }
    return 0;
}

int vodhip_http_destroy(vodhip_http_t* h) {
    if (!h) return 0;
    const int rc = vodhip_http_stop(h);
    if (rc) return rc;  // (threads still hold `h`: leak it rather than free it under them)
    if (h->wake_pipe[0] >= 0) close(h->wake_pipe[0]);
    if (h->wake_pipe[1] >= 0) close(h->wake_pipe[1]);
    delete h;
    return 0;
}

int vodhip_http_get_stat(vodhip_http_t* h, const char* key, int64_t* out) {
    if (!h || !key || !out) return sfail("NULL argument");
    if (!strcmp(key, "requests_native")) *out = h->n_native.load();
    else if (!strcmp(key, "requests_fallback")) *out = h->n_fallback.load();
    else if (!strcmp(key, "connections")) *out = h->n_connections.load();
    else if (!strcmp(key, "open_connections")) {
        std::lock_guard<std::mutex> lk(h->mu);
        *out = h->n_conn_threads;
    } else return sfail("unknown http stat '%s'", key);
    return 0;
}

}  // extern "C"

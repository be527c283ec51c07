// Serving layer of libvodhip.so, part 1 (host code only): request fusion in front of one index handle (the HTTP/1.1 front for the search
// service's hot routes is vodhip_http.hip).  Declared in include/vodhip.h, section "H6 serving".
//
// What it replaces in the reference: the single uvicorn worker that runs `faiss_index.search` synchronously for one request at a
// time (src/vod_search/faiss_search/server.py:57-98) while every DataLoader worker of every trainer rank sends its own small batch
// (src/vod_dataloaders/realm_dataloader.py:92-118, core/search.py:128-146).  A brute-force scan reads the whole corpus whatever the
// batch size, so R concurrent requests answered by ONE scan cost about what one costs.  Round 3 did the fusing in Python (two
// collector threads, a fixed wait window, off by default) and the per-request work under the GIL; here both are native:
//
//   vodhip_batcher  - thread-safe blocking `search`; a scheduler thread fuses whatever is pending into one batch, a completion
//                     thread finishes it (exactness check / recovery inside the library), the callers copy their own rows out.
//                     Policy = batch-while-busy, no fixed window (make_decision below).
//   vodhip_http     - (vodhip_http.hip) thread per connection; POST /fast-search and POST /raw-search are parsed, decoded, searched and
//                     encoded natively; everything else goes to the host's fallback callback.
#include "../../include/vodhip.h"

#include <arpa/inet.h>
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

inline int64_t ns_between(tp_t a, tp_t b) { return std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); }
inline int elem_bytes(int dtype) { return dtype == VODHIP_F32 ? 4 : 2; }

}  // namespace

// ================================================================================================================================
// vodhip_batcher
// ================================================================================================================================
namespace {

struct Batch;

struct Request {
    const void* q = nullptr;
    int q_dtype = VODHIP_F32;
    int64_t nq = 0;
    int k = 0;
    const int32_t* subset = nullptr;
    int n_subset = 0;
    uint64_t client = 0;
    float* out_s = nullptr;
    int64_t* out_i = nullptr;
    tp_t t_arrive;
    // set by the scheduler / completion side (under vodhip_batcher::mu)
    Batch* batch = nullptr;
    int64_t row0 = 0;
    bool done = false;
};

struct Slot {  // staging of one batch: fused queries (host [+ device]), result rows (host memory the device can write), subset labels
    void* q_host = nullptr;
    size_t q_host_bytes = 0;
    void* q_dev = nullptr;
    size_t q_dev_bytes = 0;
    float* s_host = nullptr;
    int64_t* i_host = nullptr;
    size_t out_elems = 0;
    int32_t* sub_host = nullptr;
    int32_t* sub_dev = nullptr;
    size_t sub_elems = 0;
    bool busy = false;  // a batch owns it (from assembly until its last caller has copied its rows out)
    int users = 0;      // callers that still have to copy their rows out of the result buffers
};

struct Batch {
    std::vector<Request*> reqs;
    int64_t nq = 0;
    int k = 0;
    int q_dtype = VODHIP_F32;
    bool subset = false;
    int n_subset = 0;
    Slot* slot = nullptr;
    tp_t t_submit;
    bool pipeline_was_empty = false;  // its duration is a clean scan time (feeds the grace estimate)
    bool submitted = false;           // the library has accepted it (the completion thread may finish it)
    int rc = 0;
    std::string err;
};

enum : int { ENGINE_INDEX = 0, ENGINE_NODE = 1, ENGINE_CALLBACK = 2 };

}  // namespace

struct vodhip_batcher {
    int kind = ENGINE_INDEX;
    vodhip_index_t* index = nullptr;
    vodhip_node_index_t* node = nullptr;
    vodhip_search_fn fn = nullptr;
    void* fn_user = nullptr;
    int device = 0;
    int64_t dim = 0;
    int64_t id_base = 0;
    hipStream_t stream = nullptr;

    // tunables (vodhip_batcher_set_param)
    int64_t max_queries = 2048;   // a fused batch never exceeds this many queries (a single larger request runs alone)
    int64_t flat_queries = 256;   // up to here a scan costs the same whatever the batch size (one query tile: HBM-bound)
    int64_t grace_us = 1000;      // cap of the wait for expected company when the engine is idle (0 = never wait)
    int64_t grace_pct = 35;       // ... and never more than this share of a measured scan
    int64_t window_us = 0;        // optional fixed window (round 3's `--micro-batch-wait-ms`): wait at least this long for company
    int64_t depth = 2;            // batches enqueued on the device at once (ENGINE_INDEX; the others run one at a time)

    std::mutex mu;
    std::condition_variable cv_sched;  // new request / completion / slot freed -> scheduler
    std::condition_variable cv_done;   // a batch completed -> its callers
    std::condition_variable cv_compl;  // a batch was submitted -> completion thread
    std::deque<Request*> pending;
    std::deque<Batch*> inflight;       // submitted, not yet completed (oldest first)
    std::vector<Slot> slots;
    struct Client {
        tp_t last_seen;          // arrival of its latest request
        tp_t last_done;          // when its latest request was answered (its rows copied out)
        double ema_gap_ns = 0.0; // how long it usually takes to come back after an answer (turn-around + think time)
        bool answered = false;
    };
    std::unordered_map<uint64_t, Client> clients;  // client tag -> what is known about its rhythm
    bool stop = false;
    bool closing = false;  // destroy has begun: new searches are refused, the ones already inside are drained
    int callers = 0;       // threads inside vodhip_batcher_search (destroy waits for them before it frees the handle)
    std::thread th_sched, th_compl;
    double ema_flat_scan_ns = 0.0;  // duration of batches of <= flat_queries queries that started on an idle engine
    int tiles_samples[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double ema_tiles_ns[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // ... and of batches of t query tiles (256 queries each) that started on an idle engine
    double estimate_ns(int64_t nq) const {  // expected duration of a batch of nq queries: its own bucket, else scaled from the nearest known one
        const int t = (int)std::min<int64_t>(8, std::max<int64_t>(1, (nq + 255) / 256));
        if (ema_tiles_ns[t] > 0.0) return ema_tiles_ns[t];
        for (int d = 1; d < 8; ++d)
            for (int u : {t - d, t + d})
                if (u >= 1 && u <= 8 && ema_tiles_ns[u] > 0.0) return ema_tiles_ns[u] * (u == 1 && t > 1 ? (double)t * 0.75 : (double)t / (double)u);
        return 0.0;
    }
    tp_t t_idle_since;              // when the engine last became idle (stats)
    tp_t t_last_completion;         // when the previous batch completed (= when a batch enqueued behind it started to run)
    // stats
    int64_t n_batches = 0, n_requests = 0, n_queries = 0, n_fused_max = 0, n_grace_waits = 0, n_grace_full = 0;
    int64_t idle_ns = 0, busy_ns = 0, last_batch_queries = 0, last_batch_requests = 0;
    tp_t t_busy_since;
};

namespace {

#define SHIP(expr)                                                                                          \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            (void)hipGetLastError();                                                                        \
            return sfail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);        \
        }                                                                                                   \
    } while (0)

int slot_reserve(vodhip_batcher* b, Slot& s, size_t q_bytes, size_t out_elems, size_t sub_elems) {
    const bool gpu = b->kind != ENGINE_CALLBACK;  // a callback engine may live in a process without a GPU (tests): plain memory
    if (s.q_host_bytes < q_bytes) {
        const size_t want = std::max(q_bytes, (size_t)1 << 20);
        if (gpu) {
            if (s.q_host) (void)hipHostFree(s.q_host);
            s.q_host = nullptr;
            SHIP(hipHostMalloc(&s.q_host, want, hipHostMallocDefault));
        } else {
            free(s.q_host);
            s.q_host = aligned_alloc(64, (want + 63) / 64 * 64);
            if (!s.q_host) return sfail("out of memory");
        }
        s.q_host_bytes = want;
    }
    if (b->kind == ENGINE_INDEX && s.q_dev_bytes < q_bytes) {
        const size_t want = std::max(q_bytes, (size_t)1 << 20);
        if (s.q_dev) (void)hipFree(s.q_dev);
        s.q_dev = nullptr;
        SHIP(hipMalloc(&s.q_dev, want));
        s.q_dev_bytes = want;
    }
    if (s.out_elems < out_elems) {
        const size_t want = std::max(out_elems, (size_t)1 << 16);
        if (gpu) {
            if (s.s_host) (void)hipHostFree(s.s_host);
            if (s.i_host) (void)hipHostFree(s.i_host);
            s.s_host = nullptr;
            s.i_host = nullptr;
            // the final select kernel of a search writes its result rows straight into this (device-visible) host memory: completing
            // a search is an event wait, not a device-to-host copy queued behind the next search's kernels
            SHIP(hipHostMalloc((void**)&s.s_host, want * sizeof(float), hipHostMallocDefault));
            SHIP(hipHostMalloc((void**)&s.i_host, want * sizeof(int64_t), hipHostMallocDefault));
        } else {
            free(s.s_host);
            free(s.i_host);
            s.s_host = (float*)aligned_alloc(64, (want * sizeof(float) + 63) / 64 * 64);
            s.i_host = (int64_t*)aligned_alloc(64, (want * sizeof(int64_t) + 63) / 64 * 64);
            if (!s.s_host || !s.i_host) return sfail("out of memory");
        }
        s.out_elems = want;
    }
    if (sub_elems && s.sub_elems < sub_elems) {
        const size_t want = std::max(sub_elems, (size_t)4096);
        free(s.sub_host);
        s.sub_host = (int32_t*)malloc(want * sizeof(int32_t));
        if (!s.sub_host) return sfail("out of memory");
        if (b->kind == ENGINE_INDEX) {
            if (s.sub_dev) (void)hipFree(s.sub_dev);
            s.sub_dev = nullptr;
            SHIP(hipMalloc((void**)&s.sub_dev, want * sizeof(int32_t)));
        }
        s.sub_elems = want;
    }
    return 0;
}

void slot_free(vodhip_batcher* b, Slot& s) {
    const bool gpu = b->kind != ENGINE_CALLBACK;
    if (gpu) {
        if (s.q_host) (void)hipHostFree(s.q_host);
        if (s.s_host) (void)hipHostFree(s.s_host);
        if (s.i_host) (void)hipHostFree(s.i_host);
    } else {
        free(s.q_host);
        free(s.s_host);
        free(s.i_host);
    }
    if (s.q_dev) (void)hipFree(s.q_dev);
    if (s.sub_dev) (void)hipFree(s.sub_dev);
    free(s.sub_host);
    s = Slot();
}

// ---- the policy ------------------------------------------------------------------------------------------------------------------
// Called with b->mu held and at least one request pending.  Returns true = assemble and submit a batch now; false = wait (until
// `*until`, or for ever when `*until` == tp_t::max(); any arrival / completion wakes the scheduler and the decision is taken again).
//
//   * engine idle, no other recently active client is missing            -> now (a lone client never waits: no fixed window)
//   * engine idle, other clients that searched a moment ago have not come back yet, and the batch still fits one query tile
//     (<= flat_queries: the scan costs the same with them aboard)          -> wait for them, at most `grace` (a share of a measured
//                                                                            scan, capped) from the first arrival / the engine going idle
//   * engine busy, pending < flat_queries                                  -> keep collecting until the running batch completes
//                                                                            (enqueuing early would freeze a small batch that costs
//                                                                            a whole scan of its own)
//   * engine busy, pending >= flat_queries (time is linear in the batch)   -> enqueue BEHIND the running batch (no gap on the device).
//                                                                            (Holding instead, to scan ONCE for both groups when they
//                                                                            alternate, was tried in round 4: the scan's fixed cost at
//                                                                            256 queries is ~0.75 ms of 3.8 on a 10 M-row store against
//                                                                            a ~0.45 ms come-back gap - 0-4 %, unstable at the margin:
//                                                                            profiles/r04_ab_merge_groups.txt)
//   * pending >= max_queries                                               -> now, whenever a pipeline slot is free
bool make_decision(vodhip_batcher* b, tp_t now, tp_t* until) {
    *until = tp_t::max();
    const int in_flight = (int)b->inflight.size();
    // (the node index takes two searches in flight: vodhip_node_index_search_async / _finish, round 6; a host callback runs one at a time)
    const int max_depth = b->kind == ENGINE_INDEX ? (int)std::max<int64_t>(1, b->depth)
                          : b->kind == ENGINE_NODE ? (int)std::min<int64_t>(2, std::max<int64_t>(1, b->depth)) : 1;
    if (in_flight >= max_depth) return false;
    bool slot_ok = false;
    for (const Slot& s : b->slots) slot_ok = slot_ok || !s.busy;
    if (!slot_ok) return false;
    int64_t pend_q = 0;
    for (const Request* r : b->pending) pend_q += r->nq;
    const Request* first = b->pending.front();
    if (first->subset || first->nq >= b->max_queries) return true;  // runs alone: nothing to wait for
    if (pend_q >= b->max_queries) return true;
    const tp_t window_end = first->t_arrive + std::chrono::microseconds(b->window_us);
    if (in_flight > 0) {
        if (pend_q < b->flat_queries) return false;  // woken by the completion (or by the arrivals that make it a tile's worth)
        // More than one query tile is waiting: it will be enqueued BEHIND the running batch - but a batch is frozen the moment it is
        // enqueued, and a scan is cheapest per query when the batch is large and tile-aligned (32 clients x 64 queries: three groups of
        // ~685 queries ran at 78 % of the device rate with the engine never idle; two groups of 1024 run at 97 %).  So it is enqueued
        //   - at once when nobody else can join anyway: every recently active client is pending or inside a batch on the device, or
        //   - just in time: when the oldest running batch is about to complete (its duration is known from earlier batches of its size).
        int missing = 0;
        const auto active_window = std::chrono::nanoseconds((int64_t)std::max(5e6, 4.0 * (b->ema_flat_scan_ns + 1e6)));
        for (const auto& kv : b->clients) {
            // (recently ACTIVE = sent a request or was ANSWERED within the window: with 32 clients x 64 queries a client's cycle is ~27 ms, its
            // last arrival older than the window when its answer leaves - and it is the next to come back)
            if (now - std::max(kv.second.last_seen, kv.second.answered ? kv.second.last_done : kv.second.last_seen) > active_window) continue;
            bool has = false;
            for (const Request* r : b->pending) has = has || r->client == kv.first;
            for (const Batch* bt : b->inflight)
                for (const Request* r : bt->reqs) has = has || r->client == kv.first;
            if (!has) ++missing;
        }
        if (missing == 0) return true;
        const Batch* oldest = b->inflight.front();
        const double est = b->estimate_ns(oldest->nq);
        if (est <= 0.0) return true;  // no estimate yet for a batch of that size: round 3's behaviour (enqueue behind at once)
        const double lead = std::max(300e3, 0.06 * est) + 0.15e3 * (double)pend_q;  // host-side submission: ~0.15 us per query to stage + copy
        // (a batch that was itself enqueued behind another started to RUN when that one completed, not when it was submitted)
        const tp_t started = oldest->pipeline_was_empty ? oldest->t_submit : std::max(oldest->t_submit, b->t_last_completion);
        const tp_t jit = started + std::chrono::nanoseconds((int64_t)(est - lead));
        if (now >= jit) return true;
        *until = jit;
        return false;
    }
    if (b->window_us > 0 && now < window_end) {
        *until = window_end;
        return false;
    }
    if (pend_q >= b->flat_queries || b->grace_us <= 0 || b->ema_flat_scan_ns <= 0.0) return true;
    // expected company: clients that (a) searched within the last few cycles, (b) have nothing pending right now and (c) are DUE - their
    // usual come-back time after an answer (an EMA per client) ends before the grace does.  A worker that tokenises for 5 ms between
    // requests is not waited for (measured: waiting for it cost 8 workers with 2 ms pauses +0.85 ms per request and 9 % throughput);
    // closed-loop workers (come-back ~0.5 ms) are - that is what keeps them in ONE batch instead of two alternating half batches.
    const double grace_ns = std::min((double)b->grace_us * 1e3, b->ema_flat_scan_ns * (double)b->grace_pct / 100.0);
    const auto active_window = std::chrono::nanoseconds((int64_t)std::max(5e6, 4.0 * (b->ema_flat_scan_ns + grace_ns)));
    const tp_t deadline = std::max(first->t_arrive, b->t_idle_since) + std::chrono::nanoseconds((int64_t)grace_ns);
    int missing = 0;
    tp_t give_up = deadline;  // the earliest moment one of the awaited clients stops being plausible
    for (const auto& kv : b->clients) {
        const vodhip_batcher::Client& c = kv.second;
        if (!c.answered || c.ema_gap_ns <= 0.0 || now - std::max(c.last_seen, c.last_done) > active_window) continue;
        bool has = false;
        for (const Request* r : b->pending) has = has || r->client == kv.first;
        if (has) continue;
        const tp_t due = c.last_done + std::chrono::nanoseconds((int64_t)c.ema_gap_ns);
        const tp_t late = c.last_done + std::chrono::nanoseconds((int64_t)(2.0 * c.ema_gap_ns + 100e3));
        if (due > deadline || now > late) continue;  // not expected in time / overdue: somebody who went to do something else
        ++missing;
        give_up = std::min(give_up, late);
    }
    if (missing == 0) return true;
    if (now >= deadline) {
        ++b->n_grace_full;
        return true;
    }
    *until = std::max(give_up, now + std::chrono::microseconds(20));
    return false;
}

// assemble a batch from the head of `pending` (b->mu held); returns nullptr when no slot is free
Batch* assemble(vodhip_batcher* b) {
    Slot* slot = nullptr;
    for (Slot& s : b->slots)
        if (!s.busy) {
            slot = &s;
            break;
        }
    if (!slot) return nullptr;
    Batch* bt = new Batch();
    bt->slot = slot;
    slot->busy = true;
    Request* first = b->pending.front();
    bt->q_dtype = first->q_dtype;
    bt->subset = first->subset != nullptr;
    bt->n_subset = first->n_subset;
    // A batch enqueued BEHIND a running one is cut to whole query tiles when it spans more than one: the kernels pad a batch to a
    // multiple of 256 queries, so 11 requests of 64 queries (704 -> 768) or 21 (1344 -> 1536) waste 8-12 % of a scan, while the
    // requests left behind lose nothing - they ride with the running batch's callers, who are about to come back.  (An idle engine
    // takes everything: nothing is gained by leaving work behind then.)
    int64_t budget = b->max_queries;
    if (!b->inflight.empty() && !first->subset) {
        int64_t pend_q = 0;
        for (const Request* r : b->pending) pend_q += r->nq;
        const int64_t tile = std::max<int64_t>(1, b->flat_queries);
        if (pend_q > tile && pend_q % tile != 0) budget = std::min(budget, pend_q / tile * tile);
    }
    // requests are taken in arrival order; one that does not fit this batch (other dtype, subset filter, no room) ends the batch
    while (!b->pending.empty()) {
        Request* r = b->pending.front();
        if (!bt->reqs.empty()) {
            if (bt->subset || r->subset || r->q_dtype != bt->q_dtype || bt->nq + r->nq > budget) break;
        }
        b->pending.pop_front();
        r->batch = bt;
        r->row0 = bt->nq;
        bt->nq += r->nq;
        bt->k = std::max(bt->k, r->k);
        bt->reqs.push_back(r);
    }
    return bt;
}

// submit (no lock held): copy the queries into the slot, hand the batch to the engine.  Synchronous engines complete it here.
int submit(vodhip_batcher* b, Batch* bt) {
    Slot& s = *bt->slot;
    const size_t es = (size_t)elem_bytes(bt->q_dtype);
    const size_t row_bytes = (size_t)b->dim * es;
    // the slot's device buffers must live on the INDEX's device: select it before the first allocation (a fresh thread starts on device 0)
    if (b->kind != ENGINE_CALLBACK) SHIP(hipSetDevice(b->device));
    if (slot_reserve(b, s, (size_t)bt->nq * row_bytes, (size_t)bt->nq * bt->k, bt->subset ? (size_t)bt->nq * bt->n_subset : 0)) return -1;
    for (Request* r : bt->reqs) memcpy((char*)s.q_host + (size_t)r->row0 * row_bytes, r->q, (size_t)r->nq * row_bytes);
    if (bt->subset) memcpy(s.sub_host, bt->reqs[0]->subset, (size_t)bt->nq * bt->n_subset * sizeof(int32_t));
    if (b->kind == ENGINE_INDEX) {
        SHIP(hipMemcpyAsync(s.q_dev, s.q_host, (size_t)bt->nq * row_bytes, hipMemcpyHostToDevice, b->stream));
        if (bt->subset) {
            SHIP(hipMemcpyAsync(s.sub_dev, s.sub_host, (size_t)bt->nq * bt->n_subset * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
            if (vodhip_index_set_query_labels(b->index, s.sub_dev, bt->n_subset)) return -1;
        }
        const int rc = vodhip_index_search_async(b->index, s.q_dev, bt->q_dtype, bt->nq, bt->k, b->id_base, s.s_host, s.i_host, b->stream);
        if (bt->subset) (void)vodhip_index_set_query_labels(b->index, nullptr, 0);  // the search captured its labels at enqueue time
        return rc;
    }
    if (b->kind == ENGINE_NODE) {
        if (bt->subset && vodhip_node_index_set_query_labels(b->node, s.sub_host, bt->n_subset, VODHIP_HOST)) return -1;
        // (enqueued on every shard, not waited for: the completion thread finishes it while the scheduler assembles the next batch)
        const int rc = vodhip_node_index_search_async(b->node, s.q_host, bt->q_dtype, bt->nq, bt->k, VODHIP_HOST, s.s_host, s.i_host, nullptr);
        if (bt->subset) (void)vodhip_node_index_set_query_labels(b->node, nullptr, 0, VODHIP_HOST);  // the search staged its labels at enqueue time
        return rc;
    }
    // callback engine: float32 queries only (the host's engine interface: `search(np.float32[nq, d], k)`)
    const float* q32 = (const float*)s.q_host;
    std::vector<float> widened;
    if (bt->q_dtype == VODHIP_F16) {
        widened.resize((size_t)bt->nq * b->dim);
        const _Float16* h = (const _Float16*)s.q_host;
        for (size_t i = 0; i < widened.size(); ++i) widened[i] = (float)h[i];
        q32 = widened.data();
    }
    const int rc = b->fn(b->fn_user, q32, bt->nq, bt->k, bt->subset ? s.sub_host : nullptr, bt->n_subset, s.s_host, s.i_host);
    if (rc) return sfail("the host's search callback failed (status %d)", rc);
    return 0;
}

void complete_locked(vodhip_batcher* b, Batch* bt, tp_t now) {
    const double dur = (double)ns_between(bt->t_submit, now);
    if (bt->rc == 0 && bt->pipeline_was_empty && bt->nq <= b->flat_queries && !bt->subset)
        b->ema_flat_scan_ns = b->ema_flat_scan_ns <= 0.0 ? dur : 0.8 * b->ema_flat_scan_ns + 0.2 * dur;
    if (bt->rc == 0 && !bt->subset && bt->nq <= 2048) {
        // a batch that started on an idle engine ran `dur`; one that was enqueued behind another ran from its predecessor's completion
        const double own = bt->pipeline_was_empty ? dur : (double)ns_between(std::max(bt->t_submit, b->t_last_completion), now);
        const int t = (int)std::min<int64_t>(8, std::max<int64_t>(1, (bt->nq + 255) / 256));
        // the FIRST batch of a size pays for growing the library's workspace and the staging slot (a 512-query batch right after
        // 256-query ones: 11.7 ms against 6.9 from then on - measured; that one sample kept the merge rule off for good): not recorded
        if (b->tiles_samples[t]++ > 0) {
            double& e = b->ema_tiles_ns[t];
            e = e <= 0.0 ? own : (own < e ? 0.5 * e + 0.5 * own : 0.7 * e + 0.3 * own);
        }
    }
    b->t_last_completion = now;
    bt->slot->users = (int)bt->reqs.size();
    for (Request* r : bt->reqs) {
        r->done = true;
        if (r->client) {  // its answer exists from NOW: the moment its come-back time is counted from (set here, under the lock, before the
                          // scheduler looks at the queue again - the callers themselves wake up later)
            auto it = b->clients.find(r->client);
            if (it != b->clients.end()) {
                it->second.last_done = now;
                it->second.answered = true;
            }
        }
    }
    if (b->inflight.empty()) {
        b->busy_ns += ns_between(b->t_busy_since, now);
        b->t_idle_since = now;
    }
    b->cv_done.notify_all();
    b->cv_sched.notify_all();
}

void scheduler_main(vodhip_batcher* b) {
    if (b->kind != ENGINE_CALLBACK) (void)hipSetDevice(b->device);  // this thread allocates the slots' buffers and enqueues the searches
    std::unique_lock<std::mutex> lk(b->mu);
    for (;;) {
        while (!b->stop && b->pending.empty()) b->cv_sched.wait(lk);
        if (b->stop) return;
        tp_t until;
        const tp_t now = clock_t_::now();
        if (!make_decision(b, now, &until)) {
            if (until == tp_t::max()) {
                b->cv_sched.wait(lk);
            } else {
                ++b->n_grace_waits;
                b->cv_sched.wait_until(lk, until);
            }
            continue;
        }
        Batch* bt = assemble(b);
        if (!bt) {
            b->cv_sched.wait(lk);
            continue;
        }
        bt->t_submit = clock_t_::now();
        bt->pipeline_was_empty = b->inflight.empty();
        if (b->inflight.empty()) {
            b->idle_ns += ns_between(b->t_idle_since, bt->t_submit);
            b->t_busy_since = bt->t_submit;
        }
        b->inflight.push_back(bt);
        ++b->n_batches;
        b->n_requests += (int64_t)bt->reqs.size();
        b->n_queries += bt->nq;
        b->n_fused_max = std::max<int64_t>(b->n_fused_max, (int64_t)bt->reqs.size());
        b->last_batch_queries = bt->nq;
        b->last_batch_requests = (int64_t)bt->reqs.size();
        lk.unlock();
        const int rc = submit(b, bt);
        lk.lock();
        if (rc) {
            bt->rc = -1;
            bt->err = vodhip_last_error();
        }
        bt->submitted = rc == 0;
        if (b->kind == ENGINE_CALLBACK || rc) {
            // the synchronous engine (and failed submissions) complete here
            auto it = std::find(b->inflight.begin(), b->inflight.end(), bt);
            if (it != b->inflight.end()) b->inflight.erase(it);
            complete_locked(b, bt, clock_t_::now());
        } else {
            b->cv_compl.notify_one();
        }
    }
}

void completion_main(vodhip_batcher* b) {
    (void)hipSetDevice(b->device);
    std::unique_lock<std::mutex> lk(b->mu);
    for (;;) {
        // the oldest batch, once the library has accepted it (`submit` runs outside the lock: a batch sits in `inflight` before that)
        while (!b->stop && !(b->inflight.size() && b->inflight.front()->submitted)) b->cv_compl.wait(lk);
        if (b->stop) return;
        Batch* bt = b->inflight.front();
        lk.unlock();
        // waits for THIS search's event only (younger batches keep the device busy); if a candidate list overflowed, the library
        // runs its per-query recovery passes here - under its own per-handle lock, so the scheduler may keep enqueuing
        const int rc = b->kind == ENGINE_NODE ? vodhip_node_index_search_finish(b->node) : vodhip_index_search_finish(b->index, b->stream);
        const std::string err = rc ? vodhip_last_error() : "";
        lk.lock();
        b->inflight.pop_front();
        if (rc) {
            bt->rc = -1;
            bt->err = err;
        }
        complete_locked(b, bt, clock_t_::now());
    }
}

}  // namespace

extern "C" {

int vodhip_batcher_create(vodhip_index_t* index, vodhip_node_index_t* node, vodhip_search_fn fn, void* user, int64_t dim, int64_t id_base,
                          vodhip_batcher_t** out) {
    if (!out) return sfail("out is NULL");
    const int n_engines = (index != nullptr) + (node != nullptr) + (fn != nullptr);
    if (n_engines != 1) return sfail("exactly one of index / node / fn selects the engine");
    if (dim <= 0) return sfail("invalid dim=%lld", (long long)dim);
    vodhip_batcher* b = new vodhip_batcher();
    b->kind = index ? ENGINE_INDEX : (node ? ENGINE_NODE : ENGINE_CALLBACK);
    b->index = index;
    b->node = node;
    b->fn = fn;
    b->fn_user = user;
    b->dim = dim;
    b->id_base = id_base;
    if (index) {
        int64_t idim = 0;
        void* data = nullptr;
        int64_t stride = 0;
        int dt = 0;
        if (vodhip_index_dim(index, &idim) || idim != dim) {
            delete b;
            return sfail("dim=%lld does not match the index (dim %lld)", (long long)dim, (long long)idim);
        }
        // the device of the handle: where its store lives
        hipPointerAttribute_t attr;
        if (vodhip_index_data(index, &data, &stride, &dt) == 0 && data && hipPointerGetAttributes(&attr, data) == hipSuccess) b->device = attr.device;
        (void)hipGetLastError();
        int caller_device = -1;
        if (hipGetDevice(&caller_device) != hipSuccess) caller_device = -1;
        hipError_t e = hipSetDevice(b->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
        if (caller_device >= 0 && caller_device != b->device) (void)hipSetDevice(caller_device);  // the caller's current device is its own
        if (e != hipSuccess) {
            delete b;
            return sfail("creating the batcher's stream failed: %s", hipGetErrorString(e));
        }
    }
    if (node) {  // the node index merges on its first shard's device: the slots' pinned buffers are allocated with that device current
        vodhip_index_t* sh = nullptr;
        int64_t base = 0;
        int dev0 = 0;
        if (vodhip_node_index_shard(node, 0, &sh, &base, &dev0) == 0) b->device = dev0;
    }
    b->slots.resize(4);  // depth (<= 3) batches on the device + one being assembled / copied out
    b->t_idle_since = clock_t_::now();
    b->th_sched = std::thread(scheduler_main, b);
    if (b->kind != ENGINE_CALLBACK) b->th_compl = std::thread(completion_main, b);
    *out = b;
    return 0;
}

int vodhip_batcher_destroy(vodhip_batcher_t* b) {
    if (!b) return 0;
    {
        std::unique_lock<std::mutex> lk(b->mu);
        b->closing = true;  // (a search that arrives from now on is refused: nothing may join `pending` behind the scheduler's back)
        // requests that were never assembled fail; batches on the device are completed first
        for (Request* r : b->pending) {
            r->batch = nullptr;
            r->done = true;
        }
        b->pending.clear();
        b->cv_done.notify_all();
        while (!b->inflight.empty()) b->cv_sched.wait_for(lk, std::chrono::milliseconds(5));
        b->stop = true;
        b->cv_sched.notify_all();
        b->cv_compl.notify_all();
    }
    if (b->th_sched.joinable()) b->th_sched.join();
    if (b->th_compl.joinable()) b->th_compl.join();
    {
        std::unique_lock<std::mutex> lk(b->mu);
        for (;;) {  // callers still copying their rows out / on their way out of vodhip_batcher_search (they touch the handle's mutex last)
            bool used = b->callers > 0;
            for (const Slot& s : b->slots) used = used || s.busy;
            if (!used) break;
            b->cv_sched.wait_for(lk, std::chrono::milliseconds(5));
        }
    }
    if (b->kind != ENGINE_CALLBACK) (void)hipSetDevice(b->device);
    for (Slot& s : b->slots) slot_free(b, s);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return 0;
}

int vodhip_batcher_set_param(vodhip_batcher_t* b, const char* key, int64_t value) {
    if (!b || !key) return sfail("NULL argument");
    std::lock_guard<std::mutex> lk(b->mu);
    if (!strcmp(key, "max_queries")) {
        if (value < 1) return sfail("max_queries must be >= 1");
        b->max_queries = value;
    } else if (!strcmp(key, "flat_queries")) {
        b->flat_queries = std::max<int64_t>(1, value);
    } else if (!strcmp(key, "grace_us")) {
        b->grace_us = std::max<int64_t>(0, value);
    } else if (!strcmp(key, "grace_pct")) {
        b->grace_pct = std::min<int64_t>(100, std::max<int64_t>(0, value));
    } else if (!strcmp(key, "window_us")) {
        b->window_us = std::max<int64_t>(0, value);
    } else if (!strcmp(key, "depth")) {
        if (value < 1 || value > 3) return sfail("depth must be in [1, 3]");
        b->depth = value;
    } else {
        return sfail("unknown batcher parameter '%s'", key);
    }
    b->cv_sched.notify_all();
    return 0;
}

int vodhip_batcher_get_stat(vodhip_batcher_t* b, const char* key, int64_t* out) {
    if (!b || !key || !out) return sfail("NULL argument");
    std::lock_guard<std::mutex> lk(b->mu);
    if (!strcmp(key, "batches")) *out = b->n_batches;
    else if (!strcmp(key, "requests")) *out = b->n_requests;
    else if (!strcmp(key, "queries")) *out = b->n_queries;
    else if (!strcmp(key, "fused_requests_max")) *out = b->n_fused_max;
    else if (!strcmp(key, "grace_waits")) *out = b->n_grace_waits;
    else if (!strcmp(key, "grace_expired")) *out = b->n_grace_full;
    else if (!strcmp(key, "idle_ns")) *out = b->idle_ns;
    else if (!strcmp(key, "busy_ns")) *out = b->busy_ns;
    else if (!strcmp(key, "last_batch_queries")) *out = b->last_batch_queries;
    else if (!strcmp(key, "last_batch_requests")) *out = b->last_batch_requests;
    else if (!strcmp(key, "flat_scan_ns")) *out = (int64_t)b->ema_flat_scan_ns;
    else if (!strncmp(key, "tiles_ns_", 9) && key[9] >= '1' && key[9] <= '8' && !key[10]) *out = (int64_t)b->ema_tiles_ns[key[9] - '0'];
    else if (!strcmp(key, "in_flight")) *out = (int64_t)b->inflight.size();
    else if (!strcmp(key, "pending")) *out = (int64_t)b->pending.size();
    else if (!strcmp(key, "active_clients")) *out = (int64_t)b->clients.size();
    else return sfail("unknown batcher stat '%s'", key);
    return 0;
}

int vodhip_batcher_forget_client(vodhip_batcher_t* b, uint64_t client) {
    if (!b) return sfail("NULL argument");
    std::lock_guard<std::mutex> lk(b->mu);
    b->clients.erase(client);
    b->cv_sched.notify_all();  // somebody may be waiting for exactly this client
    return 0;
}

int vodhip_batcher_search(vodhip_batcher_t* b, const void* queries, int q_dtype, int64_t nq, int k, const int32_t* subset, int n_subset,
                          uint64_t client, float* out_scores, int64_t* out_ids) {
    if (!b) return sfail("batcher is NULL");
    if (k < 1 || k > VODHIP_MAX_K) return sfail("k=%d out of range [1, %d]", k, VODHIP_MAX_K);
    if (nq < 0 || (nq > 0 && (!queries || !out_scores || !out_ids))) return sfail("invalid query / output pointers");
    if (q_dtype != VODHIP_F32 && q_dtype != VODHIP_F16 && !(q_dtype == VODHIP_BF16 && b->kind != ENGINE_CALLBACK)) return sfail("invalid q_dtype %d", q_dtype);
    if (subset && (n_subset < 1 || n_subset > 64)) return sfail("n_subset must be in [1, 64]");
    if (nq == 0) return 0;
    Request r;
    r.q = queries;
    r.q_dtype = q_dtype;
    r.nq = nq;
    r.k = k;
    r.subset = subset;
    r.n_subset = subset ? n_subset : 0;
    r.client = client;
    r.out_s = out_scores;
    r.out_i = out_ids;
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->stop || b->closing) return sfail("the batcher is shutting down");
    ++b->callers;
    struct Leave {  // runs before `lk` unlocks on every way out: the handle stays alive until the count is back to zero
        vodhip_batcher* b;
        std::unique_lock<std::mutex>& lk;
        ~Leave() {
            if (!lk.owns_lock()) lk.lock();
            --b->callers;
        }
    } leave{b, lk};
    r.t_arrive = clock_t_::now();
    if (client) {
        vodhip_batcher::Client& c = b->clients[client];
        if (c.answered) {  // how long this client took to come back after its last answer
            const double gap = (double)ns_between(c.last_done, r.t_arrive);
            c.ema_gap_ns = c.ema_gap_ns <= 0.0 ? gap : 0.7 * c.ema_gap_ns + 0.3 * gap;
        }
        c.last_seen = r.t_arrive;
    }
    if (b->clients.size() > 4096) {  // tags of clients long gone (no forget call): keep the table small
        for (auto it = b->clients.begin(); it != b->clients.end();)
            it = (r.t_arrive - it->second.last_seen > std::chrono::seconds(10)) ? b->clients.erase(it) : std::next(it);
    }
    b->pending.push_back(&r);
    b->cv_sched.notify_all();
    b->cv_done.wait(lk, [&] { return r.done; });
    Batch* bt = r.batch;
    if (!bt) return sfail("the batcher was shut down before the request ran");
    Slot* s = bt->slot;
    const int rc = bt->rc;
    const std::string err = bt->err;
    const int kb = bt->k;
    lk.unlock();
    if (rc == 0) {
        // every caller copies its own rows (first k of the fused batch's k_max columns: a prefix of a top-k' list is the top-k)
        for (int64_t i = 0; i < nq; ++i) {
            memcpy(out_scores + i * k, s->s_host + (size_t)(r.row0 + i) * kb, (size_t)k * sizeof(float));
            memcpy(out_ids + i * k, s->i_host + (size_t)(r.row0 + i) * kb, (size_t)k * sizeof(int64_t));
        }
    }
    lk.lock();
    if (--s->users == 0) {
        s->busy = false;
        delete bt;
        b->cv_sched.notify_all();
    }
    if (rc) {
        vodhip::set_last_error(err.c_str());
        return -1;
    }
    return 0;
}

}  // extern "C"

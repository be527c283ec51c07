// vodhip_node_index_*: the row-sharded index of ONE node driven from ONE process, for consumers that have no
// torch.distributed (C, C++, cgo, JNI): every device holds a contiguous row range, a search runs on all devices at once and the
// per-shard top-k lists meet on devices[0] (peer copies over xGMI) where vodhip_merge_topk folds them.
// Counterpart of faiss.index_cpu_to_all_gpus(index, co) with co.shard = True (/root/reference/src/vod_search/faiss_search/
// server.py:51-54, vod_configs/search.py:80).  A composition of the PUBLIC entry points of include/vodhip.h - nothing here reaches
// into a shard's internals.  (The Python host shards with one process per GPU and one RCCL all-gather instead: vod_amd/distributed.py.)
#include "../../include/vodhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "vodhip_internal.h"

namespace {

int nfail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    vodhip::set_last_error(buf);
    return -1;
}

#define NODE_HIP_OK(expr)                                                                                               \
    do {                                                                                                                \
        hipError_t _e = (expr);                                                                                         \
        if (_e != hipSuccess) {                                                                                         \
            (void)hipGetLastError();                                                                                    \
            return nfail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);                   \
        }                                                                                                               \
    } while (0)

inline size_t elem_bytes(int dtype) { return dtype == VODHIP_F32 ? 4 : 2; }

struct ShardBuffers {
    void* q = nullptr;          // the query batch on this shard's device
    float* scores = nullptr;    // this shard's top-k [nq, k]
    int64_t* ids = nullptr;
    int32_t* q_labels = nullptr;  // the batch's allowed subset labels on this shard's device
    size_t q_bytes = 0, res_elems = 0, q_label_elems = 0;
};

}  // namespace

// Everything ONE search in flight owns (round 6: two slots, so that the host enqueues batch i + 1 on every shard - 7 launches per shard -
// while batch i runs: with 8 shards the enqueue alone is ~0.3 ms of host time per batch, which a synchronous search adds to every step)
struct NodeSlot {
    std::vector<ShardBuffers> buf;     // per shard: queries, top-k list, subset labels on its device
    std::vector<hipEvent_t> arrived;   // shard g's result is on devices[0]
    hipEvent_t ready = nullptr;        // on devices[0]: the caller's queries are readable / the slot's previous merge has read `gathered`
    float* gathered_scores = nullptr;  // devices[0]: [n, nq, k]
    int64_t* gathered_ids = nullptr;
    float* merged_scores = nullptr;    // devices[0]: [nq, k] (host-located outputs)
    int64_t* merged_ids = nullptr;
    size_t gathered_elems = 0, merged_elems = 0;
    std::vector<void*> pin_res;        // per shard: pinned [scores f32 | ids i64] of its top-k list (host-staged route)
    std::vector<size_t> pin_res_elems;
    std::vector<hipEvent_t> on_host;   // shard g's list is in pin_res[g]
    void* pin_q = nullptr;             // the query batch (+ subset labels) staged from devices[0]
    size_t pin_q_bytes = 0;
    hipEvent_t q_on_host = nullptr;
    hipEvent_t merge_ev[2] = {nullptr, nullptr};   // param "profile": on devices[0]
    std::vector<hipEvent_t> copy_ev;               // [2 * n]: begin / end of shard g's copy, on its device
    bool timed = false;                            // the events above belong to this slot's last search
    std::vector<char> enqueued;        // shard g holds this slot's search in its FIFO (drained if a call fails half-way)
    // the search in flight
    bool pending = false;
    int64_t nq = 0;
    int k = 0, location = 0;
    float* out_scores = nullptr;
    int64_t* out_ids = nullptr;
    hipStream_t user = nullptr;
};

struct vodhip_node_index {
    int n = 0;
    int64_t dim = 0, capacity = 0, rows_per_shard = 0, ntotal = 0;
    int dtype = 0;
    std::vector<int> device;
    std::vector<vodhip_index_t*> shard;
    std::vector<hipStream_t> stream;   // one per shard, on its device: the shard's searches
    // a finished shard's list leaves on a stream of its own: with two searches in flight the shard's search stream already holds the NEXT
    // search's launches, and a copy enqueued behind them would hold this search's merge back by a whole batch
    std::vector<hipStream_t> copy_stream;
    hipStream_t merge_stream = nullptr;  // on devices[0]: merge + D2H of HOST-located searches (DEVICE-located ones merge on the caller's stream)
    NodeSlot slot[2];
    int next_slot = 0, n_pending = 0;  // FIFO: the oldest pending search sits in slot (next_slot - n_pending) & 1
    int last_finished = 0;             // the slot whose profile events `get_stat` reads
    bool finishing = false;            // a `search_finish` is running (outside the mutex: it waits for the devices) on the oldest slot
    const int32_t* q_labels = nullptr;  // the caller's per-query labels for the next searches (host, or devices[0])
    int q_labels_per_query = 0, q_labels_location = VODHIP_HOST;
    bool has_row_labels = false;
    // topology (read once at create): can devices[0] and shard g's device copy into each other's memory directly?  A shard without
    // peer access - or every shard but the first with `host_staging` set (bring-up, tests on a 1-GPU box) - exchanges its queries and
    // its top-k list with devices[0] through pinned host memory instead (two DMA hops over PCIe, no xGMI)
    std::vector<int> peer_ok;          // 2 = same device as devices[0], 1 = peer access both ways, 0 = none
    int64_t host_staging = 0;
    // add / reset / label changes / searches on one handle are serialised (the staging buffers and the shards' FIFOs are shared); a
    // `set_query_labels` + `search` pair is two calls: callers that filter from several threads go through a vodhip_batcher
    // param "profile" = 1: HIP events bracket the merge (recorded once every shard's list has arrived on devices[0]) and every shard's
    // copy of its list towards devices[0] - `vodhip_node_index_get_stat` reads them: "last_merge_ns", "last_copy_ns_max"
    int64_t profile = 0;
    std::mutex mu;
    bool staged(int g) const { return g > 0 && (host_staging != 0 || (device[g] != device[0] && peer_ok[g] == 0)); }
};

namespace {

int ensure(void** p, size_t* have, size_t want, size_t elem) {
    if (*have >= want && *p) return 0;
    if (*p) NODE_HIP_OK(hipFree(*p));
    *p = nullptr;
    *have = 0;
    const size_t n = want + want / 4 + 64;  // a little headroom: batch sizes wobble
    NODE_HIP_OK(hipMalloc(p, n * elem));
    *have = n;
    return 0;
}

}  // namespace

extern "C" {

int vodhip_node_index_create(int n_devices, const int* devices, int64_t dim, int store_dtype, int64_t capacity_rows,
                             vodhip_node_index_t** out) {
    if (!out) return nfail("out is NULL");
    if (n_devices < 1 || n_devices > 64 || !devices) return nfail("n_devices=%d out of range [1, 64]", n_devices);
    if (capacity_rows < 0) return nfail("invalid capacity=%lld", (long long)capacity_rows);
    vodhip_node_index* nx = new vodhip_node_index();
    nx->n = n_devices;
    nx->dim = dim;
    nx->dtype = store_dtype;
    nx->capacity = capacity_rows;
    nx->rows_per_shard = (capacity_rows + n_devices - 1) / n_devices;
    nx->device.assign(devices, devices + n_devices);
    nx->shard.assign(n_devices, nullptr);
    nx->stream.assign(n_devices, nullptr);
    nx->copy_stream.assign(n_devices, nullptr);
    for (NodeSlot& S : nx->slot) {
        S.arrived.assign(n_devices, nullptr);
        S.buf.resize(n_devices);
        S.pin_res.assign(n_devices, nullptr);
        S.pin_res_elems.assign(n_devices, 0);
        S.on_host.assign(n_devices, nullptr);
    }
    nx->peer_ok.assign(n_devices, 2);
    auto bail = [&](int rc) {
        const std::string keep = vodhip_last_error();
        vodhip_node_index_destroy(nx);
        vodhip::set_last_error(keep.c_str());
        return rc;
    };
    for (int g = 0; g < n_devices; ++g) {
        const int64_t lo = std::min(capacity_rows, g * nx->rows_per_shard), hi = std::min(capacity_rows, lo + nx->rows_per_shard);
        if (vodhip_index_create(devices[g], dim, store_dtype, hi - lo, &nx->shard[g])) return bail(-1);
        hipError_t e = hipSetDevice(devices[g]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&nx->stream[g], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&nx->copy_stream[g], hipStreamNonBlocking);
        for (NodeSlot& S : nx->slot)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&S.arrived[g], hipEventDisableTiming);
        if (e != hipSuccess) return bail(nfail("stream / event creation on device %d failed: %s", devices[g], hipGetErrorString(e)));
    }
    hipError_t e = hipSetDevice(devices[0]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&nx->merge_stream, hipStreamNonBlocking);
    for (NodeSlot& S : nx->slot) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.q_on_host, hipEventDisableTiming);
    }
    if (e != hipSuccess) return bail(nfail("event creation failed: %s", hipGetErrorString(e)));
    // topology: direct peer copies where both directions are possible (enabled here; "already enabled" is fine), host staging elsewhere
    for (int g = 0; g < n_devices; ++g) {
        if (devices[g] == devices[0]) continue;
        int to = 0, from = 0;
        if (hipDeviceCanAccessPeer(&to, devices[0], devices[g]) != hipSuccess) to = 0;
        if (hipDeviceCanAccessPeer(&from, devices[g], devices[0]) != hipSuccess) from = 0;
        nx->peer_ok[g] = (to && from) ? 1 : 0;
        if (nx->peer_ok[g]) {
            (void)hipSetDevice(devices[0]);
            hipError_t pe = hipDeviceEnablePeerAccess(devices[g], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) nx->peer_ok[g] = 0;
            (void)hipSetDevice(devices[g]);
            pe = hipDeviceEnablePeerAccess(devices[0], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) nx->peer_ok[g] = 0;
            (void)hipGetLastError();
        }
        if (nx->peer_ok[g]) {
            // the runtime said yes: prove it with a 256-byte round trip devices[0] -> devices[g] -> devices[0] before any search depends
            // on it (a node whose fabric is half configured answers "can access" and then fails the copy: host staging still works)
            void *a = nullptr, *b2 = nullptr;
            unsigned char probe[256], back[256];
            for (int i = 0; i < 256; ++i) probe[i] = (unsigned char)(i * 7 + g);
            memset(back, 0, sizeof(back));
            hipError_t pe = hipSetDevice(devices[0]);
            if (pe == hipSuccess) pe = hipMalloc(&a, 512);
            if (pe == hipSuccess) pe = hipSetDevice(devices[g]);
            if (pe == hipSuccess) pe = hipMalloc(&b2, 256);
            if (pe == hipSuccess) pe = hipSetDevice(devices[0]);
            if (pe == hipSuccess) pe = hipMemcpy(a, probe, 256, hipMemcpyHostToDevice);
            if (pe == hipSuccess) pe = hipMemcpyPeerAsync(b2, devices[g], a, devices[0], 256, nx->stream[0]);
            if (pe == hipSuccess) pe = hipStreamSynchronize(nx->stream[0]);
            if (pe == hipSuccess) pe = hipSetDevice(devices[g]);
            if (pe == hipSuccess) pe = hipMemcpyPeerAsync((char*)a + 256, devices[0], b2, devices[g], 256, nx->stream[g]);
            if (pe == hipSuccess) pe = hipStreamSynchronize(nx->stream[g]);
            if (pe == hipSuccess) pe = hipSetDevice(devices[0]);
            if (pe == hipSuccess) pe = hipMemcpy(back, (char*)a + 256, 256, hipMemcpyDeviceToHost);
            if (pe != hipSuccess || memcmp(probe, back, 256) != 0) nx->peer_ok[g] = 0;
            (void)hipSetDevice(devices[0]);
            if (a) (void)hipFree(a);
            (void)hipSetDevice(devices[g]);
            if (b2) (void)hipFree(b2);
            (void)hipGetLastError();
        }
    }
    (void)hipSetDevice(devices[0]);
    *out = nx;
    return 0;
}

int vodhip_node_index_peer_access(const vodhip_node_index_t* nx, int* out, int n) {
    if (!nx || !out || n < nx->n) return nfail("invalid arguments");
    for (int g = 0; g < nx->n; ++g) out[g] = nx->staged(g) ? 0 : nx->peer_ok[g];
    return nx->n;
}

int vodhip_node_index_destroy(vodhip_node_index_t* nx) {
    if (!nx) return 0;
    for (int g = 0; g < nx->n; ++g) {
        if (!nx->shard[g]) continue;  // create() failed before this shard existed: its device may not exist either
        (void)hipSetDevice(nx->device[g]);
        if (nx->stream[g]) (void)hipStreamSynchronize(nx->stream[g]);
        if (nx->shard[g]) (void)vodhip_index_destroy(nx->shard[g]);
        for (NodeSlot& S : nx->slot) {
            (void)hipFree(S.buf[g].q);
            (void)hipFree(S.buf[g].scores);
            (void)hipFree(S.buf[g].ids);
            (void)hipFree(S.buf[g].q_labels);
            if (S.arrived[g]) (void)hipEventDestroy(S.arrived[g]);
            if (S.on_host[g]) (void)hipEventDestroy(S.on_host[g]);
            if (S.pin_res[g]) (void)hipHostFree(S.pin_res[g]);
        }
        if (nx->copy_stream[g]) {
            (void)hipStreamSynchronize(nx->copy_stream[g]);
            (void)hipStreamDestroy(nx->copy_stream[g]);
        }
        if (nx->stream[g]) (void)hipStreamDestroy(nx->stream[g]);
    }
    if (nx->n && nx->shard[0]) (void)hipSetDevice(nx->device[0]);
    if (nx->merge_stream) {
        (void)hipStreamSynchronize(nx->merge_stream);
        (void)hipStreamDestroy(nx->merge_stream);
    }
    for (NodeSlot& S : nx->slot) {
        if (S.pin_q) (void)hipHostFree(S.pin_q);
        if (S.q_on_host) (void)hipEventDestroy(S.q_on_host);
        for (hipEvent_t e : S.copy_ev)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : S.merge_ev)
            if (e) (void)hipEventDestroy(e);
        (void)hipFree(S.gathered_scores);
        (void)hipFree(S.gathered_ids);
        (void)hipFree(S.merged_scores);
        (void)hipFree(S.merged_ids);
        if (S.ready) (void)hipEventDestroy(S.ready);
    }
    delete nx;
    (void)hipGetLastError();  // best-effort teardown: nothing of it may resurface in the caller's next launch check
    return 0;
}

int vodhip_node_index_add(vodhip_node_index_t* nx, const void* rows, int64_t n_rows, int src_dtype) {
    if (!nx) return nfail("index is NULL");
    std::lock_guard<std::mutex> guard(nx->mu);
    if (n_rows < 0 || (n_rows > 0 && !rows)) return nfail("invalid rows");
    if (src_dtype < 0 || src_dtype > 2) return nfail("invalid src_dtype %d", src_dtype);
    if (nx->n_pending) return nfail("%d searches are in flight: finish them before adding rows", nx->n_pending);
    if (nx->ntotal + n_rows > nx->capacity)
        return nfail("index full: ntotal=%lld + %lld > capacity=%lld", (long long)nx->ntotal, (long long)n_rows, (long long)nx->capacity);
    // global rows [ntotal, ntotal + n_rows): the part inside shard g's range goes to shard g; the shards ingest concurrently
    // (one host thread each: every device has its own DMA engines and PCIe link)
    struct Part { int g; int64_t first, count; };
    std::vector<Part> parts;
    for (int g = 0; g < nx->n; ++g) {
        const int64_t lo = g * nx->rows_per_shard, hi = lo + nx->rows_per_shard;
        const int64_t a = std::max(lo, nx->ntotal), b = std::min(hi, nx->ntotal + n_rows);
        if (b > a) parts.push_back({g, a - nx->ntotal, b - a});
    }
    std::vector<std::string> errors(parts.size());
    std::vector<int> rcs(parts.size(), 0);
    auto work = [&](size_t i) {
        const Part& p = parts[i];
        const char* src = (const char*)rows + (size_t)p.first * (size_t)nx->dim * elem_bytes(src_dtype);
        rcs[i] = vodhip_index_add(nx->shard[p.g], src, p.count, src_dtype, VODHIP_HOST, nx->stream[p.g]);
        if (rcs[i]) errors[i] = vodhip_last_error();  // thread-local text: carried back to the caller's thread
    };
    if (parts.size() == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t i = 0; i < parts.size(); ++i) pool.emplace_back(work, i);
        for (auto& t : pool) t.join();
    }
    for (size_t i = 0; i < parts.size(); ++i)
        if (rcs[i]) return nfail("shard %d: %s", parts[i].g, errors[i].c_str());
    nx->ntotal += n_rows;
    return 0;
}

int vodhip_node_index_reset(vodhip_node_index_t* nx) {
    if (!nx) return nfail("index is NULL");
    std::lock_guard<std::mutex> guard(nx->mu);
    if (nx->n_pending) return nfail("%d searches are in flight: finish them before resetting the index", nx->n_pending);
    for (int g = 0; g < nx->n; ++g)
        if (vodhip_index_reset(nx->shard[g])) return -1;
    nx->ntotal = 0;
    return 0;
}

int vodhip_node_index_ntotal(const vodhip_node_index_t* nx, int64_t* out) {
    if (!nx || !out) return nfail("NULL argument");
    *out = nx->ntotal;
    return 0;
}

int vodhip_node_index_n_shards(const vodhip_node_index_t* nx) { return nx ? nx->n : nfail("index is NULL"); }

int vodhip_node_index_shard(vodhip_node_index_t* nx, int g, vodhip_index_t** shard, int64_t* id_base, int* device) {
    if (!nx) return nfail("index is NULL");
    if (g < 0 || g >= nx->n) return nfail("shard %d out of range [0, %d)", g, nx->n);
    if (shard) *shard = nx->shard[g];
    if (id_base) *id_base = g * nx->rows_per_shard;
    if (device) *device = nx->device[g];
    return 0;
}

int vodhip_node_index_set_row_labels(vodhip_node_index_t* nx, const int32_t* labels, int64_t n_rows) {
    if (!nx) return nfail("index is NULL");
    if (labels && (n_rows < 0 || n_rows > nx->capacity)) return nfail("n_rows=%lld out of range", (long long)n_rows);
    std::lock_guard<std::mutex> guard(nx->mu);
    for (int g = 0; g < nx->n; ++g) {
        const int64_t lo = std::min<int64_t>(n_rows, g * nx->rows_per_shard), hi = std::min<int64_t>(n_rows, lo + nx->rows_per_shard);
        if (!labels) {
            if (vodhip_index_set_row_labels(nx->shard[g], nullptr, 0, VODHIP_HOST, nullptr)) return -1;
        } else if (vodhip_index_set_row_labels(nx->shard[g], labels + lo, hi - lo, VODHIP_HOST, nx->stream[g])) {
            return -1;
        }
    }
    nx->has_row_labels = labels != nullptr;
    if (!labels) nx->q_labels = nullptr;
    return 0;
}

int vodhip_node_index_set_query_labels(vodhip_node_index_t* nx, const int32_t* q_labels, int n_per_query, int location) {
    if (!nx) return nfail("index is NULL");
    if (q_labels && (n_per_query < 1 || n_per_query > 64)) return nfail("n_per_query must be in [1, 64]");
    if (q_labels && !nx->has_row_labels) return nfail("set the row labels first (vodhip_node_index_set_row_labels)");
    if (location != VODHIP_HOST && location != VODHIP_DEVICE) return nfail("invalid location %d", location);
    std::lock_guard<std::mutex> guard(nx->mu);
    nx->q_labels = q_labels;
    nx->q_labels_per_query = q_labels ? n_per_query : 0;
    nx->q_labels_location = location;
    return 0;
}

int vodhip_node_index_set_param(vodhip_node_index_t* nx, const char* key, int64_t value) {
    if (!nx) return nfail("index is NULL");
    std::lock_guard<std::mutex> guard(nx->mu);
    if (key && !strcmp(key, "host_staging")) {  // 1: every shard but the first exchanges with devices[0] through pinned host memory
        nx->host_staging = value;
        return 0;
    }
    if (key && !strcmp(key, "profile")) nx->profile = value;  // (and on every shard: its filter launches are bracketed too)
    for (int g = 0; g < nx->n; ++g)
        if (vodhip_index_set_param(nx->shard[g], key, value)) return -1;
    return 0;
}

}  // extern "C"

namespace {

// step 1 of a search, into slot `S`: the query batch reaches every device (replicated: nq * dim * 2-4 bytes), then every shard's search is
// enqueued - nothing is waited for, so the devices run side by side and the call returns while they do
int node_enqueue_locked(vodhip_node_index* nx, NodeSlot& S, const void* queries, int q_dtype, int64_t nq, int k, int location,
                        float* out_scores, int64_t* out_ids, void* stream_) {
    const int G = nx->n, dev0 = nx->device[0];
    hipStream_t user = location == VODHIP_DEVICE ? (hipStream_t)stream_ : nx->merge_stream;
    const size_t q_bytes = (size_t)nq * (size_t)nx->dim * elem_bytes(q_dtype), res = (size_t)nq * (size_t)k;
    S.timed = false;
    if (nx->profile && S.copy_ev.empty()) {
        S.copy_ev.assign((size_t)2 * G, nullptr);
        for (int g = 0; g < G; ++g) {
            NODE_HIP_OK(hipSetDevice(nx->device[g]));
            NODE_HIP_OK(hipEventCreate(&S.copy_ev[2 * g]));
            NODE_HIP_OK(hipEventCreate(&S.copy_ev[2 * g + 1]));
        }
        NODE_HIP_OK(hipSetDevice(dev0));
        NODE_HIP_OK(hipEventCreate(&S.merge_ev[0]));
        NODE_HIP_OK(hipEventCreate(&S.merge_ev[1]));
    }
    // buffers (grown on demand, kept)
    for (int g = 0; g < G; ++g) {
        NODE_HIP_OK(hipSetDevice(nx->device[g]));
        ShardBuffers& b = S.buf[g];
        size_t rs = b.res_elems, ri = b.res_elems;
        if (ensure(&b.q, &b.q_bytes, q_bytes, 1)) return -1;
        if (ensure((void**)&b.scores, &rs, res, sizeof(float))) return -1;
        if (ensure((void**)&b.ids, &ri, res, sizeof(int64_t))) return -1;
        b.res_elems = std::min(rs, ri);
    }
    NODE_HIP_OK(hipSetDevice(dev0));
    if (G > 1) {
        size_t gs = S.gathered_elems, gi = S.gathered_elems;
        if (ensure((void**)&S.gathered_scores, &gs, res * G, sizeof(float))) return -1;
        if (ensure((void**)&S.gathered_ids, &gi, res * G, sizeof(int64_t))) return -1;
        S.gathered_elems = std::min(gs, gi);
    }
    if (location == VODHIP_HOST) {
        size_t ms = S.merged_elems, mi = S.merged_elems;
        if (ensure((void**)&S.merged_scores, &ms, res, sizeof(float))) return -1;
        if (ensure((void**)&S.merged_ids, &mi, res, sizeof(int64_t))) return -1;
        S.merged_elems = std::min(ms, mi);
    }
    float* final_scores = location == VODHIP_HOST ? S.merged_scores : out_scores;
    int64_t* final_ids = location == VODHIP_HOST ? S.merged_ids : out_ids;

    if (location == VODHIP_DEVICE) NODE_HIP_OK(hipEventRecord(S.ready, user));  // the caller's stream has produced the queries
    bool any_staged = false;
    for (int g = 0; g < G; ++g) any_staged = any_staged || nx->staged(g);
    const size_t lab_bytes = nx->q_labels ? (size_t)nq * (size_t)nx->q_labels_per_query * sizeof(int32_t) : 0;
    const bool stage_q = any_staged && location == VODHIP_DEVICE, stage_lab = any_staged && nx->q_labels && nx->q_labels_location == VODHIP_DEVICE;
    if (stage_q || stage_lab) {
        // shards without peer access read the batch from pinned host memory: ONE copy down from devices[0], then one copy up per shard
        const size_t want = q_bytes + lab_bytes + 64;
        if (S.pin_q_bytes < want) {
            if (S.pin_q) NODE_HIP_OK(hipHostFree(S.pin_q));
            S.pin_q = nullptr;
            NODE_HIP_OK(hipHostMalloc(&S.pin_q, want + want / 4, hipHostMallocDefault));
            S.pin_q_bytes = want + want / 4;
        }
        if (stage_q) NODE_HIP_OK(hipMemcpyAsync(S.pin_q, queries, q_bytes, hipMemcpyDeviceToHost, user));
        if (stage_lab) NODE_HIP_OK(hipMemcpyAsync((char*)S.pin_q + q_bytes, nx->q_labels, lab_bytes, hipMemcpyDeviceToHost, user));
        NODE_HIP_OK(hipEventRecord(S.q_on_host, user));
    }
    for (int g = 0; g < G; ++g) {
        NODE_HIP_OK(hipSetDevice(nx->device[g]));
        const bool staged = nx->staged(g);
        if (location == VODHIP_HOST) {
            NODE_HIP_OK(hipMemcpyAsync(S.buf[g].q, queries, q_bytes, hipMemcpyHostToDevice, nx->stream[g]));
        } else if (staged) {
            NODE_HIP_OK(hipStreamWaitEvent(nx->stream[g], S.q_on_host, 0));
            NODE_HIP_OK(hipMemcpyAsync(S.buf[g].q, S.pin_q, q_bytes, hipMemcpyHostToDevice, nx->stream[g]));
        } else {
            NODE_HIP_OK(hipStreamWaitEvent(nx->stream[g], S.ready, 0));
            if (nx->device[g] == dev0) NODE_HIP_OK(hipMemcpyAsync(S.buf[g].q, queries, q_bytes, hipMemcpyDeviceToDevice, nx->stream[g]));
            else NODE_HIP_OK(hipMemcpyPeerAsync(S.buf[g].q, nx->device[g], queries, dev0, q_bytes, nx->stream[g]));
        }
        if (nx->q_labels) {  // the batch's subset labels travel with the queries
            const size_t n_lab = (size_t)nq * (size_t)nx->q_labels_per_query;
            if (ensure((void**)&S.buf[g].q_labels, &S.buf[g].q_label_elems, n_lab, sizeof(int32_t))) return -1;
            if (nx->q_labels_location == VODHIP_HOST)
                NODE_HIP_OK(hipMemcpyAsync(S.buf[g].q_labels, nx->q_labels, n_lab * sizeof(int32_t), hipMemcpyHostToDevice, nx->stream[g]));
            else if (staged) {
                NODE_HIP_OK(hipStreamWaitEvent(nx->stream[g], S.q_on_host, 0));
                NODE_HIP_OK(hipMemcpyAsync(S.buf[g].q_labels, (char*)S.pin_q + q_bytes, n_lab * sizeof(int32_t), hipMemcpyHostToDevice, nx->stream[g]));
            } else if (nx->device[g] == dev0)
                NODE_HIP_OK(hipMemcpyAsync(S.buf[g].q_labels, nx->q_labels, n_lab * sizeof(int32_t), hipMemcpyDeviceToDevice, nx->stream[g]));
            else
                NODE_HIP_OK(hipMemcpyPeerAsync(S.buf[g].q_labels, nx->device[g], nx->q_labels, dev0, n_lab * sizeof(int32_t), nx->stream[g]));
            if (vodhip_index_set_query_labels(nx->shard[g], S.buf[g].q_labels, nx->q_labels_per_query)) return -1;
        } else if (nx->has_row_labels) {
            if (vodhip_index_set_query_labels(nx->shard[g], nullptr, 0)) return -1;
        }
        // a G = 1 index writes the caller's / the merged buffers directly
        float* s_out = G == 1 ? final_scores : S.buf[g].scores;
        int64_t* i_out = G == 1 ? final_ids : S.buf[g].ids;
        if (vodhip_index_search_async(nx->shard[g], S.buf[g].q, q_dtype, nq, k, g * nx->rows_per_shard, s_out, i_out, nx->stream[g])) {
            const std::string keep = vodhip_last_error();
            return nfail("shard %d: %s", g, keep.c_str());
        }
        S.enqueued[g] = 1;
    }
    NODE_HIP_OK(hipSetDevice(dev0));
    S.nq = nq;
    S.k = k;
    S.location = location;
    S.out_scores = out_scores;
    S.out_ids = out_ids;
    S.user = user;
    return 0;
}

// steps 2 and 3 of the search in slot `S`: the exactness check of every shard (host side; a shard that needs a recovery pass runs it on its
// own stream), its list travels to devices[0], the merge there
int node_finish_locked(vodhip_node_index* nx, NodeSlot& S) {
    const int G = nx->n, dev0 = nx->device[0];
    const int64_t nq = S.nq;
    const int k = S.k, location = S.location;
    hipStream_t user = S.user;
    const size_t res = (size_t)nq * (size_t)k;
    float* final_scores = location == VODHIP_HOST ? S.merged_scores : S.out_scores;
    int64_t* final_ids = location == VODHIP_HOST ? S.merged_ids : S.out_ids;
    int rc = 0;
    std::string first_error;
    for (int g = 0; g < G; ++g) {
        S.enqueued[g] = 0;
        if (vodhip_index_search_finish(nx->shard[g], nx->stream[g])) {
            if (!rc) first_error = std::string("shard ") + std::to_string(g) + ": " + vodhip_last_error();
            rc = -1;
            continue;
        }
        if (G == 1 || rc) continue;
        NODE_HIP_OK(hipSetDevice(nx->device[g]));
        float* ds = S.gathered_scores + (size_t)g * res;
        int64_t* di = S.gathered_ids + (size_t)g * res;
        hipStream_t cs = nx->copy_stream[g];  // (the shard's search is complete - `finish` waited for it on the host - so the copy depends on nothing)
        if (nx->staged(g)) {
            // no peer access: the list goes down to pinned host memory on the shard's copy stream and up to devices[0] on the merge stream
            if (S.pin_res_elems[g] < res) {
                if (S.pin_res[g]) NODE_HIP_OK(hipHostFree(S.pin_res[g]));
                S.pin_res[g] = nullptr;
                NODE_HIP_OK(hipHostMalloc(&S.pin_res[g], (res + res / 4 + 64) * 12, hipHostMallocDefault));
                S.pin_res_elems[g] = res + res / 4 + 64;
            }
            if (!S.on_host[g]) NODE_HIP_OK(hipEventCreateWithFlags(&S.on_host[g], hipEventDisableTiming));
            char* pin = (char*)S.pin_res[g];
            if (nx->profile) NODE_HIP_OK(hipEventRecord(S.copy_ev[2 * g], cs));
            NODE_HIP_OK(hipMemcpyAsync(pin, S.buf[g].scores, res * sizeof(float), hipMemcpyDeviceToHost, cs));
            NODE_HIP_OK(hipMemcpyAsync(pin + S.pin_res_elems[g] * 4, S.buf[g].ids, res * sizeof(int64_t), hipMemcpyDeviceToHost, cs));
            if (nx->profile) NODE_HIP_OK(hipEventRecord(S.copy_ev[2 * g + 1], cs));  // (the down-leg; the up-leg runs on the merge stream)
            NODE_HIP_OK(hipEventRecord(S.on_host[g], cs));
            NODE_HIP_OK(hipSetDevice(dev0));
            NODE_HIP_OK(hipStreamWaitEvent(user, S.on_host[g], 0));
            NODE_HIP_OK(hipMemcpyAsync(ds, pin, res * sizeof(float), hipMemcpyHostToDevice, user));
            NODE_HIP_OK(hipMemcpyAsync(di, pin + S.pin_res_elems[g] * 4, res * sizeof(int64_t), hipMemcpyHostToDevice, user));
            NODE_HIP_OK(hipEventRecord(S.arrived[g], user));  // (the merge below runs on `user` anyway: this keeps the wait list uniform)
            continue;
        }
        if (nx->profile) NODE_HIP_OK(hipEventRecord(S.copy_ev[2 * g], cs));
        if (nx->device[g] == dev0) {
            NODE_HIP_OK(hipMemcpyAsync(ds, S.buf[g].scores, res * sizeof(float), hipMemcpyDeviceToDevice, cs));
            NODE_HIP_OK(hipMemcpyAsync(di, S.buf[g].ids, res * sizeof(int64_t), hipMemcpyDeviceToDevice, cs));
        } else {
            NODE_HIP_OK(hipMemcpyPeerAsync(ds, dev0, S.buf[g].scores, nx->device[g], res * sizeof(float), cs));
            NODE_HIP_OK(hipMemcpyPeerAsync(di, dev0, S.buf[g].ids, nx->device[g], res * sizeof(int64_t), cs));
        }
        if (nx->profile) NODE_HIP_OK(hipEventRecord(S.copy_ev[2 * g + 1], cs));
        NODE_HIP_OK(hipEventRecord(S.arrived[g], cs));
    }
    if (rc) return nfail("%s", first_error.c_str());
    // the merge on devices[0], on the caller's stream (DEVICE) or shard 0's (HOST)
    NODE_HIP_OK(hipSetDevice(dev0));
    if (G > 1) {
        for (int g = 0; g < G; ++g) NODE_HIP_OK(hipStreamWaitEvent(user, S.arrived[g], 0));
        if (nx->profile) NODE_HIP_OK(hipEventRecord(S.merge_ev[0], user));  // every list has arrived: what follows is the merge alone
        if (vodhip_merge_topk(S.gathered_scores, S.gathered_ids, G, nq, k, k, final_scores, final_ids, user)) return -1;
        if (nx->profile) NODE_HIP_OK(hipEventRecord(S.merge_ev[1], user));
        S.timed = nx->profile != 0;
    }
    // (G == 1: the one shard wrote the final buffers itself and its search is complete - nothing to order on the device)
    if (location == VODHIP_HOST) {
        NODE_HIP_OK(hipMemcpyAsync(S.out_scores, final_scores, res * sizeof(float), hipMemcpyDeviceToHost, user));
        NODE_HIP_OK(hipMemcpyAsync(S.out_ids, final_ids, res * sizeof(int64_t), hipMemcpyDeviceToHost, user));
        NODE_HIP_OK(hipStreamSynchronize(user));
    }
    if (location == VODHIP_DEVICE) {
        // the slot's NEXT search must not refill `gathered` / the shard buffers before this merge has read them: every shard stream waits
        // for `ready`, recorded here behind the merge (the other slot's search - already enqueued on the shard streams - is ahead of the wait)
        NODE_HIP_OK(hipEventRecord(S.ready, user));
        for (int g = 0; g < G; ++g) {
            NODE_HIP_OK(hipSetDevice(nx->device[g]));
            NODE_HIP_OK(hipStreamWaitEvent(nx->stream[g], S.ready, 0));
        }
        NODE_HIP_OK(hipSetDevice(dev0));
    }
    return 0;
}

int node_check_args(vodhip_node_index* nx, const void* queries, int q_dtype, int64_t nq, int k, int location, float* out_scores, int64_t* out_ids) {
    if (k < 1 || k > VODHIP_MAX_K) return nfail("k=%d out of range [1, %d]", k, VODHIP_MAX_K);
    if (nq < 0 || (nq > 0 && (!queries || !out_scores || !out_ids))) return nfail("invalid query / output pointers");
    if (q_dtype < 0 || q_dtype > 2) return nfail("invalid q_dtype %d", q_dtype);
    if (location != VODHIP_HOST && location != VODHIP_DEVICE) return nfail("invalid location %d", location);
    (void)nx;
    return 0;
}

// whatever failed (a copy, an allocation, one shard's search): no shard keeps a search of the failed call in its FIFO - the next call's
// `finish` must meet the next call's search
void node_drain_failed(vodhip_node_index* nx, NodeSlot& S) {
    const std::string keep = vodhip_last_error();
    for (int g = 0; g < nx->n; ++g)
        if (S.enqueued[(size_t)g]) (void)vodhip_index_search_finish(nx->shard[g], nx->stream[g]);
    S.enqueued.assign((size_t)nx->n, 0);
    (void)hipGetLastError();
    vodhip::set_last_error(keep.c_str());
}

}  // namespace

extern "C" {

int vodhip_node_index_search_async(vodhip_node_index_t* nx, const void* queries, int q_dtype, int64_t nq, int k, int location,
                                   float* out_scores, int64_t* out_ids, void* stream_) {
    if (!nx) return nfail("index is NULL");
    std::lock_guard<std::mutex> guard(nx->mu);
    if (node_check_args(nx, queries, q_dtype, nq, k, location, out_scores, out_ids)) return -1;
    if (nx->n_pending >= 2) return nfail("2 searches are already in flight on this node index: call vodhip_node_index_search_finish first");
    NodeSlot& S = nx->slot[nx->next_slot];
    S.nq = 0;
    S.enqueued.assign((size_t)nx->n, 0);
    if (nq > 0) {
        if (node_enqueue_locked(nx, S, queries, q_dtype, nq, k, location, out_scores, out_ids, stream_)) {
            node_drain_failed(nx, S);
            return -1;
        }
    }
    S.pending = true;
    nx->next_slot ^= 1;
    ++nx->n_pending;
    return 0;
}

int vodhip_node_index_search_finish(vodhip_node_index_t* nx) {
    if (!nx) return nfail("index is NULL");
    // The wait for the devices happens OUTSIDE the mutex: another thread may enqueue the next search into the other slot meanwhile (the
    // batcher's scheduler beside its completion thread).  The slot stays counted as pending until this call is over, so it is not reused.
    std::unique_lock<std::mutex> lk(nx->mu);
    if (nx->n_pending == 0) return nfail("no search is pending on this node index");
    if (nx->finishing) return nfail("another vodhip_node_index_search_finish is running on this node index");
    const int si = (nx->next_slot - nx->n_pending) & 1;
    NodeSlot& S = nx->slot[si];
    nx->finishing = true;
    lk.unlock();
    int rc = 0;
    if (S.nq > 0) {
        rc = node_finish_locked(nx, S);
        if (rc) node_drain_failed(nx, S);  // (the shards whose finish this call did not reach still hold the search)
    }
    lk.lock();
    S.pending = false;
    --nx->n_pending;
    nx->last_finished = si;
    nx->finishing = false;
    return rc;
}

int vodhip_node_index_search(vodhip_node_index_t* nx, const void* queries, int q_dtype, int64_t nq, int k, int location,
                             float* out_scores, int64_t* out_ids, void* stream_) {
    if (!nx) return nfail("index is NULL");
    {
        std::lock_guard<std::mutex> guard(nx->mu);
        if (nx->n_pending) return nfail("%d asynchronous searches are pending on this node index: finish them first", nx->n_pending);
    }
    if (vodhip_node_index_search_async(nx, queries, q_dtype, nq, k, location, out_scores, out_ids, stream_)) return -1;
    return vodhip_node_index_search_finish(nx);
}

int vodhip_node_index_get_stat(vodhip_node_index_t* nx, const char* key, int64_t* out) {
    if (!nx || !key || !out) return nfail("NULL argument");
    std::lock_guard<std::mutex> guard(nx->mu);
    *out = 0;
    if (!strcmp(key, "last_merge_ns") || !strcmp(key, "last_copy_ns_max")) {
        NodeSlot& S = nx->slot[nx->last_finished];
        if (!S.timed || nx->n < 2) return 0;  // (param "profile" was off, or one shard: no exchange, no merge)
        float ms = 0.f;
        if (!strcmp(key, "last_merge_ns")) {
            NODE_HIP_OK(hipSetDevice(nx->device[0]));
            NODE_HIP_OK(hipEventSynchronize(S.merge_ev[1]));
            NODE_HIP_OK(hipEventElapsedTime(&ms, S.merge_ev[0], S.merge_ev[1]));
            *out = (int64_t)((double)ms * 1e6);
            return 0;
        }
        for (int g = 0; g < nx->n; ++g) {
            NODE_HIP_OK(hipSetDevice(nx->device[g]));
            NODE_HIP_OK(hipEventSynchronize(S.copy_ev[2 * g + 1]));
            NODE_HIP_OK(hipEventElapsedTime(&ms, S.copy_ev[2 * g], S.copy_ev[2 * g + 1]));
            *out = std::max<int64_t>(*out, (int64_t)((double)ms * 1e6));
        }
        NODE_HIP_OK(hipSetDevice(nx->device[0]));
        return 0;
    }
    return nfail("unknown stat '%s'", key);
}

}  // extern "C"

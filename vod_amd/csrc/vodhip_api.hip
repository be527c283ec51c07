// C-ABI layer of libvodhip.so: handle management, ingest, the chunk schedule of the fused
// score + top-k search, and thin wrappers for the merge / hybrid / retrieval kernels.
// See include/vodhip.h for the contract and the reference call sites each entry point replaces.
#include "../../include/vodhip.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "vodhip_internal.h"

using namespace vodhip;

namespace {

thread_local std::string g_last_error;

int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return -1;
}

}  // namespace

namespace vodhip {
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
}  // namespace vodhip

namespace {

#define HIP_OK(expr)                                                                            \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (void)hipGetLastError(); /* reported here: must not resurface in a later launch check */ \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                       \
    } while (0)

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

constexpr int64_t ROW_ALIGN = 256;       // largest tile height; chunk boundaries and capacity padding
constexpr int64_t MAX_NQ_PER_PASS = 2048;
constexpr int64_t STAGE_BYTES = 64ll << 20;

struct PendingSearch {
    bool active = false;
    const void* queries = nullptr;
    int q_dtype = 0;
    int64_t nq = 0;
    int k = 0;
    int64_t id_base = 0;
    float* out_scores = nullptr;
    int64_t* out_ids = nullptr;
    const int* q_label = nullptr;  // subset labels in force when the search was enqueued
    int n_qlab = 0;
    const int* q_map = nullptr;    // recovery of a few queries: workspace row -> row of the caller's batch (device)
    int slot = 0;                  // overflow-flag word / completion event of this search
    size_t ev_begin = 0, ev_end = 0;  // profile events of this search in ev_pool
    int kx = 0;                    // exact mode: the scan's list length k' (> k; 0 = not an exact-mode search)
    bool band = false;             // exact mode: a BAND pass - candidates are re-scored into the caller's rows (kernels_exact.hip)
    bool defer_flags = false;      // exact mode: the flag words are fetched (and the slot's event recorded) behind the re-scoring launch
    int lane = 0;                  // which of the index's two workspaces / streams this search runs on (1 = the index's own stream)
};

constexpr int FLAG_WORDS = 4;     // flag words of a workspace / of a slot's host copy
constexpr int MAX_IN_FLIGHT = 4;  // searches that may be enqueued before the oldest is finished
constexpr int64_t OVF_ROWS = 65536;  // per-query overflow flags are kept for batches up to this many queries

}  // namespace

struct vodhip_index {
    int device = 0;
    int64_t dim = 0, dim_pad = 0;
    int64_t capacity = 0, capacity_pad = 0;
    int64_t ntotal = 0;
    int dtype = VODHIP_F16;
    uint16_t* data = nullptr;
    int* row_label = nullptr;        // [capacity_pad] subset label per row (optional)
    const int* q_label = nullptr;    // caller-owned device [nq, n_qlab] for the next searches (optional)
    int n_qlab = 0;
    // host ingest pipeline: two slots of (pinned host staging, device staging of the raw dtype, completion event), one stream
    void* stage_dev[2] = {nullptr, nullptr};
    void* stage_pin[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    int64_t ingest_threads = 0;  // CPU threads that fill the pinned staging slots (0 = auto: min(8, hardware threads))
    int64_t last_ingest_pinned_src = 0;
    // VODHIP_EXACT_F32 (kernels_exact.hip): the float32 rows, the statistics of the error bound, per-slot lists of the scan
    bool exact = false;
    float* data32 = nullptr;                 // [capacity + 1][dim_pad]
    unsigned int* norm_stats = nullptr;      // device [EXS_WORDS]: max |x|^2, max |x - x~|^2 (atomics at ingest), the ordinary maxima, the outliers
    float* row_n2 = nullptr;                 // device [capacity + 1]: |x|^2 of every row (written at ingest)
    float* row_d2 = nullptr;                 // device [capacity + 1]: |x - x~|^2
    bool stats_dirty = true;                 // rows were added / the store was reset since the statistics below were derived
    float ord_n2 = 0.f, ord_d2 = 0.f, cut_n2 = 0.f, cut_d2 = 0.f;  // host copies (refresh_exact_stats): the bound's maxima over the ordinary rows, the outlier cuts
    int n_out = 0;                           // outlier rows (scored by every query instead of bounded)
    int64_t exact_adapt = 1;                 // 1: the list length k' follows what the last searches needed (0: always the formula)
    int adapt_k = 0, adapt_kx = 0;           // the k the adapted k' belongs to / the adapted k' (0 = none yet)
    int64_t last_exact_need = 0;             // list entries within eps of the k-th exact score, maximum over the last search's queries
    float* x_list_s[4] = {nullptr, nullptr, nullptr, nullptr};     // per in-flight slot: the scan's top-k' list [nq][k']
    int64_t* x_list_i[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t x_list_cap[4] = {0, 0, 0, 0};     // elements
    float* x_eps = nullptr;                  // device [MAX_IN_FLIGHT][OVF_ROWS]: the per-query bound of a slot's search
    unsigned int* x_flag_q = nullptr;        // device [MAX_IN_FLIGHT][OVF_ROWS]: the queries whose list did not prove complete
    int64_t exact_expand_x100 = 0;           // k' = k * this / 100 (+ 16); 0 = by store dtype (fp16: 110, bf16: 200)
    int64_t last_exact_kx = 0, last_exact_band_queries = 0, last_exact_band_passes = 0;
    // Two LANES: workspace 0 is used on the caller's stream; workspace 1 on a stream of the index's own, so that two consecutive searches
    // of a caller that runs one search ahead (the bench, the batcher) overlap on the device: while one search's select / prepare launches
    // and the partly filled last round of its stage leave CUs idle, the other search's FILTER workgroups take them.  Used for batches of
    // one query tile (param "lanes": 0 = auto, 1, 2), where those gaps are ~12 % of a 1 M-row search (profiles/r05_ab_lanes.txt).
    SearchWorkspace ws_lane[2];
    hipStream_t lane_stream = nullptr;
    hipEvent_t lane_in[4] = {};              // recorded on the caller's stream at enqueue: lane 1 starts behind the caller's pending work
    int64_t lanes = 0;
    unsigned int* overflow_host = nullptr;  // pinned, FLAG_WORDS per in-flight slot: [0] a candidate list overflowed, [1] exact mode: some list did not prove complete, [2] exact mode: max over queries of the list entries within eps of the k-th exact score
    unsigned int* ovf_q = nullptr;          // device [MAX_IN_FLIGHT][OVF_ROWS]: which queries of a slot's search overflowed
    int* q_map = nullptr;                   // device [MAX_IN_FLIGHT][OVF_ROWS]: the rows a slot's recovery pass re-searches
    hipEvent_t done[MAX_IN_FLIGHT] = {};    // recorded after a search's overflow word is copied back
    std::deque<PendingSearch> inflight;     // oldest first
    int unfinished = 0;                     // searches popped by a vodhip_index_search_finish that is still waiting for their event
    int next_slot = 0;
    // tunables
    int64_t cand_cap = 16384;
    int64_t dense_rows = 2048;   // indexes up to this many rows are scored densely in one launch
    int64_t growth_x100 = 0;     // FILTER stage = growth x the rows its threshold was calibrated on; 0 = auto (8; 3 for batches above 512 queries on stores of 4 M rows and more)
    int64_t sample_div = 0;      // GMAX bootstrap scores ~ ntotal / sample_div sampled rows; 0 = auto (96; 192 where growth is 3)
    int64_t force_safe = 0;
    int64_t tile = 0;
    int64_t kflags = 0;
    int64_t tile_order = 0;      // 0 = FILTER stages walk the store's super-tiles in a low-discrepancy order (default); 1 = in row order
    int64_t small_chunk_tiles = 256;  // launches with fewer 256x256 tiles than this (less than one per CU) use the 128x128 kernel
    int n_cu = 256;       // compute units of `device` (read once at create; the planner never touches the runtime)
    int64_t profile = 0;  // 1: bracket every filter launch with HIP events (bench / roofline accounting)
    // stats
    int64_t last_overflow = 0, last_chunks = 0, last_safe_reruns = 0, last_recovered_queries = 0;
    int64_t last_filter_launches = 0, last_filter_ns = 0;
    int64_t last_recovery_launches = 0, last_recovery_ns = 0;  // filter launches of the recovery passes (with "profile")
    std::vector<hipEvent_t> ev_pool;  // pairs (start, stop), reused across searches
    size_t ev_used = 0;
    // Every entry point that touches `inflight`, the shared workspace (`ws`, incl. its host-side `extra`) or the event pool holds
    // this: a server thread may enqueue search i + 1 while another completes (and, after an overflow, RE-ENQUEUES recovery passes of)
    // search i on the same handle.  The wait for a search's completion event happens outside it.
    std::mutex mu;
};

namespace {

int elem_size(int dtype) { return dtype == VODHIP_F32 ? 4 : 2; }

int free_workspace(vodhip_index* ix, int lane) {
    SearchWorkspace& w = ix->ws_lane[lane];
    (void)hipFree(w.q_pad);
    (void)hipFree(w.topk);
    (void)hipFree(w.cand);
    (void)hipFree(w.cnt);
    (void)hipFree(w.thr_s);
    (void)hipFree(w.thr_key);
    (void)hipFree(w.overflow);
    const int n_cu = w.n_cu;
    w = SearchWorkspace();
    w.n_cu = n_cu;
    return 0;
}

int ensure_workspace(vodhip_index* ix, int lane, int64_t nq_pad, int64_t cap, int64_t kp) {
    SearchWorkspace& w = ix->ws_lane[lane];
    // (kp: the running top-k's row stride of THIS search, passed to the launches by value; the buffer is sized for the widest seen - an
    // exact-mode scan (k') and its band pass (k) alternate between two widths: reallocating per search is a device-wide hipFree each time)
    if (w.nq_cap >= nq_pad && w.cap == cap && w.kp_cap >= kp && w.q_pad) {
        w.kp = kp;
        return 0;
    }
    const int64_t nq_cap = std::max(nq_pad, w.nq_cap);
    const int64_t kp_cap = std::max(kp, w.kp_cap);
    free_workspace(ix, lane);
    HIP_OK(hipMalloc((void**)&w.q_pad, (size_t)nq_cap * ix->dim_pad * 2));
    HIP_OK(hipMalloc((void**)&w.topk, (size_t)nq_cap * kp_cap * sizeof(key_t64)));
    HIP_OK(hipMalloc((void**)&w.cand, (size_t)nq_cap * cap * sizeof(key_t64)));
    HIP_OK(hipMalloc((void**)&w.cnt, (size_t)nq_cap * CNT_STRIDE * sizeof(unsigned int)));
    HIP_OK(hipMalloc((void**)&w.thr_s, (size_t)nq_cap * sizeof(float)));
    HIP_OK(hipMalloc((void**)&w.thr_key, (size_t)nq_cap * sizeof(key_t64)));
    HIP_OK(hipMalloc((void**)&w.overflow, FLAG_WORDS * sizeof(unsigned int)));  // [0] overflow, [1] exact mode's "incomplete" word, [2] its "needed list length"
    w.nq_cap = nq_cap;
    w.cap = cap;
    w.kp = kp;
    w.kp_cap = kp_cap;
    return 0;
}

// ---- stage schedule ------------------------------------------------------------------------------------------
// A search is a list of stages, each one filter launch + one select launch:
//   GMAX    threshold bootstrap.  S sampled rows (8-row groups spread evenly over the whole store) are scored and every
//           lane writes the maximum of its group of 16 / 32 rows; the k-th largest group maximum is a lower bound of the
//           k-th best score (k distinct rows reach it) however the rows are ordered.  Emits no candidates.
//   FILTER  rows [b, e) filtered against the running threshold; survivors -> candidate lists -> running top-k.
//           Stage i covers `growth` x the rows the threshold was calibrated on, so it emits ~ growth * k survivors per
//           query for exchangeable row order, and never more than the GMAX bound allows (~ k * rows / S) for any order.
//   DENSE   every score of <= cap rows becomes a candidate (indexes of a few thousand rows; the exhaustive fallback).
enum : int { ST_FILTER = 0, ST_DENSE = 1, ST_GMAX = 2 };
struct Stage {
    int kind;
    int64_t b, e;        // rows (FILTER / DENSE)
    int64_t n_tiles;     // sampled tiles (GMAX)
    int64_t rstride;     // store rows between consecutive sampled rows (GMAX)
    int64_t n_groups;    // lane groups of the sample = candidate slots per query (GMAX)
};

void make_safe_schedule(int64_t n, int64_t cap, std::vector<Stage>& st) {
    const int64_t step = std::max<int64_t>(ROW_ALIGN, cap / ROW_ALIGN * ROW_ALIGN);
    for (int64_t b = 0; b < n; b += step) st.push_back({ST_DENSE, b, std::min(n, b + step), 0, 0, 0});
}

// dense head of <= cap rows, then FILTER stages that grow by 1 + cap / 4k (each emits ~ (growth - 1) k survivors per query for
// exchangeable row order): the schedule for searches that cannot use the bootstrap (subset filters; k too large for cap)
void make_geometric_schedule(int64_t n, int k, int64_t cap, std::vector<Stage>& st) {
    int64_t b = std::min(n, std::max<int64_t>(ROW_ALIGN, std::min<int64_t>(cap, 2048) / ROW_ALIGN * ROW_ALIGN));
    if (b < k && b < n) return make_safe_schedule(n, cap, st);
    st.push_back({ST_DENSE, 0, b, 0, 0, 0});
    const double growth = std::min(8.0, std::max(1.25, 1.0 + (double)cap / (4.0 * k)));
    while (b < n) {
        const int64_t e = std::min(n, round_up(std::max((int64_t)((double)b * growth), b + ROW_ALIGN), ROW_ALIGN));
        st.push_back({ST_FILTER, b, e, 0, 0, 0});
        b = e;
    }
}

// The multiplier of the low-discrepancy stage order over T super-tiles: position p -> super-tile (p * P) mod T with P the largest
// integer <= T / golden ratio that is coprime to T (a bijection of [0, T); consecutive positions land ~0.618 T apart, so any run of L
// positions leaves gaps of O(T / L) - three-distance theorem).  <= 1: no permutation.
int64_t tile_order_multiplier(int64_t T) {
    if (T < 8) return 0;
    // Candidates around T / golden ratio; among those coprime to T the one whose continued fraction P / T has the smallest largest
    // partial quotient: a run of L consecutive multiples of P (mod T) then leaves gaps within a small factor of T / L at EVERY scale L
    // (three-distance theorem; a candidate that merely is coprime can sit next to a fraction with a small denominator and leave gaps
    // 12x the mean - seen at T = 99,684).
    const int64_t P0 = (int64_t)((double)T * 0.6180339887498949);
    int64_t best = 0, best_q = INT64_MAX;
    for (int64_t d = 0; d <= 64; ++d) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            const int64_t P = sgn ? P0 - d : P0 + d;
            if (P <= 1 || P >= T || (d == 0 && sgn)) continue;
            int64_t a = T, b = P, worst = 0;
            bool first = true;
            while (b) {  // Euclid: the partial quotients of T / P
                const int64_t quo = a / b, rem = a % b;
                if (!first) worst = std::max(worst, quo);  // (the first is floor(T / P) = 1 by construction)
                first = false;
                a = b;
                b = rem;
            }
            if (a != 1) continue;  // not coprime
            if (worst < best_q) {
                best_q = worst;
                best = P;
            }
        }
    }
    return best;
}

// `recovery` > 0: pass number after a candidate-list overflow - no bootstrap (the thresholds are seeded from the previous
// result), 2^(recovery-1) equal FILTER stages, and the exhaustive schedule once a stage would be <= cap rows.
void make_schedule(const vodhip_index* ix, int k, int gmax_tile, int64_t nq_pad, bool safe, int recovery, std::vector<Stage>& st) {
    st.clear();
    const int64_t n = ix->ntotal, cap = ix->cand_cap;
    if (n <= 0) return;
    int64_t dense_limit = std::max(ix->dense_rows, round_up(k, ROW_ALIGN));
    dense_limit = std::min(dense_limit, cap / ROW_ALIGN * ROW_ALIGN);
    if (safe || n <= dense_limit) return make_safe_schedule(n, cap, st);
    if (recovery > 0) {
        const int64_t n_st = 1ll << std::min(recovery - 1, 30);
        const int64_t rows = round_up((n + n_st - 1) / n_st, ROW_ALIGN);
        if (rows <= cap) return make_safe_schedule(n, cap, st);
        for (int64_t b = 0; b < n; b += rows) st.push_back({ST_FILTER, b, std::min(n, b + rows), 0, 0, 0});
        return;
    }
    const int64_t bm = filter_tile_rows(gmax_tile), rg = filter_group_rows(gmax_tile);
    // 4k groups wanted, 2k needed: the k-th largest of G group maxima is exceeded by a fraction -ln(1 - k/G) / rg of the rows,
    // ~1.15 k/S at G = 4k, 1.39 k/S at G = 2k (the stage-size bound below allows 1.6 k/S), and it blows up as G approaches k
    int64_t kp = 64;
    while (kp < k) kp <<= 1;
    // one candidate slot per group, and the select kernel takes the group maxima in ONE round of its largest buffer
    const int64_t g_max = std::min<int64_t>(cap, 8192 - kp);
    const int64_t s_max = std::min(rg * g_max, n / 2) / bm * bm;
    const int64_t s_need = round_up(std::max<int64_t>(2 * rg * (int64_t)k, 2048), bm);
    if (s_max < s_need) {  // no usable bootstrap (few rows, or k too large for cap): all dense when that is a few launches
        if (n <= 64 * cap) return make_safe_schedule(n, cap, st);
        return make_geometric_schedule(n, k, cap, st);
    }
    const int64_t s_min = std::min(s_max, round_up(std::max<int64_t>(4 * rg * (int64_t)k, 2048), bm));
    // Round 6: the survivor path of the FILTER epilogue is 7-10 % of a C3 batch (kernels_mips_8phase.hip), and a stage lets ~ growth * k
    // rows per query pass: on a large store searched with a large batch, more and smaller stages behind a smaller bootstrap pay (10 M x
    // 768, nq 1024: growth 3 + N / 192 = 6 launches against growth 8 + N / 96 = 4: -2.0 %, exact-f32 -2.1 %, clustered rows -1.9 %).  At
    // 512 queries and fewer (half the survivors per corpus tile) and on small stores a stage's own cost - a select launch, a partial round
    // of the persistent grid - weighs as much: measured +-0.4 % (5 M / 40 M x 1024 at nq 512, 10 M at nq 256, 1.25 M), the round-3 rule
    // stays there (profiles/r06_ab_epilogue.txt).
    const bool many_small_stages = n >= 4000000 && nq_pad > 512;
    const int64_t sdiv = ix->sample_div > 0 ? ix->sample_div : (many_small_stages ? 192 : 96);
    int64_t s = std::min(s_max, std::max(s_min, round_up(n / std::max<int64_t>(sdiv, 2), bm)));
    int64_t round_rows = ROW_ALIGN;  // corpus rows ONE round of the persistent grid covers (one 256 x 256 tile per CU)
    if (filter_tile_is_persistent(gmax_tile)) {
        round_rows = std::max<int64_t>(1, std::max(1, ix->n_cu) / std::max<int64_t>(1, nq_pad / 256)) * bm;
        // the persistent kernel runs one workgroup per CU: a bootstrap of r.x "rounds" of tiles costs as much as r+1 full ones.
        // Whole rounds only: down when that keeps >= 4k groups (a cheaper bootstrap), up otherwise (a tighter bound for free)
        const int64_t n_cu = std::max(1, ix->n_cu);
        const int64_t per_round = std::max<int64_t>(1, n_cu / std::max<int64_t>(1, nq_pad / 256)) * bm;  // sampled rows per round
        const int64_t down = s / per_round * per_round, up = round_up(s, per_round);
        if (down >= s_min) s = down;
        else if (up <= s_max) s = up;
    }
    // S sampled rows at stride (n-1)/(S-1): the last one is row (S-1)*rstride <= n-1, all distinct (S <= n/2)
    st.push_back({ST_GMAX, 0, 0, s / bm, (n - 1) / (s - 1), s / rg});
    // however the rows are ordered, rows [b, e) hold about k * (e - b) / S scores above the bootstrap bound (the sample is
    // stratified over the whole store): a stage never covers more rows than the candidate lists can take with 60 % headroom
    const int64_t rows_safe = std::max<int64_t>(ROW_ALIGN, (int64_t)((double)cap * (double)s / (1.6 * (double)k)) / ROW_ALIGN * ROW_ALIGN);
    const size_t n_head = st.size();
    auto plan = [&](double growth) {
        st.resize(n_head);
        int64_t b = 0, calibrated = s;
        while (b < n) {
            int64_t rows = std::min(rows_safe, round_up((int64_t)((double)calibrated * growth), ROW_ALIGN));
            // stages held down by the capacity bound share what is left evenly (40 M x 1024 at growth 3: 3 x 10.06 M + a 0.38 M tail otherwise)
            if (rows == rows_safe && n - b > rows) {
                const int64_t m = (n - b + rows_safe - 1) / rows_safe;
                rows = std::min(rows_safe, round_up((n - b + m - 1) / m, ROW_ALIGN));
            }
            // whole rounds of the persistent grid: a stage of r.x rounds costs r + 1 (its last round runs on a fraction of the CUs), so
            // only the LAST stage of a search may end inside a round (round 4; the capacity bound rows_safe only ever rounds DOWN)
            if (rows > round_rows) rows = rows / round_rows * round_rows;
            int64_t e = std::min(n, b + rows);
            if (n - e < rows / 4 && n - b <= rows_safe) e = n;  // no short tail stage
            st.push_back({ST_FILTER, b, e, 0, 0, 0});
            b = e;
            calibrated = e;
        }
    };
    plan(std::min(256.0, std::max(1.25, ix->growth_x100 > 0 ? ix->growth_x100 / 100.0 : (many_small_stages ? 3.0 : 8.0))));
    // A store of ~10-20 sample sizes comes out as a short first stage followed by ONE stage with all the rest (the 1.25 M-row shard
    // of the headline: 131 k + 1,119 k rows).  Three stages at growth 4 (65 k + 327 k + 858 k) measure 1.3 % faster there, six out of
    // six interleaved runs (profiles/r03_ab_growth.txt); stores that already get three or more stages are unaffected (10 M rows: growth
    // 4 is 0.5 % slower than 8, so the default stays).
    if (ix->growth_x100 <= 0 && st.size() == n_head + 2 && (st[n_head + 1].e - st[n_head + 1].b) > 6 * (st[n_head].e - st[n_head].b)) plan(4.0);
}

// exact mode: the statistics of the error bound as the re-scoring launches take them (host copies: refresh_exact_stats)
void exact_bound_args(const vodhip_index* ix, ExactArgs& xa) {
    xa.ord_n2 = ix->ord_n2;
    xa.ord_d2 = ix->ord_d2;
    xa.cut_n2 = ix->cut_n2;
    xa.cut_d2 = ix->cut_d2;
    xa.n_out = ix->n_out;
    xa.out_rows = ix->norm_stats + EXS_OUT_ROWS;
    xa.row_n2 = ix->row_n2;
    xa.row_d2 = ix->row_d2;
}

// exact mode, before the first search after rows were added (no search is in flight then: `add` refuses otherwise): re-derive the maxima
// of the bound over the ORDINARY rows and the list of outliers from the per-row planes (8 bytes per row: ~30 us for 10 M rows), and
// fetch them.  ONE host sync per batch of adds.
int refresh_exact_stats(vodhip_index* ix, hipStream_t stream) {
    if (!ix->exact || !ix->stats_dirty) return 0;
    HIP_OK(launch_exact_stats(ix->row_n2, ix->row_d2, ix->ntotal, ix->norm_stats, stream));
    unsigned int w[8];
    HIP_OK(hipMemcpyAsync(w, ix->norm_stats, sizeof(w), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    auto f = [](unsigned int u) { float v; memcpy(&v, &u, 4); return v; };
    ix->ord_n2 = f(w[EXS_ORD_N2]);
    ix->ord_d2 = f(w[EXS_ORD_D2]);
    ix->cut_n2 = f(w[EXS_CUT_N2]);
    ix->cut_d2 = f(w[EXS_CUT_D2]);
    ix->n_out = (int)std::min<unsigned int>(w[EXS_N_OUT], 2u * EXACT_MAX_OUTLIERS);
    if (w[EXS_N_OUT] > 2u * EXACT_MAX_OUTLIERS) return fail("internal error: %u outlier rows", w[EXS_N_OUT]);
    ix->stats_dirty = false;
    return 0;
}

int enqueue_search_impl(vodhip_index* ix, const PendingSearch& ps, bool safe, int recovery, hipStream_t stream) {
    SearchWorkspace& W = ix->ws_lane[ps.lane];  // this search's lane
    const int k = ps.k;
    int64_t kp = 64;
    while (kp < k) kp <<= 1;
    const int64_t cap = ix->cand_cap;
    if (cap / ROW_ALIGN * ROW_ALIGN < k) return fail("cand_cap=%lld is too small for k=%d", (long long)cap, k);

    int tile = (int)ix->tile;
    // auto: up to 128 queries the search is HBM-bound: 256 corpus rows x 64 / 128 queries per workgroup on a 3-slot LDS
    // ring (few query bytes per corpus byte through the LDS-DMA path); above, the persistent 256x256 tile on
    // v_mfma_f32_16x16x32 (8; variant 9 staggers the two waves of every SIMD by one k-step: measured equal or 1-2 % slower)
    if (tile == 0) tile = ps.nq > 128 ? 8 : (ps.nq > 64 ? 46 : 42);
    // ... and the FILTER stages of those batches run the 8-phase K loop (tile 14: C3 -2.4 %, C4 shard -4.0 %; with ONE query tile and the
    // corpus stream on the `nt` policy C2 -3.5 %, nq 256 on 10 M rows -2.7 %: profiles/r05_ab_8phase.txt, r05_ab_one_query_tile.txt); its
    // subset instantiation spilled through round 6a and is 4-5 % slower than tile 8 on filtered searches since it no longer does (2.5 M x
    // 768, a quarter of the rows eligible: 3.95-3.99 vs 3.78-3.80 ms, experiments/tools/probe_subset_tile.py): they stay on tile 8, as does
    // the bootstrap
    const bool auto_8phase = ix->tile == 0 && ps.nq > 128 && !(ix->row_label && ps.q_label);
    const bool persistent = filter_tile_is_persistent(tile);
    const int64_t bn = filter_tile_cols(tile);

    // The bootstrap of a SMALL store (C2, a 1.25 M-row shard) samples fewer 256-row tiles than the chip has CUs: on the persistent
    // kernel ~40-200 workgroups would each run one whole 256 x 256 x dim tile (a ~25 us latency chain) while the other CUs idle.
    // Those bootstraps run on the 128 x 128 kernel instead (4x the workgroups, a quarter of the chain each; 16-row groups).
    std::vector<Stage> stages;
    make_schedule(ix, k, tile, round_up(std::min(MAX_NQ_PER_PASS, ps.nq), bn), safe, recovery, stages);
    ix->last_chunks = (int64_t)stages.size();

    const int q_es = elem_size(ps.q_dtype);
    if (ensure_workspace(ix, ps.lane, round_up(std::min(MAX_NQ_PER_PASS, ps.nq), 256), cap, kp)) return -1;
    W.extra.flags = (int)ix->kflags << 8;  // timing knobs of diagnostic builds (ignored by production kernels)
    // ONE q-tile: every corpus line is read by exactly one workgroup, once - fetch it with the `nt` policy so it does not
    // push the query tile out of L2 (measured -2 % at nq = 256 on 10 M rows; +8 % with 4 q-tiles sharing the lines, so only here)
    if (persistent && round_up(std::min(MAX_NQ_PER_PASS, ps.nq), bn) == 256) W.extra.flags |= FILTER_FLAG_CORPUS_NT;
    // the subset labels in force when THIS search was enqueued (a recovery pass may run after younger searches changed them)
    W.extra.row_label = (ix->row_label && ps.q_label) ? ix->row_label : nullptr;
    W.extra.n_qlab = ps.n_qlab;
    const bool track_ovf = ps.nq <= OVF_ROWS;  // per-query overflow flags of this search's slot
    const bool subset = W.extra.row_label != nullptr;
    if (subset && !safe && recovery == 0) {
        // group maxima would include ineligible rows: a subset search runs the exhaustive-free geometric schedule instead
        // (dense head of <= cap rows, then FILTER stages growing by `growth`)
        stages.clear();
        make_geometric_schedule(ix->ntotal, k, cap, stages);
        ix->last_chunks = (int64_t)stages.size();
    }
    // The order in which the FILTER stages walk the store's 256-row super-tiles: a low-discrepancy permutation (position p ->
    // super-tile p * P mod T, P ~ 0.618 T coprime to T), so that every stage - any run of consecutive positions - is spread evenly
    // over the whole store.  The reference ingests documents in corpus order (build.py:65-73): with contiguous stages a topic that
    // only the LAST stage contains meets a threshold calibrated without it, and all its tiles are scanned at the same moment
    // (bench.py --data clustered: 1.22x the i.i.d. time, L2-miss traffic 1.9x).  Only when every stage after the bootstrap is a FILTER
    // stage (they must tile the store together); results do not depend on the order.
    W.extra.perm_mul = W.extra.perm_mod = 0;
    W.extra.row_bound = (int)ix->ntotal;
    {
        bool all_filter = !stages.empty();
        for (const Stage& sg : stages) all_filter = all_filter && (sg.kind == ST_FILTER || sg.kind == ST_GMAX);
        const int64_t T = (ix->ntotal + ROW_ALIGN - 1) / ROW_ALIGN;
        if (all_filter && ix->tile_order == 0 && T >= 8 && !(tile >= 10 && tile <= 12)) {
            const int64_t P = tile_order_multiplier(T);
            if (P > 1) {
                W.extra.perm_mul = (int)P;
                W.extra.perm_mod = (int)T;
            }
        }
    }
    const SearchWorkspace& ws = W;
    for (int64_t qb = 0; qb < ps.nq; qb += MAX_NQ_PER_PASS) {
        const int64_t nq = std::min(MAX_NQ_PER_PASS, ps.nq - qb);
        const int64_t nq_pad = round_up(nq, bn);
        // one launch: queries -> store dtype with zero padded rows / columns, running top-k and counters cleared,
        // thresholds -inf (or seeded from the previous result in a recovery pass), overflow word cleared at the first pass
        // with a query map (recovery of a few queries) workspace row r is row q_map[qb + r] of the caller's arrays
        const int* q_map = ps.q_map ? ps.q_map + qb : nullptr;
        const int64_t row0 = ps.q_map ? 0 : qb;
        W.ovf_q = track_ovf ? ix->ovf_q + (size_t)ps.slot * OVF_ROWS + qb : nullptr;
        const float* margin = ps.band ? ix->x_eps + (size_t)ps.slot * OVF_ROWS + row0 : nullptr;  // BAND pass: seeded k-th exact score - eps
        HIP_OK(launch_search_prepare(ws, (const char*)ps.queries + (size_t)row0 * ix->dim * q_es, ps.q_dtype, nq, ix->dim,
                                     ix->dtype, nq_pad, ix->dim_pad, qb == 0, recovery > 0 ? ps.out_scores + row0 * k : nullptr,
                                     recovery > 0 ? ps.out_ids + row0 * k : nullptr, k, q_map, stream, margin));
        W.extra.q_label = ps.q_label ? ps.q_label + (size_t)row0 * ps.n_qlab : nullptr;
        W.extra.q_map = q_map;
        for (size_t c = 0; c < stages.size(); ++c) {
            const Stage& sg = stages[c];
            // one filter launch, bracketed by profile events (bench / roofline accounting)
            auto launch_one = [&](int tile_c, int64_t b, int64_t e) -> int {
                hipEvent_t ev1 = nullptr;
                if (ix->profile) {
                    while (ix->ev_pool.size() < ix->ev_used + 2) {
                        hipEvent_t ev;
                        HIP_OK(hipEventCreate(&ev));
                        ix->ev_pool.push_back(ev);
                    }
                    hipEvent_t ev0 = ix->ev_pool[ix->ev_used++];
                    ev1 = ix->ev_pool[ix->ev_used++];
                    HIP_OK(hipEventRecord(ev0, stream));
                }
                W.extra.sample_rstride = (int)sg.rstride;
                // S sampled rows at offset + i * rstride, i < S: the (ntotal - 1) % rstride-ish rows the integer stride leaves out
                // are split between the head and the tail of the store
                W.extra.sample_offset = sg.kind == ST_GMAX ? (int)(((ix->ntotal - 1) - (sg.n_tiles * filter_tile_rows(tile_c) - 1) * sg.rstride) / 2) : 0;
                W.extra.sample_groups = (int)sg.n_groups;
                HIP_OK(launch_filter(ix->dtype, tile_c, sg.kind, ix->data, ws.q_pad, ix->dim_pad, b, e, sg.n_tiles, nq, nq_pad, ws, stream));
                if (ix->profile) HIP_OK(hipEventRecord(ev1, stream));
                return 0;
            };
            const bool last = c + 1 == stages.size();
            int tile_c = (auto_8phase && sg.kind == ST_FILTER) ? 14 : tile;
            if (persistent && ix->tile == 0 && sg.kind == ST_FILTER) {  // (nq_pad is a multiple of 256 there, which the 128-wide tile divides)
                const int64_t q_tiles = nq_pad / 256;
                const int64_t x_tiles = (sg.e - sg.b + 255) / 256;
                // short FILTER stages do not fill the CUs with 256x256 tiles: those launches run on 128x128 tiles, 2 workgroups per CU
                if (x_tiles * q_tiles < ix->small_chunk_tiles) {
                    tile_c = 1;
                }
            }
            if (launch_one(tile_c, sg.b, sg.e)) return -1;
            int64_t dense_n = -1;
            int flags = last ? 1 : 0;
            if (sg.kind == ST_DENSE) dense_n = sg.e - sg.b;
            if (sg.kind == ST_GMAX) {
                dense_n = sg.n_groups;
                flags |= 2;
            }
            if (ps.band) {  // exact mode: the stage's candidates are re-scored from the float32 plane into the caller's rows
                ExactArgs xa;
                xa.plane = ix->data32;
                xa.stride = ix->dim_pad;
                xa.dim = (int)ix->dim;
                xa.dim_pad = (int)ix->dim_pad;
                xa.q_src = (const char*)ps.queries + (size_t)row0 * ix->dim * q_es;
                xa.q_dtype = ps.q_dtype;
                xa.q_map = q_map;
                xa.store_dtype = ix->dtype;
                exact_bound_args(ix, xa);
                xa.row_label = W.extra.row_label;
                xa.q_label = W.extra.q_label;
                xa.n_qlab = W.extra.n_qlab;
                xa.mode = EXACT_CAND | (c == 0 ? EXACT_FIRST : 0);
                xa.cand = ws.cand;
                xa.cnt = ws.cnt;
                xa.cap = (int)ws.cap;
                xa.dense_n = sg.kind == ST_DENSE ? (int)(sg.e - sg.b) : -1;
                xa.thr_s = ws.thr_s;
                xa.thr_key = ws.thr_key;
                xa.kr = 2;
                while (xa.kr < k) xa.kr <<= 1;
                xa.P = std::max(2 * xa.kr, 1024);
                xa.k = k;
                xa.id_base = ps.id_base;
                xa.out_scores = ps.out_scores + row0 * k;
                xa.out_ids = ps.out_ids + row0 * k;
                xa.flag_word = ws.overflow;
                HIP_OK(launch_exact_rescore(xa, nq, stream));
                continue;
            }
            HIP_OK(launch_select(ws, nq, k, dense_n, flags, stream, ps.id_base, ps.out_scores + row0 * k, ps.out_ids + row0 * k, q_map));
        }
        if (stages.empty())  // empty index: nothing was selected, the cleared top-k leaves as pads
            HIP_OK(launch_output(ws, nq, k, ps.id_base, ps.out_scores + qb * k, ps.out_ids + qb * k, stream));
    }
    if (ps.defer_flags) return 0;
    HIP_OK(hipMemcpyAsync(ix->overflow_host + FLAG_WORDS * ps.slot, ws.overflow, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipEventRecord(ix->done[ps.slot], stream));
    return 0;
}

int enqueue_search(vodhip_index* ix, const PendingSearch& ps, bool safe, int recovery, hipStream_t stream) {
    const int rc = enqueue_search_impl(ix, ps, safe, recovery, stream);
    // on failure some kernels of this search may already run on buffers the caller is about to release
    if (rc) (void)hipStreamSynchronize(stream);
    return rc;
}

// ---- exact mode (kernels_exact.hip) ----------------------------------------------------------------------------------------
// the scan's list length k' for the caller's k: enough rows that the list usually PROVES itself complete (rows within eps of the
// k-th score: ~5 % more than k for an fp16 store of N(0, 1) x 768 rows, ~70 % more for bf16 x 1024 - profiles/r05_exact_*.json);
// a query whose list does not is searched again as a band pass, so this is a speed knob, never a correctness one
int exact_kx(const vodhip_index* ix, int k, bool upper_limit = false) {
    // upper_limit: what the adaptive list length may grow to when lists keep failing their proof (the bf16 formula, whatever the scan dtype:
    // fp16 at dim 1024 / k 200 needs ~1.3 k, more than its 1.1 k + 16 starting point)
    const int64_t x100 = ix->exact_expand_x100 > 0 ? ix->exact_expand_x100 : ((ix->dtype == VODHIP_F16 && !upper_limit) ? 110 : 200);
    const int64_t kx = ((int64_t)k * x100 + 99) / 100 + 16;
    const int64_t fits = std::max<int64_t>(k, ix->cand_cap / ROW_ALIGN * ROW_ALIGN);  // the scan needs cand_cap >= its list length
    return (int)std::min<int64_t>(std::min<int64_t>(VODHIP_MAX_K, fits), std::max<int64_t>(kx, k));
}

// the scan behind an exact-mode search: same queries, k' results into the slot's list buffers, LOCAL ids
PendingSearch exact_inner(const vodhip_index* ix, const PendingSearch& ps) {
    PendingSearch in = ps;
    in.k = ps.kx;
    in.kx = 0;
    in.id_base = 0;
    in.out_scores = ix->x_list_s[ps.slot];
    in.out_ids = ix->x_list_i[ps.slot];
    in.defer_flags = false;
    return in;
}

// re-score the list of every query of `ps` (LIST mode), then fetch BOTH flag words (the scan's overflow word and the "some query is
// incomplete" word the re-scoring sets: the head of every search pass clears both) and record the slot's event
int enqueue_exact_list(vodhip_index* ix, const PendingSearch& ps, hipStream_t stream) {
    ExactArgs xa;
    xa.plane = ix->data32;
    xa.stride = ix->dim_pad;
    xa.dim = (int)ix->dim;
    xa.dim_pad = (int)ix->dim_pad;
    xa.q_src = ps.queries;
    xa.q_dtype = ps.q_dtype;
    xa.store_dtype = ix->dtype;
    exact_bound_args(ix, xa);
    xa.row_label = (ix->row_label && ps.q_label) ? ix->row_label : nullptr;
    xa.q_label = ps.q_label;
    xa.n_qlab = ps.n_qlab;
    xa.mode = EXACT_LIST;
    xa.list_s = ix->x_list_s[ps.slot];
    xa.list_i = ix->x_list_i[ps.slot];
    xa.kx = ps.kx;
    xa.P = 64;
    while (xa.P < ps.kx + xa.n_out) xa.P <<= 1;
    xa.k = ps.k;
    xa.id_base = ps.id_base;
    xa.out_scores = ps.out_scores;
    xa.out_ids = ps.out_ids;
    xa.eps = ix->x_eps + (size_t)ps.slot * OVF_ROWS;
    xa.flag_word = ix->ws_lane[ps.lane].overflow + 1;
    xa.flag_q = ix->x_flag_q + (size_t)ps.slot * OVF_ROWS;
    HIP_OK(launch_exact_rescore(xa, ps.nq, stream));
    HIP_OK(hipMemcpyAsync(ix->overflow_host + FLAG_WORDS * ps.slot, ix->ws_lane[ps.lane].overflow, 3 * sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipEventRecord(ix->done[ps.slot], stream));
    return 0;
}

int exact_reserve_lists(vodhip_index* ix, int slot, size_t elems) {
    if (ix->x_list_cap[slot] >= elems) return 0;
    (void)hipFree(ix->x_list_s[slot]);
    (void)hipFree(ix->x_list_i[slot]);
    ix->x_list_s[slot] = nullptr;
    ix->x_list_i[slot] = nullptr;
    ix->x_list_cap[slot] = 0;
    HIP_OK(hipMalloc((void**)&ix->x_list_s[slot], elems * sizeof(float)));
    HIP_OK(hipMalloc((void**)&ix->x_list_i[slot], elems * sizeof(int64_t)));
    ix->x_list_cap[slot] = elems;
    return 0;
}

}  // namespace

extern "C" {

const char* vodhip_last_error(void) { return g_last_error.c_str(); }
int vodhip_version(void) { return VODHIP_VERSION; }

int vodhip_index_create(int device, int64_t dim, int store_dtype, int64_t capacity_rows, vodhip_index_t** out) {
    if (!out) return fail("out is NULL");
    if (dim <= 0 || capacity_rows < 0) return fail("invalid dim=%lld / capacity=%lld", (long long)dim, (long long)capacity_rows);
    const bool exact = (store_dtype & VODHIP_EXACT_F32) != 0;
    store_dtype &= ~VODHIP_EXACT_F32;
    if (store_dtype != VODHIP_F16 && store_dtype != VODHIP_BF16) return fail("store dtype must be F16 or BF16 (optionally | VODHIP_EXACT_F32)");
    if (exact && (dim + 63) / 64 * 64 > 16384) return fail("VODHIP_EXACT_F32 stores take dim <= 16384");
    if (capacity_rows >= (1ll << 31) - 2 * ROW_ALIGN) return fail("capacity must be < 2^31 rows per device");
    HIP_OK(hipSetDevice(device));
    vodhip_index* ix = new vodhip_index();
    ix->device = device;
    ix->dim = dim;
    ix->dim_pad = round_up(dim, 64);
    ix->capacity = capacity_rows;
    ix->capacity_pad = round_up(capacity_rows, ROW_ALIGN) + 2 * ROW_ALIGN;  // the last tile (up to 384 rows from a 256-aligned start) never reads past the allocation
    ix->dtype = store_dtype;
    ix->exact = exact;
    if (hipDeviceGetAttribute(&ix->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ix->n_cu < 1) ix->n_cu = 256;
    ix->ws_lane[0].n_cu = ix->ws_lane[1].n_cu = ix->n_cu;
    const size_t bytes = (size_t)ix->capacity_pad * ix->dim_pad * 2;
    hipError_t e = hipMalloc((void**)&ix->data, bytes);
    if (e != hipSuccess) {
        delete ix;
        return fail("hipMalloc of the %zu-byte vector store failed: %s", bytes, hipGetErrorString(e));
    }
    e = hipMemset(ix->data, 0, bytes);
    // hipMemset of device memory returns before the fill has run; rows are ingested (and searched) on streams that do not order
    // themselves behind the null stream (the batcher's and the node index's are non-blocking): the fill must be over before any of that
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e == hipSuccess) e = hipMalloc((void**)&ix->ovf_q, MAX_IN_FLIGHT * OVF_ROWS * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc((void**)&ix->q_map, MAX_IN_FLIGHT * OVF_ROWS * sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc((void**)&ix->overflow_host, FLAG_WORDS * MAX_IN_FLIGHT * sizeof(unsigned int), hipHostMallocDefault);
    for (int i = 0; i < MAX_IN_FLIGHT && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ix->done[i], hipEventDisableTiming);
    for (int i = 0; i < MAX_IN_FLIGHT && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ix->lane_in[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->lane_stream, hipStreamNonBlocking);
    if (exact && e == hipSuccess) {
        // the float32 plane (rows are written whole, zero padded columns included: no fill needed) + the bound's bookkeeping
        const size_t bytes32 = ((size_t)capacity_rows + 1) * ix->dim_pad * sizeof(float);
        e = hipMalloc((void**)&ix->data32, bytes32);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(ix->data);
            (void)hipFree(ix->ovf_q);
            (void)hipFree(ix->q_map);
            (void)hipHostFree(ix->overflow_host);
            delete ix;
            return fail("hipMalloc of the %zu-byte float32 plane (VODHIP_EXACT_F32) failed: %s", bytes32, hipGetErrorString(e));
        }
        if (e == hipSuccess) e = hipMalloc((void**)&ix->norm_stats, EXS_WORDS * sizeof(unsigned int));
        if (e == hipSuccess) e = hipMemset(ix->norm_stats, 0, EXS_WORDS * sizeof(unsigned int));
        if (e == hipSuccess) e = hipMalloc((void**)&ix->row_n2, ((size_t)capacity_rows + 1) * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&ix->row_d2, ((size_t)capacity_rows + 1) * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&ix->x_eps, MAX_IN_FLIGHT * OVF_ROWS * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&ix->x_flag_q, MAX_IN_FLIGHT * OVF_ROWS * sizeof(unsigned int));
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(ix->data);
        (void)hipFree(ix->data32);
        delete ix;
        return fail("store initialisation failed: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < FLAG_WORDS * MAX_IN_FLIGHT; ++i) ix->overflow_host[i] = 0;
    *out = ix;
    return 0;
}

int vodhip_index_destroy(vodhip_index_t* ix) {
    if (!ix) return 0;
    (void)hipSetDevice(ix->device);
    if (!ix->inflight.empty()) (void)hipDeviceSynchronize();  // enqueued searches still use the store and the workspace
    free_workspace(ix, 0);
    free_workspace(ix, 1);
    if (ix->lane_stream) (void)hipStreamDestroy(ix->lane_stream);
    for (int i = 0; i < MAX_IN_FLIGHT; ++i)
        if (ix->lane_in[i]) (void)hipEventDestroy(ix->lane_in[i]);
    for (hipEvent_t e : ix->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < MAX_IN_FLIGHT; ++i)
        if (ix->done[i]) (void)hipEventDestroy(ix->done[i]);
    (void)hipFree(ix->data);
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(ix->stage_dev[b]);
        (void)hipHostFree(ix->stage_pin[b]);
        if (ix->stage_done[b]) (void)hipEventDestroy(ix->stage_done[b]);
    }
    (void)hipFree(ix->data32);
    (void)hipFree(ix->norm_stats);
    (void)hipFree(ix->row_n2);
    (void)hipFree(ix->row_d2);
    (void)hipFree(ix->x_eps);
    (void)hipFree(ix->x_flag_q);
    for (int i = 0; i < MAX_IN_FLIGHT; ++i) {
        (void)hipFree(ix->x_list_s[i]);
        (void)hipFree(ix->x_list_i[i]);
    }
    (void)hipFree(ix->row_label);
    (void)hipFree(ix->ovf_q);
    (void)hipFree(ix->q_map);
    (void)hipHostFree(ix->overflow_host);
    delete ix;
    return 0;
}

int vodhip_index_add(vodhip_index_t* ix, const void* rows, int64_t n_rows, int src_dtype, int src_location, void* stream_) {
    if (!ix) return fail("index is NULL");
    if (n_rows < 0 || (n_rows > 0 && !rows)) return fail("invalid rows");
    if (src_dtype < 0 || src_dtype > 2) return fail("invalid src_dtype %d", src_dtype);
    if (ix->ntotal + n_rows > ix->capacity)
        return fail("index full: ntotal=%lld + %lld > capacity=%lld", (long long)ix->ntotal, (long long)n_rows, (long long)ix->capacity);
    if (!ix->inflight.empty()) return fail("%d searches are in flight: finish them before adding rows", (int)ix->inflight.size());
    hipStream_t stream = (hipStream_t)stream_;
    HIP_OK(hipSetDevice(ix->device));
    ix->stats_dirty = true;  // exact mode: the bound's maxima / outliers are re-derived before the next search
    const int es = elem_size(src_dtype);
    // one launch per slice: the rounded plane, and for VODHIP_EXACT_F32 stores also the float32 plane + the row statistics
    auto ingest = [&](const void* dev_src, int64_t n, int64_t first_row) -> hipError_t {
        if (ix->exact)
            return launch_ingest_exact(dev_src, src_dtype, n, ix->dim, ix->data + (size_t)first_row * ix->dim_pad, ix->dtype,
                                       ix->data32 + (size_t)first_row * ix->dim_pad, ix->dim_pad, ix->norm_stats, ix->row_n2 + first_row,
                                       ix->row_d2 + first_row, stream);
        return launch_convert_rows(dev_src, src_dtype, n, ix->dim, ix->data + (size_t)first_row * ix->dim_pad, ix->dtype, ix->dim_pad, stream);
    };
    if (src_location == VODHIP_DEVICE) {
        HIP_OK(ingest(rows, n_rows, ix->ntotal));
        ix->ntotal += n_rows;
        return 0;
    }
    if (src_location != VODHIP_HOST) return fail("invalid src_location %d", src_location);
    // Host rows -> HBM, pipelined (round 3; the index is rebuilt every training period, /root/reference/src/vod_exps/recipes/
    // periodic_training.py:53-96, so ingest time is on the trainer's critical path):
    //   pageable source (a NumPy array, an .npy memory map, decoded zarr chunks): CPU threads copy slice i + 1 into a pinned staging
    //   slot while the DMA engine moves slice i and the convert kernel rounds it into the store - two slots, one stream;
    //   pinned / registered source: the DMA reads it in place, no CPU copy at all.
    // Round 2 pushed 64 MB slices of pageable memory through hipMemcpyAsync (staged by the runtime, synchronous) and waited after each.
    hipPointerAttribute_t attr;
    bool src_pinned = false;
    if (hipPointerGetAttributes(&attr, rows) == hipSuccess) src_pinned = attr.type == hipMemoryTypeHost;
    else (void)hipGetLastError();  // an ordinary host pointer is "invalid value" to the runtime: not an error here
    ix->last_ingest_pinned_src = src_pinned;
    const int64_t row_bytes = ix->dim * es;
    const int64_t rows_per_stage = std::max<int64_t>(1, STAGE_BYTES / row_bytes);
    for (int b = 0; b < 2; ++b) {
        if (!ix->stage_dev[b]) HIP_OK(hipMalloc(&ix->stage_dev[b], STAGE_BYTES));
        if (!src_pinned && !ix->stage_pin[b]) HIP_OK(hipHostMalloc(&ix->stage_pin[b], STAGE_BYTES, hipHostMallocDefault));
        if (!ix->stage_done[b]) HIP_OK(hipEventCreateWithFlags(&ix->stage_done[b], hipEventDisableTiming));
    }
    int n_thr = (int)ix->ingest_threads;
    if (n_thr <= 0) n_thr = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    bool used[2] = {false, false};
    int slot = 0;
    for (int64_t r = 0; r < n_rows; r += rows_per_stage, slot ^= 1) {
        const int64_t n = std::min(rows_per_stage, n_rows - r);
        const size_t bytes = (size_t)n * row_bytes;
        const char* src = (const char*)rows + (size_t)r * row_bytes;
        if (used[slot]) HIP_OK(hipEventSynchronize(ix->stage_done[slot]));  // the slot's previous slice has left the staging buffers
        const void* dma_src = src;
        if (!src_pinned) {
            char* dst = (char*)ix->stage_pin[slot];
            const int t_use = (int)std::min<size_t>((size_t)n_thr, std::max<size_t>(1, bytes >> 20));  // >= 1 MB per thread
            if (t_use <= 1) {
                memcpy(dst, src, bytes);
            } else {
                std::vector<std::thread> pool;
                const size_t per = ((bytes + t_use - 1) / t_use + 4095) & ~(size_t)4095;
                for (int t = 0; t < t_use; ++t) {
                    const size_t lo = std::min(bytes, (size_t)t * per), hi = std::min(bytes, lo + per);
                    if (hi > lo) pool.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
                }
                for (auto& th : pool) th.join();
            }
            dma_src = dst;
        }
        HIP_OK(hipMemcpyAsync(ix->stage_dev[slot], dma_src, bytes, hipMemcpyHostToDevice, stream));
        HIP_OK(ingest(ix->stage_dev[slot], n, ix->ntotal + r));
        HIP_OK(hipEventRecord(ix->stage_done[slot], stream));
        used[slot] = true;
    }
    HIP_OK(hipStreamSynchronize(stream));  // synchronous for host sources: the caller may free `rows` on return
    ix->ntotal += n_rows;
    return 0;
}

int vodhip_index_reset(vodhip_index_t* ix) {
    if (!ix) return fail("index is NULL");
    if (!ix->inflight.empty()) return fail("%d searches are in flight: finish them before resetting the index", (int)ix->inflight.size());
    if (ix->exact) {  // the row maxima of the error bound start over with the store (stale maxima would only loosen it)
        HIP_OK(hipSetDevice(ix->device));
        HIP_OK(hipDeviceSynchronize());  // ingests of the old rows may still run on some stream
        HIP_OK(hipMemset(ix->norm_stats, 0, EXS_WORDS * sizeof(unsigned int)));
        HIP_OK(hipStreamSynchronize(nullptr));
        ix->stats_dirty = true;
        ix->adapt_k = ix->adapt_kx = 0;
    }
    ix->ntotal = 0;
    return 0;
}

int vodhip_index_ntotal(const vodhip_index_t* ix, int64_t* out) {
    if (!ix || !out) return fail("NULL argument");
    *out = ix->ntotal;
    return 0;
}
int vodhip_index_dim(const vodhip_index_t* ix, int64_t* out) {
    if (!ix || !out) return fail("NULL argument");
    *out = ix->dim;
    return 0;
}
int vodhip_index_capacity(const vodhip_index_t* ix, int64_t* out) {
    if (!ix || !out) return fail("NULL argument");
    *out = ix->capacity;
    return 0;
}
int vodhip_index_data(const vodhip_index_t* ix, void** dev_ptr, int64_t* row_stride_elems, int* store_dtype) {
    if (!ix) return fail("index is NULL");
    if (dev_ptr) *dev_ptr = ix->data;
    if (row_stride_elems) *row_stride_elems = ix->dim_pad;
    if (store_dtype) *store_dtype = ix->dtype;
    return 0;
}

int vodhip_index_get_rows(const vodhip_index_t* ix, int64_t row_begin, int64_t n_rows, void* dst, int dst_location,
                          void* stream_) {
    if (!ix) return fail("index is NULL");
    if (row_begin < 0 || n_rows < 0 || row_begin + n_rows > ix->ntotal) return fail("row range out of bounds");
    if (n_rows == 0) return 0;
    if (!dst) return fail("dst is NULL");
    HIP_OK(hipSetDevice(ix->device));
    hipStream_t stream = (hipStream_t)stream_;
    const hipMemcpyKind kind = dst_location == VODHIP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    HIP_OK(hipMemcpy2DAsync(dst, (size_t)ix->dim * 2, ix->data + (size_t)row_begin * ix->dim_pad, (size_t)ix->dim_pad * 2,
                            (size_t)ix->dim * 2, (size_t)n_rows, kind, stream));
    if (dst_location != VODHIP_DEVICE) HIP_OK(hipStreamSynchronize(stream));
    return 0;
}

int vodhip_index_data_f32(const vodhip_index_t* ix, void** dev_ptr, int64_t* row_stride_elems) {
    if (!ix) return fail("index is NULL");
    if (!ix->exact) return fail("the index was not created with VODHIP_EXACT_F32: it keeps no float32 rows");
    if (dev_ptr) *dev_ptr = ix->data32;
    if (row_stride_elems) *row_stride_elems = ix->dim_pad;
    return 0;
}

int vodhip_index_get_rows_f32(const vodhip_index_t* ix, int64_t row_begin, int64_t n_rows, void* dst, int dst_location, void* stream_) {
    if (!ix) return fail("index is NULL");
    if (!ix->exact) return fail("the index was not created with VODHIP_EXACT_F32: it keeps no float32 rows");
    if (row_begin < 0 || n_rows < 0 || row_begin + n_rows > ix->ntotal) return fail("row range out of bounds");
    if (n_rows == 0) return 0;
    if (!dst) return fail("dst is NULL");
    HIP_OK(hipSetDevice(ix->device));
    hipStream_t stream = (hipStream_t)stream_;
    const hipMemcpyKind kind = dst_location == VODHIP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    HIP_OK(hipMemcpy2DAsync(dst, (size_t)ix->dim * 4, ix->data32 + (size_t)row_begin * ix->dim_pad, (size_t)ix->dim_pad * 4,
                            (size_t)ix->dim * 4, (size_t)n_rows, kind, stream));
    if (dst_location != VODHIP_DEVICE) HIP_OK(hipStreamSynchronize(stream));
    return 0;
}

int vodhip_index_set_row_labels(vodhip_index_t* ix, const int32_t* labels, int64_t n_rows, int location, void* stream_) {
    if (!ix) return fail("index is NULL");
    std::lock_guard<std::mutex> guard(ix->mu);
    // searches in flight read the labels - on the caller's stream or, with two lanes, on the index's own: like add / reset, refuse
    // (round-5 advisor: a lane-1 subset search could still be reading `row_label` while this call overwrote / freed it)
    if (!ix->inflight.empty() || ix->unfinished) return fail("%d searches are in flight: finish them before changing the row labels", (int)ix->inflight.size() + ix->unfinished);
    HIP_OK(hipSetDevice(ix->device));
    if (!labels) {  // clear
        (void)hipFree(ix->row_label);
        ix->row_label = nullptr;
        return 0;
    }
    if (n_rows < 0 || n_rows > ix->capacity) return fail("n_rows=%lld out of range", (long long)n_rows);
    hipStream_t stream = (hipStream_t)stream_;
    if (!ix->row_label) {
        HIP_OK(hipMalloc((void**)&ix->row_label, (size_t)ix->capacity_pad * sizeof(int)));
        // -1: matches no query label.  On the SAME stream as the copy below: a null-stream hipMemset is not ordered before work on a
        // non-blocking stream (the node index's shard streams), and the late fill wiped the labels just copied - every restricted query
        // of the FIRST filtered search of a process came back empty (round 4, fuzz_search seed 404 trial 1051: found, fixed, pinned by
        // tests/test_node_index_gpu.py::test_row_labels_set_on_a_non_blocking_stream_survive_the_initial_fill)
        HIP_OK(hipMemsetAsync(ix->row_label, 0xFF, (size_t)ix->capacity_pad * sizeof(int), stream));
    }
    HIP_OK(hipMemcpyAsync(ix->row_label, labels, (size_t)n_rows * sizeof(int),
                          location == VODHIP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return 0;
}

int vodhip_index_set_query_labels(vodhip_index_t* ix, const int32_t* q_labels_dev, int n_per_query) {
    if (!ix) return fail("index is NULL");
    if (q_labels_dev && (n_per_query < 1 || n_per_query > 64)) return fail("n_per_query must be in [1, 64]");
    if (q_labels_dev && !ix->row_label) return fail("set the row labels first (vodhip_index_set_row_labels)");
    std::lock_guard<std::mutex> guard(ix->mu);
    ix->q_label = q_labels_dev;
    ix->n_qlab = q_labels_dev ? n_per_query : 0;
    return 0;
}

int vodhip_index_search_async(vodhip_index_t* ix, const void* queries, int q_dtype, int64_t nq, int k, int64_t id_base,
                              float* out_scores, int64_t* out_ids, void* stream_) {
    if (!ix) return fail("index is NULL");
    if (k < 1 || k > VODHIP_MAX_K) return fail("k=%d out of range [1, %d]", k, VODHIP_MAX_K);
    if (nq < 0 || (nq > 0 && (!queries || !out_scores || !out_ids))) return fail("invalid query / output pointers");
    if (q_dtype < 0 || q_dtype > 2) return fail("invalid q_dtype %d", q_dtype);
    std::lock_guard<std::mutex> guard(ix->mu);
    if (ix->cand_cap < ROW_ALIGN || ix->cand_cap < k) return fail("cand_cap too small");
    if ((int)ix->inflight.size() + ix->unfinished >= MAX_IN_FLIGHT)
        return fail("%d searches are already in flight on this index: call vodhip_index_search_finish first", MAX_IN_FLIGHT);
    if (ix->exact && nq > OVF_ROWS) return fail("VODHIP_EXACT_F32: at most %lld queries per search", (long long)OVF_ROWS);
    HIP_OK(hipSetDevice(ix->device));
    PendingSearch ps;
    ps.active = true;
    ps.queries = queries;
    ps.q_dtype = q_dtype;
    ps.nq = nq;
    ps.k = k;
    ps.id_base = id_base;
    ps.out_scores = out_scores;
    ps.out_ids = out_ids;
    ps.q_label = ix->q_label;
    ps.n_qlab = ix->n_qlab;
    ps.slot = ix->next_slot;
    // lane: batches of one query tile alternate between the two workspaces (auto), so that a caller running one search ahead has two
    // searches on the device at once; lane 1 runs on the index's own stream, behind what the caller's stream holds now (the queries)
    hipStream_t stream = (hipStream_t)stream_;
    if (ix->exact && nq > 0 && refresh_exact_stats(ix, stream)) return -1;
    const int64_t lanes_eff = ix->lanes > 0 ? ix->lanes : (nq <= 256 ? 2 : 1);
    if (lanes_eff == 2 && nq > 0) {
        // the lane the youngest search in flight does NOT use; a caller that finishes every search before the next (nothing in flight)
        // stays on lane 0 = its own stream, with no event hop
        ps.lane = ix->inflight.empty() ? 0 : 1 - ix->inflight.back().lane;
        if (ps.lane == 1) {
            HIP_OK(hipEventRecord(ix->lane_in[ps.slot], stream));
            HIP_OK(hipStreamWaitEvent(ix->lane_stream, ix->lane_in[ps.slot], 0));
            stream = ix->lane_stream;
        }
    }
    if (ix->inflight.empty() && ix->unfinished == 0) ix->ev_used = 0;  // profile events are recycled once nothing refers to them
    ps.ev_begin = ix->ev_used;
    if (ix->exact && nq > 0) {
        // exact mode: the scan fills the slot's top-k' list, the re-scoring kernel behind it writes the caller's rows (no host round
        // trip in between: the list is re-scored even if it later turns out that a candidate list overflowed - finish repeats it then)
        const int kx_formula = exact_kx(ix, k);
        // (adapted to what the last searches of this k needed - finish() - unless the caller fixed the expansion or switched it off)
        const bool adapt = ix->exact_adapt && ix->exact_expand_x100 == 0 && ix->adapt_k == k && ix->adapt_kx >= k;
        const int kx_limit = exact_kx(ix, k, true);
        ps.kx = adapt ? std::min(ix->adapt_kx, kx_limit) : kx_formula;
        if (exact_reserve_lists(ix, ps.slot, (size_t)nq * kx_limit)) return -1;
        PendingSearch in = exact_inner(ix, ps);
        in.defer_flags = true;  // ONE copy of both flag words, behind the re-scoring launch
        if (enqueue_search(ix, in, ix->force_safe != 0, 0, stream)) return -1;
        if (enqueue_exact_list(ix, ps, stream)) {
            (void)hipStreamSynchronize(stream);
            return -1;
        }
    } else if (nq > 0 && enqueue_search(ix, ps, ix->force_safe != 0, 0, stream)) {
        return -1;
    }
    ps.ev_end = ix->ev_used;
    ix->next_slot = (ix->next_slot + 1) % MAX_IN_FLIGHT;
    ix->inflight.push_back(ps);
    return 0;
}

namespace {

// Recovery of a search whose candidate lists overflowed (overflow_host[slot] set): the result is valid (real rows, real scores)
// but may miss hits.  Recovery passes re-scan the store against thresholds seeded from that result - its k-th score is a lower
// bound of the true k-th best, so few rows survive - in 1, 2, 4, ... FILTER stages, and in the exhaustive schedule (dense chunks
// of <= cap rows, cannot overflow) once the stages are that short.  Returns the number of passes run, or -1.
int recover_overflow(vodhip_index* ix, const PendingSearch& ps, hipStream_t stream) {
    PendingSearch rs = ps;  // what the recovery passes search: the whole batch, or only the queries that overflowed
    int pass = 0;
    while (ix->overflow_host[FLAG_WORDS * ps.slot]) {
        if (++pass > 40) return fail("internal error: the exhaustive schedule overflowed");
        ix->last_overflow = 1;
        ix->last_safe_reruns += 1;
        if (pass == 1 && ps.nq <= OVF_ROWS && !ps.band) {
            // the select kernels flagged the queries whose lists overflowed: usually a few (their neighbours are
            // clustered in the store): only they are searched again, as a small batch on the small-batch kernels
            std::vector<unsigned int> flags((size_t)ps.nq);
            HIP_OK(hipMemcpy(flags.data(), ix->ovf_q + (size_t)ps.slot * OVF_ROWS, (size_t)ps.nq * sizeof(unsigned int), hipMemcpyDeviceToHost));
            std::vector<int> rows;
            for (int64_t q = 0; q < ps.nq; ++q)
                if (flags[(size_t)q]) rows.push_back((int)q);
            ix->last_recovered_queries = rows.empty() ? ps.nq : (int64_t)rows.size();
            if (!rows.empty() && (int64_t)rows.size() < ps.nq) {
                int* dmap = ix->q_map + (size_t)ps.slot * OVF_ROWS;
                HIP_OK(hipMemcpy(dmap, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
                rs.q_map = dmap;
                rs.nq = (int64_t)rows.size();
            }
        }
        const size_t ev_first = ix->ev_used;  // behind the events of every younger search in flight
        // (a BAND pass is itself "recovery" pass 1: its overflow passes start at 2 stages)
        if (enqueue_search(ix, rs, false, pass + (ps.band ? 1 : 0), stream)) return -1;
        HIP_OK(hipStreamSynchronize(stream));
        // the recovery launches are accounted separately ("last_recovery_ns"): the time they take is real
        for (size_t e = ev_first; e + 1 < ix->ev_used; e += 2) {
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, ix->ev_pool[e], ix->ev_pool[e + 1]));
            ix->last_recovery_ns += (int64_t)((double)ms * 1e6);
            ++ix->last_recovery_launches;
        }
        ix->ev_used = ev_first;
    }
    return pass;
}

}  // namespace

int vodhip_index_search_finish(vodhip_index_t* ix, void* stream_) {
    if (!ix) return fail("index is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    std::unique_lock<std::mutex> guard(ix->mu);
    if (ix->inflight.empty()) return fail("no search is pending on this index");
    HIP_OK(hipSetDevice(ix->device));
    PendingSearch ps = ix->inflight.front();
    ix->inflight.pop_front();
    if (ps.lane == 1) stream = ix->lane_stream;  // recovery / band passes of a lane-1 search run where the search ran
    if (ps.nq > 0) {
        // this search only: younger ones keep the device busy - and other threads may enqueue more meanwhile (the slot and its
        // event are not reused before MAX_IN_FLIGHT further searches, which cannot be accepted while this one counts as unfinished:
        // `unfinished` below)
        ++ix->unfinished;
        guard.unlock();
        const hipError_t ev_rc = hipEventSynchronize(ix->done[ps.slot]);
        guard.lock();
        --ix->unfinished;
        HIP_OK(ev_rc);
    }
    ix->last_overflow = 0;
    ix->last_safe_reruns = 0;
    ix->last_recovered_queries = 0;
    ix->last_recovery_launches = 0;
    ix->last_recovery_ns = 0;
    ix->last_exact_kx = ps.kx;
    ix->last_exact_band_queries = 0;
    ix->last_exact_band_passes = 0;
    ix->last_exact_need = ps.kx > 0 && ps.nq > 0 ? (int64_t)ix->overflow_host[FLAG_WORDS * ps.slot + 2] : 0;
    if (ps.kx > 0 && ps.nq > 0 && ix->exact_adapt && ix->exact_expand_x100 == 0) {
        // The list length of the NEXT searches of this k: what this one needed (the list entries within eps of the k-th exact score,
        // maximum over the queries) + 1/16 + 8, rounded up to 8; it rises at once and falls by a quarter of the gap per search.  A list
        // that did not prove complete (need == k') makes the next one a quarter longer, up to the bf16 formula whatever the scan dtype.  A speed knob only: any k' >= k returns
        // the same result (the band pass covers what a short list misses).
        const int need = (int)ix->overflow_host[FLAG_WORDS * ps.slot + 2];
        const int formula = exact_kx(ix, ps.k), limit = exact_kx(ix, ps.k, true);
        // (a list that did not prove complete - need == k' - grows by a quarter, up to the limit; one that did settles at need + 1/16 + 8)
        int target = need >= ps.kx ? std::min(limit, (ps.kx + ps.kx / 4 + 7) / 8 * 8) : std::min(limit, (need + need / 16 + 8 + 7) / 8 * 8);
        target = std::max(target, std::min(formula, ps.k + 8));
        if (ix->adapt_k != ps.k || ix->adapt_kx <= 0) ix->adapt_kx = formula;
        ix->adapt_k = ps.k;
        ix->adapt_kx = target >= ix->adapt_kx ? target : ix->adapt_kx - std::max(8, (ix->adapt_kx - target) / 4 / 8 * 8);
        if (ix->adapt_kx < target) ix->adapt_kx = target;
    }
    if (ps.nq > 0 && ps.kx == 0) {
        if (recover_overflow(ix, ps, stream) < 0) return -1;
    } else if (ps.nq > 0) {
        // exact mode.  (1) the scan's own exactness: a recovered list is re-scored again (the first re-scoring saw the incomplete one)
        const int passes = recover_overflow(ix, exact_inner(ix, ps), stream);
        if (passes < 0) return -1;
        if (passes > 0) {
            if (enqueue_exact_list(ix, ps, stream)) return -1;
            HIP_OK(hipStreamSynchronize(stream));
        }
        // (2) the lists that did not prove complete: those queries run a BAND pass (every row whose scan score is within eps of
        // the k-th exact score is re-scored); its own candidate lists may overflow, which splits the pass like any recovery
        if (ix->overflow_host[FLAG_WORDS * ps.slot + 1]) {
            std::vector<unsigned int> flags((size_t)ps.nq);
            HIP_OK(hipMemcpy(flags.data(), ix->x_flag_q + (size_t)ps.slot * OVF_ROWS, (size_t)ps.nq * sizeof(unsigned int), hipMemcpyDeviceToHost));
            std::vector<int> rows;
            for (int64_t q = 0; q < ps.nq; ++q)
                if (flags[(size_t)q]) rows.push_back((int)q);
            ix->last_exact_band_queries = (int64_t)rows.size();
            if (!rows.empty()) {
                PendingSearch bs = ps;
                bs.band = true;
                bs.kx = 0;
                int* dmap = ix->q_map + (size_t)ps.slot * OVF_ROWS;
                if ((int64_t)rows.size() < ps.nq) {
                    HIP_OK(hipMemcpy(dmap, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
                    bs.q_map = dmap;
                    bs.nq = (int64_t)rows.size();
                }
                const size_t ev_first = ix->ev_used;
                if (enqueue_search(ix, bs, false, 1, stream)) return -1;
                HIP_OK(hipStreamSynchronize(stream));
                for (size_t e = ev_first; e + 1 < ix->ev_used; e += 2) {
                    float ms = 0.f;
                    HIP_OK(hipEventElapsedTime(&ms, ix->ev_pool[e], ix->ev_pool[e + 1]));
                    ix->last_recovery_ns += (int64_t)((double)ms * 1e6);
                    ++ix->last_recovery_launches;
                }
                ix->ev_used = ev_first;
                const int more = recover_overflow(ix, bs, stream);
                if (more < 0) return -1;
                ix->last_exact_band_passes = 1 + more;
            }
        }
    }
    ix->last_filter_launches = (int64_t)((ps.ev_end - ps.ev_begin) / 2);
    ix->last_filter_ns = 0;
    for (size_t e = ps.ev_begin; e + 1 < ps.ev_end; e += 2) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, ix->ev_pool[e], ix->ev_pool[e + 1]));
        ix->last_filter_ns += (int64_t)((double)ms * 1e6);
    }
    return 0;
}

int vodhip_index_search(vodhip_index_t* ix, const void* queries, int q_dtype, int64_t nq, int k, int64_t id_base,
                        float* out_scores, int64_t* out_ids, void* stream) {
    if (vodhip_index_search_async(ix, queries, q_dtype, nq, k, id_base, out_scores, out_ids, stream)) return -1;
    return vodhip_index_search_finish(ix, stream);
}

int vodhip_debug_schedule(int64_t ntotal, int k, int64_t nq, int64_t cand_cap, int64_t dense_rows, int64_t sample_div,
                          int64_t growth_x100, int tile, int recovery_pass, int n_cu, int64_t* out, int max_stages) {
    if (!out || max_stages < 1 || k < 1 || nq < 1 || ntotal < 0) return fail("invalid arguments");
    vodhip_index tmp;  // host-side planning only: no device call uses it
    tmp.ntotal = ntotal;
    tmp.n_cu = n_cu > 0 ? n_cu : 256;
    if (cand_cap > 0) tmp.cand_cap = cand_cap;
    if (dense_rows > 0) tmp.dense_rows = round_up(dense_rows, ROW_ALIGN);
    if (sample_div > 0) tmp.sample_div = sample_div;
    tmp.growth_x100 = growth_x100;
    if (tile == 0) tile = nq > 128 ? 8 : (nq > 64 ? 46 : 42);
    std::vector<Stage> st;
    make_schedule(&tmp, k, tile, round_up(std::min(MAX_NQ_PER_PASS, nq), filter_tile_cols(tile)), false, recovery_pass, st);
    if ((int)st.size() > max_stages) return fail("%d stages do not fit max_stages=%d", (int)st.size(), max_stages);
    for (size_t i = 0; i < st.size(); ++i) {
        int64_t* o = out + i * 6;
        o[0] = st[i].kind;
        o[1] = st[i].b;
        o[2] = st[i].e;
        o[3] = st[i].n_tiles;
        o[4] = st[i].rstride;
        o[5] = st[i].n_groups;
    }
    return (int)st.size();
}

int vodhip_debug_tile_order(int64_t ntotal, int64_t* perm_mul, int64_t* perm_mod) {
    if (!perm_mul || !perm_mod || ntotal < 0) return fail("invalid arguments");
    *perm_mod = (ntotal + ROW_ALIGN - 1) / ROW_ALIGN;
    *perm_mul = tile_order_multiplier(*perm_mod);
    return 0;
}

int vodhip_index_set_param(vodhip_index_t* ix, const char* key, int64_t value) {
    if (!ix || !key) return fail("NULL argument");
    if (!strcmp(key, "cand_cap")) {
        if (value < ROW_ALIGN || value > (1 << 16)) return fail("cand_cap must be in [256, 65536]");
        ix->cand_cap = value;
    } else if (!strcmp(key, "dense_rows")) {
        if (value < ROW_ALIGN) return fail("dense_rows must be >= 256");
        ix->dense_rows = round_up(value, ROW_ALIGN);
    } else if (!strcmp(key, "growth")) {
        ix->growth_x100 = value;  // growth factor * 100; 0 = derive from k
    } else if (!strcmp(key, "force_safe")) {
        ix->force_safe = value;
    } else if (!strcmp(key, "small_chunk_tiles")) {
        ix->small_chunk_tiles = value;
    } else if (!strcmp(key, "kflags")) {
        ix->kflags = value;
    } else if (!strcmp(key, "sample_div")) {
        if (value != 0 && value < 2) return fail("sample_div must be 0 (auto) or >= 2");
        ix->sample_div = value;
    } else if (!strcmp(key, "profile")) {
        ix->profile = value;
    } else if (!strcmp(key, "ingest_threads")) {
        if (value < 0 || value > 256) return fail("ingest_threads must be in [0, 256] (0 = auto)");
        ix->ingest_threads = value;
    } else if (!strcmp(key, "lanes")) {
        if (value < 0 || value > 2) return fail("lanes must be 0 (auto: 2 for batches of one query tile), 1 or 2");
        if (!ix->inflight.empty()) return fail("searches are in flight: finish them before changing the lanes");
        ix->lanes = value;
    } else if (!strcmp(key, "tile_order")) {
        if (value != 0 && value != 1) return fail("tile_order must be 0 (low-discrepancy stage order) or 1 (row order)");
        ix->tile_order = value;
    } else if (!strcmp(key, "exact_adapt")) {
        if (value < 0 || value > 1) return fail("exact_adapt must be 0 or 1");
        ix->exact_adapt = value;
        ix->adapt_k = ix->adapt_kx = 0;
    } else if (!strcmp(key, "exact_expand")) {
        if (value < 0 || value > 100000) return fail("exact_expand (x100) must be in [0, 100000]");
        ix->exact_expand_x100 = value;
    } else if (!strcmp(key, "tile")) {
#ifdef VODHIP_EXPERIMENTS
        const bool ring_ok = true;  // tiles 10 / 11 / 12: the FILTER kernels of experiment builds
#else
        const bool ring_ok = false;
#endif
        if (value != 0 && value != 1 && value != 8 && value != 9 && value != 14 && value != 42 && value != 46 && !(ring_ok && ((value >= 10 && value <= 13) || (value >= 15 && value <= 18))))
            return fail("tile must be 0 (auto) or a filter-kernel variant id: 1, 8, 9, 14, 42, 46 (DESIGN.md 4)");
        ix->tile = value;
    } else {
        return fail("unknown parameter '%s'", key);
    }
    return 0;
}

int vodhip_index_get_stat(const vodhip_index_t* ix, const char* key, int64_t* out) {
    if (!ix || !key || !out) return fail("NULL argument");
    if (!strcmp(key, "last_overflow"))
        *out = ix->last_overflow;
    else if (!strcmp(key, "last_chunks"))
        *out = ix->last_chunks;
    else if (!strcmp(key, "last_safe_reruns"))
        *out = ix->last_safe_reruns;
    else if (!strcmp(key, "last_recovered_queries"))
        *out = ix->last_recovered_queries;
    else if (!strcmp(key, "last_filter_launches"))
        *out = ix->last_filter_launches;
    else if (!strcmp(key, "last_filter_ns"))
        *out = ix->last_filter_ns;
    else if (!strcmp(key, "last_recovery_launches"))
        *out = ix->last_recovery_launches;
    else if (!strcmp(key, "last_recovery_ns"))
        *out = ix->last_recovery_ns;
    else if (!strcmp(key, "exact"))
        *out = ix->exact ? 1 : 0;
    else if (!strcmp(key, "last_exact_kx"))
        *out = ix->last_exact_kx;
    else if (!strcmp(key, "last_exact_need"))
        *out = ix->last_exact_need;
    else if (!strcmp(key, "exact_outliers")) {  // (derived lazily: before the first search after rows were added this is the previous value)
        *out = ix->n_out;
    } else if (!strcmp(key, "last_exact_band_queries"))
        *out = ix->last_exact_band_queries;
    else if (!strcmp(key, "last_exact_band_passes"))
        *out = ix->last_exact_band_passes;
    else if (!strcmp(key, "cand_cap"))
        *out = ix->cand_cap;
    else if (!strcmp(key, "dense_rows"))
        *out = ix->dense_rows;
    else if (!strcmp(key, "sample_div"))
        *out = ix->sample_div;
    else if (!strcmp(key, "dim_pad"))
        *out = ix->dim_pad;
    else if (!strcmp(key, "last_ingest_pinned_src"))
        *out = ix->last_ingest_pinned_src;
    else
        return fail("unknown stat '%s'", key);
    return 0;
}

// One merge launch holds n_shards * k entries of a query in LDS (<= 8192).  More than that (e.g. 8 shards x top-2048) is
// merged in levels: groups of floor(8192 / k) shards -> their top-min(k_out, group size * k) in a stream-ordered temporary
// (top-k of a union = top-k of the per-part top-k's), then the groups.
static int merge_topk_levels(const float* scores, int64_t stride_s, const int64_t* ids, int64_t stride_i, int n_shards, int64_t nq,
                             int k, int k_out, float* out_scores, int64_t* out_ids, hipStream_t stream) {
    if ((int64_t)n_shards * k <= 8192) {
        HIP_OK(launch_merge_topk(scores, ids, stride_s, stride_i, n_shards, nq, k, k_out, out_scores, out_ids, stream));
        return 0;
    }
    const int per_group = 8192 / k;
    if (per_group < 2) return fail("k = %d: lists longer than 4096 entries cannot be merged", k);
    const int n_groups = (n_shards + per_group - 1) / per_group;
    const int k_mid = (int)std::min<int64_t>(k_out, (int64_t)per_group * k);
    float* mid_s = nullptr;
    int64_t* mid_i = nullptr;
    const size_t n_mid = (size_t)n_groups * (size_t)nq * (size_t)k_mid;
    HIP_OK(hipMallocAsync((void**)&mid_s, n_mid * sizeof(float), stream));
    if (hipError_t e = hipMallocAsync((void**)&mid_i, n_mid * sizeof(int64_t), stream); e != hipSuccess) {
        (void)hipFreeAsync(mid_s, stream);
        return fail("HIP error: %s", hipGetErrorString(e));
    }
    int rc = 0;
    for (int g = 0; g < n_groups && rc == 0; ++g) {
        const int s0 = g * per_group, ns = std::min(per_group, n_shards - s0);
        if (hipError_t e = launch_merge_topk(scores + (size_t)s0 * stride_s, ids + (size_t)s0 * stride_i, stride_s, stride_i, ns, nq, k, k_mid,
                                             mid_s + (size_t)g * nq * k_mid, mid_i + (size_t)g * nq * k_mid, stream);
            e != hipSuccess)
            rc = fail("HIP error: %s", hipGetErrorString(e));
    }
    if (rc == 0) rc = merge_topk_levels(mid_s, nq * k_mid, mid_i, nq * k_mid, n_groups, nq, k_mid, k_out, out_scores, out_ids, stream);
    (void)hipFreeAsync(mid_s, stream);
    (void)hipFreeAsync(mid_i, stream);
    return rc;
}

int vodhip_merge_topk(const float* scores, const int64_t* ids, int n_shards, int64_t nq, int k, int k_out,
                      float* out_scores, int64_t* out_ids, void* stream) {
    if (n_shards < 1 || k < 1 || k_out < 1 || nq < 0) return fail("invalid sizes");
    if (nq > 0 && (!scores || !ids || !out_scores || !out_ids)) return fail("NULL argument");
    return merge_topk_levels(scores, nq * k, ids, nq * k, n_shards, nq, k, k_out, out_scores, out_ids, (hipStream_t)stream);
}

int vodhip_merge_topk_strided(const float* scores, int64_t shard_stride_scores, const int64_t* ids, int64_t shard_stride_ids,
                              int n_shards, int64_t nq, int k, int k_out, float* out_scores, int64_t* out_ids, void* stream) {
    if (n_shards < 1 || k < 1 || k_out < 1 || nq < 0) return fail("invalid sizes");
    if (nq > 0 && (!scores || !ids || !out_scores || !out_ids)) return fail("NULL argument");
    if (shard_stride_scores < nq * k || shard_stride_ids < nq * k) return fail("shard strides smaller than nq * k");
    return merge_topk_levels(scores, shard_stride_scores, ids, shard_stride_ids, n_shards, nq, k, k_out, out_scores, out_ids,
                             (hipStream_t)stream);
}

int vodhip_merge_hybrid(const int64_t* lookup_idx, const int64_t* lookup_lbl, int k_lookup, int n_engines,
                        const int64_t* const* engine_idx, const float* const* engine_scr, const int* engine_k,
                        const float* engine_weight, int64_t nq, int64_t* out_idx, float* out_scr, int64_t* out_lbl,
                        float* const* out_raw, int out_stride, int32_t* out_width, int32_t* out_row_cursor, void* stream) {
    if (n_engines < 0 || n_engines > VODHIP_MAX_ENGINES) return fail("n_engines=%d out of range", n_engines);
    if (k_lookup < 0 || nq < 0) return fail("invalid sizes");
    HybridArgs a;
    memset(&a, 0, sizeof(a));
    a.lookup_idx = lookup_idx;
    a.lookup_lbl = lookup_lbl;
    a.k_lookup = k_lookup;
    a.n_engines = n_engines;
    int64_t width = k_lookup;
    for (int e = 0; e < n_engines; ++e) {
        if (engine_k[e] < 0 || (engine_k[e] > 0 && (!engine_idx[e] || !engine_scr[e]))) return fail("engine %d: invalid arguments", e);
        a.engine_idx[e] = engine_idx[e];
        a.engine_scr[e] = engine_scr[e];
        a.engine_k[e] = engine_k[e];
        a.engine_w[e] = engine_weight[e];
        a.out_raw[e] = out_raw ? out_raw[e] : nullptr;
        width += engine_k[e];
    }
    if (out_stride < width + 1) return fail("out_stride=%d < %lld", out_stride, (long long)width + 1);
    if (width > 4096) return fail("total width %lld exceeds 4096", (long long)width);
    a.nq = nq;
    a.out_idx = out_idx;
    a.out_scr = out_scr;
    a.out_lbl = out_lbl;
    a.out_stride = out_stride;
    a.out_width = out_width;
    a.out_row_cursor = out_row_cursor;
    if (!out_idx || !out_scr) return fail("NULL output");
    HIP_OK(launch_merge_hybrid(a, (hipStream_t)stream));
    return 0;
}

int vodhip_retrieval_forward(const void* q, const void* s, int enc_dtype, int sections_3d, int64_t B, int64_t D, int64_t H,
                             const float* score, const int64_t* relevance, const float* sparse, const float* dense,
                             float* retriever_scores, float* d_scores, float* loss, float* kl, float* workspace,
                             void* stream) {
    if (!q || !s || !score || !relevance || !retriever_scores || !d_scores || !loss || !kl || !workspace)
        return fail("NULL argument");
    if (B <= 0 || D <= 0 || H <= 0) return fail("invalid sizes");
    if (D > 16384) return fail("D=%lld exceeds 16384 sections per row", (long long)D);
    if (enc_dtype < 0 || enc_dtype > 2) return fail("invalid enc_dtype");
    HIP_OK(launch_retrieval_forward(q, s, enc_dtype, sections_3d, B, D, H, score, relevance, sparse, dense,
                                    retriever_scores, d_scores, loss, kl, workspace, RetrievalAux(), (hipStream_t)stream));
    return 0;
}

int vodhip_retrieval_forward_aux(const void* q, const void* s, int enc_dtype, int sections_3d, int64_t B, int64_t D, int64_t H,
                                 const float* score, const int64_t* relevance, const float* sparse, const float* dense,
                                 int guidance_type, float guidance_weight, float self_supervision_weight, float score_decay,
                                 float* retriever_scores, float* d_scores, float* loss, float* kl, float* aux_losses,
                                 float* aux_grad, float* workspace, int64_t workspace_floats, void* stream) {
    if (workspace_floats < 16 * B) return fail("workspace_floats=%lld < 16 * B", (long long)workspace_floats);
    if (!q || !s || !score || !relevance || !retriever_scores || !d_scores || !loss || !kl || !workspace || !aux_losses)
        return fail("NULL argument");
    if (B <= 0 || D <= 0 || H <= 0) return fail("invalid sizes");
    if (D > 16384) return fail("D=%lld exceeds 16384 sections per row", (long long)D);
    if (enc_dtype < 0 || enc_dtype > 2) return fail("invalid enc_dtype");
    if (guidance_type != 0 && guidance_type != 1) return fail("guidance_type must be 0 (zero) or 1 (sparse)");
    if (guidance_weight < 0 || self_supervision_weight < 0 || score_decay < 0) return fail("negative auxiliary weight");
    const bool any = guidance_weight > 0 || self_supervision_weight > 0 || score_decay > 0;
    if (any && !aux_grad) return fail("aux_grad (3*B*D floats) is required when an auxiliary weight is > 0");
    if (guidance_weight > 0 && guidance_type == 1 && !sparse) return fail("guidance='sparse' needs the sparse scores");
    RetrievalAux aux;
    aux.enabled = 1;
    aux.guidance_type = guidance_type;
    aux.w_guidance = guidance_weight;
    aux.w_self = self_supervision_weight;
    aux.w_decay = score_decay;
    aux.grad = aux_grad;
    aux.out = aux_losses;
    HIP_OK(launch_retrieval_forward(q, s, enc_dtype, sections_3d, B, D, H, score, relevance, sparse, dense,
                                    retriever_scores, d_scores, loss, kl, workspace, aux, (hipStream_t)stream, workspace_floats));
    return 0;
}

int vodhip_retrieval_backward(const void* q, const void* s, int enc_dtype, int sections_3d, int64_t B, int64_t D, int64_t H,
                              const float* d_scores, const float* grad_out, float* dq, float* ds, void* stream) {
    if (!q || !s || !d_scores || !grad_out || !dq || !ds) return fail("NULL argument");
    if (B <= 0 || D <= 0 || H <= 0) return fail("invalid sizes");
    if (enc_dtype < 0 || enc_dtype > 2) return fail("invalid enc_dtype");
    HIP_OK(launch_retrieval_backward(q, s, enc_dtype, sections_3d, B, D, H, d_scores, grad_out, dq, ds, (hipStream_t)stream));
    return 0;
}

int vodhip_priority_sample(const float* scores, const uint8_t* labels, const float* noise, int64_t nq, int width,
                           int k_positive, int k_total, float temperature, int max_support_size, int normalized,
                           int64_t* out_samples, float* out_log_weights, uint8_t* out_labels, float* out_lse,
                           void* stream) {
    if (nq < 0 || width < 0 || k_total < 1 || k_positive < 0) return fail("invalid sizes");
    if (width > 4096) return fail("width=%d exceeds 4096 candidates per row", width);
    if (k_total > 4096) return fail("k_total=%d exceeds 4096", k_total);
    if (nq == 0) return 0;
    if (!scores || !labels || !noise || !out_samples || !out_log_weights || !out_labels || !out_lse) return fail("NULL argument");
    HIP_OK(launch_priority_sample(scores, labels, noise, nq, width, k_positive, k_total, temperature, max_support_size,
                                  normalized, out_samples, out_log_weights, out_labels, out_lse, (hipStream_t)stream));
    return 0;
}

int vodhip_priority_sample_merged(const int64_t* ids, const float* scores, const int64_t* labels, int n_raw, const float* const* raw,
                                  const float* noise, int64_t noise_stride, int64_t nq, int stride, int width,
                                  const int32_t* merge_width, const int32_t* merge_row_cursor, int k_lookup, int n_engines,
                                  const int* engine_k, int k_positive,
                                  int k_total, float temperature, int max_support_size, int normalized, int64_t* out_samples,
                                  int64_t* out_ids, float* out_scores, float* out_log_weights, uint8_t* out_labels,
                                  float* const* out_raw, float* out_lse, float* out_max_sampling_id, void* stream) {
    if (nq < 0 || stride < 0 || k_total < 1 || k_positive < 0) return fail("invalid sizes");
    if (stride > 4096 || width > stride) return fail("stride=%d / width=%d: at most 4096 candidates per row, width <= stride", stride, width);
    if (k_total > 4096) return fail("k_total=%d exceeds 4096", k_total);
    if (n_raw < 0 || n_raw > VODHIP_MAX_ENGINES || (n_raw && (!raw || !out_raw))) return fail("n_raw=%d out of range / NULL raw arrays", n_raw);
    if (noise_stride < (width >= 0 ? width : stride)) return fail("noise rows are shorter than the candidate rows");
    if (nq == 0) return 0;
    if (!ids || !scores || !labels || !noise || !out_samples || !out_ids || !out_scores || !out_log_weights || !out_labels || !out_lse)
        return fail("NULL argument");
    SampleMergedArgs m{};
    if (width < 0) {
        if ((n_engines > 0 && !merge_width && !merge_row_cursor) || !engine_k || n_engines < 0 || n_engines > VODHIP_MAX_ENGINES || k_lookup < 0)
            return fail("width < 0 needs merge_width or merge_row_cursor, k_lookup and engine_k of the merge that produced the rows");
        int64_t w_max = k_lookup;
        for (int e = 0; e < n_engines; ++e) {
            if (engine_k[e] < 0) return fail("engine %d: negative k", e);
            m.engine_k[e] = engine_k[e];
            w_max += engine_k[e];
        }
        if (n_engines > 0 && stride < w_max + 1) return fail("stride=%d < k_lookup + sum(k_e) + 1 = %lld", stride, (long long)w_max + 1);
        if (n_engines == 0 && stride < w_max) return fail("stride=%d < k_lookup", stride);
    }
    m.ids = ids;
    m.scores = scores;
    m.labels = labels;
    m.noise = noise;
    m.nq = nq;
    m.stride = stride;
    m.noise_stride = noise_stride;
    m.width = width;
    m.merge_width = merge_width;
    m.row_cursor = merge_row_cursor;
    m.k_lookup = k_lookup;
    m.n_engines = width < 0 ? n_engines : 0;
    m.k_positive = k_positive;
    m.k_total = k_total;
    m.temperature = temperature;
    m.max_support = max_support_size;
    m.normalized = normalized;
    m.n_raw = n_raw;
    for (int e = 0; e < n_raw; ++e) {
        if (!raw[e] || !out_raw[e]) return fail("NULL raw array %d", e);
        m.raw[e] = raw[e];
        m.out_raw[e] = out_raw[e];
    }
    m.out_samples = out_samples;
    m.out_ids = out_ids;
    m.out_scores = out_scores;
    m.out_logw = out_log_weights;
    m.out_labels = out_labels;
    m.out_lse = out_lse;
    m.lse_row_stride = 2;
    m.lse_cls_stride = 1;
    m.out_max_sampling_id = out_max_sampling_id;
    HIP_OK(launch_priority_sample_merged(m, (hipStream_t)stream));
    return 0;
}

int vodhip_flatten_inbatch(const int64_t* ids, int64_t n_rows, int n_keys, int n_values, const float* const* values, const float* fill,
                           float* const* outs, const uint8_t* labels, uint8_t* out_labels, int64_t* out_unique, int32_t* out_n_unique,
                           void* stream) {
    if (n_rows < 0 || n_keys < 0 || n_values < 0 || n_values > 8) return fail("invalid sizes (n_values <= 8)");
    if ((labels == nullptr) != (out_labels == nullptr)) return fail("labels and out_labels go together");
    if (n_rows * (int64_t)n_keys > 8192) return fail("%lld ids in the batch: the one-launch flattening holds at most 8192", (long long)(n_rows * n_keys));
    if (n_keys > 1024) return fail("%d ids per row: the one-launch flattening stages at most 1024 per row in LDS (use vodhip_gather_by_id)", n_keys);
    if (n_rows == 0 || n_keys == 0) return 0;
    if (!ids || !out_unique || (n_values && (!values || !fill || !outs))) return fail("NULL argument");
    for (int v = 0; v < n_values; ++v)
        if (!outs[v] || !values[v]) return fail("NULL value / output array %d", v);
    HIP_OK(launch_flatten_inbatch(ids, n_rows, n_keys, n_values, values, fill, outs, labels, out_labels, out_unique, out_n_unique,
                                  (hipStream_t)stream));
    return 0;
}

int vodhip_collate(const vodhip_collate_args_t* c, void* stream_) {
    if (!c) return fail("args is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    if (c->n_engines < 1 || c->n_engines > VODHIP_MAX_ENGINES) return fail("n_engines=%d out of range [1, %d]", c->n_engines, VODHIP_MAX_ENGINES);
    if (c->k_lookup < 0 || c->nq < 0 || c->k_total < 1 || c->k_positive < 0) return fail("invalid sizes");
    int64_t stride = c->k_lookup + 1;
    for (int e = 0; e < c->n_engines; ++e) {
        if (c->engine_k[e] < 0 || (c->engine_k[e] > 0 && (!c->engine_idx[e] || !c->engine_scr[e]))) return fail("engine %d: invalid arguments", e);
        if (!c->merged_raw[e] || !c->out_raw[e]) return fail("engine %d: NULL raw-score workspace / output", e);
        stride += c->engine_k[e];
    }
    if (stride - 1 > 4096) return fail("total width %lld exceeds 4096", (long long)stride - 1);
    if (c->k_total > 4096) return fail("k_total=%d exceeds 4096", c->k_total);
    if (c->noise_stride < stride) return fail("noise rows (%lld) are shorter than the merged rows (%lld)", (long long)c->noise_stride, (long long)stride);
    if (c->nq == 0) return 0;
    if ((c->k_lookup && !c->lookup_idx) || !c->noise || !c->merged_idx || !c->merged_lbl || !c->merged_scr || !c->row_cursor || !c->out_local ||
        !c->out_ids || !c->out_scores || !c->out_log_weights || !c->out_labels || !c->out_lse_pos || !c->out_lse_neg)
        return fail("NULL argument");
    HybridArgs h;
    memset(&h, 0, sizeof(h));
    h.lookup_idx = c->lookup_idx;
    h.lookup_lbl = c->lookup_lbl;
    h.k_lookup = c->k_lookup;
    h.n_engines = c->n_engines;
    for (int e = 0; e < c->n_engines; ++e) {
        h.engine_idx[e] = c->engine_idx[e];
        h.engine_scr[e] = c->engine_scr[e];
        h.engine_k[e] = c->engine_k[e];
        h.engine_w[e] = c->engine_weight[e];
        h.out_raw[e] = c->merged_raw[e];
    }
    h.nq = c->nq;
    h.out_idx = c->merged_idx;
    h.out_scr = c->merged_scr;
    h.out_lbl = c->merged_lbl;
    h.out_stride = (int)stride;
    h.out_row_cursor = c->row_cursor;
    HIP_OK(launch_merge_hybrid(h, stream));
    SampleMergedArgs m{};
    m.ids = c->merged_idx;
    m.scores = c->merged_scr;
    m.labels = c->merged_lbl;
    m.noise = c->noise;
    m.nq = c->nq;
    m.stride = stride;
    m.noise_stride = c->noise_stride;
    m.width = -1;
    m.row_cursor = c->row_cursor;
    m.k_lookup = c->k_lookup;
    m.n_engines = c->n_engines;
    m.n_raw = c->n_engines;
    for (int e = 0; e < c->n_engines; ++e) {
        m.engine_k[e] = c->engine_k[e];
        m.raw[e] = c->merged_raw[e];
        m.out_raw[e] = c->out_raw[e];
    }
    m.k_positive = c->k_positive;
    m.k_total = c->k_total;
    m.temperature = c->temperature;
    m.max_support = c->max_support_size;
    m.normalized = 1 | (c->flags & VODHIP_SAMPLE_KEEP_TOP_SUPPORT);
    m.out_samples = c->out_local;
    m.out_ids = c->out_ids;
    m.out_scores = c->out_scores;
    m.out_logw = c->out_log_weights;
    m.out_labels = c->out_labels;
    // lse_pos / lse_neg are two [nq] arrays: class stride = their distance (any two arrays of the same type can be addressed so)
    m.out_lse = c->out_lse_pos;
    m.lse_row_stride = 1;
    m.lse_cls_stride = c->out_lse_neg - c->out_lse_pos;
    m.out_max_sampling_id = c->out_max_sampling_id;
    HIP_OK(launch_priority_sample_merged(m, stream));
    if (!c->in_batch_negatives) return 0;
    if (c->nq * (int64_t)c->k_total > 8192) return fail("%lld ids in the batch: the one-launch flattening holds at most 8192", (long long)(c->nq * c->k_total));
    if (c->k_total > 1024) return fail("k_total=%d: the one-launch flattening stages at most 1024 ids per row", c->k_total);
    if (!c->flat_ids || !c->flat_scores || !c->flat_log_weights || !c->flat_labels) return fail("NULL flattened output");
    const float* values[8];
    float* outs[8];
    float fill[8];
    int nv = 0;
    values[nv] = c->out_scores, outs[nv] = c->flat_scores, fill[nv++] = __builtin_nanf("");
    values[nv] = c->out_log_weights, outs[nv] = c->flat_log_weights, fill[nv++] = __builtin_nanf("");
    for (int e = 0; e < c->n_engines; ++e) {
        if (!c->flat_raw[e]) return fail("engine %d: NULL flattened raw-score output", e);
        values[nv] = c->out_raw[e], outs[nv] = c->flat_raw[e], fill[nv++] = __builtin_nanf("");
    }
    HIP_OK(launch_flatten_inbatch(c->out_ids, c->nq, c->k_total, nv, values, fill, outs, c->out_labels, c->flat_labels, c->flat_ids,
                                  c->flat_n_unique, stream));
    return 0;
}

int vodhip_gather_by_id(const int64_t* queries, int64_t n_queries, const int64_t* keys, int64_t n_rows, int n_keys,
                        int n_values, const float* const* values, const float* fill, float* const* outs, void* stream) {
    if (n_queries < 0 || n_rows < 0 || n_keys < 0 || n_values < 1 || n_values > 8) return fail("invalid sizes (1 <= n_values <= 8)");
    if (n_keys > 4096) return fail("n_keys=%d exceeds 4096 candidates per row", n_keys);
    if (n_queries >= (1ll << 31)) return fail("too many query ids");
    if (n_queries == 0 || n_rows == 0) return 0;
    if (!queries || !values || !fill || !outs || (n_keys && !keys)) return fail("NULL argument");
    for (int v = 0; v < n_values; ++v)
        if (!outs[v] || (n_keys && !values[v])) return fail("NULL value / output array %d", v);
    HIP_OK(launch_gather_by_id(queries, n_queries, keys, n_rows, n_keys, n_values, values, fill, outs, (hipStream_t)stream));
    return 0;
}

int vodhip_debug_read_probe(int which, int64_t* out, int n) {
    if (!out || n < 256) return fail("out must hold 256 values");
    static_assert(sizeof(long long) == sizeof(int64_t), "probe words are 64-bit");
    const hipError_t e = which == 0 ? read_probe_hybrid((long long*)out) : which == 3 ? read_probe_select((long long*)out) : read_probe_sample(which, (long long*)out);
    if (e == hipErrorNotSupported) return fail("phase stamps exist in diagnostic builds only (make ABLATION=1)");
    if (e != hipSuccess) return fail("HIP error: %s", hipGetErrorString(e));
    return 0;
}

// (the wire codec - vodhip_b64url_encode / vodhip_b64url_decode - lives in wire_codec.cpp: host C++ with AVX2 paths)

}  // extern "C"


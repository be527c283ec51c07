// Internal declarations shared by the kernel translation units and the C-ABI layer of libvodhip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Diagnostic builds (`make ABLATION=1`): phase stamps of workgroup 0 of the collate-side kernels - shader-clock cycles
// (s_memtime) and the constant 100 MHz counter (s_memrealtime) - read back with vodhip_debug_read_probe.  Nothing in production.
#ifdef VODHIP_ABLATION
#define VODHIP_PROBE(buf, i)                                                  \
    do {                                                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                            \
            (buf)[2 * (i)] = (long long)__builtin_amdgcn_s_memtime();         \
            (buf)[2 * (i) + 1] = (long long)__builtin_amdgcn_s_memrealtime(); \
        }                                                                     \
    } while (0)
// begin / end stamp of EVERY workgroup (blocks 0..63 of the launch): buf[128 + 2 * block] = begin tick, + 1 = end tick
#define VODHIP_PROBE_WG(buf, end)                                                                                   \
    do {                                                                                                            \
        if (blockIdx.x < 64 && threadIdx.x == 0) (buf)[128 + 2 * blockIdx.x + (end)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define VODHIP_PROBE_WG(buf, end) \
    do {                          \
    } while (0)
#define VODHIP_PROBE(buf, i) \
    do {                     \
    } while (0)
#endif

namespace vodhip {

void set_last_error(const char* msg);  // the calling thread's vodhip_last_error() text (vodhip_node.hip composes public entry points)

// which: 0 = merge_hybrid_kernel, 1 = priority_sample_kernel, 2 = flatten_inbatch_kernel; out: 256 words = 64 (cycles, 10 ns ticks) phase pairs + 64 (begin, end) workgroup pairs
hipError_t read_probe_hybrid(long long* out);
hipError_t read_probe_sample(int which, long long* out);
hipError_t read_probe_select(long long* out);  // which = 3: mips_select_kernel (final | threshold-only | middle launch at [0], [8], [16])

// 64-bit order-preserving result key: high 32 bits = monotone image of the fp32 score, low 32 bits =
// 0xFFFFFFFF - local row id.  Larger key == better hit (higher score; on equal score the smaller id).
// Key 0 is reserved for "no entry" and sorts below every real key.
typedef unsigned long long key_t64;

// Per-query candidate counters sit one per 128-byte line: every survivor is one returning atomic on its query's counter, and
// atomics on one line execute one after the other at the memory side (~3 ns each) - with the counters packed (8-32 lines for
// 256-1024 queries) the ~800 k atomics of a stage took 70 us whatever the stage's size (measured: a 166 k-row stage at
// nq = 256 ran 143 us with survivors, 74 us without).
constexpr int CNT_STRIDE = 32;  // in unsigned ints

enum : int { FILTER_FLAG_CORPUS_NT = 1 };  // FilterExtra::flags bit 0; bits 8.. = timing knobs of diagnostic builds

// Extra, optional inputs of the filter kernels (passed by value).
struct FilterExtra {
    int flags = 0;                    // FILTER_FLAG_* | (diagnostic knobs << 8)
    const int* row_label = nullptr;   // [ntotal] subset label of every stored row, or NULL (no subset filtering)
    const int* q_label = nullptr;     // [nq, n_qlab] allowed labels per query (-1 = empty slot; all -1 = unrestricted)
    int n_qlab = 0;
    int sample_rstride = 0;           // GMAX launches: store rows between consecutive sampled rows (>= 1)
    int sample_offset = 0;            // GMAX launches: first sampled row (centres the sample: the unsampled rows split between head and tail)
    int sample_groups = 0;            // GMAX launches: number of lane groups of the sample (= candidate slots per query)
    const int* q_map = nullptr;       // recovery of a few queries: workspace row -> row of the caller's batch (q_label is indexed by the latter)
    // FILTER launches: the order in which a search's stages walk the store's 256-row super-tiles.  perm_mod = T > 0: position p of the
    // sequence is super-tile (p * perm_mul) % T (perm_mul ~ 0.618 T, coprime to T: a low-discrepancy order - ANY run of consecutive
    // positions, i.e. any stage, is spread evenly over the whole store, so its threshold is calibrated on every topic of a
    // document-ordered corpus and a hot topic's tiles are not all scanned at once); 0 = positions are tiles.  row_bound = ntotal (the
    // partly filled last super-tile can sit at any position: every launch masks rows >= row_bound).
    int perm_mul = 0, perm_mod = 0, row_bound = 0;
};

struct SearchWorkspace {
    uint16_t* q_pad = nullptr;       // [nq_pad][dim_pad] queries rounded to the store dtype, zero padded
    key_t64* topk = nullptr;         // [nq_pad][kp] running top-k keys, sorted descending
    key_t64* cand = nullptr;         // [nq_pad][cap] candidate keys appended by the filter kernel
    unsigned int* cnt = nullptr;     // [nq_pad * CNT_STRIDE] candidates appended in the current stage (one counter per 128-B line)
    float* thr_s = nullptr;          // [nq_pad] score of the current k-th best (-inf until k hits exist)
    key_t64* thr_key = nullptr;      // [nq_pad] key of the current k-th best (0 until k hits exist)
    unsigned int* overflow = nullptr;  // [4] flag words: [0] a candidate buffer overflowed; exact mode: [1] some list did not prove complete, [2] max needed list length
    unsigned int* ovf_q = nullptr;     // [nq] per query: its candidate list overflowed in some stage (set by the select kernel), or NULL
    int n_cu = 256;                    // compute units of the device (read once at index create)
    int64_t nq_cap = 0;
    int64_t cap = 0;
    int64_t kp = 0;       // row stride of `topk` for the search being enqueued (a power of two >= its k)
    int64_t kp_cap = 0;   // ... and the stride the buffer was allocated for (>= kp)
    FilterExtra extra;  // flags + optional subset filter of the current search
};

// ---- launchers (kernels_mips.hip) -------------------------------------------------------------
// store_dtype: 0 = f16, 1 = bf16.  tile: filter-kernel variant (kernels_mips.hip: 1, 42, 46 generic; 8, 9 persistent).
// mode: 0 = FILTER (threshold survivors -> candidate lists), 1 = DENSE (every score of rows [row_begin, row_end) is a
// candidate), 2 = GMAX (threshold bootstrap: group maxima of `n_sample_tiles` sampled tiles, see FilterExtra::sample_rstride).
hipError_t launch_convert_rows(const void* src, int src_dtype, int64_t n_rows, int64_t dim, void* dst, int dst_dtype,
                               int64_t dst_stride, hipStream_t stream);
// seed_scores / seed_ids ([nq, k] device, or NULL): a previous valid-but-incomplete result whose k-th score seeds the thresholds
// q_map ([nq] device, or NULL): workspace row r is row q_map[r] of q_src / of the seed arrays
// seed_margin ([rows of the caller's batch] device, or NULL): subtracted from the seeding score (the BAND pass of exact mode seeds
// its scan thresholds with `k-th exact score - eps`)
hipError_t launch_search_prepare(const SearchWorkspace& ws, const void* q_src, int q_dtype, int64_t nq, int64_t dim,
                                 int store_dtype, int64_t nq_pad, int64_t dim_pad, bool clear_overflow,
                                 const float* seed_scores, const int64_t* seed_ids, int k, const int* q_map, hipStream_t stream,
                                 const float* seed_margin = nullptr);
hipError_t launch_filter(int store_dtype, int tile, int mode, const void* store, const void* q_pad, int64_t dim_pad,
                         int64_t row_begin, int64_t row_end, int64_t n_sample_tiles, int64_t nq, int64_t nq_pad,
                         const SearchWorkspace& ws, hipStream_t stream);
// flags: 1 = final (sort; the top-k also leaves as float32 scores / int64 ids (+ id_base) in out_scores / out_ids [nq, k]),
//        2 = threshold only (the candidates are GMAX group maxima: nothing enters the running top-k)
//        q_map: result row r is written to row q_map[r] of out_scores / out_ids
hipError_t launch_select(const SearchWorkspace& ws, int64_t nq, int k, int64_t dense_n, int flags, hipStream_t stream,
                         int64_t id_base = 0, float* out_scores = nullptr, int64_t* out_ids = nullptr, const int* q_map = nullptr);
hipError_t launch_output(const SearchWorkspace& ws, int64_t nq, int k, int64_t id_base, float* out_scores,
                         int64_t* out_ids, hipStream_t stream);
hipError_t launch_merge_topk(const float* scores, const int64_t* ids, int64_t stride_s, int64_t stride_i, int n_shards,
                             int64_t nq, int k, int k_out, float* out_scores, int64_t* out_ids, hipStream_t stream);
// the FILTER stage on the deeper LDS ring (kernels_mips_ring.hip; tiles 10 = plain, 11 = fragments read one k-step ahead)
hipError_t launch_filter_ring(int store_dtype, bool pipe, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin,
                              int64_t row_end, int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream);
// the FILTER stage on the 384 x 256 workgroup tile (kernels_mips_wide.hip; tile 12)
hipError_t launch_filter_wide(int store_dtype, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end,
                              int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel): it is a driver call on the launch path
hipError_t allow_dynamic_lds(const void* kernel, int bytes);
// the FILTER stage on the guide's 8-phase K loop (experiments/csrc/kernels_mips_8phase.hip; tiles 13 / 14)
hipError_t launch_filter_8phase(int store_dtype, int variant /* 14 = production; 13, 15, 16: experiment builds */, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end,
                                int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream);
// the FILTER stage with the query tile resident in registers (experiments/csrc/kernels_mips_qres.hip; tile 17; dim_pad 384 / 768 only)
bool filter_qres_supports(int64_t dim_pad);
hipError_t launch_filter_qres(int store_dtype, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end,
                              int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream);
// ... and with a K-split wave pair, two waves per SIMD (experiments/csrc/kernels_mips_ksplit.hip; tile 18; dim_pad 768 only)
bool filter_ksplit_supports(int64_t dim_pad);
hipError_t launch_filter_ksplit(int store_dtype, const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin, int64_t row_end,
                                int64_t nq, int64_t nq_pad, const SearchWorkspace& ws, hipStream_t stream);
inline bool filter_tile_is_persistent(int tile) { return tile >= 8 && tile <= 18; }  // one 256 x 256 workgroup per CU walking tiles
int filter_tile_rows(int tile);  // BM of the tile config
int filter_tile_cols(int tile);  // BN of the tile config
int filter_group_rows(int tile); // rows per GMAX group (one lane's rows of one column block)

// ---- launchers (kernels_exact.hip): the float32 plane of VODHIP_EXACT_F32 stores ----------------
enum : int { EXACT_LIST = 0, EXACT_CAND = 1, EXACT_FIRST = 2 };
struct ExactArgs {
    const float* plane = nullptr;     // [rows][stride] float32 rows (zero padded columns)
    int64_t stride = 0;
    int dim = 0, dim_pad = 0;
    const void* q_src = nullptr;      // the caller's queries [batch rows, dim] of q_dtype (unrounded)
    int q_dtype = 0;
    const int* q_map = nullptr;       // workspace row -> row of the caller's batch, or NULL
    int store_dtype = 0;              // what the scan rounded to (the bound needs |q - q~|)
    // the bound's statistics, host copies (refreshed before the first search after rows were added: launch_exact_stats)
    float ord_n2 = 0.f, ord_d2 = 0.f;     // max |x|^2, max |x - x~|^2 over the ORDINARY rows
    float cut_n2 = 0.f, cut_d2 = 0.f;     // a row with |x|^2 > cut_n2 or |x - x~|^2 > cut_d2 is an OUTLIER: scored by every query, not bounded
    int n_out = 0;                        // outliers (<= 2 * EXACT_MAX_OUTLIERS)
    const unsigned int* out_rows = nullptr;  // [n_out] their local rows (device)
    const float* row_n2 = nullptr;        // [rows] |x|^2 of every stored row (device)
    const float* row_d2 = nullptr;        // [rows] |x - x~|^2
    const int* row_label = nullptr;       // the subset filter in force (outliers are injected past the scan: they are filtered here), or NULL
    const int* q_label = nullptr;         // [batch rows, n_qlab]
    int n_qlab = 0;
    int mode = EXACT_LIST;            // EXACT_LIST | EXACT_CAND [| EXACT_FIRST]
    // LIST: the scan's top-kx list, rows of the caller's batch, LOCAL ids (pads -1)
    const float* list_s = nullptr;
    const int64_t* list_i = nullptr;
    int kx = 0;
    // CAND: a stage's candidate list (workspace rows)
    const key_t64* cand = nullptr;
    unsigned int* cnt = nullptr;
    int cap = 0, dense_n = -1;
    float* thr_s = nullptr;
    key_t64* thr_key = nullptr;
    int kr = 0;                       // slots of the running top-k in the key buffer (power of two >= k)
    int P = 0;                        // key-buffer slots (power of two; LIST: >= kx, CAND: >= 2 * kr)
    // results: rows of the caller's batch
    int k = 0;
    int64_t id_base = 0;
    float* out_scores = nullptr;
    int64_t* out_ids = nullptr;
    float* eps = nullptr;             // [batch rows] the per-query bound (written by LIST)
    unsigned int* flag_word = nullptr;  // LIST: some query is incomplete; CAND: some list lost candidates
    unsigned int* flag_q = nullptr;     // LIST: [batch rows] 0 / 1 per query; CAND: [workspace rows] or NULL
};
// words of an exact-mode store's statistics buffer (device)
constexpr int EXACT_MAX_OUTLIERS = 32;  // per statistic
enum : int { EXS_MAX_N2 = 0, EXS_MAX_D2 = 1, EXS_ORD_N2 = 2, EXS_ORD_D2 = 3, EXS_N_OUT = 4, EXS_CUT_N2 = 5, EXS_CUT_D2 = 6,
             EXS_HIST_N2 = 8, EXS_HIST_D2 = 24, EXS_OUT_ROWS = 64, EXS_WORDS = 64 + 2 * EXACT_MAX_OUTLIERS };
// row_n2 / row_d2: [n_rows] planes of the ingested rows' |x|^2 and |x - x~|^2 (already offset to the first ingested row)
hipError_t launch_ingest_exact(const void* src, int src_dtype, int64_t n_rows, int64_t dim, void* dst16, int dst_dtype, float* dst32,
                               int64_t stride, unsigned int* stats, float* row_n2, float* row_d2, hipStream_t stream);
// re-derive words [EXS_ORD_N2 ..) from the per-row planes of rows [0, n_rows) and the global maxima in words [0], [1]
hipError_t launch_exact_stats(const float* row_n2, const float* row_d2, int64_t n_rows, unsigned int* stats, hipStream_t stream);
hipError_t launch_exact_rescore(const ExactArgs& a, int64_t nq, hipStream_t stream);

// ---- launchers (kernels_hybrid.hip) -----------------------------------------------------------
struct HybridArgs {
    const int64_t* lookup_idx;
    const int64_t* lookup_lbl;
    int k_lookup;
    int n_engines;
    const int64_t* engine_idx[4];
    const float* engine_scr[4];
    int engine_k[4];
    float engine_w[4];
    int64_t nq;
    int64_t* out_idx;
    float* out_scr;
    int64_t* out_lbl;
    float* out_raw[4];
    int out_stride;
    int32_t* out_width;        // [4] maximum over rows of the cursor after each engine (atomics; cleared by the launcher), or NULL
    int32_t* out_row_cursor;   // [nq, 4] the same cursors per row (plain stores), or NULL
};
hipError_t launch_merge_hybrid(const HybridArgs& a, hipStream_t stream);

// ---- launchers (kernels_retrieval.hip) --------------------------------------------------------
// auxiliary losses of the retrieval objective (passed by value; `enabled` = 0 leaves the plain loss and an 8-float row stride)
struct RetrievalAux {
    int enabled = 0;
    int guidance_type = 0;   // 0 = "zero", 1 = "sparse"
    float w_guidance = 0.f, w_self = 0.f, w_decay = 0.f;
    float* grad = nullptr;   // [3][B*D] scratch: unnormalised gradients of the three terms w.r.t. the scores
    float* out = nullptr;    // [3] loss values: guidance, self-supervision, score decay (NaN where the weight is 0)
};
hipError_t launch_retrieval_forward(const void* q, const void* s, int enc_dtype, int sections_3d, int64_t B, int64_t D,
                                    int64_t H, const float* score, const int64_t* relevance, const float* sparse,
                                    const float* dense, float* retriever_scores, float* d_scores, float* loss,
                                    float* kl, float* workspace, const RetrievalAux& aux, hipStream_t stream,
                                    int64_t workspace_floats = 0);
hipError_t launch_retrieval_backward(const void* q, const void* s, int enc_dtype, int sections_3d, int64_t B, int64_t D,
                                     int64_t H, const float* d_scores, const float* grad_out, float* dq, float* ds,
                                     hipStream_t stream);

// ---- launchers (kernels_sample.hip) -----------------------------------------------------------
hipError_t launch_priority_sample(const float* scores, const uint8_t* labels, const float* noise, int64_t nq, int width,
                                  int k_positive, int k_total, float temperature, int max_support_size, int normalized,
                                  int64_t* out_samples, float* out_log_weights, uint8_t* out_labels, float* out_lse,
                                  hipStream_t stream);

// sampling straight from the merge's full-stride outputs + the gather epilogue (device-resident collate)
struct SampleMergedArgs {
    const int64_t* ids;
    const float* scores;
    const int64_t* labels;   // > 0 = positive
    const float* noise;
    int64_t nq, stride, noise_stride;
    int width;               // >= 0: columns in use; < 0: derived on the device from merge_width / k_lookup / engine_k
    const int* merge_width;  // [n_engines] maxima over rows, or NULL
    const int* row_cursor;   // [nq, 4] per-row cursors (the kernel takes the maximum), or NULL
    int k_lookup, n_engines, engine_k[4];
    int k_positive, k_total;
    float temperature;
    int max_support, normalized;
    int n_raw;
    const float* raw[4];
    int64_t* out_samples;
    int64_t* out_ids;
    float* out_scores;
    float* out_logw;
    uint8_t* out_labels;
    float* out_raw[4];
    float* out_lse;          // lse of class c of row r at out_lse[r * lse_row_stride + c * lse_cls_stride]
    int64_t lse_row_stride, lse_cls_stride;
    float* out_max_sampling_id;
};
hipError_t launch_priority_sample_merged(const SampleMergedArgs& m, hipStream_t stream);
hipError_t launch_flatten_inbatch(const int64_t* ids, int64_t n_rows, int n_keys, int n_values, const float* const* values,
                                  const float* fill, float* const* outs, const uint8_t* labels, uint8_t* out_labels,
                                  int64_t* out_unique, int* out_n_unique, hipStream_t stream);

hipError_t launch_gather_by_id(const int64_t* queries, int64_t n_queries, const int64_t* keys, int64_t n_rows, int n_keys,
                               int n_values, const float* const* values, const float* fill, float* const* outs,
                               hipStream_t stream);

}  // namespace vodhip

/*
 * vodhip.h -- C-ABI of the MI355X-native dense-retrieval scoring library (libvodhip.so).
 *
 * This is the drop-in boundary for the ONE hot path of VodLM/vod that this project replaces:
 * the corpus vector store + batched query x section inner-product top-k that the reference
 * reaches through a faiss server process, plus the hybrid score merge and the in-batch
 * retrieval scoring glued to it.  Plain pointers and sizes only; no torch / C++ types.
 * Reference citations are relative to /root/reference/.
 *
 * Conventions
 *   - every function returns 0 on success, < 0 on error; `vodhip_last_error()` returns a
 *     thread-local, NUL-terminated description of the last failure on the calling thread;
 *   - no exception crosses the boundary; the caller owns every buffer it passes in;
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream).  Kernels are
 *     enqueued on it; functions documented "async" return before the GPU work is done;
 *   - device pointers must belong to the device the handle was created on;
 *   - result layout everywhere: scores float32 [nq,k] sorted descending, ids int64 [nq,k],
 *     ties -> smaller id first, missing entries padded with score -inf / id -1
 *     (the repo-wide padding convention, src/vod_types/retrieval.py:284-285).
 */
#ifndef VODHIP_H
#define VODHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VODHIP_VERSION 1

/* element types of vectors handed to / stored by the library */
enum { VODHIP_F16 = 0, VODHIP_BF16 = 1, VODHIP_F32 = 2 };
/* OR-ed into the store dtype at create: the store ALSO keeps the float32 rows and every search returns what a float32 brute force
 * over the unrounded rows and queries returns (H2 below) - the reference's own arithmetic: faiss IndexFlat holds float32
 * (`index.add(vectors.astype(float32))`, src/vod_search/faiss_search/build.py:65-73; float32 queries, faiss_search/server.py:71,81).
 * Costs 4 more bytes per element of HBM and ~1-3 % of search time. */
#define VODHIP_EXACT_F32 0x100
/* where a caller buffer lives */
enum { VODHIP_HOST = 0, VODHIP_DEVICE = 1 };

typedef struct vodhip_index vodhip_index_t;

const char* vodhip_last_error(void);
int vodhip_version(void);

/* ---------------------------------------------------------------------------------------------
 * H1  corpus vector store.
 * Replaces: faiss.index_factory(D, "Flat", METRIC_INNER_PRODUCT) + index.add(float32 batch)
 *           (src/vod_search/faiss_search/build.py:51-81) and the faiss.write_index/read_index
 *           round trip (src/vod_search/factory.py:167, src/vod_search/faiss_search/server.py:42).
 * The store is a row-major [capacity, dim_padded] fp16/bf16 matrix in HBM on `device`
 * (dim padded with zeros to a multiple of 64).  Rows get ids 0,1,2,... in insertion order.
 * ------------------------------------------------------------------------------------------- */
int vodhip_index_create(int device, int64_t dim, int store_dtype /* VODHIP_F16 | VODHIP_BF16, optionally | VODHIP_EXACT_F32 */,
                        int64_t capacity_rows, vodhip_index_t** out);
int vodhip_index_destroy(vodhip_index_t* index);

/* Append `n_rows` row-major [n_rows, dim] vectors of `src_dtype` living in host or device memory.
 * Values are rounded to the store dtype (round-to-nearest-even).  Synchronous for host sources
 * (the call returns when the rows are resident); async on `stream` for device sources.
 * Host sources are pipelined in 64 MB slices: a pinned / registered source is read in place by the DMA engine; a pageable one
 * (NumPy array, .npy memory map) is copied into two pinned staging slots by CPU threads (param "ingest_threads", 0 = auto)
 * while the previous slice is in flight.  Pass the WHOLE array in one call: slicing it on the caller's side serialises the stages. */
int vodhip_index_add(vodhip_index_t* index, const void* rows, int64_t n_rows, int src_dtype,
                     int src_location, void* stream);
int vodhip_index_reset(vodhip_index_t* index);               /* ntotal := 0 (capacity kept) */
int vodhip_index_ntotal(const vodhip_index_t* index, int64_t* out);
int vodhip_index_dim(const vodhip_index_t* index, int64_t* out);
int vodhip_index_capacity(const vodhip_index_t* index, int64_t* out);
/* raw view of the store (device pointer, row stride in elements) for persistence / tests */
int vodhip_index_data(const vodhip_index_t* index, void** dev_ptr, int64_t* row_stride_elems, int* store_dtype);
/* copy rows [row_begin, row_begin + n_rows) of the store, as stored (fp16/bf16, unpadded [n_rows, dim]),
 * to `dst` in host or device memory; async on `stream` for device destinations. */
int vodhip_index_get_rows(const vodhip_index_t* index, int64_t row_begin, int64_t n_rows, void* dst,
                          int dst_location, void* stream);
/* VODHIP_EXACT_F32 stores only: the float32 plane (what faiss.write_index would persist, factory.py:167) - raw view, and rows
 * [row_begin, row_begin + n_rows) unpadded [n_rows, dim] float32 into host or device memory.  -1 for a store without the plane. */
int vodhip_index_data_f32(const vodhip_index_t* index, void** dev_ptr, int64_t* row_stride_elems);
int vodhip_index_get_rows_f32(const vodhip_index_t* index, int64_t row_begin, int64_t n_rows, void* dst, int dst_location, void* stream);

/* ---------------------------------------------------------------------------------------------
 * H2  exact brute-force inner-product top-k over the store.
 * Replaces: faiss_index.search(query_vec, k)
 *           (src/vod_search/faiss_search/server.py:72 and :84).
 * `queries` is a DEVICE pointer to row-major [nq, dim] values of `q_dtype`; they are rounded to the
 * store dtype.  Scores are fp32-accumulated dot products of the stored (rounded) values.
 * `id_base` is added to every valid id (row offset of this shard inside a sharded corpus,
 * the reference's `indices += offset` at src/vod_search/sharded_search.py:103,155 -- pads stay -1).
 * out_scores / out_ids are DEVICE pointers [nq, k].  1 <= k <= VODHIP_MAX_K.
 *
 * VODHIP_EXACT_F32 stores: the fp16 / bf16 scan is only the FILTER.  Scores are float32 dot products of the UNROUNDED query and row
 * (one fixed summation order: the same bits from every path, batch and shard), the k best by (score desc, id asc) over ALL rows -
 * guaranteed, not statistical: the scan's top-k' list (k' > k, param "exact_expand") is re-scored from the float32 plane, and the
 * device checks per query that no row outside the list can reach the k-th re-scored score, using the bound
 * |s - s~| <= |q - q~| max|x| + |q~| max|x - x~| (+ summation slack) with the norms of the actual data; a query that fails the check
 * is searched again with every row inside that band as a candidate ("Exact-f32 mode").  The queries must stay valid until finish, as
 * always; q_dtype F32 is the point of the mode, F16 / BF16 queries are taken as exact values.
 *
 * vodhip_index_search        enqueue + wait + exactness check (if a query's candidate list overflowed, that query is
 *                            searched again with thresholds seeded from its incomplete result; the exhaustive
 *                            schedule is the last resort); results are final on return.
 * vodhip_index_search_async  enqueue only.  Must be followed by vodhip_index_search_finish on the
 *                            same index before the outputs are trusted.  Up to 4 searches may be in flight
 *                            (same stream; they share the workspace in stream order); finish completes the
 *                            OLDEST one and waits for it alone, so a caller that runs one search ahead never
 *                            leaves the device idle while the host checks the exactness flag.  The query and
 *                            output buffers of a search must stay valid and unmodified until its finish.
 * ------------------------------------------------------------------------------------------- */
#define VODHIP_MAX_K 2048
int vodhip_index_search(vodhip_index_t* index, const void* queries, int q_dtype, int64_t nq, int k,
                        int64_t id_base, float* out_scores, int64_t* out_ids, void* stream);
int vodhip_index_search_async(vodhip_index_t* index, const void* queries, int q_dtype, int64_t nq, int k,
                              int64_t id_base, float* out_scores, int64_t* out_ids, void* stream);
int vodhip_index_search_finish(vodhip_index_t* index, void* stream);

/* Optional subset filter.  The reference's SearchClient.search carries `subset_ids`; its Elasticsearch and
 * Qdrant engines restrict hits to sections whose subset id is listed (src/vod_search/es_search/client.py:185-191,
 * qdrant_search/client.py:124-136) while its faiss client ignores it (faiss_search/client.py:67-72).  Here every stored
 * row may carry an int32 label (`labels` [n_rows], host or device; NULL clears), and the next searches may pass, per
 * query, up to `n_per_query` allowed labels (DEVICE int32 [nq, n_per_query], caller-owned until cleared with NULL;
 * -1 = empty slot; a query whose slots are all -1 is unrestricted; any other value, e.g. -2 for an unknown id,
 * restricts the query).  A row is eligible iff its label is listed.
 * Results stay exact top-k over the eligible rows. */
int vodhip_index_set_row_labels(vodhip_index_t* index, const int32_t* labels, int64_t n_rows, int location, void* stream);
int vodhip_index_set_query_labels(vodhip_index_t* index, const int32_t* q_labels_dev, int n_per_query);

/* Tunables / introspection (tests and bench).
 * params: "cand_cap" (candidate slots per query and stage, default 16384), "dense_rows" (indexes up to this many rows are
 *   scored densely, default 2048), "sample_div" (the threshold bootstrap scores ~ntotal / sample_div sampled rows; 0 = auto, the default: 96, and 192
 *   for batches above 512 queries on stores of 4 M rows and more), "growth" (x100: a filter stage covers growth x the rows its threshold
 *   was calibrated on; 0 = auto, the default: 800, and 300 where sample_div is 192),
 *   "force_safe" (1 = exhaustive schedule: dense chunks of <= cand_cap rows), "tile" (0 = auto; 1 = 128x128, 42 / 46 = small-batch
 *   rings, 8 / 9 = persistent 256x256 two-slot kernel without / with the wave stagger, 14 = persistent 256x256 on the 8-phase K loop: the
 *   auto choice above 128 queries), "small_chunk_tiles", "profile" (1 = HIP events around
 *   every filter launch), "ingest_threads" (CPU threads staging pageable host rows, 0 = auto), "kflags" (timing knobs of
 *   diagnostic builds), "tile_order" (0 = default: the FILTER stages of a search walk the store's 256-row tiles in a low-discrepancy
 *   order, so that every stage samples the whole store whatever order the rows were added in; 1 = in row order; results are identical),
 *   "lanes" (0 = auto, 1, 2: with 2 lanes consecutive searches alternate between two workspaces, the second on a stream of the index's
 *   own that starts behind the work the caller's stream holds at enqueue - two searches of a caller that runs one search ahead overlap
 *   on the device; auto = 2 for batches of up to 256 queries, where a search's select / prepare launches and partly filled last
 *   rounds leave ~12 % of the device idle; results are unchanged.  STREAM ORDERING with 2 lanes: every second search in flight runs on
 *   the index's stream, so work the caller enqueues on ITS stream behind vodhip_index_search_async is NOT ordered behind that search:
 *   only vodhip_index_search_finish completes a search - a hipStreamSynchronize of the caller's stream does not; set "lanes" = 1 to keep
 *   every search on the caller's stream.  add / reset / set_row_labels refuse while searches are in flight),
 *   "exact_expand" (x100, VODHIP_EXACT_F32 stores: the scan lists k' = k * exact_expand / 100 + 16 rows per
 *   query; 0 = default: 110 for an fp16 store, 200 for bf16 as the upper limit, with k' following what the last searches of the same k
 *   needed ("exact_adapt" = 1, default; 0 = always the formula); speed only - results are exact for any value).
 * VODHIP_EXACT_F32 input range: finite float32 rows and queries of ANY magnitude whose squared norm is finite in float32 (|x| < 1.8e19)
 *   give the float32 brute-force result; values beyond the scan dtype's range (fp16: |v| > 65504) saturate in the scan copy only, and
 *   the up to 64 rows whose norm / rounding error exceeds the rest of the store's by 2x or more ("outliers") are scored exactly by
 *   every query instead of widening the error bound.  Rows with NaN / inf components never enter a result (NaN scores are dropped).
 * stats (of the search completed by the last vodhip_index_search_finish): "last_overflow" (a candidate list overflowed),
 *   "last_safe_reruns" (recovery passes run), "last_recovered_queries" (queries the first recovery pass re-searched),
 *   "last_chunks" (stages), "last_filter_launches", "last_filter_ns" (with "profile"), "last_recovery_launches",
 *   "last_recovery_ns" (the filter launches of the recovery passes, accounted separately); "exact" (1 for a VODHIP_EXACT_F32 store),
 *   "last_exact_kx" (k' of the last search), "last_exact_band_queries" (queries whose list did not prove complete and ran a band
 *   pass), "last_exact_band_passes". */
int vodhip_index_set_param(vodhip_index_t* index, const char* key, int64_t value);
/* Host-side planning only (no device and no HIP runtime call): the stage list a search of `nq` queries for the top `k` of
 * `ntotal` rows would run on a device with `n_cu` compute units (<= 0 = 256, the MI355X), with the given tunables (<= 0 =
 * library default; tile 0 = auto; recovery_pass 0 = the normal schedule).  An index reads its device's CU count once, at create.
 * out: int64 [max_stages][6] = {kind (0 FILTER, 1 DENSE, 2 GMAX bootstrap), row_begin, row_end, sampled tiles, sample row
 * stride, sample groups}.  Returns the number of stages, or -1. */
int vodhip_debug_schedule(int64_t ntotal, int k, int64_t nq, int64_t cand_cap, int64_t dense_rows, int64_t sample_div,
                          int64_t growth_x100, int tile, int recovery_pass, int n_cu, int64_t* out, int max_stages);
int vodhip_index_get_stat(const vodhip_index_t* index, const char* key, int64_t* out);
/* Host-side planning only: the order in which a search's FILTER stages walk the 256-row tiles of a store of `ntotal` rows - position p
 * (stages are runs of consecutive positions) is tile (p * perm_mul) mod perm_mod; perm_mul <= 1 = positions are tiles (small stores). */
int vodhip_debug_tile_order(int64_t ntotal, int64_t* perm_mul, int64_t* perm_mod);
/* Diagnostic builds only (make ABLATION=1; production returns -1): phase stamps of workgroup 0 of the last launch of a
 * kernel (which: 0 = hybrid merge, 1 = priority sampling, 2 = in-batch flattening, 3 = the search's select kernel) as 64 pairs
 * (shader-clock cycles, 10 ns ticks of the constant 100 MHz counter), then the (begin, end) ticks of workgroups 0..63;
 * out holds n >= 256 values. */
int vodhip_debug_read_probe(int which, int64_t* out, int n);

/* ---------------------------------------------------------------------------------------------
 * H3  merge of per-shard top-k lists (multi-GPU exchange step, after the RCCL all-gather).
 * Replaces: the row-offset + stack of ShardedSearchClient (src/vod_search/sharded_search.py:92-106,
 *           198-203) and faiss IndexShards' host merge (src/vod_search/faiss_search/server.py:51-54).
 * scores/ids: DEVICE [n_shards, nq, k] (ids already global, pads -1).  Output [nq, k_out].
 * One launch merges up to 8192 entries per query; more (e.g. 8 shards x top-2048) are merged in levels through a
 * stream-ordered temporary (hipMallocAsync on `stream`); k <= 4096.
 * ------------------------------------------------------------------------------------------- */
int vodhip_merge_topk(const float* scores, const int64_t* ids, int n_shards, int64_t nq, int k,
                      int k_out, float* out_scores, int64_t* out_ids, void* stream);
/* Same, with explicit element strides between shards: lets ONE all-gather move a packed per-rank record
 * [scores f32 nq*k | ids i64 nq*k] and the merge read both parts in place. */
int vodhip_merge_topk_strided(const float* scores, int64_t shard_stride_scores, const int64_t* ids,
                              int64_t shard_stride_ids, int n_shards, int64_t nq, int k, int k_out,
                              float* out_scores, int64_t* out_ids, void* stream);

/* ---------------------------------------------------------------------------------------------
 * H1 + H2 + H3 on several devices of one node, from ONE process: for consumers without torch.distributed (C, C++, cgo, JNI).
 * Replaces: faiss.index_cpu_to_all_gpus(index, co) with co.shard = True, i.e. faiss IndexShards
 *           (src/vod_search/faiss_search/server.py:51-54, src/vod_configs/search.py:80).
 * Shard g (on devices[g]; a device may be listed more than once) holds the global rows [g * R, (g + 1) * R), R = ceil(capacity /
 * n_devices): rows keep their insertion ids, the shards are balanced when the store is full.  A search replicates the query batch,
 * runs on every device at once, copies the per-shard top-k lists to devices[0] (peer copies) and merges them there with
 * vodhip_merge_topk: results are identical to one index holding all the rows.  (The Python host uses one process per GPU and one
 * RCCL all-gather instead - vod_amd/distributed.py - because that is how torch.distributed programs are launched.)
 *   add     HOST rows only (row-major [n_rows, dim] of src_dtype); the shards a batch straddles ingest concurrently; synchronous.
 *   search  location = VODHIP_HOST: queries / outputs are host buffers, the call returns with the results in place (`stream` unused);
 *           location = VODHIP_DEVICE: they live on devices[0]; the queries must be complete on `stream` and the outputs are complete
 *           on `stream` when the call returns (the exactness check of every shard has already been done on the host).
 *   shard   the per-device handle (for params, stats, subset labels, persistence), its id offset and its device.
 * Calls on one handle are serialised by the library (a mutex per handle); a search that fails half-way leaves no shard with a search
 * in flight.  `set_query_labels` + `search` are two calls: callers that filter from several threads go through a vodhip_batcher.
 * ------------------------------------------------------------------------------------------- */
typedef struct vodhip_node_index vodhip_node_index_t;
int vodhip_node_index_create(int n_devices, const int* devices, int64_t dim, int store_dtype, int64_t capacity_rows,
                             vodhip_node_index_t** out);
int vodhip_node_index_destroy(vodhip_node_index_t* index);
int vodhip_node_index_add(vodhip_node_index_t* index, const void* rows, int64_t n_rows, int src_dtype);
int vodhip_node_index_reset(vodhip_node_index_t* index);
int vodhip_node_index_ntotal(const vodhip_node_index_t* index, int64_t* out);
int vodhip_node_index_n_shards(const vodhip_node_index_t* index);
int vodhip_node_index_shard(vodhip_node_index_t* index, int g, vodhip_index_t** shard, int64_t* id_base, int* device);
int vodhip_node_index_set_param(vodhip_node_index_t* index, const char* key, int64_t value);  /* on every shard; plus "host_staging" (below) */
/* Topology, read once at create (hipDeviceCanAccessPeer both ways; peer access is enabled where possible): out[g] = 2 when shard g lives
 * on devices[0] itself, 1 when it exchanges with devices[0] by direct peer copies (xGMI), 0 when it goes through pinned host memory
 * (no peer access, or param "host_staging" = 1: every shard but the first takes that route - bring-up / tests on a 1-GPU box).
 * Returns the number of shards. */
int vodhip_node_index_peer_access(const vodhip_node_index_t* index, int* out, int n);
/* The subset filter of vodhip_index_set_row_labels / _set_query_labels behind the one handle: `labels` = HOST int32 [n_rows] for the
 * global rows 0 .. n_rows-1 (NULL clears); `q_labels` = int32 [nq, n_per_query] of the NEXT searches' batches, host memory or on
 * devices[0] (`location`), caller-owned until cleared with NULL; replicated to every device with the queries. */
int vodhip_node_index_set_row_labels(vodhip_node_index_t* index, const int32_t* labels, int64_t n_rows);
int vodhip_node_index_set_query_labels(vodhip_node_index_t* index, const int32_t* q_labels, int n_per_query, int location);
int vodhip_node_index_search(vodhip_node_index_t* index, const void* queries, int q_dtype, int64_t nq, int k, int location,
                             float* out_scores, int64_t* out_ids, void* stream);
/* The same search in two halves (round 6), so that the host enqueues batch i + 1 on every shard while batch i runs - with 8 shards the
 * enqueue alone is ~0.3 ms of host time per batch: `search_async` replicates the queries and enqueues every shard's search (nothing is
 * waited for), `search_finish` completes the OLDEST pending search (every shard's exactness check, the lists' copies to devices[0], the
 * merge; HOST outputs are complete on return, DEVICE outputs on `stream`).  Up to 2 searches may be pending; queries, outputs and subset
 * labels of a pending search must stay valid until its finish.  `vodhip_node_index_search` = async + finish. */
int vodhip_node_index_search_async(vodhip_node_index_t* index, const void* queries, int q_dtype, int64_t nq, int k, int location,
                                   float* out_scores, int64_t* out_ids, void* stream);
int vodhip_node_index_search_finish(vodhip_node_index_t* index);
/* With param "profile" = 1 (also set on every shard: their filter launches are bracketed, read per shard through vodhip_index_get_stat on
 * the handle vodhip_node_index_shard returns): "last_merge_ns" = the merge launch on devices[0], from the moment every shard's list had
 * arrived; "last_copy_ns_max" = the slowest shard's copy of its list towards devices[0] (peer copy over xGMI, or the down-leg to pinned
 * host memory when staged).  HIP events of the last search; 0 with one shard.  Waits for that search. */
int vodhip_node_index_get_stat(vodhip_node_index_t* index, const char* key, int64_t* out);

/* ---------------------------------------------------------------------------------------------
 * H4  hybrid score merge (lookup + up to VODHIP_MAX_ENGINES scored engines), one query row per wavefront.
 * Replaces: _merge_search_results (src/vod_dataloaders/core/search.py:79-125) =
 *           normalize_search_scores_ (core/normalize.py:6-20) + merge_search_results
 *           (core/merge.py:8-164) + gather_values_by_indices (core/numpy_ops.py:126-143).
 * All pointers are DEVICE pointers.  lookup_idx/lookup_lbl [nq, k_lookup] (lookup scores are discarded
 * by the reference, search.py:92).  engine e: idx[e] int64 [nq, k_e], scr[e] float32 [nq, k_e].
 * Outputs have `out_stride` = k_lookup + sum(k_e) + 1 columns allocated.  The reference cuts its buffer to
 * `[: max_cursor + 1]` after every pairwise fold (merge.py:160-162); out_width (device int32[VODHIP_MAX_ENGINES], may be
 * NULL) receives, per engine e, the maximum over rows of the cursor after engine e was folded in, from which the
 * caller derives the reference's final width: W = k_lookup; for e: W = min(out_width[e] + 1, W + k_e).
 * out_row_cursor (device int32 [nq, VODHIP_MAX_ENGINES], may be NULL) receives the same cursors per row with plain stores
 * (no cleared buffer, no atomics: one launch): vodhip_priority_sample_merged takes the maximum itself.
 *   out_idx  int64  : union of ids in first-seen order, -1 padded
 *   out_scr  float32: sum_e w_e * (s_e - rowmin_e), -inf padded
 *   out_lbl  int64  : lookup label of the id, -1 if absent (pad column: see SURVEY quirk Q3)
 *   out_raw[e] float32: min-subtracted score of the id in engine e, NaN if absent
 * ------------------------------------------------------------------------------------------- */
#define VODHIP_MAX_ENGINES 4
int vodhip_merge_hybrid(const int64_t* lookup_idx, const int64_t* lookup_lbl, int k_lookup,
                        int n_engines, const int64_t* const* engine_idx, const float* const* engine_scr,
                        const int* engine_k, const float* engine_weight, int64_t nq,
                        int64_t* out_idx, float* out_scr, int64_t* out_lbl, float* const* out_raw,
                        int out_stride, int32_t* out_width, int32_t* out_row_cursor, void* stream);

/* ---------------------------------------------------------------------------------------------
 * H5  in-batch retrieval scoring + log-prob / loss combination, forward and backward fused.
 * Replaces: RetrievalGradients.__call__ (src/vod_models/vod_gradients/retrieval.py:30-92) with
 *           _compute_retriever_scores (:186-203), _cast_data_targets (:206-215), _compute_loss
 *           (:153-177), _compute_kld (:225-243) and the autograd backward of those.
 * q [B,H], s [D,H] (sections_3d = 0) or [B,D,H] (sections_3d = 1), of `enc_dtype` (F32 | F16 | BF16).
 * score/sparse/dense float32 [B,D] (sparse/dense may be NULL), relevance int64 [B,D].
 * Outputs (DEVICE): retriever_scores float32 [B,D]; d_scores float32 [B,D] (= dLoss/dScores);
 * loss float32 [1]; kl float32 [3] (score, sparse, dense; NaN where the input is NULL).
 * vodhip_retrieval_backward turns d_scores into dq [B,H] and ds ([D,H] | [B,D,H]) in float32,
 * scaled by *grad_out (device float32 scalar).
 * ------------------------------------------------------------------------------------------- */
int vodhip_retrieval_forward(const void* q, const void* s, int enc_dtype, int sections_3d,
                             int64_t B, int64_t D, int64_t H,
                             const float* score, const int64_t* relevance, const float* sparse, const float* dense,
                             float* retriever_scores, float* d_scores, float* loss, float* kl,
                             float* workspace /* DEVICE scratch, >= 8*B floats */, void* stream);
/* The same with the reference's auxiliary losses (RetrievalGradients._auxiliary_losses, retrieval.py:94-150):
 *   guidance          (weight > 0): huber(logp - ref) over entries finite in both, ref = sparse scores (guidance_type 1)
 *                                   or zeros (guidance_type 0)                                    (:116-126,180-183)
 *   self supervision  (weight > 0): cross entropy of the positives' log-probs against their own arg-max   (:129-140)
 *   score decay       (weight > 0): mean of the squared finite scores                                     (:143-145)
 * loss = KL term + sum(weight * term); d_scores carries every term's gradient.  aux_losses float32 [3] (DEVICE) receives the
 * three terms (NaN where the weight is 0); aux_grad is DEVICE scratch of 3*B*D floats (may be NULL when all weights are 0);
 * workspace holds workspace_floats >= 16*B floats; with >= 16*B + 4*B*D (2-D sections) the in-batch contraction is split over K
 * into slabs behind the row words and uses 4x the workgroups. */
int vodhip_retrieval_forward_aux(const void* q, const void* s, int enc_dtype, int sections_3d,
                                 int64_t B, int64_t D, int64_t H,
                                 const float* score, const int64_t* relevance, const float* sparse, const float* dense,
                                 int guidance_type, float guidance_weight, float self_supervision_weight, float score_decay,
                                 float* retriever_scores, float* d_scores, float* loss, float* kl, float* aux_losses,
                                 float* aux_grad, float* workspace /* DEVICE scratch */, int64_t workspace_floats, void* stream);
int vodhip_retrieval_backward(const void* q, const void* s, int enc_dtype, int sections_3d,
                              int64_t B, int64_t D, int64_t H, const float* d_scores, const float* grad_out,
                              float* dq, float* ds, void* stream);

/* ---------------------------------------------------------------------------------------------
 * H7  labeled priority sampling of the merged candidates (the collate stage right after the merge).
 * Replaces: _labeled_priority_sampling_2d_ / _labeled_priority_sampling_1d_ / _priority_sampling_1d
 *           (src/vod_dataloaders/core/sample.py:160-219,245-352) and the numba log-softmax helpers
 *           (src/vod_dataloaders/core/numpy_ops.py:162-216).
 * DEVICE pointers.  scores float32 [nq, width]; labels uint8 [nq, width] (non-zero = positive);
 * noise float32 [nq, width] = Exp(1) draws supplied by the caller (the reference draws them with
 * np.random.exponential on the host, sample.py:398), width <= 4096.
 * Outputs: samples int64 [nq, k_total] = column index into the row (-1 padded), log_weights float32
 * [nq, k_total] (-inf padded), out_labels uint8 [nq, k_total], lse float32 [nq, 2] (positives, negatives).
 * Reference quirks kept: support truncation masks entries >= the
 * `max_support_size`-th largest - it REMOVES the best entries of each class (src/vod_dataloaders/core/sample.py:176-178); ties in
 * the priority keys are broken by the smaller column.  `normalized` is a flag word: bit 0 = log-softmax the selected weights
 * (the reference's `normalized`), bit 1 = VODHIP_SAMPLE_KEEP_TOP_SUPPORT: the corrected truncation - KEEP the `max_support_size`
 * best entries of each class, mask the rest (what the parameter's name and the shipped `support_size: 100` intend).  Off by
 * default everywhere: bit parity with the reference is the default.
 * ------------------------------------------------------------------------------------------- */
#define VODHIP_SAMPLE_KEEP_TOP_SUPPORT 2
int vodhip_priority_sample(const float* scores, const uint8_t* labels, const float* noise, int64_t nq, int width,
                           int k_positive, int k_total, float temperature, int max_support_size, int normalized,
                           int64_t* out_samples, float* out_log_weights, uint8_t* out_labels, float* out_lse,
                           void* stream);

/* The same sampling fed straight from vodhip_merge_hybrid's full-stride outputs, with the gathers and the rank diagnostic of
 * sample_search_results (src/vod_dataloaders/core/sample.py:22-84) as an epilogue: the collate's merge -> sample chain stays on
 * the device (reference flow src/vod_dataloaders/realm_collate.py:110-122).
 * DEVICE pointers.  ids int64 / scores float32 / labels int64 (> 0 = positive; the merge's lookup labels) / raw[e] float32, all
 * [nq, stride]; noise float32 rows of `noise_stride` elements.  `width` >= 0: the columns in use; `width` < 0: the reference's cut
 * (merge.py:160-162) is derived ON THE DEVICE from merge_width (the merge's out_width) or merge_row_cursor (its out_row_cursor;
 * one of the two), k_lookup and engine_k - no host sync between the two launches.  `raw` / `out_raw` are HOST arrays of n_raw (<= VODHIP_MAX_ENGINES) device pointers.
 * Outputs [nq, k_total]: out_samples (column index, -1 pad), out_ids / out_scores / out_raw[e] = take_along_axis by the sampled
 * columns (a -1 pad takes the LAST column in use, as NumPy does), out_log_weights, out_labels uint8; out_lse float32 [nq, 2];
 * out_max_sampling_id float32 [nq] = number of finite negatives of the pool scoring >= the lowest sampled finite negative. */
int vodhip_priority_sample_merged(const int64_t* ids, const float* scores, const int64_t* labels, int n_raw,
                                  const float* const* raw, const float* noise, int64_t noise_stride, int64_t nq, int stride,
                                  int width, const int32_t* merge_width, const int32_t* merge_row_cursor, int k_lookup,
                                  int n_engines, const int* engine_k,
                                  int k_positive, int k_total, float temperature, int max_support_size, int normalized,
                                  int64_t* out_samples, int64_t* out_ids, float* out_scores, float* out_log_weights,
                                  uint8_t* out_labels, float* const* out_raw, float* out_lse, float* out_max_sampling_id,
                                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gather by id: flattening of the sampled sections of a batch into ONE in-batch section set.
 * Replaces: gather_values_by_indices / gather_values_2d / _nopy_gather_values_{1d,2d}
 *           (src/vod_dataloaders/core/numpy_ops.py:24-143) as called by flatten_samples
 *           (src/vod_dataloaders/core/in_batch_negatives.py:10-52) for the scores, labels, log-weights and every
 *           raw engine score.
 * DEVICE pointers.  queries int64 [n_queries] (one id list shared by all rows: the sorted unique ids of the batch,
 * padded as the reference pads it); keys int64 [n_rows, n_keys] (n_keys <= 4096); `values` / `outs` are HOST arrays
 * of n_values (<= 8) DEVICE pointers to float32 [n_rows, n_keys] / [n_rows, n_queries]; fill[v] is written where
 * the id does not occur in the row (NaN for scores, 0 for labels in the reference).  First occurrence wins.
 * ------------------------------------------------------------------------------------------- */
int vodhip_gather_by_id(const int64_t* queries, int64_t n_queries, const int64_t* keys, int64_t n_rows, int n_keys,
                        int n_values, const float* const* values, const float* fill, float* const* outs, void* stream);

/* flatten_samples (src/vod_dataloaders/core/in_batch_negatives.py:10-52) in ONE launch: the sorted distinct ids of the whole
 * [n_rows, n_keys] batch, padded to U = n_rows * n_keys entries with the reference's constant 1 (out_unique int64 [U];
 * *out_n_unique = number of distinct ids, may be NULL), and every value array gathered onto that list (outs[v] float32
 * [n_rows, U], first occurrence wins, fill[v] where the row does not hold the id).  `labels` uint8 [n_rows, n_keys] (non-zero =
 * positive; may be NULL) is gathered the same way into out_labels uint8 [n_rows, U] with the reference's fill 0.
 * DEVICE pointers; `values` / `outs` HOST arrays of n_values (<= 8) device pointers; U <= 8192. */
int vodhip_flatten_inbatch(const int64_t* ids, int64_t n_rows, int n_keys, int n_values, const float* const* values,
                           const float* fill, float* const* outs, const uint8_t* labels, uint8_t* out_labels,
                           int64_t* out_unique, int32_t* out_n_unique, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The collate-side chain in ONE call: hybrid merge -> labeled priority sampling (+ gathers, rank diagnostic) -> optional
 * in-batch flattening = the three launches above enqueued back to back on `stream`, no host synchronisation.
 * Replaces the sequence RealmCollate.__call__ runs on the host (src/vod_dataloaders/realm_collate.py:110-139):
 *   _merge_search_results (core/search.py:79-125) -> sample_search_results (core/sample.py:22-84) -> flatten_samples
 *   (core/in_batch_negatives.py:10-52).
 * Every pointer is a DEVICE pointer owned by the caller.  stride = k_lookup + sum(engine_k) + 1 (n_engines >= 1).
 * ------------------------------------------------------------------------------------------- */
typedef struct vodhip_collate_args {
    /* the three engines' replies (core/search.py:42-62) */
    const int64_t* lookup_idx;                        /* [nq, k_lookup] */
    const int64_t* lookup_lbl;                        /* [nq, k_lookup] or NULL */
    const int64_t* engine_idx[VODHIP_MAX_ENGINES];    /* [nq, engine_k[e]] */
    const float* engine_scr[VODHIP_MAX_ENGINES];      /* [nq, engine_k[e]] */
    float engine_weight[VODHIP_MAX_ENGINES];
    int32_t engine_k[VODHIP_MAX_ENGINES];
    int32_t k_lookup, n_engines;
    int64_t nq;
    const float* noise;                               /* Exp(1) draws, rows of noise_stride >= stride elements (sample.py:398) */
    int64_t noise_stride;
    /* sampling parameters (sample_search_results' arguments) */
    int32_t k_positive, k_total, max_support_size, in_batch_negatives;
    float temperature;
    int32_t flags;                                    /* 0, or VODHIP_SAMPLE_KEEP_TOP_SUPPORT */
    /* workspace: the merged rows at full stride (also an output: what _merge_search_results returns, uncut) */
    int64_t* merged_idx;                              /* [nq, stride] */
    int64_t* merged_lbl;                              /* [nq, stride] */
    float* merged_scr;                                /* [nq, stride] */
    float* merged_raw[VODHIP_MAX_ENGINES];            /* [nq, stride] each */
    int32_t* row_cursor;                              /* [nq, VODHIP_MAX_ENGINES] */
    /* sampled sections [nq, k_total] */
    int64_t* out_local;                               /* column of the merged row, -1 pad */
    int64_t* out_ids;
    float* out_scores;
    float* out_log_weights;
    uint8_t* out_labels;
    float* out_raw[VODHIP_MAX_ENGINES];
    float* out_lse_pos;                               /* [nq] */
    float* out_lse_neg;                               /* [nq] */
    float* out_max_sampling_id;                       /* [nq] */
    /* flattened batch (in_batch_negatives != 0): U = nq * k_total <= 8192 */
    int64_t* flat_ids;                                /* [U] */
    float* flat_scores;                               /* [nq, U] */
    float* flat_log_weights;                          /* [nq, U] */
    uint8_t* flat_labels;                             /* [nq, U] */
    float* flat_raw[VODHIP_MAX_ENGINES];              /* [nq, U] each */
    int32_t* flat_n_unique;                           /* [1] or NULL */
} vodhip_collate_args_t;
int vodhip_collate(const vodhip_collate_args_t* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Wire codec helper (HOST memory, no device work): urlsafe base64 of `head || data` and back.
 * Replaces the inner loops of: serialize_np_array / deserialize_np_array (src/vod_search/io.py:17-32:
 *   base64.urlsafe_b64encode(np.save(...)) / np.load(base64.urlsafe_b64decode(...))), which cost ~9 ms per
 *   direction for a 1024 x 768 float32 batch in CPython; output is byte-identical ('-' '_' alphabet, '=' padding).
 * encode: writes 4*ceil((n_head+n_data)/3) chars to `out`, returns that count.
 * decode: accepts both alphabets, stops at '=' padding; returns the number of bytes written to `out`
 *         (capacity >= 3*n/4), or -1 if a character outside the alphabets is met (the caller falls back to
 *         the lenient library decoder) - never reads or writes out of bounds.
 * ------------------------------------------------------------------------------------------- */
int64_t vodhip_b64url_encode(const uint8_t* head, int64_t n_head, const uint8_t* data, int64_t n_data, char* out);
int64_t vodhip_b64url_decode(const char* src, int64_t n, uint8_t* out);

/* ---------------------------------------------------------------------------------------------
 * H6  serving: request fusion in front of one index + an HTTP/1.1 front for the search service's hot routes
 *     (host code only; native threads, no interpreter on the request path).
 * Replaces: the reference's single uvicorn worker, which runs `faiss_index.search` for ONE request at a time
 *           (src/vod_search/faiss_search/server.py:57-98) while every DataLoader worker of every trainer rank sends its own small
 *           batch (src/vod_dataloaders/realm_dataloader.py:92-118, core/search.py:128-146); and the wire codec of
 *           src/vod_search/io.py:17-32 on the server side of /fast-search.
 *
 * vodhip_batcher   fuses concurrent searches into shared corpus scans (a brute-force scan reads the whole store whatever the batch
 *   size).  Exactly one engine: `index` (searches are pipelined on the batcher's own stream, up to "depth" batches on the device,
 *   result rows written straight into device-visible host memory), `node` (the node index; two batches in flight since round 6: vodhip_node_index_search_async / _finish) or `fn` (a host
 *   callback `fn(user, float32 queries [nq, dim], nq, k, subset | NULL, n_subset, out_scores, out_ids) -> 0 | error`, e.g. a
 *   multi-process group; needs no GPU in this process).  The batcher OWNS the engine's search path while it exists: do not call
 *   vodhip_index_search* on the same handle from elsewhere.
 *   search: blocking, thread-safe; host pointers; q_dtype F32 | F16 (| BF16 for index / node); `subset` = int32 [nq, n_subset]
 *     allowed row labels (vodhip_index_set_query_labels semantics; such a request runs as its own batch) or NULL; `client` = a tag
 *     of the requester (connection id; 0 = anonymous).  Results are identical to separate vodhip_index_search calls: the fused batch
 *     runs with k = max(k_i) and a prefix of a top-k' list is the top-k.
 *   policy (no fixed wait window): a request that finds the engine idle runs at once, unless other clients that searched a moment
 *     ago have nothing pending yet and the batch still fits one query tile ("flat_queries", 256: the scan costs the same with them
 *     aboard) and are DUE (their usual come-back time after an answer, an EMA per client, ends before the grace does) - then it waits for
 *     them at most min("grace_us", "grace_pct" % of a measured scan); requests that arrive while a batch
 *     is on the device are fused, and enqueued behind it once they exceed one query tile (time is linear from there) or when it
 *     completes.  params: "max_queries" (2048), "flat_queries", "grace_us" (1000; 0 = never wait), "grace_pct" (35), "window_us"
 *     (0; > 0 = additionally wait this long for company, round 3's --micro-batch-wait-ms), "depth" (2).
 *   stats: "batches", "requests", "queries", "fused_requests_max", "grace_waits", "grace_expired", "idle_ns", "busy_ns",
 *     "last_batch_queries", "last_batch_requests", "flat_scan_ns", "in_flight", "pending", "active_clients".
 * ------------------------------------------------------------------------------------------- */
typedef struct vodhip_batcher vodhip_batcher_t;
typedef int (*vodhip_search_fn)(void* user, const float* queries, int64_t nq, int k, const int32_t* subset, int n_subset,
                                float* out_scores, int64_t* out_ids);
int vodhip_batcher_create(vodhip_index_t* index, vodhip_node_index_t* node, vodhip_search_fn fn, void* user, int64_t dim,
                          int64_t id_base, vodhip_batcher_t** out);
int vodhip_batcher_destroy(vodhip_batcher_t* batcher);
int vodhip_batcher_set_param(vodhip_batcher_t* batcher, const char* key, int64_t value);
int vodhip_batcher_get_stat(vodhip_batcher_t* batcher, const char* key, int64_t* out);
int vodhip_batcher_search(vodhip_batcher_t* batcher, const void* queries, int q_dtype, int64_t nq, int k, const int32_t* subset,
                          int n_subset, uint64_t client, float* out_scores, int64_t* out_ids);
int vodhip_batcher_forget_client(vodhip_batcher_t* batcher, uint64_t client);  /* the requester went away (connection closed) */

/* vodhip_http  HTTP/1.1 server, one native thread per (keep-alive) connection, in front of a batcher:
 *   POST /fast-search  {"vectors": "<urlsafe-b64(np.save(float32|float16 [nq, dim]))>", "top_k": K}
 *                      -> {"scores": "<b64(npy f32 [nq, K])>", "indices": "<b64(npy i64 [nq, K])>"}   (server.py:76-91, io.py:17-32;
 *                      reply bytes identical to the reference's FastSearchResponse as this package's Python shell writes it)
 *   POST /raw-search?top_k=K   body = raw .npy bytes -> scores f32 [nq, K] || ids i64 [nq, K], shapes in x-nq / x-k
 *   GET  /stats        batcher + front counters as JSON (not in the reference)
 * are parsed, decoded, searched and encoded natively WHEN the request is the plain hot case (known keys, escape-free payload, version
 * 1.0 C-order 2-D .npy of the index dimension, 1 <= top_k <= VODHIP_MAX_K, no subset filter).  Every other request - GET /, POST
 * /search, subset filters, unknown fields, malformed payloads, out-of-range top_k - is handed unchanged to `fallback`, which answers
 * through vodhip_http_reply_set (status, content type, payload, extra "name: value\r\n" header lines): validation and error mapping
 * (422 / 500 + trace, models.py:43-79, server.py:89-91) live in ONE place, the host's.  `fallback` is called on the connection's
 * thread; NULL answers 404.  listen_tcp binds every address `host` resolves to and returns the port (port 0 = pick one);
 * start spawns the accept thread and returns; stop closes listeners and connections and waits for the connection threads. */
typedef struct vodhip_http vodhip_http_t;
typedef struct vodhip_http_reply vodhip_http_reply_t;
typedef void (*vodhip_http_fallback_fn)(void* user, const char* method, const char* target, const uint8_t* body, int64_t n_body,
                                        uint64_t client, vodhip_http_reply_t* reply);
int vodhip_http_reply_set(vodhip_http_reply_t* reply, int status, const char* content_type, const uint8_t* payload, int64_t n,
                          const char* extra_headers);
int vodhip_http_create(vodhip_batcher_t* batcher, int64_t dim, vodhip_http_fallback_fn fallback, void* user, int64_t max_body_bytes,
                       vodhip_http_t** out);
int vodhip_http_listen_tcp(vodhip_http_t* http, const char* host, int port);
int vodhip_http_listen_unix(vodhip_http_t* http, const char* path);
int vodhip_http_start(vodhip_http_t* http);
int vodhip_http_stop(vodhip_http_t* http);
int vodhip_http_destroy(vodhip_http_t* http);
int vodhip_http_get_stat(vodhip_http_t* http, const char* key, int64_t* out);  /* "requests_native", "requests_fallback", "connections", "open_connections" */

/* vodhip_client  the CLIENT side of the search service: one kept-alive HTTP/1.1 connection (TCP, or a Unix-domain socket when `unix_path`
 * is given) that speaks POST /fast-search - the reference's documents, byte for byte what its client sends and parses - and POST
 * /raw-search.  Replaces: FaissClient.search (src/vod_search/faiss_search/client.py:64-110) + io.serialize_np_array /
 * deserialize_np_array (src/vod_search/io.py:17-32) for consumers that are not Python (cgo / JNI: same C-ABI as the in-process index) and
 * for Python workers (one call with the GIL released instead of ~150 us of interpreter per exchange).
 *   One handle = one connection = one caller at a time (a handle per thread).  A connection the server closed while idle is re-opened once.
 *   search: queries host [nq, dim] F32 | F16; route 0 = /fast-search, 1 = /raw-search; timeout_s <= 0 = 120 s.
 *           returns 0 with out_scores / out_ids [nq, k] filled; > 0 = the HTTP status of an error reply (its body: vodhip_client_last_body -
 *           the server's `{"detail": ...}`); -1 = transport / protocol failure (vodhip_last_error; "timed out ..." for a timeout). */
typedef struct vodhip_client vodhip_client_t;
int vodhip_client_create(const char* host, int port, const char* unix_path, vodhip_client_t** out);
int vodhip_client_destroy(vodhip_client_t* client);
int vodhip_client_search(vodhip_client_t* client, const void* queries, int q_dtype, int64_t nq, int64_t dim, int k, int route, double timeout_s,
                         float* out_scores, int64_t* out_ids);
const char* vodhip_client_last_body(vodhip_client_t* client);

/* The native front's wire pieces, exposed so that they can be checked byte for byte against NumPy / the Python codec without a
 * socket or a GPU (tests/test_host_logic.py):
 *   npy_header         np.lib.format.write_array_header_1_0 for a C-ordered [rows, cols] array; dtype VODHIP_F32 | VODHIP_F16 | 3 (int64);
 *                      out >= 192 bytes; returns the header length (the data offset)
 *   parse_npy          0 + dtype / rows / cols / data offset for a version-1.0 little-endian float32 | float16 | int64 (dtype code 3)
 *                      C-order 2-D array whose data is complete; -1 for anything else (not an error: the host's reader takes over)
 *   parse_fast_search  0 + the [begin, end) span of the "vectors" payload and top_k (default 3) for the plain hot document; 1 otherwise
 *   fast_search_reply  the /fast-search reply body; out = NULL returns the size needed */
int64_t vodhip_wire_npy_header(int dtype, int64_t rows, int64_t cols, uint8_t* out, int64_t cap);
int vodhip_wire_parse_npy(const uint8_t* data, int64_t n, int* dtype, int64_t* rows, int64_t* cols, int64_t* data_offset);
int vodhip_wire_parse_fast_search(const char* body, int64_t n, int64_t* vec_begin, int64_t* vec_end, int64_t* top_k);
int64_t vodhip_wire_fast_search_reply(const float* scores, const int64_t* ids, int64_t nq, int k, char* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* VODHIP_H */

// Hybrid score merge on gfx950: lookup + N scored engines -> union in first-seen order with weighted,
// min-subtracted score accumulation, lookup labels and per-engine raw scores.
//
// Replaces (paths relative to /root/reference/src/vod_dataloaders/core):
//   _merge_search_results          search.py:79-125
//   _subtract_min_score            normalize.py:17-20
//   _nopy_merge_two_search_results merge.py:108-164 (folded pairwise over the engines, merge.py:31-43)
//   gather_values_by_indices       numpy_ops.py:24-143 (raw scores + labels, merge.py:45-60)
//
// One 256-thread workgroup per query row; the row's concatenated (id, weighted score) list lives in LDS.
// The reference's sequential fold is reproduced exactly: an entry is a "first occurrence" when no
// earlier entry of the concatenation carries its id; its output column is the number of first
// occurrences before it; its score is the left fold, in sequence order, of every occurrence's weighted
// score.
//
// Round 3: every "where does this id occur" question - the reference's O(K) `_search_1d_arr` scans, which round 2
// transliterated into O(W) serial LDS scans per thread (~0.5 M dependent LDS reads per row at W = 385) - is answered by
// ONE open-addressing table in LDS keyed by (id, segment) -> the smallest position of that id inside that segment
// (segment 0 = the lookup list, segment e + 1 = engine e; an entry is one 64-bit word, fingerprint | position, so a probe
// is one LDS read).  Segments are folded in order, so
//   first occurrence   = smallest position in the first segment that holds the id,
//   folded score       = w[first] then + w[segment-first of each later segment]      (one probe per later segment),
//   label / raw scores = value at the segment-first position                         (one probe each; pad columns probe id -1).
// A row costs O(W) LDS operations.  The only case the table does not order is an id REPEATED INSIDE one segment (legal,
// pinned by the reference fixture `merge_corners`): the slot of such an (id, segment) is flagged while the table is built and
// the fold of THAT id walks THAT segment in position order (round 3, first version: the whole row fell back to linear scans -
// one row with one repeated id took 33 us and with it the whole launch).  All float arithmetic uses the non-contracting intrinsics so that results are
// bit-identical to the reference's float32 NumPy arithmetic (no FMA contraction).  This is latency-bound integer work
// (< 1 MB per batch); it is deliberately not shaped into a GEMM.
#include <cstdlib>

#include "vodhip_internal.h"

namespace vodhip {

#ifdef VODHIP_ABLATION
__device__ long long g_probe_hybrid[256];
#endif
#ifdef VODHIP_ABLATION
#define HY_PROBE(i)                        \
    do {                                   \
        VODHIP_PROBE(g_probe_hybrid, i);   \
        if (hy_stop == (i)) return;        \
    } while (0)
#else
#define HY_PROBE(i) VODHIP_PROBE(g_probe_hybrid, i)
#endif

constexpr int HY_THREADS = 512;  // 8 wavefronts: one entry per thread up to W = 512 (the probes are latency chains: width beats depth)
constexpr int HY_WAVES = HY_THREADS / 64;
constexpr int HY_MAX_SEG = 5;  // lookup + VODHIP_MAX_ENGINES
typedef unsigned long long hy_u64;
constexpr hy_u64 HY_EMPTY = ~0ull;

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// Segment s covers positions [off(s), off(s + 1)).  The offsets are NAMED scalars (SGPRs), not an array: LLVM turns a select
// chain over array elements back into an indexed load from a scratch copy of the array - 32 bytes of private memory that made
// every probe pay two scratch round trips AND made the dispatch itself wait for a scratch set-up (~25 us per launch, measured:
// profiles/r03_c5_*).
struct HySegs {
    int o1, o2, o3, o4, o5;  // off(0) = 0
    int n;                   // number of segments
};

__device__ __forceinline__ int hy_off(const HySegs& sg, int s) {
    int v = 0;
    v = s >= 1 ? sg.o1 : v;
    v = s >= 2 ? sg.o2 : v;
    v = s >= 3 ? sg.o3 : v;
    v = s >= 4 ? sg.o4 : v;
    v = s >= 5 ? sg.o5 : v;
    return v;
}

__device__ __forceinline__ int hy_seg_of(const HySegs& sg, int p) {
    return (sg.n > 1 && p >= sg.o1) + (sg.n > 2 && p >= sg.o2) + (sg.n > 3 && p >= sg.o3) + (sg.n > 4 && p >= sg.o4);
}

// 32-bit hash of (id, segment): the table slot is its low bits, the whole word is the fingerprint stored with the entry
__device__ __forceinline__ unsigned hy_hash(int64_t id, int seg) {
    const unsigned long long x = ((unsigned long long)id + 0x632BE59BD9B4E019ull * (unsigned)(seg + 1)) * 0x9E3779B97F4A7C15ull;
    unsigned h = (unsigned)(x >> 32);
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}

// Table entry = (fingerprint << 32) | position.  A probe reads ONE LDS word; `ids[]` is read only when the fingerprint matches
// (a hit, or a 2^-32 collision that the id comparison rejects).
// smallest position of `id` inside segment `seg`, or -1.  Only called after the table is complete.
__device__ __forceinline__ int hy_find(const hy_u64* table, unsigned mask, const int64_t* ids, const HySegs& sg, int64_t id, int seg,
                                       unsigned* slot_out = nullptr) {
    const unsigned h = hy_hash(id, seg);
    unsigned slot = h & mask;
    const int lo = hy_off(sg, seg), hi = hy_off(sg, seg + 1);
    while (true) {
        const hy_u64 e = table[slot];
        if (e == HY_EMPTY) return -1;
        if ((unsigned)(e >> 32) == h) {
            const int cur = (int)(unsigned)e;
            if (cur >= lo && cur < hi && ids[cur] == id) {
                if (slot_out) *slot_out = slot;
                return cur;
            }
        }
        slot = (slot + 1) & mask;
    }
}

__global__ __launch_bounds__(HY_THREADS) void merge_hybrid_kernel(HybridArgs a, int T_) {
    const int T = T_ & 0xFFFFFF;
#ifdef VODHIP_ABLATION
    const int hy_stop = (T_ >> 24) - 1;  // diagnostic builds: return after phase stamp `hy_stop` (VODHIP_HY_STOP), -1 = run everything
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    HySegs sg;
    sg.n = a.n_engines + 1;
    sg.o1 = a.k_lookup;
    sg.o2 = sg.o1 + (a.n_engines > 0 ? a.engine_k[0] : 0);
    sg.o3 = sg.o2 + (a.n_engines > 1 ? a.engine_k[1] : 0);
    sg.o4 = sg.o3 + (a.n_engines > 2 ? a.engine_k[2] : 0);
    sg.o5 = sg.o4 + (a.n_engines > 3 ? a.engine_k[3] : 0);
    const int W = sg.o5;
    const unsigned mask = (unsigned)T - 1u;
    // LDS carve-up
    int64_t* ids = (int64_t*)smem;                 // [W] concatenated ids
    hy_u64* table = (hy_u64*)(ids + W);            // [T] (id, segment) -> (fingerprint, smallest position); HY_EMPTY = free
    float* wsc = (float*)(table + T);              // [W] weighted, min-subtracted scores
    float* nsc = wsc + W;                          // [W] min-subtracted (unweighted) scores
    int* pre = (int*)(nsc + W);                    // [W + 1] exclusive prefix of the first-occurrence flags (pre[W] = cursor)
    float(*s_min)[4] = (float(*)[4])(pre + W + 1); // [HY_WAVES][4 engines] partial minima
    int* s_misc = (int*)(s_min + HY_WAVES);        // [0..HY_WAVES) wave sums of the scan, [HY_WAVES] row has an id repeated inside a segment
    unsigned char* multi = (unsigned char*)(s_misc + HY_WAVES + 1);  // [T] the slot's (id, segment) occurs more than once

    HY_PROBE(0);
    VODHIP_PROBE_WG(g_probe_hybrid, 0);
    // ---- 1. per-engine row minimum over finite scores (normalize.py:17-20) ----
    for (int e = 0; e < a.n_engines; ++e) {
        const float* sc = a.engine_scr[e] + row * a.engine_k[e];
        float m = __builtin_inff();
        for (int j = tid; j < a.engine_k[e]; j += HY_THREADS) {
            const float v = sc[j];
            if (!(__builtin_isinf(v) || v != v)) m = fminf(m, v);
        }
        m = wave_min(m);
        if (lane == 0) s_min[wave][e] = m;
    }
    for (int h = tid; h < T; h += HY_THREADS) {
        table[h] = HY_EMPTY;
        multi[h] = 0;
    }
    if (tid == 0) s_misc[HY_WAVES] = 0;
    __syncthreads();
    HY_PROBE(1);

    // ---- 2. concatenate: lookup (score 0 * 0), then each engine ((s - min) * w); every entry goes into the table on the way.
    //         A slot, once claimed, only ever holds entries of ONE (id, segment): equal fingerprints, so atomicMin lowers the
    //         position and nothing else ----
    auto insert = [&](int p, int64_t id, int seg) {
        const unsigned h = hy_hash(id, seg);
        const hy_u64 mine = ((hy_u64)h << 32) | (unsigned)p;
        const int lo = hy_off(sg, seg), hi = hy_off(sg, seg + 1);
        unsigned slot = h & mask;
        while (true) {
            hy_u64 e = __atomic_load_n(&table[slot], __ATOMIC_RELAXED);
            if (e == HY_EMPTY) {
                e = atomicCAS(&table[slot], HY_EMPTY, mine);
                if (e == HY_EMPTY) return;  // claimed
            }
            if ((unsigned)(e >> 32) == h) {
                const int cur = (int)(unsigned)e;
                if (cur >= lo && cur < hi && ids[cur] == id) {
                    atomicMin(&table[slot], mine);
                    multi[slot] = 1;  // the id occurs twice in this segment: its fold walks that segment in position order
                    if (id >= 0) s_misc[HY_WAVES] = 1;
                    return;
                }
            }
            slot = (slot + 1) & mask;
        }
    };
    for (int j = tid; j < a.k_lookup; j += HY_THREADS) {
        const int64_t id = a.lookup_idx[row * a.k_lookup + j];
        ids[j] = id;
        wsc[j] = 0.0f;  // lookup scores are zeroed, min-subtracted (0 - 0) and weighted by 0.0 (search.py:92,117)
        nsc[j] = 0.0f;
    }
    for (int e = 0; e < a.n_engines; ++e) {
        float mn = s_min[0][e];
#pragma unroll
        for (int w = 1; w < HY_WAVES; ++w) mn = fminf(mn, s_min[w][e]);
        const int ke = a.engine_k[e], off = hy_off(sg, e + 1);
        const float w = a.engine_w[e];
        for (int j = tid; j < ke; j += HY_THREADS) {
            const float n = __fsub_rn(a.engine_scr[e][row * ke + j], mn);  // offset 0.0 adds nothing
            ids[off + j] = a.engine_idx[e][row * ke + j];
            nsc[off + j] = n;
            wsc[off + j] = __fmul_rn(n, w);
        }
    }
    __syncthreads();
    HY_PROBE(2);
    for (int p = tid; p < W; p += HY_THREADS) insert(p, ids[p], hy_seg_of(sg, p));
    __syncthreads();
    HY_PROBE(3);

    // ---- 3. first-occurrence flags (into pre[]) and their exclusive prefix = output columns.  An entry is a first occurrence
    //         when no EARLIER segment holds its id (one probe per earlier segment: none for the lookup list) and, in rows with
    //         an id repeated inside a segment, it is also the first of its own segment ----
    const bool has_dup = s_misc[HY_WAVES] != 0;
    const int chunk = (W + HY_THREADS - 1) / HY_THREADS;
    const int c_lo = min(W, tid * chunk), c_hi = min(W, c_lo + chunk);
    int cnt = 0;
    for (int p = c_lo; p < c_hi; ++p) {
        const int64_t id = ids[p];
        const int seg = hy_seg_of(sg, p);
        int f = id >= 0 && (!has_dup || hy_find(table, mask, ids, sg, id, seg) == p);
        for (int s = 0; f && s < seg; ++s) f = hy_find(table, mask, ids, sg, id, s) < 0;
        pre[p] = f;
        cnt += f;
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_misc[wave] = incl;
    __syncthreads();
    int run = incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < HY_WAVES; ++w) {
        run += w < wave ? s_misc[w] : 0;
        total += s_misc[w];
    }
    for (int p = c_lo; p < c_hi; ++p) {
        const int f = pre[p];
        pre[p] = run;
        run += f;
    }
    if (tid == 0) pre[W] = total;
    __syncthreads();
    const int cursor = total;
    HY_PROBE(4);

    // ---- 4. every first occurrence writes its column: folded score, lookup label, raw score per engine ----
    int64_t* o_idx = a.out_idx + row * a.out_stride;
    float* o_scr = a.out_scr + row * a.out_stride;
    int64_t* o_lbl = a.out_lbl ? a.out_lbl + row * a.out_stride : nullptr;
    for (int p = tid; p < W; p += HY_THREADS) {
        const int col = pre[p];
        if (pre[p + 1] == col) continue;
        const int64_t id = ids[p];
        const int seg = hy_seg_of(sg, p);
        // segment-first position of the id in every LATER segment (-1: absent): one probe each.  The fold adds every occurrence in
        // position order: one per segment - except where the table saw the (id, segment) more than once (rare: an engine that
        // returns a section twice), where that segment is walked from its first occurrence
        int q_seg[HY_MAX_SEG];
        float acc = wsc[p];
        if (has_dup) {
            unsigned own = 0;
            (void)hy_find(table, mask, ids, sg, id, seg, &own);
            if (multi[own])
                for (int r = p + 1, hi = hy_off(sg, seg + 1); r < hi; ++r)
                    if (ids[r] == id) acc = __fadd_rn(wsc[r], acc);
        }
#pragma unroll
        for (int s = 1; s < HY_MAX_SEG; ++s) {
            q_seg[s] = -1;
            if (s > seg && s < sg.n) {
                unsigned slot = 0;
                const int q = hy_find(table, mask, ids, sg, id, s, &slot);
                q_seg[s] = q;
                if (q >= 0) {
                    acc = __fadd_rn(wsc[q], acc);  // scores[found] = score + scores[found]
                    if (has_dup && multi[slot])
                        for (int r = q + 1, hi = hy_off(sg, s + 1); r < hi; ++r)
                            if (ids[r] == id) acc = __fadd_rn(wsc[r], acc);
                }
            }
        }
        o_idx[col] = id;
        o_scr[col] = acc;
        if (o_lbl) o_lbl[col] = (seg == 0 && a.lookup_lbl) ? a.lookup_lbl[row * a.k_lookup + p] : -1;  // first match in the lookup list
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e >= a.n_engines || !a.out_raw[e]) continue;
            const int q = seg == e + 1 ? p : q_seg[e + 1];
            a.out_raw[e][row * a.out_stride + col] = q >= 0 ? nsc[q] : __builtin_nanf("");
        }
    }
    HY_PROBE(5);
    // columns beyond the cursor carry id -1, which MATCHES the -1 pads of the lookup / engine lists: their
    // label and raw scores are those of the first -1 entry of each list
    int64_t pad_lbl = -1;
    if (o_lbl && a.lookup_lbl) {
        const int q = hy_find(table, mask, ids, sg, -1, 0);
        if (q >= 0) pad_lbl = a.lookup_lbl[row * a.k_lookup + q];
    }
    float pad_raw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int q = (e < a.n_engines && a.out_raw[e]) ? hy_find(table, mask, ids, sg, -1, e + 1) : -1;
        pad_raw[e] = q >= 0 ? nsc[q] : __builtin_nanf("");
    }
    for (int c = cursor + tid; c < a.out_stride; c += HY_THREADS) {
        o_idx[c] = -1;
        o_scr[c] = -__builtin_inff();
        if (o_lbl) o_lbl[c] = pad_lbl;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < a.n_engines && a.out_raw[e]) a.out_raw[e][row * a.out_stride + c] = pad_raw[e];
    }
    // The reference folds the engines pairwise and cuts the buffer to `[: max_cursor + 1]` after EVERY fold
    // (merge.py:160-162), so the final width depends on the per-stage maxima over rows of the cursor.
    // stage e = "after engine e has been folded in": cursor_e = #first occurrences among the first off[e + 2] entries.
    if (tid < a.n_engines) {
        const int c = pre[hy_off(sg, tid + 2)];
        if (a.out_width) atomicMax(&a.out_width[tid], c);
        if (a.out_row_cursor) a.out_row_cursor[row * 4 + tid] = c;  // plain stores: a consumer on the device takes the maximum itself
    }
    HY_PROBE(6);
    VODHIP_PROBE_WG(g_probe_hybrid, 1);
}

hipError_t read_probe_hybrid(long long* out) {
#ifdef VODHIP_ABLATION
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe_hybrid), sizeof(long long) * 256);
#else
    (void)out;
    return hipErrorNotSupported;
#endif
}

hipError_t launch_merge_hybrid(const HybridArgs& a, hipStream_t stream) {
    if (a.out_width) {
        hipError_t e = hipMemsetAsync(a.out_width, 0, sizeof(int32_t) * 4, stream);
        if (e != hipSuccess) return e;
    }
    if (a.nq == 0) return hipSuccess;
    int W = a.k_lookup;
    for (int i = 0; i < a.n_engines; ++i) W += a.engine_k[i];
    // load factor <= 1/4 where LDS allows it (short probe chains: a wavefront waits for its longest one), never above 1/2
    int T = 64;
    while (T < 4 * W) T <<= 1;
    const size_t fixed = (size_t)W * (8 + 4 + 4) + (size_t)(W + 1) * 4 + HY_WAVES * 4 * 4 + (HY_WAVES + 1) * 4 + 16;
    while (fixed + (size_t)T * 9 > 150 * 1024 && T > 2 * W) T >>= 1;
    const size_t lds = fixed + (size_t)T * 9;
    if (lds > 64 * 1024) {
        hipError_t e = allow_dynamic_lds((const void*)merge_hybrid_kernel, (int)lds);
        if (e != hipSuccess) return e;
    }
    int t_arg = T;
#ifdef VODHIP_ABLATION
    if (const char* st = getenv("VODHIP_HY_STOP")) t_arg |= (atoi(st) + 1) << 24;
#endif
    hipLaunchKernelGGL(merge_hybrid_kernel, dim3((unsigned)a.nq), dim3(HY_THREADS), lds, stream, a, t_arg);
    return hipGetLastError();
}

}  // namespace vodhip

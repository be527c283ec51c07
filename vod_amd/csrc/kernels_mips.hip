// Fused inner-product scoring + streaming top-k for gfx950 (MI355X, CDNA4).
//
// Replaces the arithmetic behind `faiss_index.search(query_vec, k)`
// (/root/reference/src/vod_search/faiss_search/server.py:72,84) for a Flat / inner-product index.
//
// Data layout in HBM
//   store X : [capacity][dim_pad] fp16|bf16, row-major, dim_pad % 64 == 0, rows >= ntotal are zero
//   queries : [nq_pad][dim_pad] same dtype (workspace), nq_pad % BN == 0, padding rows are zero
//
// mips_filter_kernel: one workgroup owns a BM(corpus rows) x BN(queries) score tile.  The K loop stages
// 64-deep slices of both operands into LDS with LDS-DMA (`global_load_lds_dwordx4`, 16 B per lane, the
// XOR swizzle applied on the per-lane SOURCE address so the LDS image stays lane-linear) and runs
// v_mfma_f32_32x32x16_{f16,bf16} with the corpus as the A operand and the queries as the B operand, so
// that in the accumulator a lane's column is ONE query and its 16 registers are 16 corpus rows.
// The score tile never leaves registers: each lane compares its scores with its query's running
// threshold (the score of the k-th best hit found in earlier chunks) and only survivors are appended
// (packed 64-bit keys) to the query's candidate list.  mips_select_kernel folds the candidates
// into the running sorted top-k and tightens the threshold between chunks.
//
// Roofline: 2*nq*N*D flop per batch on MFMA vs N*D*2 bytes of HBM; arithmetic intensity = nq flop/B.
#include "vodhip_internal.h"

namespace vodhip {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

// ------------------------------------------------------------------------------------------------
// key packing
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int flip_f32(float s) {
    unsigned int u = __float_as_uint(s + 0.0f);  // -0.0 -> +0.0 so that equal floats get equal keys
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unflip_f32(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ key_t64 make_key(float s, unsigned int local_row) {
    return ((key_t64)flip_f32(s) << 32) | (key_t64)(0xFFFFFFFFu - local_row);
}

template <int DT>
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (DT == 0) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const AS1 void*)gsrc, (AS3 void*)lds_dst, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// filter kernel
// ------------------------------------------------------------------------------------------------
constexpr int BK = 64;            // K elements per LDS stage (128 B per row)
constexpr int ROW_BYTES = BK * 2;  // 128

template <int DT, int BM, int BN, int WM, int WN, bool DENSE>
__global__ __launch_bounds__(WM* WN * 64, 2) void mips_filter_kernel(
    const uint16_t* __restrict__ X, const uint16_t* __restrict__ Q, int dim_pad, int row_begin, int row_end,
    int n_xtiles, int n_qtiles, int nq, const float* __restrict__ thr_s, const key_t64* __restrict__ thr_key,
    key_t64* __restrict__ cand, unsigned int* __restrict__ cnt, int cap, unsigned int* __restrict__ overflow) {
    constexpr int NWAVES = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tile
    constexpr int MI = TM / 32, NJ = TN / 32;  // 32x32 blocks per wave
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NA = BM / 8 / NWAVES;  // LDS-DMA wave-instructions per wave per stage (8 rows each)
    constexpr int NB = BN / 8 / NWAVES;
    static_assert(BM % (8 * NWAVES) == 0 && BN % (8 * NWAVES) == 0, "tile/wave mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the n_qtiles
    // workgroups that re-read one corpus tile are dealt to the same XCD back to back (L2 reuse only;
    // correctness does not depend on placement).
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jj = bid >> 3;
    const int qt = jj % n_qtiles;
    const int xt = (jj / n_qtiles) * 8 + xcd;
    if (xt >= n_xtiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int x0 = row_begin + xt * BM;  // first corpus row of the tile
    const int q0 = qt * BN;              // first query of the tile

    // ---- per-lane LDS-DMA source pointers: lane -> (row = base + lane/8, 16-B slot = lane%8) ----
    // slot s of row r holds logical chunk c = s ^ ((r >> 1) & 7)  (conflict-free ds_read_b128, see below)
    const int st_row = lane >> 3, st_slot = lane & 7;
    const char* a_src[NA];
    const char* b_src[NB];
#pragma unroll
    for (int t = 0; t < NA; ++t) {
        const int r = (wave * NA + t) * 8 + st_row;
        const int c = st_slot ^ ((r >> 1) & 7);
        a_src[t] = (const char*)X + ((size_t)(x0 + r) * dim_pad + c * 8) * 2;
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int r = (wave * NB + t) * 8 + st_row;
        const int c = st_slot ^ ((r >> 1) & 7);
        b_src[t] = (const char*)Q + ((size_t)(q0 + r) * dim_pad + c * 8) * 2;
    }

    auto stage = [&](int buf, int kbyte) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int t = 0; t < NA; ++t) glds16(a_src[t] + kbyte, sa + (wave * NA + t) * 8 * ROW_BYTES);
#pragma unroll
        for (int t = 0; t < NB; ++t) glds16(b_src[t] + kbyte, sb + (wave * NB + t) * 8 * ROW_BYTES);
    };

    // ---- fragment read addressing --------------------------------------------------------------
    // MFMA 32x32x16: lane l supplies A[row l&31][k = 8h..8h+7] and B[k = 8h..8h+7][col l&31], h = l>>5.
    // Logical chunk of k-substep kk is 2*kk + h; its slot is chunk ^ ((row>>1)&7) and, because every
    // block starts at a multiple of 32 rows, (row>>1)&7 == (l>>1)&7 for every block.
    const int fr = lane & 31, fh = lane >> 5;
    const int swz = (lane >> 1) & 7;
    const int a_row_off = (wm * TM + fr) * ROW_BYTES;
    const int b_row_off = A_BYTES + (wn * TN + fr) * ROW_BYTES;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // thresholds of this lane's queries (loaded early; latency hidden behind the K loop)
    float thr[NJ];
    if constexpr (!DENSE) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = q0 + wn * TN + j * 32 + fr;
            thr[j] = (q < nq) ? thr_s[q] : __builtin_inff();
        }
    }

    const int nk = dim_pad / BK;
    stage(0, 0);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();  // vmcnt(0): slice t has landed; barrier: everyone is done reading the other buffer
        if (t + 1 < nk) stage((t + 1) & 1, (t + 1) * ROW_BYTES);
        const char* base = smem + (t & 1) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int slot_off = ((2 * kk + fh) ^ swz) << 4;
            u32x4 af[MI], bf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const u32x4*)(base + a_row_off + i * 32 * ROW_BYTES + slot_off);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = *(const u32x4*)(base + b_row_off + j * 32 * ROW_BYTES + slot_off);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<DT>(af[i], bf[j], acc[i][j]);
        }
    }

    // ---- epilogue: threshold filter --------------------------------------------------------------
    // C layout of v_mfma_f32_32x32x16: col = lane&31 (query), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = q0 + wn * TN + j * 32 + fr;
        const bool q_ok = q < nq;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rbase = x0 + wm * TM + i * 32 + 4 * fh;
            if constexpr (DENSE) {
                if (q_ok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        const float s = acc[i][j][r];
                        if (row < row_end) {
                            const key_t64 key = (s == s) ? make_key(s, (unsigned)row) : 0ull;  // NaN never enters
                            cand[(size_t)q * cap + (row - row_begin)] = key;
                        }
                    }
                }
            } else {
                float m = acc[i][j][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
                const bool hit = m >= thr[j];  // false for NaN and for padded queries (thr = +inf)
                if (__any(hit)) {
                    if (hit) {
                        const key_t64 tk = thr_key[q];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = rbase + (r & 3) + 8 * (r >> 2);
                            const float s = acc[i][j][r];
                            if (s >= thr[j] && row < row_end) {
                                const key_t64 key = make_key(s, (unsigned)row);
                                if (key > tk) {
                                    const unsigned slot = atomicAdd(&cnt[q], 1u);
                                    if (slot < (unsigned)cap)
                                        cand[(size_t)q * cap + slot] = key;
                                    else
                                        atomicOr(overflow, 1u);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// select kernel: fold the chunk's candidates into the running sorted top-k, tighten the threshold
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mips_select_kernel(key_t64* __restrict__ topk, int kp, int k,
                                                          const key_t64* __restrict__ cand,
                                                          unsigned int* __restrict__ cnt, int cap, int dense_n,
                                                          float* __restrict__ thr_s, key_t64* __restrict__ thr_key,
                                                          unsigned int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    key_t64* keys = (key_t64*)smem;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    unsigned n = dense_n >= 0 ? (unsigned)dense_n : cnt[q];
    if (n > (unsigned)cap) {
        if (tid == 0) atomicOr(overflow, 1u);
        n = cap;
    }
    const int total = kp + (int)n;
    int P = 64;
    while (P < total) P <<= 1;
    for (int i = tid; i < P; i += 256) {
        key_t64 v = 0;
        if (i < kp)
            v = topk[(size_t)q * kp + i];
        else if (i < total)
            v = cand[(size_t)q * cap + (i - kp)];
        keys[i] = v;
    }
    // bitonic sort, descending
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (P >> 1); t += 256) {
                const int pos = 2 * t - (t & (stride - 1));
                const key_t64 a = keys[pos], b = keys[pos + stride];
                const bool desc = (pos & size) == 0;
                if ((a < b) == desc) {
                    keys[pos] = b;
                    keys[pos + stride] = a;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < kp; i += 256) topk[(size_t)q * kp + i] = keys[i];
    if (tid == 0) {
        const key_t64 kth = keys[k - 1];
        thr_key[q] = kth;
        thr_s[q] = kth ? unflip_f32((unsigned)(kth >> 32)) : -__builtin_inff();
        cnt[q] = 0;
    }
}

__global__ void mips_init_kernel(key_t64* topk, int64_t n_topk, unsigned int* cnt, float* thr_s, key_t64* thr_key,
                                 int64_t nq_pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_topk) topk[i] = 0;
    if (i < nq_pad) {
        cnt[i] = 0;
        thr_s[i] = -__builtin_inff();
        thr_key[i] = 0;
    }
}

__global__ void mips_output_kernel(const key_t64* __restrict__ topk, int kp, int k, int64_t nq, int64_t id_base,
                                   float* __restrict__ out_scores, int64_t* __restrict__ out_ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * k) return;
    const int64_t q = i / k;
    const int c = (int)(i % k);
    const key_t64 key = topk[q * kp + c];
    if (key == 0) {
        out_scores[i] = -__builtin_inff();
        out_ids[i] = -1;
    } else {
        out_scores[i] = unflip_f32((unsigned)(key >> 32));
        out_ids[i] = id_base + (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
    }
}

// ------------------------------------------------------------------------------------------------
// row conversion (ingest + query staging): src [n, dim] of f16|bf16|f32 -> dst [n, dst_stride] f16|bf16,
// columns >= dim zero-filled.  One thread per 8 destination elements (16-B stores).
// ------------------------------------------------------------------------------------------------
template <int SRC, int DST>
__global__ void convert_rows_kernel(const void* __restrict__ src, int64_t n_rows, int64_t dim, uint16_t* __restrict__ dst,
                                    int64_t dst_stride) {
    const int64_t chunks_per_row = dst_stride / 8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * chunks_per_row) return;
    const int64_t row = i / chunks_per_row;
    const int64_t c0 = (i % chunks_per_row) * 8;
    uint16_t out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t c = c0 + e;
        uint16_t v = 0;
        if (c < dim) {
            float f;
            if constexpr (SRC == 2) {
                f = ((const float*)src)[row * dim + c];
            } else if constexpr (SRC == 0) {
                f = (float)(((const _Float16*)src)[row * dim + c]);
            } else {
                f = (float)(((const __bf16*)src)[row * dim + c]);
            }
            if constexpr (DST == 0) {
                const _Float16 h = (_Float16)f;
                v = __builtin_bit_cast(uint16_t, h);
            } else {
                const __bf16 h = (__bf16)f;
                v = __builtin_bit_cast(uint16_t, h);
            }
        }
        out[e] = v;
    }
    *(uint4*)(dst + row * dst_stride + c0) = *(const uint4*)out;
}

hipError_t launch_convert_rows(const void* src, int src_dtype, int64_t n_rows, int64_t dim, void* dst, int dst_dtype,
                               int64_t dst_stride, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    const int64_t total = n_rows * (dst_stride / 8);
    const int threads = 256;
    const unsigned blocks = (unsigned)((total + threads - 1) / threads);
    uint16_t* d = (uint16_t*)dst;
#define VOD_CONV(S, D)                                                                                         \
    if (src_dtype == S && dst_dtype == D) {                                                                    \
        hipLaunchKernelGGL((convert_rows_kernel<S, D>), dim3(blocks), dim3(threads), 0, stream, src, n_rows, dim, d, \
                           dst_stride);                                                                        \
        return hipGetLastError();                                                                              \
    }
    VOD_CONV(0, 0) VOD_CONV(1, 0) VOD_CONV(2, 0) VOD_CONV(0, 1) VOD_CONV(1, 1) VOD_CONV(2, 1)
#undef VOD_CONV
    return hipErrorInvalidValue;
}

hipError_t launch_search_init(const SearchWorkspace& ws, int64_t nq_pad, hipStream_t stream) {
    const int64_t n_topk = nq_pad * ws.kp;
    const int threads = 256;
    const unsigned blocks = (unsigned)((n_topk + threads - 1) / threads);
    hipLaunchKernelGGL(mips_init_kernel, dim3(blocks), dim3(threads), 0, stream, ws.topk, n_topk, ws.cnt, ws.thr_s,
                       ws.thr_key, nq_pad);
    return hipGetLastError();
}

int filter_tile_rows(int tile) { return tile == 2 ? 256 : 128; }
int filter_tile_cols(int tile) { return tile == 2 ? 256 : 128; }

template <int DT, int BM, int BN, int WM, int WN, bool DENSE>
static hipError_t launch_filter_cfg(const void* store, const void* q_pad, int64_t dim_pad, int64_t row_begin,
                                    int64_t row_end, int64_t nq, int64_t nq_pad, const SearchWorkspace& ws,
                                    hipStream_t stream) {
    const int n_xtiles = (int)((row_end - row_begin + BM - 1) / BM);
    const int n_qtiles = (int)(nq_pad / BN);
    const int xgroups = (n_xtiles + 7) / 8;
    const unsigned grid = (unsigned)xgroups * 8u * (unsigned)n_qtiles;
    constexpr int threads = WM * WN * 64;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * ROW_BYTES;
    auto kern = mips_filter_kernel<DT, BM, BN, WM, WN, DENSE>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, (const uint16_t*)store, (const uint16_t*)q_pad,
                       (int)dim_pad, (int)row_begin, (int)row_end, n_xtiles, n_qtiles, (int)nq, ws.thr_s, ws.thr_key,
                       ws.cand, ws.cnt, (int)ws.cap, ws.overflow);
    return hipGetLastError();
}

hipError_t launch_filter(int store_dtype, int tile, bool dense, const void* store, const void* q_pad, int64_t dim_pad,
                         int64_t row_begin, int64_t row_end, int64_t nq, int64_t nq_pad, const SearchWorkspace& ws,
                         hipStream_t stream) {
    if (row_end <= row_begin) return hipSuccess;
#define VOD_FILTER(DT, TILE, BM, BN, WM, WN)                                                                          \
    if (store_dtype == DT && tile == TILE) {                                                                          \
        return dense ? launch_filter_cfg<DT, BM, BN, WM, WN, true>(store, q_pad, dim_pad, row_begin, row_end, nq,     \
                                                                   nq_pad, ws, stream)                                \
                     : launch_filter_cfg<DT, BM, BN, WM, WN, false>(store, q_pad, dim_pad, row_begin, row_end, nq,    \
                                                                    nq_pad, ws, stream);                              \
    }
    VOD_FILTER(0, 1, 128, 128, 2, 2)
    VOD_FILTER(1, 1, 128, 128, 2, 2)
    VOD_FILTER(0, 2, 256, 256, 2, 4)
    VOD_FILTER(1, 2, 256, 256, 2, 4)
#undef VOD_FILTER
    return hipErrorInvalidValue;
}

hipError_t launch_select(const SearchWorkspace& ws, int64_t nq, int k, int64_t dense_n, hipStream_t stream) {
    int64_t total = ws.kp + ws.cap;
    size_t P = 64;
    while ((int64_t)P < total) P <<= 1;
    const size_t lds = P * sizeof(key_t64);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)mips_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(mips_select_kernel, dim3((unsigned)nq), dim3(256), lds, stream, ws.topk, (int)ws.kp, k, ws.cand,
                       ws.cnt, (int)ws.cap, (int)dense_n, ws.thr_s, ws.thr_key, ws.overflow);
    return hipGetLastError();
}

hipError_t launch_output(const SearchWorkspace& ws, int64_t nq, int k, int64_t id_base, float* out_scores,
                         int64_t* out_ids, hipStream_t stream) {
    const int64_t total = nq * k;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(mips_output_kernel, dim3(blocks), dim3(256), 0, stream, ws.topk, (int)ws.kp, k, nq, id_base,
                       out_scores, out_ids);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// merge of per-shard top-k lists: one workgroup per query, bitonic sort of (score, id) pairs
// ordered by (score desc, id asc); invalid entries (id < 0) sink to the end.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pair_before(float sa, int64_t ia, float sb, int64_t ib) {
    // true when (sa, ia) must come before (sb, ib)
    const bool va = ia >= 0, vb = ib >= 0;
    if (va != vb) return va;
    if (!va) return false;
    if (sa != sb) return sa > sb;
    return ia < ib;
}

__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ scores, const int64_t* __restrict__ ids,
                                                         int n_shards, int64_t nq, int k, int k_out,
                                                         float* __restrict__ out_scores, int64_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int total = n_shards * k;
    int P = 64;
    while (P < total) P <<= 1;
    int64_t* sid = (int64_t*)smem;
    float* ssc = (float*)(smem + (size_t)P * 8);
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < P; i += 256) {
        float s = -__builtin_inff();
        int64_t id = -1;
        if (i < total) {
            const int sh = i / k, c = i % k;
            s = scores[((int64_t)sh * nq + q) * k + c];
            id = ids[((int64_t)sh * nq + q) * k + c];
            if (id < 0 || s != s) {
                id = -1;
                s = -__builtin_inff();
            }
        }
        ssc[i] = s;
        sid[i] = id;
    }
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (P >> 1); t += 256) {
                const int pos = 2 * t - (t & (stride - 1));
                const float sa = ssc[pos], sb = ssc[pos + stride];
                const int64_t ia = sid[pos], ib = sid[pos + stride];
                const bool fwd = (pos & size) == 0;  // this run must end up best-first
                const bool swap = fwd ? pair_before(sb, ib, sa, ia) : pair_before(sa, ia, sb, ib);
                if (swap) {
                    ssc[pos] = sb;
                    ssc[pos + stride] = sa;
                    sid[pos] = ib;
                    sid[pos + stride] = ia;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < k_out; i += 256) {
        float s = -__builtin_inff();
        int64_t id = -1;
        if (i < P) {
            s = ssc[i];
            id = sid[i];
        }
        out_scores[q * k_out + i] = s;
        out_ids[q * k_out + i] = id;
    }
}

hipError_t launch_merge_topk(const float* scores, const int64_t* ids, int n_shards, int64_t nq, int k, int k_out,
                             float* out_scores, int64_t* out_ids, hipStream_t stream) {
    if (nq == 0) return hipSuccess;
    const int total = n_shards * k;
    size_t P = 64;
    while ((int)P < total) P <<= 1;
    const size_t lds = P * 12;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)merge_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)nq), dim3(256), lds, stream, scores, ids, n_shards, nq, k, k_out,
                       out_scores, out_ids);
    return hipGetLastError();
}

}  // namespace vodhip

"""HBM-resident corpus vector store + exact inner-product top-k (host wrapper over the C-ABI).

Mirrors what the reference gets from a faiss `IndexFlat` with METRIC_INNER_PRODUCT:
`index.add(float32 batch)` (/root/reference/src/vod_search/faiss_search/build.py:65-73) and
`index.search(query_vec, k)` (/root/reference/src/vod_search/faiss_search/server.py:72,84).
PyTorch is used only to own the query / result buffers and the stream; the store itself is owned by
libvodhip.so.
"""
from __future__ import annotations

import collections
import ctypes

import numpy as np
import torch

from vod_amd import _native


class HipFlatIndex:
    """Exact MIPS index: row-major fp16/bf16 rows in HBM, searched by the fused MFMA + top-k kernels.

    `exact_f32=True` (VODHIP_EXACT_F32): the store also keeps the float32 rows and every search returns what a float32 brute force
    over the UNROUNDED rows and queries returns - the reference's own arithmetic (faiss IndexFlat holds float32,
    /root/reference/src/vod_search/faiss_search/build.py:65-73) - with the fp16 / bf16 scan as the filter.  Without it scores are
    dot products of the values as rounded to `dtype`."""

    def __init__(self, dim: int, capacity: int, dtype: torch.dtype = torch.float16, device: int | torch.device = 0,
                 exact_f32: bool = False):
        self._lib = _native.load_library()
        if not torch.cuda.is_available():
            raise _native.NativeLibraryError("HipFlatIndex needs a ROCm device (torch.cuda.is_available() is False)")
        if dtype not in (torch.float16, torch.bfloat16):
            raise TypeError("store dtype must be torch.float16 or torch.bfloat16")
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.dim = int(dim)
        self.capacity = int(capacity)
        self.dtype = dtype
        self.exact_f32 = bool(exact_f32)
        handle = ctypes.c_void_p()
        _native.check(
            self._lib.vodhip_index_create(
                self.device.index or 0, self.dim, _native.torch_dtype_code(dtype) | (_native.EXACT_F32 if self.exact_f32 else 0),
                self.capacity, ctypes.byref(handle)
            )
        )
        self._h = handle
        self._keep: collections.deque = collections.deque()
        self._keep_subset = None

    # -- lifecycle ---------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vodhip_index_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self) -> "HipFlatIndex":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    # -- store -------------------------------------------------------------------------------------
    @property
    def ntotal(self) -> int:
        out = ctypes.c_int64()
        _native.check(self._lib.vodhip_index_ntotal(self._h, ctypes.byref(out)))
        return out.value

    def reset(self) -> None:
        _native.check(self._lib.vodhip_index_reset(self._h))

    def add(self, vectors: np.ndarray | torch.Tensor) -> None:
        """Append rows (float32/float16 NumPy on the host, or float32/float16/bfloat16 tensors on this device)."""
        if isinstance(vectors, torch.Tensor) and vectors.is_cuda:
            if vectors.device != self.device:
                raise ValueError(f"vectors live on {vectors.device}, the index on {self.device}")
            v = vectors.contiguous()
            if v.ndim != 2 or v.shape[1] != self.dim:
                raise ValueError(f"expected [n, {self.dim}] vectors, got {tuple(v.shape)}")
            _native.check(
                self._lib.vodhip_index_add(
                    self._h, v.data_ptr(), v.shape[0], _native.torch_dtype_code(v.dtype), _native.DEVICE,
                    _native.current_stream_ptr(self.device),
                )
            )
            torch.cuda.current_stream(self.device).synchronize()  # `v` may be freed by the caller right after
            return
        if isinstance(vectors, torch.Tensor):
            vectors = vectors.numpy() if vectors.dtype != torch.bfloat16 else vectors.float().numpy()
        v = np.ascontiguousarray(vectors)
        if v.dtype not in (np.float16, np.float32):
            v = v.astype(np.float32)
        if v.ndim != 2 or v.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}] vectors, got {v.shape}")
        with torch.cuda.device(self.device):
            _native.check(
                self._lib.vodhip_index_add(
                    self._h, v.ctypes.data, v.shape[0], _native.numpy_dtype_code(v.dtype), _native.HOST,
                    _native.current_stream_ptr(self.device),
                )
            )

    def stored_rows(self, begin: int = 0, n: int | None = None) -> torch.Tensor:
        """Copy of the stored (rounded) rows [begin, begin+n) as a device tensor -- tests / persistence."""
        n = self.ntotal - begin if n is None else int(n)
        out = torch.empty((n, self.dim), dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self._lib.vodhip_index_get_rows(self._h, int(begin), n, out.data_ptr(), _native.DEVICE,
                                                _native.current_stream_ptr(self.device))
            )
        return out

    def stored_rows_f32(self, begin: int = 0, n: int | None = None) -> torch.Tensor:
        """Copy of the float32 rows [begin, begin+n) of an `exact_f32` store (what was added, bit for bit)."""
        n = self.ntotal - begin if n is None else int(n)
        out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self._lib.vodhip_index_get_rows_f32(self._h, int(begin), n, out.data_ptr(), _native.DEVICE,
                                                    _native.current_stream_ptr(self.device))
            )
        return out

    # -- subset filter ---------------------------------------------------------------
    def set_row_labels(self, labels: np.ndarray | torch.Tensor | None) -> None:
        """Attach an int32 subset label to every stored row (None clears).  Needed before `search(..., subset=...)`."""
        with torch.cuda.device(self.device):
            if labels is None:
                _native.check(self._lib.vodhip_index_set_row_labels(self._h, None, 0, _native.HOST, None))
                return
            if isinstance(labels, torch.Tensor) and labels.is_cuda:
                lab = labels.to(torch.int32).contiguous()
                _native.check(self._lib.vodhip_index_set_row_labels(self._h, lab.data_ptr(), lab.numel(), _native.DEVICE,
                                                                    _native.current_stream_ptr(self.device)))
                return
            lab = np.ascontiguousarray(np.asarray(labels), dtype=np.int32)
            _native.check(self._lib.vodhip_index_set_row_labels(self._h, lab.ctypes.data, lab.size, _native.HOST,
                                                                _native.current_stream_ptr(self.device)))

    # -- persistence (the store itself is the on-disk format, no faiss file round trip) -----------
    def save(self, path, chunk: int = 1 << 20) -> None:
        """Write the stored rows (as stored: fp16, or bf16 widened to fp32; the float32 rows of an `exact_f32` store - what
        `faiss.write_index` persists for the reference, factory.py:167) to a `.npy`, slice by slice."""
        n = self.ntotal
        np_dtype = np.float16 if (self.dtype == torch.float16 and not self.exact_f32) else np.float32
        out = np.lib.format.open_memmap(path, mode="w+", dtype=np_dtype, shape=(n, self.dim))
        for lo in range(0, n, chunk):
            if self.exact_f32:
                rows = self.stored_rows_f32(lo, min(chunk, n - lo))
                out[lo : lo + rows.shape[0]] = rows.cpu().numpy()
                continue
            rows = self.stored_rows(lo, min(chunk, n - lo))
            out[lo : lo + rows.shape[0]] = (rows if self.dtype == torch.float16 else rows.float()).cpu().numpy()
        out.flush()
        del out

    @classmethod
    def load(cls, path, dtype: torch.dtype = torch.float16, device: int | torch.device = 0, capacity: int | None = None,
             chunk: int = 1 << 18, exact_f32: bool = False) -> "HipFlatIndex":
        """Build an index from a 2-D float16/float32 `.npy` (memory-mapped, streamed to HBM in slices)."""
        arr = np.load(path, mmap_mode="r", allow_pickle=False)
        if arr.ndim != 2:
            raise ValueError(f"expected a 2-D vector file, got shape {arr.shape}")
        ix = cls(arr.shape[1], max(capacity or arr.shape[0], 1), dtype=dtype, device=device, exact_f32=exact_f32)
        if arr.flags.c_contiguous:
            ix.add(arr)  # ONE call: the library pipelines page-cache reads, pinned staging and DMA itself (64 MB slices)
        else:
            for lo in range(0, arr.shape[0], chunk):
                ix.add(np.ascontiguousarray(arr[lo : lo + chunk]))
        return ix

    # -- search ------------------------------------------------------------------------------------
    def set_param(self, key: str, value: int) -> None:
        _native.check(self._lib.vodhip_index_set_param(self._h, key.encode(), int(value)))

    def get_stat(self, key: str) -> int:
        out = ctypes.c_int64()
        _native.check(self._lib.vodhip_index_get_stat(self._h, key.encode(), ctypes.byref(out)))
        return out.value

    def _prep_queries(self, queries) -> torch.Tensor:
        if isinstance(queries, np.ndarray):
            q = np.ascontiguousarray(queries)
            if q.dtype not in (np.float16, np.float32):
                q = q.astype(np.float32)
            queries = torch.from_numpy(q).to(self.device, non_blocking=False)
        q = queries
        if q.device != self.device:
            q = q.to(self.device)
        if q.dtype not in (torch.float16, torch.bfloat16, torch.float32):
            q = q.float()
        q = q.contiguous()
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] queries, got {tuple(q.shape)}")
        return q

    def search_async(self, queries, k: int, id_base: int = 0, out: tuple[torch.Tensor, torch.Tensor] | None = None,
                     subset: np.ndarray | torch.Tensor | None = None):
        """Enqueue a search on the current stream; call `finish()` before trusting the outputs.

        `subset`: optional int32 [nq, S] allowed row labels per query (-1 = empty slot, all -1 = unrestricted)."""
        q = self._prep_queries(queries)
        nq = q.shape[0]
        if subset is not None:
            sub = subset if isinstance(subset, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(subset))
            sub = sub.to(self.device, torch.int32).contiguous()
            if sub.ndim != 2 or sub.shape[0] != nq:
                raise ValueError(f"expected subset labels of shape [{nq}, S], got {tuple(sub.shape)}")
            self._keep_subset = sub
            _native.check(self._lib.vodhip_index_set_query_labels(self._h, sub.data_ptr(), int(sub.shape[1])))
        else:
            self._keep_subset = None
            _native.check(self._lib.vodhip_index_set_query_labels(self._h, None, 0))
        if out is None:
            scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            ids = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        else:
            scores, ids = out
        _native.check(
            self._lib.vodhip_index_search_async(
                self._h, q.data_ptr(), _native.torch_dtype_code(q.dtype), nq, int(k), int(id_base),
                scores.data_ptr(), ids.data_ptr(), _native.current_stream_ptr(self.device),
            )
        )
        # only a search the library accepted is tracked (a refused one synchronised the stream before returning);
        # its buffers stay alive until its finish()
        self._keep.append((q, self._keep_subset, scores, ids))
        return scores, ids

    def finish(self) -> None:
        """Complete the OLDEST enqueued search (up to 4 may be in flight): waits for it alone and, if a candidate list
        overflowed, runs recovery passes seeded from its incomplete result.  Pipelined searches need their own `out` buffers if their
        results are read after younger searches were enqueued."""
        _native.check(self._lib.vodhip_index_search_finish(self._h, _native.current_stream_ptr(self.device)))
        if self._keep:
            self._keep.popleft()

    def search(self, queries, k: int, id_base: int = 0, out=None, subset=None) -> tuple[torch.Tensor, torch.Tensor]:
        """Exact top-k by inner product: (scores f32 [nq,k] desc, ids i64 [nq,k]); ties -> smaller id; pad -inf/-1."""
        with torch.cuda.device(self.device):
            res = self.search_async(queries, k, id_base, out, subset)
            self.finish()
        return res


def merge_topk(scores: torch.Tensor, ids: torch.Tensor, k_out: int | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """Merge per-shard top-k lists [n_shards, nq, k] (global ids, pads -1) into [nq, k_out] on the GPU."""
    lib = _native.load_library()
    if scores.ndim != 3 or ids.shape != scores.shape:
        raise ValueError("expected scores/ids of shape [n_shards, nq, k]")
    n_shards, nq, k = scores.shape
    k_out = k if k_out is None else int(k_out)
    scores = scores.contiguous().float()
    ids = ids.contiguous().long()
    out_s = torch.empty((nq, k_out), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k_out), dtype=torch.int64, device=scores.device)
    with torch.cuda.device(scores.device):
        _native.check(
            lib.vodhip_merge_topk(scores.data_ptr(), ids.data_ptr(), n_shards, nq, k, k_out, out_s.data_ptr(),
                                  out_i.data_ptr(), _native.current_stream_ptr(scores.device))
        )
    return out_s, out_i


class PackedTopk:
    """One contiguous per-rank record `[scores f32 nq*k | ids i64 nq*k]` so that the multi-GPU exchange is ONE all-gather.

    `scores` / `ids` are views into `buffer` (uint8); `merge_gathered` merges `world` such records laid end to end.
    """

    def __init__(self, nq: int, k: int, device: torch.device):
        self.nq, self.k = int(nq), int(k)
        n = self.nq * self.k
        self._s_bytes = (n * 4 + 7) // 8 * 8  # keep the id block 8-byte aligned
        self.nbytes = self._s_bytes + n * 8
        self.buffer = torch.empty((self.nbytes,), dtype=torch.uint8, device=device)
        self.scores = self.buffer[: n * 4].view(torch.float32).view(self.nq, self.k)
        self.ids = self.buffer[self._s_bytes :].view(torch.int64).view(self.nq, self.k)

    def merge_gathered(self, gathered: torch.Tensor, world: int, k_out: int | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """`gathered`: uint8 [world * nbytes] (rank-major).  Returns merged (scores [nq,k_out], ids [nq,k_out])."""
        lib = _native.load_library()
        k_out = self.k if k_out is None else int(k_out)
        dev = gathered.device
        out_s = torch.empty((self.nq, k_out), dtype=torch.float32, device=dev)
        out_i = torch.empty((self.nq, k_out), dtype=torch.int64, device=dev)
        base = gathered.data_ptr()
        with torch.cuda.device(dev):
            _native.check(
                lib.vodhip_merge_topk_strided(base, self.nbytes // 4, base + self._s_bytes, self.nbytes // 8, int(world), self.nq,
                                              self.k, k_out, out_s.data_ptr(), out_i.data_ptr(), _native.current_stream_ptr(dev))
            )
        return out_s, out_i


class HipNodeIndex:
    """The row-sharded index of one node in ONE process (`vodhip_node_index_*`): shard g on `devices[g]` holds a contiguous row
    range, a search runs on every device at once and the per-shard top-k lists are merged on `devices[0]`.

    Counterpart of faiss `index_cpu_to_all_gpus(..., shard=True)` (/root/reference/src/vod_search/faiss_search/server.py:51-54).
    The collective variant (one process per GPU, one RCCL all-gather) is `vod_amd.distributed.ShardedFlatIndex`; this one is what a
    C / C++ consumer of the library gets, and what a single-process Python host can use without `torch.distributed`.
    """

    def __init__(self, dim: int, capacity: int, devices: list[int], dtype: torch.dtype = torch.float16, exact_f32: bool = False):
        self._lib = _native.load_library()
        if not torch.cuda.is_available():
            raise _native.NativeLibraryError("HipNodeIndex needs a ROCm device (torch.cuda.is_available() is False)")
        if dtype not in (torch.float16, torch.bfloat16):
            raise TypeError("store dtype must be torch.float16 or torch.bfloat16")
        if not devices:
            raise ValueError("devices must list at least one device ordinal")
        self.dim, self.capacity, self.dtype, self.devices = int(dim), int(capacity), dtype, [int(d) for d in devices]
        self.device = torch.device("cuda", self.devices[0])
        self.exact_f32 = bool(exact_f32)  # every shard keeps its float32 rows (HipFlatIndex): the merged result is the float32 brute force
        handle = ctypes.c_void_p()
        arr = (ctypes.c_int32 * len(self.devices))(*self.devices)
        code = _native.torch_dtype_code(dtype) | (_native.EXACT_F32 if self.exact_f32 else 0)
        _native.check(self._lib.vodhip_node_index_create(len(self.devices), arr, self.dim, code, self.capacity, ctypes.byref(handle)))
        self._h = handle

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vodhip_node_index_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self) -> "HipNodeIndex":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    @property
    def ntotal(self) -> int:
        out = ctypes.c_int64()
        _native.check(self._lib.vodhip_node_index_ntotal(self._h, ctypes.byref(out)))
        return out.value

    def reset(self) -> None:
        _native.check(self._lib.vodhip_node_index_reset(self._h))

    def set_param(self, key: str, value: int) -> None:
        _native.check(self._lib.vodhip_node_index_set_param(self._h, key.encode(), int(value)))

    def get_stat(self, key: str) -> int:
        """Node-level stats of the last search with `set_param("profile", 1)`: "last_merge_ns", "last_copy_ns_max" (HIP events)."""
        out = ctypes.c_int64()
        _native.check(self._lib.vodhip_node_index_get_stat(self._h, key.encode(), ctypes.byref(out)))
        return out.value

    def shard_stat(self, g: int, key: str) -> int:
        """`vodhip_index_get_stat` of shard g's own handle (e.g. "last_filter_ns" with profiling on)."""
        out = ctypes.c_int64()
        _native.check(self._lib.vodhip_index_get_stat(ctypes.c_void_p(self.shard(g)[0]), key.encode(), ctypes.byref(out)))
        return out.value

    def peer_access(self) -> list[int]:
        """Per shard: 2 = on `devices[0]`, 1 = direct peer copies with it, 0 = staged through pinned host memory (no peer access, or
        `set_param("host_staging", 1)`)."""
        out = (ctypes.c_int32 * len(self.devices))()
        _native.check(min(0, self._lib.vodhip_node_index_peer_access(self._h, out, len(self.devices))))
        return list(out)

    def shard(self, g: int) -> tuple[int, int, int]:
        """(raw `vodhip_index_t*` of shard g, its id offset, its device) - for stats / params of one shard."""
        h, base, dev = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int32()
        _native.check(self._lib.vodhip_node_index_shard(self._h, int(g), ctypes.byref(h), ctypes.byref(base), ctypes.byref(dev)))
        return h.value, base.value, dev.value

    def add(self, vectors: np.ndarray) -> None:
        """Append host rows (float32 / float16 NumPy, [n, dim]); the shards the batch straddles ingest concurrently."""
        v = np.ascontiguousarray(vectors)
        if v.ndim != 2 or v.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}] vectors, got {tuple(v.shape)}")
        if v.dtype not in (np.float32, np.float16):
            v = v.astype(np.float32)
        code = 2 if v.dtype == np.float32 else 0
        _native.check(self._lib.vodhip_node_index_add(self._h, v.ctypes.data, v.shape[0], code))

    def set_row_labels(self, labels: np.ndarray | None) -> None:
        """One int32 subset label per stored row (global row order; None clears): see `HipFlatIndex.set_row_labels`."""
        if labels is None:
            _native.check(self._lib.vodhip_node_index_set_row_labels(self._h, None, 0))
            return
        lab = np.ascontiguousarray(labels, dtype=np.int32)
        _native.check(self._lib.vodhip_node_index_set_row_labels(self._h, lab.ctypes.data, lab.size))

    def _set_subset(self, subset: np.ndarray | None, nq: int):
        if subset is None:
            _native.check(self._lib.vodhip_node_index_set_query_labels(self._h, None, 0, 0))
            return None
        sub = np.ascontiguousarray(subset, dtype=np.int32)
        if sub.ndim != 2 or sub.shape[0] != nq:
            raise ValueError(f"subset must be int32 [nq, n_per_query], got {tuple(sub.shape)}")
        _native.check(self._lib.vodhip_node_index_set_query_labels(self._h, sub.ctypes.data, int(sub.shape[1]), 0))
        return sub  # kept alive by the caller's frame until the search has returned

    def search_async(self, queries: np.ndarray | torch.Tensor, k: int, subset: np.ndarray | None = None) -> None:
        """The first half of `search` (round 6): the batch is replicated and every shard's search enqueued; `finish()` completes the OLDEST
        pending search and returns its (scores, ids).  Up to two searches may be pending, so a caller that runs one batch ahead keeps every
        device busy while the host enqueues the next batch's launches on all shards.  Inputs / outputs as in `search`."""
        keep = self._set_subset(subset, int(queries.shape[0]))
        if isinstance(queries, torch.Tensor):
            if queries.device != self.device:
                raise ValueError(f"queries live on {queries.device}, the merge device is {self.device}")
            q = queries.contiguous()
            if q.dtype not in (torch.float16, torch.bfloat16, torch.float32):
                q = q.float()
            if q.ndim != 2 or q.shape[1] != self.dim:
                raise ValueError(f"expected [nq, {self.dim}] queries, got {tuple(q.shape)}")
            out_s = torch.empty((q.shape[0], k), dtype=torch.float32, device=self.device)
            out_i = torch.empty((q.shape[0], k), dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device):
                _native.check(self._lib.vodhip_node_index_search_async(self._h, q.data_ptr(), _native.torch_dtype_code(q.dtype), q.shape[0], int(k), 1,
                                                                       out_s.data_ptr(), out_i.data_ptr(), _native.current_stream_ptr(self.device)))
        else:
            q = np.ascontiguousarray(queries)
            if q.dtype not in (np.float32, np.float16):
                q = q.astype(np.float32)
            if q.ndim != 2 or q.shape[1] != self.dim:
                raise ValueError(f"expected [nq, {self.dim}] queries, got {tuple(q.shape)}")
            out_s = np.empty((q.shape[0], k), dtype=np.float32)
            out_i = np.empty((q.shape[0], k), dtype=np.int64)
            _native.check(self._lib.vodhip_node_index_search_async(self._h, q.ctypes.data, 2 if q.dtype == np.float32 else 0, q.shape[0], int(k), 0,
                                                                   out_s.ctypes.data, out_i.ctypes.data, None))
        if not hasattr(self, "_pending"):
            self._pending = []
        self._pending.append((q, keep, out_s, out_i))  # (queries, labels and outputs stay alive until the finish)

    def finish(self):
        """Complete the oldest pending `search_async`; returns its (scores, ids)."""
        if not getattr(self, "_pending", None):
            raise RuntimeError("no search is pending on this node index")
        with torch.cuda.device(self.device):
            _native.check(self._lib.vodhip_node_index_search_finish(self._h))
        _q, _keep, out_s, out_i = self._pending.pop(0)
        return out_s, out_i

    def search(self, queries: np.ndarray | torch.Tensor, k: int, subset: np.ndarray | None = None):
        """NumPy in -> NumPy out (host buffers, synchronous); a tensor on `devices[0]` in -> tensors there, complete on the current stream.
        `subset`: int32 [nq, n_per_query] allowed row labels per query (-1 = empty slot; host array)."""
        _keep = self._set_subset(subset, int(queries.shape[0]))  # noqa: F841
        if isinstance(queries, torch.Tensor):
            if queries.device != self.device:
                raise ValueError(f"queries live on {queries.device}, the merge device is {self.device}")
            q = queries.contiguous()
            if q.dtype not in (torch.float16, torch.bfloat16, torch.float32):
                q = q.float()
            if q.ndim != 2 or q.shape[1] != self.dim:
                raise ValueError(f"expected [nq, {self.dim}] queries, got {tuple(q.shape)}")
            out_s = torch.empty((q.shape[0], k), dtype=torch.float32, device=self.device)
            out_i = torch.empty((q.shape[0], k), dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device):
                _native.check(self._lib.vodhip_node_index_search(self._h, q.data_ptr(), _native.torch_dtype_code(q.dtype), q.shape[0], int(k), 1,
                                                                 out_s.data_ptr(), out_i.data_ptr(), _native.current_stream_ptr(self.device)))
            return out_s, out_i
        q = np.ascontiguousarray(queries)
        if q.dtype not in (np.float32, np.float16):
            q = q.astype(np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] queries, got {tuple(q.shape)}")
        out_s = np.empty((q.shape[0], k), dtype=np.float32)
        out_i = np.empty((q.shape[0], k), dtype=np.int64)
        _native.check(self._lib.vodhip_node_index_search(self._h, q.ctypes.data, 2 if q.dtype == np.float32 else 0, q.shape[0], int(k), 0,
                                                         out_s.ctypes.data, out_i.ctypes.data, None))
        return out_s, out_i

"""HTTP server that owns the GPU-resident index (counterpart of
/root/reference/src/vod_search/faiss_search/server.py:39-98; wire models :30-79 of models.py).

Routes and payloads are the reference's:
  GET  /             -> "OK" | "ERROR: Index is empty"                 (server.py:57-65)
  POST /search       {"vectors": [[...]], "top_k": k} -> {"scores": [[...]], "indices": [[...]]}   (:68-73)
  POST /fast-search  {"vectors": b64(npy), "top_k": k} -> {"scores": b64(npy f32), "indices": b64(npy i64)}  (:76-91)
Unknown fields are rejected (`extra = "forbid"`), a non-2-D query is an error, failures come back as
HTTP 500 with the formatted trace in `detail`.

`create_app(engine)` takes any object with `.ntotal` and `.search(np.ndarray[nq, d], k) -> (scores, ids)`;
`main()` -- the only production entry point -- builds the HIP engine and fails loudly without a GPU or
without libvodhip.so (there is no CPU engine in this package).
"""
from __future__ import annotations

import argparse
import re
import threading
import traceback

import numpy as np
import pydantic
from fastapi import FastAPI, Request, Response
from fastapi.concurrency import run_in_threadpool

from vod_amd import io


class SearchQuery(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="forbid")
    vectors: list = pydantic.Field(..., description="A batch of vectors: list[list[float]].")
    top_k: int = 3


class FastSearchQuery(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="forbid")
    vectors: str = pydantic.Field(..., description="A batch of vectors, np.save bytes in urlsafe base64.")
    top_k: int = 3
    # extension (absent in the reference's model, whose faiss client drops `subset_ids`): per-query allowed subset ids
    subset_ids: None | list[list[str]] = None


class SearchResponse(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="forbid")
    scores: list
    indices: list


class FastSearchResponse(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="forbid")
    scores: str
    indices: str


class _CallbackEngine:
    """What the batcher's callback engine calls: `search(float32 [nq, d], k)` and - for subset-filtered batches - the engine's
    `search_encoded(q, k, subset=int32 [nq, S])` (the labels were encoded by `engine.encode_subset` before they entered the batcher)."""

    def __init__(self, engine):
        self.engine = engine

    def search(self, q: np.ndarray, k: int, subset: np.ndarray | None = None):
        if subset is None:
            return self.engine.search(q, k)
        if not hasattr(self.engine, "search_encoded"):
            raise ValueError("this engine cannot filter by subset id")
        return self.engine.search_encoded(q, k, subset)


class Endpoints:
    """Routes, validation and error mapping of the search service, independent of the HTTP shell around them.

    `handle(method, path, query, body)` -> (status, content type, payload bytes-like, extra headers).  Every shell - libvodhip's native
    front (production: it answers the plain hot requests itself and hands everything else here) and the FastAPI app of `create_app` (ASGI hosting, tests) - answers with exactly what this returns, so
    the contract is the reference's whatever carries it:
      422 + `{"detail": [...]}` for a document that fails the pydantic model (`extra="forbid"`, wrong types: models.py:43-79),
      500 + `{"detail": <trace>}` for a failing search (server.py:89-91), 404 / 405 for unknown routes.
    Searches go through the library's request fusion (`vodhip_batcher`, include/vodhip.h H6): concurrent requests share corpus scans,
    a lone request runs at once (the reference's single uvicorn worker runs faiss for one request at a time, server.py:69,78,98).
    `micro_batch_wait_ms > 0` additionally makes every batch wait that long for company (round 3's fixed window; not needed)."""

    def __init__(self, engine, micro_batch_wait_ms: float = 0.0):
        from vod_amd.search.native import NativeBatcher

        self.engine = engine
        self._window_us = int(micro_batch_wait_ms * 1e3)
        self._make = NativeBatcher
        self._lock = threading.Lock()
        self._generic: dict[int, "NativeBatcher"] = {}  # engines without a native handle: one callback batcher per query dimension
        # `HipEngine` / `NodeHipEngine` own the batcher that owns their index's search path; any other engine (the multi-process group,
        # test doubles) is called back by one
        self.batcher = getattr(engine, "batcher", None)
        if self.batcher is None and getattr(engine, "dim", None):
            self.batcher = self._batcher_for(int(engine.dim))
        if self.batcher is not None and self._window_us > 0:
            self.batcher.set_param("window_us", self._window_us)

    def _batcher_for(self, dim: int):
        if self.batcher is not None:
            if dim != self.batcher.dim:
                raise ValueError(f"query dimension {dim} != index dimension {self.batcher.dim}")
            return self.batcher
        with self._lock:
            b = self._generic.get(dim)
            if b is None:
                b = self._generic[dim] = self._make(engine=_CallbackEngine(self.engine), dim=dim)
                if self._window_us > 0:
                    b.set_param("window_us", self._window_us)
            return b

    def close(self) -> None:
        for b in self._generic.values():
            b.close()
        self._generic.clear()

    # -- the search itself --------------------------------------------------------------------------------------------
    def search(self, query_vec: np.ndarray, top_k: int, subset_ids=None, client: int = 0) -> tuple[np.ndarray, np.ndarray]:
        if query_vec.ndim != 2:
            raise ValueError(f"Expected 2D array, got {query_vec.ndim}D array")
        subset = None
        if subset_ids is not None and any(len(s) for s in subset_ids):
            if len(subset_ids) != len(query_vec):
                raise ValueError("`subset_ids` must have one list per query")
            subset = self.engine.encode_subset(subset_ids)
        scores, indices = self._batcher_for(int(query_vec.shape[1])).search(query_vec, top_k, subset=subset, client=client)
        return np.asarray(scores, dtype=np.float32), np.asarray(indices, dtype=np.int64)

    # -- routes -------------------------------------------------------------------------------------------------------
    def health(self) -> str:
        return "ERROR: Index is empty" if self.engine.ntotal == 0 else "OK"

    def legacy_search(self, document: dict, client: int = 0) -> dict:
        """POST /search (server.py:68-73): JSON lists in, JSON lists out."""
        query = SearchQuery(**document)
        scores, indices = self.search(np.asarray(query.vectors, dtype=np.float32), query.top_k, client=client)
        rows = scores.astype(np.float64).tolist()  # (float(v) of every float32, in C)
        if np.isneginf(scores).any():  # pads of a store with fewer than top_k rows: JSON has no -inf literal in strict mode -> null
            rows = [[(None if v == -np.inf else v) for v in r] for r in rows]
        return SearchResponse(scores=rows, indices=indices.tolist()).model_dump()

    def fast_search(self, body, client: int = 0) -> bytearray:
        """POST /fast-search (server.py:76-91).  The multi-megabyte base64 field is located in the body, validated as the
        `str` the model asks for, and decoded where it lies; the reply's base64 text is written straight into its buffer."""
        small, spans = io.find_payload_spans(body, ("vectors",))
        query = FastSearchQuery(**small)  # raises pydantic.ValidationError -> 422
        try:
            vectors = io.deserialize_np_array_span(body, *spans["vectors"]) if "vectors" in spans else io.deserialize_np_array(query.vectors)
            scores, indices = self.search(vectors, query.top_k, query.subset_ids, client=client)
            return io.json_body_with_arrays({"scores": scores, "indices": indices})
        except Exception as exc:
            raise _SearchFailed(traceback.format_exc()) from exc

    def raw_search(self, body, top_k: int, client: int = 0) -> tuple[bytes, dict]:
        """POST /raw-search?top_k=K (not in the reference): body = raw `.npy` bytes of the [nq, d] queries (float32
        or float16), reply = raw bytes `scores float32 [nq, k]` followed by `indices int64 [nq, k]`; shapes in the headers."""
        try:
            scores, indices = self.search(io.load_npy_view(body), top_k, client=client)
            scores, indices = np.ascontiguousarray(scores), np.ascontiguousarray(indices)
            payload = bytearray(scores.nbytes + indices.nbytes)
            if scores.nbytes:
                np.frombuffer(payload, dtype=np.uint8, count=scores.nbytes)[:] = scores.reshape(-1).view(np.uint8)
                np.frombuffer(payload, dtype=np.uint8, offset=scores.nbytes)[:] = indices.reshape(-1).view(np.uint8)
            return payload, {"x-nq": str(scores.shape[0]), "x-k": str(scores.shape[1])}
        except Exception as exc:
            raise _SearchFailed(traceback.format_exc()) from exc

    def handle(self, method: str, path: str, query: dict, body, client: int = 0) -> tuple[int, str, "bytes | bytearray", dict]:
        import json

        js = "application/json"
        try:
            if path == "/":
                if method != "GET":
                    return 405, js, b'{"detail":"Method Not Allowed"}', {}
                return 200, js, json.dumps(self.health()).encode(), {}
            if path == "/stats" and method == "GET":
                return 200, js, json.dumps({} if self.batcher is None else self.batcher.stats()).encode(), {}
            if path not in ("/search", "/fast-search", "/raw-search"):
                return 404, js, b'{"detail":"Not Found"}', {}
            if method != "POST":
                return 405, js, b'{"detail":"Method Not Allowed"}', {}
            if path == "/fast-search":
                return 200, js, self.fast_search(body, client), {}
            if path == "/raw-search":
                try:
                    top_k = int(query.get("top_k", 3))
                except ValueError:
                    return 422, js, b'{"detail":"top_k must be an integer"}', {}
                payload, headers = self.raw_search(body, top_k, client)
                return 200, "application/octet-stream", payload, headers
            try:
                document = json.loads(bytes(body))
                if not isinstance(document, dict):
                    raise ValueError("expected a JSON object")
            except ValueError as exc:
                return 422, js, json.dumps({"detail": f"invalid request body: {exc}"}).encode(), {}
            try:
                return 200, js, json.dumps(self.legacy_search(document, client)).encode(), {}
            except pydantic.ValidationError:
                raise
            except Exception:
                return 500, js, json.dumps({"detail": traceback.format_exc()}).encode(), {}
        except pydantic.ValidationError as exc:
            return 422, js, json.dumps({"detail": exc.errors(include_url=False, include_input=False)}, default=str).encode(), {}
        except _SearchFailed as exc:
            return 500, js, json.dumps({"detail": str(exc)}).encode(), {}
        except ValueError as exc:  # a body that is not a JSON object
            return 422, js, json.dumps({"detail": f"invalid request body: {exc}"}).encode(), {}


class _SearchFailed(RuntimeError):
    """A request that parsed but whose search (or payload decoding) raised: HTTP 500 with the formatted trace."""


def create_app(engine, micro_batch_wait_ms: float = 0.0) -> FastAPI:
    """The same service as an ASGI app (FastAPI): `Endpoints` behind starlette's request / response objects."""
    app = FastAPI()
    endpoints = Endpoints(engine, micro_batch_wait_ms)
    app.state.endpoints = endpoints

    def _respond(result) -> Response:
        status, ctype, payload, headers = result
        return Response(content=bytes(payload) if isinstance(payload, bytearray) else payload, status_code=status, media_type=ctype, headers=headers)

    @app.get("/")
    def health_check() -> str:
        return endpoints.health()

    @app.post("/search")
    async def search(request: Request) -> Response:
        return _respond(await run_in_threadpool(endpoints.handle, "POST", "/search", {}, await request.body()))

    @app.post("/fast-search", response_model=FastSearchResponse)
    async def fast_search(request: Request) -> Response:
        return _respond(await run_in_threadpool(endpoints.handle, "POST", "/fast-search", {}, await request.body()))

    @app.post("/raw-search")
    async def raw_search(request: Request) -> Response:
        return _respond(await run_in_threadpool(endpoints.handle, "POST", "/raw-search", dict(request.query_params), await request.body()))

    return app


def synthetic_rows(torch, dev, lo: int, hi: int, dim: int, seed: int, chunk: int = 250_000):
    """Rows [lo, hi) of the `synthetic:` store, chunk by chunk (chunk c is seeded seed + c: any row range of any process agrees)."""
    for c in range(lo // chunk, (hi + chunk - 1) // chunk):
        g = torch.Generator(device=dev).manual_seed(seed + c)
        rows = torch.randn((chunk, dim), generator=g, device=dev, dtype=torch.float32)
        yield rows[max(lo - c * chunk, 0) : min(hi - c * chunk, chunk)]


class HipEngine:
    """The production engine: a `HipFlatIndex` fed from a vector file, searched on the GPU.

    `row_range=(lo, hi)`: hold only rows [lo, hi) of the store (one shard of a multi-GPU group); ids stay global."""

    def __init__(self, vectors_path: str, dtype: str = "float16", device: int = 0, subset_ids_path: str | None = None,
                 row_range: tuple[int, int] | None = None, exact_f32: bool = False):
        import torch

        self.exact_f32 = bool(exact_f32)  # keep the float32 rows too: results of a float32 brute force (HipFlatIndex)

        from vod_amd import store
        from vod_amd.index import HipFlatIndex

        self._torch = torch
        if str(vectors_path).startswith("synthetic:"):
            # measurement aid (tools/bench_http_load.py): `synthetic:ROWSxDIM[:SEED]` = N(0, 1) rows generated on the device, chunk
            # by chunk - the boundary can be measured in front of a 10 M-row store without writing 15 GB to disk first
            return self._init_synthetic(str(vectors_path), dtype, device, row_range)
        vectors = store.open_vectors(vectors_path)
        n, d = vectors.shape
        lo, hi = (0, n) if row_range is None else (int(row_range[0]), int(row_range[1]))
        self.n_store, self.row_lo, self.row_hi = n, lo, hi
        self.index = HipFlatIndex(d, max(hi - lo, 1), dtype=getattr(torch, dtype), device=device, exact_f32=self.exact_f32)
        step = 262144
        if hasattr(vectors, "iter_row_blocks") and row_range is None:
            # zarr store: blocks aligned to its chunk grid, each chunk decoded once, on a thread pool running ahead of the ingest
            for _lo, rows in vectors.iter_row_blocks():
                self.index.add(rows if rows.dtype != np.float64 else rows.astype(np.float32))
        elif isinstance(vectors, np.ndarray) and vectors.dtype != np.float64 and vectors[lo:hi].flags.c_contiguous:
            # .npy memory map: ONE call - the library overlaps the page-cache reads (CPU threads -> pinned staging), the DMA and
            # the on-device rounding to fp16 / bf16 over 64 MB slices
            self.index.add(vectors[lo:hi])
        else:
            for b0 in range(lo, hi, step):
                rows = np.ascontiguousarray(vectors[b0 : min(hi, b0 + step)])
                self.index.add(rows if rows.dtype != np.float64 else rows.astype(np.float32))
        self.vocab: dict[str, int] = {}
        if subset_ids_path:  # one subset id (string) per stored row -> int32 labels on the device
            ids = np.load(subset_ids_path, allow_pickle=False)
            if len(ids) != n:
                raise ValueError(f"{subset_ids_path}: {len(ids)} subset ids for {n} vectors")
            uniq, codes = np.unique(ids.astype(str), return_inverse=True)  # the vocabulary is global: same codes on every shard
            self.vocab = {str(u): i for i, u in enumerate(uniq)}
            self.index.set_row_labels(codes[lo:hi].astype(np.int32))

    def _init_synthetic(self, spec: str, dtype: str, device: int, row_range) -> None:
        import torch

        from vod_amd.index import HipFlatIndex

        shape, _, seed = spec[len("synthetic:"):].partition(":")
        n, d = (int(v) for v in shape.lower().split("x"))
        lo, hi = (0, n) if row_range is None else (int(row_range[0]), int(row_range[1]))
        self.n_store, self.row_lo, self.row_hi = n, lo, hi
        self.index = HipFlatIndex(d, max(hi - lo, 1), dtype=getattr(torch, dtype), device=device, exact_f32=self.exact_f32)
        self.vocab = {}
        for rows in synthetic_rows(torch, torch.device("cuda", device), lo, hi, d, int(seed or 0)):
            self.index.add(rows if self.exact_f32 else rows.to(getattr(torch, dtype)))

    @property
    def ntotal(self) -> int:
        return self.index.ntotal

    def encode_subset(self, subset_ids: list[list[str]] | None) -> np.ndarray | None:
        if subset_ids is None:
            return None
        if not self.vocab:
            raise ValueError("the server was started without --subset-ids-path: cannot filter by subset id")
        width = max(1, max(len(s) for s in subset_ids))
        subset = np.full((len(subset_ids), width), -1, dtype=np.int32)
        for r, names in enumerate(subset_ids):
            # an unknown subset id matches no row: -2 keeps the query restricted (and empty) instead of unrestricted
            subset[r, : len(names)] = [self.vocab.get(str(nm), -2) for nm in names]
        return subset

    # -- searching -------------------------------------------------------------------------------------------------------------
    # The library's batcher owns the index's search path: callers from any number of threads are fused into shared scans, up to two
    # batches are enqueued back to back on the batcher's stream, the final select kernel of a search writes its result rows straight
    # into device-visible host memory and every caller copies its own rows out (vodhip_serve.hip).  Round 3 did this here in Python
    # (tickets over the library's FIFO, a ring of pinned torch buffers) - and raced with itself (advisor, round 3).
    @property
    def dim(self) -> int:
        return self.index.dim

    @property
    def batcher(self):
        b = getattr(self, "_batcher", None)
        if b is None:
            with HipEngine._batcher_create:
                b = getattr(self, "_batcher", None)
                if b is None:
                    from vod_amd.search.native import NativeBatcher

                    b = self._batcher = NativeBatcher(index=self.index, dim=self.index.dim, id_base=self.row_lo)
        return b

    _batcher_create = threading.Lock()

    def search_encoded(self, query_vec: np.ndarray, top_k: int, subset: np.ndarray | None = None, client: int = 0):
        if query_vec.ndim != 2 or query_vec.shape[1] != self.index.dim:
            raise ValueError(f"query dimension {query_vec.shape[-1]} != index dimension {self.index.dim}")
        return self.batcher.search(query_vec, top_k, subset=subset, client=client)

    def search(self, query_vec: np.ndarray, top_k: int, subset_ids: list[list[str]] | None = None) -> tuple[np.ndarray, np.ndarray]:
        return self.search_encoded(query_vec, top_k, self.encode_subset(subset_ids))

    def close(self) -> None:
        b = getattr(self, "_batcher", None)
        if b is not None:
            b.close()
            self._batcher = None
        self.index.close()


class NodeHipEngine:
    """`--devices a,b,... --group-backend node`: ONE server process drives every GPU through the library's node index
    (`vodhip_node_index_*`: per-device row shards, peer copies of the per-shard top-k, merge on the first device) - no worker
    processes and no process group.  This is the shape of the reference's own server, whose single process holds
    `faiss.index_cpu_to_all_gpus(index, co)` with `co.shard = True` (/root/reference/src/vod_search/faiss_search/server.py:51-54).
    Same `.ntotal` / `.search` as `HipEngine`."""

    def __init__(self, vectors_path: str, devices: list[int], dtype: str = "float16", subset_ids_path: str | None = None,
                 exact_f32: bool = False):
        import torch

        from vod_amd import store
        from vod_amd.index import HipNodeIndex

        self.vocab: dict[str, int] = {}
        if str(vectors_path).startswith("synthetic:"):
            shape, _, seed = str(vectors_path)[len("synthetic:"):].partition(":")
            n, d = (int(v) for v in shape.lower().split("x"))
            self.index = HipNodeIndex(d, max(n, 1), devices, dtype=getattr(torch, dtype), exact_f32=exact_f32)
            for rows in synthetic_rows(torch, torch.device("cuda", devices[0]), 0, n, d, int(seed or 0)):
                self.index.add((rows if exact_f32 else rows.to(torch.float16)).cpu().numpy())
            self.n_store = n
            self._log_topology()
            return
        vectors = store.open_vectors(vectors_path)
        n, d = vectors.shape
        self.n_store = n
        self.index = HipNodeIndex(d, max(n, 1), devices, dtype=getattr(torch, dtype), exact_f32=exact_f32)
        if hasattr(vectors, "iter_row_blocks"):
            for _lo, rows in vectors.iter_row_blocks():
                self.index.add(rows)
        elif isinstance(vectors, np.ndarray) and vectors.dtype in (np.float32, np.float16) and vectors.flags.c_contiguous:
            self.index.add(vectors)  # one call: every shard ingests its row range concurrently
        else:
            for b0 in range(0, n, 262144):
                self.index.add(np.ascontiguousarray(vectors[b0 : min(n, b0 + 262144)]))
        if subset_ids_path:
            ids = np.load(subset_ids_path, allow_pickle=False)
            if len(ids) != n:
                raise ValueError(f"{subset_ids_path}: {len(ids)} subset ids for {n} vectors")
            uniq, codes = np.unique(ids.astype(str), return_inverse=True)
            self.vocab = {str(u): i for i, u in enumerate(uniq)}
            self.index.set_row_labels(codes.astype(np.int32))
        self._log_topology()

    def _log_topology(self) -> None:
        import logging

        how = {2: "on the merge device", 1: "direct peer copies", 0: "staged through pinned host memory (no peer access)"}
        for g, (dev, p) in enumerate(zip(self.index.devices, self.index.peer_access())):
            logging.getLogger(__name__).info("node index: shard %d on device %d: %s", g, dev, how.get(p, p))

    @property
    def ntotal(self) -> int:
        return self.index.ntotal

    encode_subset = HipEngine.encode_subset

    @property
    def dim(self) -> int:
        return self.index.dim

    @property
    def batcher(self):
        b = getattr(self, "_batcher", None)
        if b is None:
            with HipEngine._batcher_create:
                b = getattr(self, "_batcher", None)
                if b is None:
                    from vod_amd.search.native import NativeBatcher

                    b = self._batcher = NativeBatcher(node=self.index, dim=self.index.dim)
        return b

    def search_encoded(self, query_vec: np.ndarray, top_k: int, subset: np.ndarray | None = None, client: int = 0):
        if query_vec.ndim != 2 or query_vec.shape[1] != self.index.dim:
            raise ValueError(f"query dimension {query_vec.shape[-1]} != index dimension {self.index.dim}")
        return self.batcher.search(query_vec, top_k, subset=subset, client=client)

    def search(self, query_vec: np.ndarray, top_k: int, subset_ids: list[list[str]] | None = None) -> tuple[np.ndarray, np.ndarray]:
        return self.search_encoded(query_vec, top_k, self.encode_subset(subset_ids))

    def close(self) -> None:
        b = getattr(self, "_batcher", None)
        if b is not None:
            b.close()
            self._batcher = None
        self.index.close()


class GroupHipEngine:
    """Rank 0's engine of a multi-GPU group (`--devices`): same `.ntotal` / `.search` as `HipEngine`, but every search is
    broadcast to the N ranks, each searching its row shard on its own GPU, and merged (vod_amd.search.group)."""

    def __init__(self, local: HipEngine, dispatcher):
        self.local, self.dispatcher = local, dispatcher

    @property
    def ntotal(self) -> int:
        return self.local.n_store

    @property
    def dim(self) -> int:
        return self.local.index.dim

    def encode_subset(self, subset_ids):
        return self.local.encode_subset(subset_ids)

    def search_encoded(self, query_vec: np.ndarray, top_k: int, subset: np.ndarray | None = None) -> tuple[np.ndarray, np.ndarray]:
        # called by the batcher's callback engine, ONE fused batch at a time: the collective sequence of the group stays serial
        if query_vec.shape[1] != self.local.index.dim:
            raise ValueError(f"query dimension {query_vec.shape[1]} != index dimension {self.local.index.dim}")
        return self.dispatcher.search(query_vec, top_k, subset=subset)

    def search(self, query_vec: np.ndarray, top_k: int, subset_ids: list[list[str]] | None = None) -> tuple[np.ndarray, np.ndarray]:
        return self.search_encoded(query_vec, top_k, self.local.encode_subset(subset_ids))


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser()
    p.add_argument("--vectors-path", type=str, required=True)
    p.add_argument("--host", type=str, default="localhost")
    p.add_argument("--port", type=int, default=7678)
    p.add_argument("--logging-level", type=str, default="INFO")
    p.add_argument("--dtype", type=str, default="float16", choices=["float16", "bfloat16"])
    p.add_argument("--exact-f32", action="store_true",
                   help="keep the float32 rows next to the fp16 / bf16 scan copy and answer with the float32 brute-force result on the "
                        "unrounded vectors and queries - the reference's arithmetic (faiss IndexFlat holds float32, build.py:65-73); "
                        "+4 bytes per element of HBM, ~1-3 %% of search time")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--devices", type=str, default=None,
                   help="comma-separated GPU ids: the store is row-sharded over them, one worker process per GPU on an RCCL "
                        "group, rank 0 answers HTTP (the reference's `--serve-on-gpu` = faiss index_cpu_to_all_gpus, server.py:51-54)")
    p.add_argument("--subset-ids-path", type=str, default=None, help=".npy with one subset id (string) per vector")
    p.add_argument("--http", type=str, default="native", choices=["native", "uvicorn"],
                   help="HTTP shell: libvodhip's native front (default: the hot routes never enter the interpreter), or uvicorn + FastAPI "
                        "(fallback: ASGI hosting; every route runs in the interpreter)")
    p.add_argument("--http-workers", type=int, default=64, help="(unused by the native front: one native thread per connection; kept for command-line compatibility)")
    p.add_argument("--max-body-mb", type=int, default=512, help="largest request body the native front accepts (413 above it)")
    p.add_argument("--uds", type=str, default=None, help="also serve on this Unix-domain socket path (native front; clients on the same host)")
    p.add_argument("--micro-batch-wait-ms", type=float, default=0.0,
                   help="> 0: every batch additionally waits this long for company.  Not needed: concurrent requests are fused by default "
                        "(batch-while-busy, no fixed window: vodhip_batcher in include/vodhip.h)")
    p.add_argument("--batcher-param", action="append", default=[], metavar="KEY=VALUE",
                   help="vodhip_batcher_set_param, repeatable (max_queries, flat_queries, grace_us, grace_pct, window_us, depth)")
    # set by the owner process for its workers
    p.add_argument("--group-backend", type=str, default="nccl", choices=["nccl", "gloo", "node"],
                   help="with --devices: the workers' process group.  nccl = RCCL over xGMI (one GPU per worker); gloo = requests "
                        "and the per-shard top-k travel through host memory, so several workers may share a GPU (bring-up, tests); "
                        "node = no workers: this one process drives every GPU through the library's node index (the reference server's shape)")
    p.add_argument("--rank", type=int, default=None, help=argparse.SUPPRESS)
    p.add_argument("--master-port", type=int, default=0, help=argparse.SUPPRESS)
    return p.parse_args(argv)


def _die_with_parent() -> None:  # child side of Popen: SIGTERM when the owner goes away, however it goes
    import ctypes
    import signal

    ctypes.CDLL(None).prctl(1, signal.SIGTERM)  # PR_SET_PDEATHSIG


def run_owner(args: argparse.Namespace, argv: list[str]) -> int:
    """`--devices a,b,...`: start one fresh worker per GPU and wait.  The owner never touches a GPU (it does not import
    torch), so the workers are ordinary children - nothing that has initialised HIP is forked or replaced."""
    import os
    import signal
    import subprocess
    import sys
    import time

    from vod_amd.search.socket import find_available_port

    devices = [int(x) for x in args.devices.split(",") if x.strip() != ""]
    if not devices:
        raise SystemExit("--devices needs at least one GPU id")
    port = find_available_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    procs = [subprocess.Popen([sys.executable, "-m", "vod_amd.search.server", *argv, "--rank", str(r), "--master-port", str(port)],
                              env=env, preexec_fn=_die_with_parent) for r in range(len(devices))]

    state = {"stop_at": None}

    def _stop(*_):
        # Orderly shutdown: only rank 0 is signalled.  Its uvicorn drains, then it broadcasts OP_STOP and the workers leave
        # `worker_loop` on their own (terminating every rank at once left rank 0 broadcasting to dead peers: a hang under
        # RCCL, an error exit under gloo).  Whoever is still alive after the grace period is terminated below.
        if state["stop_at"] is None:
            state["stop_at"] = time.monotonic() + 10.0
            if procs[0].poll() is None:
                procs[0].terminate()

    def _kill_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()

    signal.signal(signal.SIGTERM, _stop)
    signal.signal(signal.SIGINT, _stop)
    rc = 0
    while any(p.poll() is None for p in procs):
        time.sleep(0.05)
        if state["stop_at"] is not None and time.monotonic() > state["stop_at"]:
            _kill_all()
            state["stop_at"] = time.monotonic() + 5.0
        for p in procs:
            if p.poll() not in (None, 0) and rc == 0 and state["stop_at"] is None:  # a worker died: the group cannot answer any more
                rc = p.returncode if p.returncode > 0 else 1
                _kill_all()
    return rc


def run_worker(args: argparse.Namespace) -> None:
    import torch
    import torch.distributed as dist

    from vod_amd import store
    from vod_amd.distributed import ShardedFlatIndex, shard_bounds
    from vod_amd.search.group import GroupDispatcher

    devices = [int(x) for x in args.devices.split(",") if x.strip() != ""]
    rank, world = args.rank, len(devices)
    torch.cuda.set_device(devices[rank])
    dev = torch.device("cuda", devices[rank])
    if args.group_backend == "nccl":
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{args.master_port}", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{args.master_port}", rank=rank, world_size=world)
    if str(args.vectors_path).startswith("synthetic:"):
        n = int(str(args.vectors_path)[len("synthetic:"):].partition(":")[0].lower().split("x")[0])
    else:
        n = store.open_vectors(args.vectors_path).shape[0]
    bounds = shard_bounds(n, world, align=256)
    local = HipEngine(args.vectors_path, dtype=args.dtype, device=devices[rank], subset_ids_path=args.subset_ids_path,
                      row_range=(bounds[rank], bounds[rank + 1]), exact_f32=args.exact_f32)
    sharded = ShardedFlatIndex(local.index, row_offset=bounds[rank], always_exchange=True)
    dispatcher = GroupDispatcher(sharded, rank, world, dev if args.group_backend == "nccl" else torch.device("cpu"), search_device=dev,
                                 dim=local.index.dim)
    dist.barrier()  # every shard is resident before rank 0 starts answering (the master's ping loop waits for that)
    if rank != 0:
        dispatcher.worker_loop()
    else:
        host = re.sub(r"^(http|https)://", "", args.host)
        try:
            _serve(GroupHipEngine(local, dispatcher), args, host)
        finally:
            _bounded(dispatcher.stop, 5.0)  # peers that already died (owner gone: PDEATHSIG reaches every rank) cannot hang the exit
    _bounded(dist.destroy_process_group, 5.0)


def _bounded(fn, seconds: float) -> None:
    """Run a shutdown step that talks to the other ranks; if they are gone and it blocks, leave without it."""
    import os

    t = threading.Thread(target=fn, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        os._exit(0)


def _serve(engine, args: argparse.Namespace, host: str) -> None:
    """Run the HTTP shell around `engine` until SIGTERM: libvodhip's native front (default) or uvicorn + FastAPI (`--http uvicorn`, the
    documented fallback: every route through the interpreter)."""
    if args.http == "uvicorn":
        import uvicorn

        if args.uds:
            raise SystemExit("--uds needs --http native (the uvicorn shell does not open the socket)")
        app = create_app(engine, micro_batch_wait_ms=args.micro_batch_wait_ms)
        _apply_batcher_params(app.state.endpoints, args)
        uvicorn.run(app, host=host, port=args.port, workers=1, log_level=args.logging_level.lower())
        return
    endpoints = Endpoints(engine, micro_batch_wait_ms=args.micro_batch_wait_ms)
    _apply_batcher_params(endpoints, args)
    from vod_amd.search import native

    native.run(endpoints, host, args.port, max_body=args.max_body_mb << 20, uds=args.uds)


def _apply_batcher_params(endpoints, args: argparse.Namespace) -> None:
    for kv in args.batcher_param:
        key, _, val = kv.partition("=")
        if endpoints.batcher is None:
            raise SystemExit("--batcher-param needs an engine with a fixed query dimension")
        endpoints.batcher.set_param(key, int(val))


def main(argv=None) -> None:
    import sys

    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.devices is None or args.rank is not None or args.group_backend == "node":
        # every process that serves: thread pools sized for the CPUs the cgroup grants, not for the cores the host shows (see
        # vod_amd/hostcpu.py: 200 ms stalls per request otherwise).  (The owner of a worker group only exports the defaults.)
        from vod_amd.hostcpu import limit_cpu_threads, usable_cpus

        n_workers = len([x for x in args.devices.split(",") if x.strip() != ""]) if (args.devices and args.rank is not None) else 1
        limit_cpu_threads(max(1, usable_cpus() // max(1, n_workers)))  # the workers of a group share the grant
    if args.devices is not None and args.group_backend == "node":
        devices = [int(x) for x in args.devices.split(",") if x.strip() != ""]
        engine = NodeHipEngine(args.vectors_path, devices, dtype=args.dtype, subset_ids_path=args.subset_ids_path, exact_f32=args.exact_f32)
        return _serve(engine, args, re.sub(r"^(http|https)://", "", args.host))
    if args.devices is not None and args.rank is None:
        raise SystemExit(run_owner(args, argv))
    if args.devices is not None:
        return run_worker(args)
    engine = HipEngine(args.vectors_path, dtype=args.dtype, device=args.device, subset_ids_path=args.subset_ids_path, exact_f32=args.exact_f32)
    host = re.sub(r"^(http|https)://", "", args.host)
    _serve(engine, args, host)


if __name__ == "__main__":
    main()

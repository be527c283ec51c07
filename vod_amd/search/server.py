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


class _NoLock:
    """Stands in for the engine lock when the engine serialises its callers itself."""

    def acquire(self, blocking: bool = True) -> bool:  # noqa: ARG002
        return True

    def release(self) -> None:
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc) -> None:
        pass


class MicroBatcher:
    """Fuse concurrent requests into one GPU batch.

    Every dataloader worker of every trainer rank sends its own small batch (the reference serialises them on one
    uvicorn worker, server.py:69,78,98).  A brute-force scan reads the whole corpus per batch whatever its size, so
    answering R waiting requests with ONE scan costs about as much as answering one.  Requests wait at most
    `max_wait_s` for company; the fused batch runs with k = max(k_i) and each caller gets its own rows and columns
    (a top-k prefix of a top-k' list is the top-k, so results are identical to separate searches).
    """

    def __init__(self, engine, max_wait_s: float = 0.001, max_queries: int = 2048, lock: "threading.Lock | None" = None, lanes: int = 2):
        import queue

        self.engine = engine
        # the engine (one index handle, one stream, shared workspace) is not re-entrant: every call into it - the fused
        # batches here and the subset / serialised searches of `create_app` - runs under ONE lock
        self.lock = lock or threading.Lock()
        self.max_wait_s = max_wait_s
        self.max_queries = max_queries
        self._q: "queue.Queue" = queue.Queue()
        # TWO collector threads ("lanes"): while one lane's batch is on the GPU the other gathers the requests that arrive meanwhile -
        # already copied into its fused buffer - and searches the moment the engine is free.  With one lane the GPU idled while the
        # batch was fused, split and answered and while the (closed-loop) clients turned around: 32 dataloader workers x 64 queries
        # reached 69 % of the device-resident rate; the clients now fall into two alternating groups.
        self._threads = [threading.Thread(target=self._run, daemon=True, name=f"vodhip-microbatch-{i}") for i in range(max(1, lanes))]
        for t in self._threads:
            t.start()

    def search(self, query_vec: np.ndarray, top_k: int) -> tuple[np.ndarray, np.ndarray]:
        import concurrent.futures

        fut: "concurrent.futures.Future" = concurrent.futures.Future()
        self._q.put((query_vec, top_k, fut))
        return fut.result()

    def _run(self) -> None:
        import queue
        import time

        fused = None  # this lane's [max_queries, dim] float32 buffer: requests are copied in as they arrive
        while True:
            batch = [self._q.get()]
            deadline = time.monotonic() + self.max_wait_s
            n = 0
            held = False
            try:
                first = np.asarray(batch[0][0])
                dim = first.shape[1]
                if fused is None or fused.shape[1] != dim or fused.shape[0] < max(self.max_queries, len(first)):
                    fused = np.empty((max(self.max_queries, len(first)), dim), dtype=np.float32)
                fused[: len(first)] = first
                n = len(first)
                # company: until the window closes AND the engine is free (a busy engine means waiting costs nothing), or the batch is full
                while n < self.max_queries:
                    remaining = deadline - time.monotonic()
                    if remaining <= 0:
                        # a pipelined engine is "free" while fewer than two batches are on it: the next one is enqueued BEHIND the
                        # running one (no gap on the GPU) and this lane keeps collecting only when two are already queued
                        busy = getattr(self.engine, "pipelined", False) and self.engine.in_flight() >= 2
                        if not busy and self.lock.acquire(blocking=False):
                            held = True
                            break
                        remaining = 0.0005
                    try:
                        item = self._q.get(timeout=remaining)
                    except queue.Empty:
                        continue
                    vec = np.asarray(item[0])
                    if vec.shape[1] != dim or n + len(vec) > fused.shape[0]:
                        self._q.put(item)  # another dimension (its own error / batch) or no room: the next batch takes it
                        if vec.shape[1] != dim:
                            time.sleep(0)  # let the other lane pick it up
                        if not held:
                            self.lock.acquire()
                            held = True
                        break
                    fused[n : n + len(vec)] = vec
                    n += len(vec)
                    batch.append(item)
                if not held:
                    self.lock.acquire()
                    held = True
                k_max = max(b[1] for b in batch)
                try:
                    scores, indices = self.engine.search(fused[:n], k_max)
                finally:
                    self.lock.release()
                    held = False
                lo = 0
                for vec, k, fut in batch:
                    hi = lo + len(vec)
                    fut.set_result((np.asarray(scores[lo:hi, :k]), np.asarray(indices[lo:hi, :k])))
                    lo = hi
            except Exception as exc:  # every waiting caller gets the error (HTTP 500 with the trace)
                if held:
                    self.lock.release()
                for _, _, fut in batch:
                    if not fut.done():
                        fut.set_exception(exc)


class Endpoints:
    """Routes, validation and error mapping of the search service, independent of the HTTP shell around them.

    `handle(method, path, query, body)` -> (status, content type, payload bytes-like, extra headers).  Both shells - the asyncio
    server of `vod_amd.search.fastserver` (production) and the FastAPI app of `create_app` (ASGI hosting, tests) - answer with
    exactly what this returns, so the contract is the reference's whatever carries it:
      422 + `{"detail": [...]}` for a document that fails the pydantic model (`extra="forbid"`, wrong types: models.py:43-79),
      500 + `{"detail": <trace>}` for a failing search (server.py:89-91), 404 / 405 for unknown routes.
    Requests are serialised (one GPU stream, like the reference's single uvicorn worker running faiss synchronously,
    server.py:69,78,98) or, with `micro_batch_wait_ms > 0`, fused into shared GPU batches by `MicroBatcher`."""

    def __init__(self, engine, micro_batch_wait_ms: float = 0.0):
        self.engine = engine
        # an engine that orders its own callers (`HipEngine`: tickets over the library's FIFO of in-flight searches) needs no lock here:
        # concurrent requests are enqueued back to back on the GPU instead of waiting for each other's host-side work
        self.lock = _NoLock() if getattr(engine, "pipelined", False) else threading.Lock()
        self.batcher = MicroBatcher(engine, max_wait_s=micro_batch_wait_ms / 1e3, lock=self.lock) if micro_batch_wait_ms > 0 else None

    # -- the search itself --------------------------------------------------------------------------------------------
    def search(self, query_vec: np.ndarray, top_k: int, subset_ids=None) -> tuple[np.ndarray, np.ndarray]:
        if query_vec.ndim != 2:
            raise ValueError(f"Expected 2D array, got {query_vec.ndim}D array")
        if subset_ids is not None and any(len(s) for s in subset_ids):
            if len(subset_ids) != len(query_vec):
                raise ValueError("`subset_ids` must have one list per query")
            with self.lock:
                scores, indices = self.engine.search(query_vec, top_k, subset_ids=subset_ids)
        elif self.batcher is not None:
            scores, indices = self.batcher.search(query_vec, top_k)
        else:
            with self.lock:
                scores, indices = self.engine.search(query_vec, top_k)
        return np.asarray(scores, dtype=np.float32), np.asarray(indices, dtype=np.int64)

    # -- routes -------------------------------------------------------------------------------------------------------
    def health(self) -> str:
        return "ERROR: Index is empty" if self.engine.ntotal == 0 else "OK"

    def legacy_search(self, document: dict) -> dict:
        """POST /search (server.py:68-73): JSON lists in, JSON lists out."""
        query = SearchQuery(**document)
        scores, indices = self.search(np.asarray(query.vectors, dtype=np.float32), query.top_k)
        rows = [[(None if np.isneginf(v) else float(v)) for v in r] for r in scores]
        return SearchResponse(scores=rows, indices=indices.tolist()).model_dump()

    def fast_search(self, body) -> bytearray:
        """POST /fast-search (server.py:76-91).  The multi-megabyte base64 field is located in the body, validated as the
        `str` the model asks for, and decoded where it lies; the reply's base64 text is written straight into its buffer."""
        small, spans = io.find_payload_spans(body, ("vectors",))
        query = FastSearchQuery(**small)  # raises pydantic.ValidationError -> 422
        try:
            vectors = io.deserialize_np_array_span(body, *spans["vectors"]) if "vectors" in spans else io.deserialize_np_array(query.vectors)
            scores, indices = self.search(vectors, query.top_k, query.subset_ids)
            return io.json_body_with_arrays({"scores": scores, "indices": indices})
        except Exception as exc:
            raise _SearchFailed(traceback.format_exc()) from exc

    def raw_search(self, body, top_k: int) -> tuple[bytes, dict]:
        """POST /raw-search?top_k=K (not in the reference): body = raw `.npy` bytes of the [nq, d] queries (float32
        or float16), reply = raw bytes `scores float32 [nq, k]` followed by `indices int64 [nq, k]`; shapes in the headers."""
        try:
            scores, indices = self.search(io.load_npy_view(body), top_k)
            scores, indices = np.ascontiguousarray(scores), np.ascontiguousarray(indices)
            payload = bytearray(scores.nbytes + indices.nbytes)
            if scores.nbytes:
                np.frombuffer(payload, dtype=np.uint8, count=scores.nbytes)[:] = scores.reshape(-1).view(np.uint8)
                np.frombuffer(payload, dtype=np.uint8, offset=scores.nbytes)[:] = indices.reshape(-1).view(np.uint8)
            return payload, {"x-nq": str(scores.shape[0]), "x-k": str(scores.shape[1])}
        except Exception as exc:
            raise _SearchFailed(traceback.format_exc()) from exc

    def handle(self, method: str, path: str, query: dict, body) -> tuple[int, str, "bytes | bytearray", dict]:
        import json

        js = "application/json"
        try:
            if path == "/":
                if method != "GET":
                    return 405, js, b'{"detail":"Method Not Allowed"}', {}
                return 200, js, json.dumps(self.health()).encode(), {}
            if path not in ("/search", "/fast-search", "/raw-search"):
                return 404, js, b'{"detail":"Not Found"}', {}
            if method != "POST":
                return 405, js, b'{"detail":"Method Not Allowed"}', {}
            if path == "/fast-search":
                return 200, js, self.fast_search(body), {}
            if path == "/raw-search":
                try:
                    top_k = int(query.get("top_k", 3))
                except ValueError:
                    return 422, js, b'{"detail":"top_k must be an integer"}', {}
                payload, headers = self.raw_search(body, top_k)
                return 200, "application/octet-stream", payload, headers
            try:
                document = json.loads(bytes(body))
                if not isinstance(document, dict):
                    raise ValueError("expected a JSON object")
            except ValueError as exc:
                return 422, js, json.dumps({"detail": f"invalid request body: {exc}"}).encode(), {}
            try:
                return 200, js, json.dumps(self.legacy_search(document)).encode(), {}
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
                 row_range: tuple[int, int] | None = None):
        import torch

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
        self.index = HipFlatIndex(d, max(hi - lo, 1), dtype=getattr(torch, dtype), device=device)
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
        self.index = HipFlatIndex(d, max(hi - lo, 1), dtype=getattr(torch, dtype), device=device)
        self.vocab = {}
        for rows in synthetic_rows(torch, torch.device("cuda", device), lo, hi, d, int(seed or 0)):
            self.index.add(rows.to(getattr(torch, dtype)))

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

    # -- searching: thread-safe and pipelined ---------------------------------------------------------------------------
    # Up to MAX_IN_FLIGHT searches are enqueued back to back on the one stream (`vodhip_index_search_async`); the library completes
    # them in FIFO order, so every caller takes a ticket at enqueue time and completes when its ticket is up.  The final select kernel
    # of a search writes its result rows STRAIGHT INTO PINNED HOST MEMORY (a slot of a small ring of pinned buffers): completing a
    # search is an event wait, no device-to-host copy that would queue behind the younger searches' kernels.
    pipelined = True
    MAX_IN_FLIGHT = 3

    _pipe_create = threading.Lock()

    def _pipeline(self):
        st = getattr(self, "_pipe", None)
        if st is None:
            with HipEngine._pipe_create:  # first use from several handler threads at once: ONE state object
                st = getattr(self, "_pipe", None)
                if st is None:
                    st = self._pipe = {"submit": threading.Lock(), "turn": threading.Condition(), "next_ticket": 0, "next_done": 0, "slots": {}}
        return st

    def in_flight(self) -> int:
        st = self._pipeline()
        return st["next_ticket"] - st["next_done"]

    def _pinned_out(self, ticket: int, nq: int, k: int):
        torch = self._torch
        st = self._pipeline()
        slot = ticket % (self.MAX_IN_FLIGHT + 1)
        have = st["slots"].get(slot)
        if have is None or have[0].numel() < nq * k:
            n = max(nq * k, 4096)
            have = (torch.empty((n,), dtype=torch.float32, pin_memory=True), torch.empty((n,), dtype=torch.int64, pin_memory=True))
            st["slots"][slot] = have
        return have[0][: nq * k].view(nq, k), have[1][: nq * k].view(nq, k)

    def search_enqueue(self, query_vec: np.ndarray, top_k: int, subset_ids: list[list[str]] | None = None):
        if query_vec.shape[1] != self.index.dim:
            raise ValueError(f"query dimension {query_vec.shape[1]} != index dimension {self.index.dim}")
        subset = self.encode_subset(subset_ids)
        st = self._pipeline()
        with st["submit"]:
            with st["turn"]:
                st["turn"].wait_for(lambda: st["next_ticket"] - st["next_done"] < self.MAX_IN_FLIGHT)
            ticket = st["next_ticket"]
            out = self._pinned_out(ticket, int(query_vec.shape[0]), int(top_k))
            with self._torch.cuda.device(self.index.device):
                self.index.search_async(query_vec, top_k, id_base=self.row_lo, out=out, subset=subset)
            st["next_ticket"] = ticket + 1  # only a search the library accepted holds a ticket
        return ticket, out

    def search_complete(self, handle) -> tuple[np.ndarray, np.ndarray]:
        ticket, (scores, ids) = handle
        st = self._pipeline()
        with st["turn"]:
            st["turn"].wait_for(lambda: st["next_done"] == ticket)
        try:
            with self._torch.cuda.device(self.index.device):
                self.index.finish()
            # copies: the pinned slot is written again MAX_IN_FLIGHT + 1 searches from now
            return scores.numpy().copy(), ids.numpy().copy()
        finally:
            with st["turn"]:
                st["next_done"] = ticket + 1
                st["turn"].notify_all()

    def search(self, query_vec: np.ndarray, top_k: int, subset_ids: list[list[str]] | None = None) -> tuple[np.ndarray, np.ndarray]:
        return self.search_complete(self.search_enqueue(query_vec, top_k, subset_ids))


class NodeHipEngine:
    """`--devices a,b,... --group-backend node`: ONE server process drives every GPU through the library's node index
    (`vodhip_node_index_*`: per-device row shards, peer copies of the per-shard top-k, merge on the first device) - no worker
    processes and no process group.  This is the shape of the reference's own server, whose single process holds
    `faiss.index_cpu_to_all_gpus(index, co)` with `co.shard = True` (/root/reference/src/vod_search/faiss_search/server.py:51-54).
    Same `.ntotal` / `.search` as `HipEngine`."""

    def __init__(self, vectors_path: str, devices: list[int], dtype: str = "float16", subset_ids_path: str | None = None):
        import torch

        from vod_amd import store
        from vod_amd.index import HipNodeIndex

        self.vocab: dict[str, int] = {}
        if str(vectors_path).startswith("synthetic:"):
            shape, _, seed = str(vectors_path)[len("synthetic:"):].partition(":")
            n, d = (int(v) for v in shape.lower().split("x"))
            self.index = HipNodeIndex(d, max(n, 1), devices, dtype=getattr(torch, dtype))
            for rows in synthetic_rows(torch, torch.device("cuda", devices[0]), 0, n, d, int(seed or 0)):
                self.index.add(rows.to(torch.float16).cpu().numpy())
            self.n_store = n
            return
        vectors = store.open_vectors(vectors_path)
        n, d = vectors.shape
        self.n_store = n
        self.index = HipNodeIndex(d, max(n, 1), devices, dtype=getattr(torch, dtype))
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

    @property
    def ntotal(self) -> int:
        return self.index.ntotal

    encode_subset = HipEngine.encode_subset

    def search(self, query_vec: np.ndarray, top_k: int, subset_ids: list[list[str]] | None = None) -> tuple[np.ndarray, np.ndarray]:
        if query_vec.shape[1] != self.index.dim:
            raise ValueError(f"query dimension {query_vec.shape[1]} != index dimension {self.index.dim}")
        return self.index.search(query_vec, top_k, subset=self.encode_subset(subset_ids))


class GroupHipEngine:
    """Rank 0's engine of a multi-GPU group (`--devices`): same `.ntotal` / `.search` as `HipEngine`, but every search is
    broadcast to the N ranks, each searching its row shard on its own GPU, and merged (vod_amd.search.group)."""

    def __init__(self, local: HipEngine, dispatcher):
        self.local, self.dispatcher = local, dispatcher

    @property
    def ntotal(self) -> int:
        return self.local.n_store

    def search(self, query_vec: np.ndarray, top_k: int, subset_ids: list[list[str]] | None = None) -> tuple[np.ndarray, np.ndarray]:
        if query_vec.shape[1] != self.local.index.dim:
            raise ValueError(f"query dimension {query_vec.shape[1]} != index dimension {self.local.index.dim}")
        return self.dispatcher.search(query_vec, top_k, subset=self.local.encode_subset(subset_ids))


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser()
    p.add_argument("--vectors-path", type=str, required=True)
    p.add_argument("--host", type=str, default="localhost")
    p.add_argument("--port", type=int, default=7678)
    p.add_argument("--logging-level", type=str, default="INFO")
    p.add_argument("--dtype", type=str, default="float16", choices=["float16", "bfloat16"])
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--devices", type=str, default=None,
                   help="comma-separated GPU ids: the store is row-sharded over them, one worker process per GPU on an RCCL "
                        "group, rank 0 answers HTTP (the reference's `--serve-on-gpu` = faiss index_cpu_to_all_gpus, server.py:51-54)")
    p.add_argument("--subset-ids-path", type=str, default=None, help=".npy with one subset id (string) per vector")
    p.add_argument("--http", type=str, default="asyncio", choices=["asyncio", "uvicorn"],
                   help="HTTP shell: the in-tree asyncio server (socket reads land in the request buffer, codec in place) or uvicorn + FastAPI")
    p.add_argument("--http-workers", type=int, default=64, help="handler threads = requests that may be in flight at once")
    p.add_argument("--max-body-mb", type=int, default=512, help="largest request body the asyncio shell accepts (413 above it)")
    p.add_argument("--uds", type=str, default=None, help="also serve on this Unix-domain socket path (asyncio shell; clients on the same host)")
    p.add_argument("--micro-batch-wait-ms", type=float, default=0.0,
                   help="> 0: fuse requests that arrive within this window into one GPU batch (default: serialise, as the reference)")
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
                      row_range=(bounds[rank], bounds[rank + 1]))
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
    """Run the HTTP shell around `engine` until SIGTERM: the asyncio server (default) or uvicorn + FastAPI (`--http uvicorn`)."""
    if args.http == "uvicorn":
        import uvicorn

        uvicorn.run(create_app(engine, micro_batch_wait_ms=args.micro_batch_wait_ms), host=host, port=args.port, workers=1,
                    log_level=args.logging_level.lower())
    else:
        from vod_amd.search import fastserver

        fastserver.run(Endpoints(engine, micro_batch_wait_ms=args.micro_batch_wait_ms), host, args.port, workers=args.http_workers,
                       max_body=args.max_body_mb << 20, uds=args.uds)


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
        engine = NodeHipEngine(args.vectors_path, devices, dtype=args.dtype, subset_ids_path=args.subset_ids_path)
        return _serve(engine, args, re.sub(r"^(http|https)://", "", args.host))
    if args.devices is not None and args.rank is None:
        raise SystemExit(run_owner(args, argv))
    if args.devices is not None:
        return run_worker(args)
    engine = HipEngine(args.vectors_path, dtype=args.dtype, device=args.device, subset_ids_path=args.subset_ids_path)
    host = re.sub(r"^(http|https)://", "", args.host)
    _serve(engine, args, host)


if __name__ == "__main__":
    main()

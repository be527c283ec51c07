"""The native serving layer of libvodhip (`vodhip_batcher_*`, `vodhip_http_*`: include/vodhip.h, section H6) from Python.

`NativeBatcher` puts the library's request fusion in front of an engine - a `HipFlatIndex` (pipelined on the batcher's own stream), a
`HipNodeIndex`, or any object with `.search(np.float32[nq, d], k[, subset=...])` through a callback (a multi-process group, a test double;
needs no GPU).  `NativeHttpFront` runs the library's HTTP/1.1 server: the hot routes never enter the interpreter, everything else is
answered by `Endpoints.handle` through the fallback callback.

Counterpart of the reference's uvicorn + FastAPI process (/root/reference/src/vod_search/faiss_search/server.py:57-98), whose single worker
serialises `faiss_index.search` calls; see HISTORY.md 6b (summary: DESIGN.md 7) for the policy and the measurements.
"""
from __future__ import annotations

import collections
import ctypes
import re
import threading
import urllib.parse

import numpy as np

from vod_amd import _native


class NativeBatcher:
    """Thread-safe blocking `search(queries, k, subset=None, client=0)` over ONE engine, fusing concurrent calls into shared scans."""

    def __init__(self, *, index=None, node=None, engine=None, dim: int, id_base: int = 0, **params: int):
        self._lib = _native.load_library()
        self.dim = int(dim)
        self._keep = (index, node, engine)  # the handles must outlive the batcher
        self._cb = None
        h_index = h_node = None
        if index is not None:
            h_index = index._h
        elif node is not None:
            h_node = node._h
        elif engine is not None:
            self._cb = _native.SEARCH_FN(self._call_engine)
            self._engine = engine
            # a failing engine call fails its whole fused batch: every caller of that batch re-raises the engine's OWN exception, found by
            # the status number the callback returned (it travels through the library's error message)
            self._failures: "collections.OrderedDict[int, BaseException]" = collections.OrderedDict()
            self._failure_seq = 0
        else:
            raise ValueError("one of index / node / engine is required")
        handle = ctypes.c_void_p()
        _native.check(self._lib.vodhip_batcher_create(h_index, h_node, self._cb, None, self.dim, int(id_base), ctypes.byref(handle)))
        self._h = handle
        # `close` may race with searches on other threads: a call is counted from before it reads the handle until it has returned, `close`
        # refuses new calls, lets the counted ones finish normally and only then destroys the handle (the library drains its own callers
        # too - vodhip_batcher_destroy - but cannot protect a Python thread that has read the handle and not yet entered the library)
        self._gate = threading.Condition()
        self._active = 0
        self._closed = False
        for key, value in params.items():
            self.set_param(key, value)

    # -- callback engine ---------------------------------------------------------------------------------------------------------
    def _call_engine(self, _user, q_ptr, nq, k, sub_ptr, n_sub, out_s, out_i) -> int:  # runs on the batcher's scheduler thread
        try:
            q = np.ctypeslib.as_array(ctypes.cast(q_ptr, ctypes.POINTER(ctypes.c_float)), shape=(nq, self.dim))
            if sub_ptr:
                sub = np.ctypeslib.as_array(ctypes.cast(sub_ptr, ctypes.POINTER(ctypes.c_int32)), shape=(nq, n_sub)).copy()
                scores, ids = self._engine.search(q, k, subset=sub)
            else:
                scores, ids = self._engine.search(q, k)
            scores = np.asarray(scores, dtype=np.float32)
            ids = np.asarray(ids, dtype=np.int64)
            if scores.shape != (nq, k) or ids.shape != (nq, k):
                raise ValueError(f"the engine returned {scores.shape} / {ids.shape} for {nq} queries, top-{k}")
            np.ctypeslib.as_array(ctypes.cast(out_s, ctypes.POINTER(ctypes.c_float)), shape=(nq, k))[:] = scores
            np.ctypeslib.as_array(ctypes.cast(out_i, ctypes.POINTER(ctypes.c_int64)), shape=(nq, k))[:] = ids
            return 0
        except BaseException as exc:  # noqa: BLE001 - nothing may propagate into the C thread
            self._failure_seq = self._failure_seq % 1_000_000 + 1
            self._failures[self._failure_seq] = exc
            while len(self._failures) > 64:
                self._failures.popitem(last=False)
            return self._failure_seq

    # -- API ---------------------------------------------------------------------------------------------------------------------
    def set_param(self, key: str, value: int) -> None:
        _native.check(self._lib.vodhip_batcher_set_param(self._h, key.encode(), int(value)))

    def get_stat(self, key: str) -> int:
        out = ctypes.c_int64()
        _native.check(self._lib.vodhip_batcher_get_stat(self._h, key.encode(), ctypes.byref(out)))
        return out.value

    def stats(self) -> dict[str, int]:
        keys = ("batches", "requests", "queries", "fused_requests_max", "grace_waits", "grace_expired", "idle_ns", "busy_ns",
                "last_batch_queries", "last_batch_requests", "flat_scan_ns", "in_flight", "pending", "active_clients",
                "tiles_ns_1", "tiles_ns_2", "tiles_ns_4", "tiles_ns_8")
        return {k: self.get_stat(k) for k in keys}

    def forget_client(self, client: int) -> None:
        self._lib.vodhip_batcher_forget_client(self._h, int(client))

    def search(self, queries: np.ndarray, k: int, subset: np.ndarray | None = None, client: int = 0) -> tuple[np.ndarray, np.ndarray]:
        q = np.ascontiguousarray(queries)
        if q.dtype not in (np.float32, np.float16):
            q = q.astype(np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] queries, got {tuple(q.shape)}")
        nq, k = int(q.shape[0]), int(k)
        sub_ptr, n_sub = None, 0
        if subset is not None:
            sub = np.ascontiguousarray(subset, dtype=np.int32)
            if sub.ndim != 2 or sub.shape[0] != nq:
                raise ValueError(f"expected subset labels of shape [{nq}, S], got {tuple(sub.shape)}")
            sub_ptr, n_sub = sub.ctypes.data, int(sub.shape[1])
        scores = np.empty((nq, max(k, 0)), dtype=np.float32)
        ids = np.empty((nq, max(k, 0)), dtype=np.int64)
        with self._gate:
            if self._closed:
                raise RuntimeError("the batcher is closed")
            self._active += 1
        try:
            rc = self._lib.vodhip_batcher_search(self._h, q.ctypes.data, _native.numpy_dtype_code(q.dtype), nq, k, sub_ptr, n_sub, int(client),
                                                 scores.ctypes.data, ids.ctypes.data)  # (ctypes releases the GIL for the whole wait)
            msg = (self._lib.vodhip_last_error() or b"").decode("utf-8", "replace") if rc != 0 else ""
        finally:
            with self._gate:
                self._active -= 1
                if self._active == 0:
                    self._gate.notify_all()
        if rc != 0:
            found = re.search(r"search callback failed \(status (\d+)\)", msg) if self._cb is not None else None
            if found and int(found.group(1)) in self._failures:
                raise self._failures[int(found.group(1))]  # the engine's own exception (ValueError from a bad argument, ...): what a direct call would raise
            raise _native.NativeLibraryError(msg or f"native call failed with status {rc}")
        return scores, ids

    def close(self) -> None:
        if not getattr(self, "_h", None):
            return
        with self._gate:
            self._closed = True
            self._gate.wait_for(lambda: self._active == 0, timeout=60.0)
            h, self._h = self._h, None
        if h:
            self._lib.vodhip_batcher_destroy(h)

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass


class NativeHttpFront:
    """libvodhip's HTTP server in front of `batcher`; requests it does not take natively are answered by `endpoints.handle`."""

    def __init__(self, batcher: NativeBatcher, endpoints, max_body: int = 512 << 20):
        self._lib = _native.load_library()
        self.batcher, self.endpoints = batcher, endpoints
        self._cb = _native.HTTP_FALLBACK_FN(self._fallback)
        handle = ctypes.c_void_p()
        _native.check(self._lib.vodhip_http_create(batcher._h, batcher.dim, self._cb, None, int(max_body), ctypes.byref(handle)))
        self._h = handle
        self.port: int | None = None

    def _fallback(self, _user, method, target, body_ptr, n_body, client, reply) -> None:  # a connection thread, GIL taken by ctypes
        try:
            url = urllib.parse.urlsplit(target.decode("latin-1"))
            body = ctypes.string_at(body_ptr, n_body) if n_body else b""
            status, ctype, payload, extra = self.endpoints.handle(method.decode("latin-1"), url.path, dict(urllib.parse.parse_qsl(url.query)),
                                                                  body, client=int(client))
            payload = bytes(payload) if not isinstance(payload, bytes) else payload
            lines = "".join(f"{k}: {v}\r\n" for k, v in extra.items()).encode("latin-1")
            self._lib.vodhip_http_reply_set(reply, int(status), ctype.encode("latin-1"), payload, len(payload), lines or None)
        except BaseException as exc:  # noqa: BLE001 - `Endpoints.handle` maps its own errors; this is a bug in the shell
            msg = b'{"detail": "internal error: %s"}' % type(exc).__name__.encode()
            self._lib.vodhip_http_reply_set(reply, 500, b"application/json", msg, len(msg), None)

    def listen(self, host: str, port: int, uds: str | None = None) -> int:
        bound = self._lib.vodhip_http_listen_tcp(self._h, host.encode(), int(port))
        if bound < 0:
            _native.check(bound)
        self.port = bound
        if uds:
            _native.check(self._lib.vodhip_http_listen_unix(self._h, str(uds).encode()))
        return bound

    def start(self) -> None:
        _native.check(self._lib.vodhip_http_start(self._h))

    def get_stat(self, key: str) -> int:
        out = ctypes.c_int64()
        _native.check(self._lib.vodhip_http_get_stat(self._h, key.encode(), ctypes.byref(out)))
        return out.value

    def stop(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vodhip_http_stop(self._h)

    def close(self) -> None:
        if getattr(self, "_h", None):
            if self._lib.vodhip_http_destroy(self._h) == 0:
                self._h = None


def run(endpoints, host: str, port: int, max_body: int = 512 << 20, uds: str | None = None) -> None:
    """Blocking entry point of the server process: serve until SIGTERM / SIGINT.  The native threads do the work; this (main)
    thread only sleeps, so Python signal handlers run promptly."""
    import signal

    stop = threading.Event()
    front = NativeHttpFront(endpoints.batcher, endpoints, max_body=max_body)
    front.listen(host, port, uds)
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, lambda *_: stop.set())
        except ValueError:  # pragma: no cover - not the main thread
            pass
    front.start()
    try:
        while not stop.wait(0.5):
            pass
    finally:
        front.close()

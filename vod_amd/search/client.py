"""`HipMipsClient` / `HipMipsMaster`: drop-in for `FaissClient` / `FaissMaster`
(/root/reference/src/vod_search/faiss_search/client.py:18-186).

Same constructor fields (`host`, `port`), same keyword-only `search()` that ignores
`text / subset_ids / ids / shard` (client.py:64-74), same `/fast-search` request and response encoding,
same `RetrievalBatch` result with `meta["time"]`.  The master spawns `python -m vod_amd.search.server`,
which owns the GPU-resident index; clients stay plain HTTP so they can be pickled into DataLoader workers.
"""
from __future__ import annotations

import http.client
import os
import pathlib
import sys
import threading
import time
from copy import copy

import numpy as np
import requests

from vod_amd import io
from vod_amd import types as vt
from vod_amd.search import base
from vod_amd.search.socket import find_available_port, private_socket_dir


class _UnixHTTPConnection(http.client.HTTPConnection):
    """HTTP/1.1 over a Unix-domain socket (the server's `--uds` listener)."""

    def __init__(self, path: str, timeout: float = 120):
        super().__init__("localhost", timeout=timeout)
        self._uds_path = path

    def connect(self) -> None:
        import socket

        sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        sock.settimeout(self.timeout)
        sock.connect(self._uds_path)
        self.sock = sock


class _Headers(dict):
    """Response headers with lower-case names (`h["x-nq"]`, `h.get("Content-Length")`)."""

    def __getitem__(self, key):
        return super().__getitem__(key.lower())

    def get(self, key, default=None):
        return super().get(key.lower(), default)


class _UnframedReply(Exception):
    """The reply carries no Content-Length (chunked / close-delimited): `http.client` reads it instead."""


class _LeanConnection:
    """One kept-alive HTTP/1.1 connection (TCP with TCP_NODELAY, or a Unix-domain socket) that speaks exactly what the search service
    needs: POST with Content-Length, a reply with Content-Length.  `http.client` spends ~100 us per exchange building the request
    through its header machinery and parsing the reply through `email.parser` - a third of a DataLoader worker's turn-around between
    two searches, which is time the GPU idles when a handful of workers run in lock-step.  Anything this class does not understand
    (no Content-Length in the reply) is handed back to `http.client`."""

    def __init__(self, host: str | None, port: int | None, uds: str | None, timeout: float):
        import socket

        if uds:
            self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            self.sock.settimeout(timeout)
            self.sock.connect(uds)
            self.host_header = b"localhost"
        else:
            self.sock = socket.create_connection((host, port), timeout=timeout)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.host_header = f"{host}:{port}".encode("latin-1")
        self.buf = bytearray(1 << 16)
        self.used = False  # a reply has been read on this connection (a failure on a USED connection may be a stale keep-alive)

    def settimeout(self, timeout: float) -> None:
        self.sock.settimeout(timeout)

    def close(self) -> None:
        try:
            self.sock.close()
        except OSError:
            pass

    def post(self, path: str, body, content_type: str) -> tuple[int, _Headers, bytes]:
        head = b"POST %s HTTP/1.1\r\nHost: %s\r\nContent-Type: %s\r\nContent-Length: %d\r\nConnection: keep-alive\r\n\r\n" % (
            path.encode("latin-1"), self.host_header, content_type.encode("latin-1"), len(body))
        if len(body) <= 16384:
            self.sock.sendall(head + bytes(body))
        else:
            self.sock.sendall(head)
            self.sock.sendall(body)
        # status line + headers
        buf, view, have = self.buf, memoryview(self.buf), 0
        while True:
            end = buf.find(b"\r\n\r\n", 0, have)
            if end >= 0:
                break
            if have == len(buf):
                raise http.client.HTTPException("reply headers too large")
            n = self.sock.recv_into(view[have:])
            if n == 0:
                raise ConnectionResetError("the server closed the connection")
            have += n
        lines = bytes(buf[:end]).split(b"\r\n")
        parts = lines[0].split(None, 2)
        if len(parts) < 2 or not parts[0].startswith(b"HTTP/1."):
            raise http.client.HTTPException(f"malformed status line {lines[0][:80]!r}")
        status = int(parts[1])
        headers = _Headers()
        for ln in lines[1:]:
            name, _, value = ln.partition(b":")
            headers[name.strip().lower().decode("latin-1")] = value.strip().decode("latin-1")
        length = headers.get("content-length")
        if length is None or "chunked" in headers.get("transfer-encoding", "").lower():
            raise _UnframedReply()
        n_body = int(length)
        payload = bytearray(n_body)
        got = min(have - (end + 4), n_body)
        payload[:got] = buf[end + 4 : end + 4 + got]
        pv = memoryview(payload)
        while got < n_body:
            n = self.sock.recv_into(pv[got:])
            if n == 0:
                raise ConnectionResetError("the server closed the connection inside a reply")
            got += n
        self.used = True
        if headers.get("connection", "").lower() == "close":
            self.close()
        return status, headers, bytes(payload) if n_body < 4096 else payload


class _NativeClientHandle:
    """Owner of one `vodhip_client` handle (one kept-alive connection): lives in a thread's local storage and closes the connection
    when the thread - or the client object - goes away.  A forked child never uses (or frees) its parent's handle."""

    def __init__(self, lib, handle):
        self.lib, self.handle, self.pid = lib, handle, os.getpid()

    def __del__(self):  # pragma: no cover - best effort
        try:
            if self.handle and self.pid == os.getpid():
                self.lib.vodhip_client_destroy(self.handle)
        except Exception:
            pass


class HipMipsClient(base.SearchClient):
    """HTTP client of the HIP MIPS server."""

    requires_vectors = True

    def __init__(self, host: str = "http://localhost", port: int = 7678, binary: bool = False, forward_subset_ids: bool = False,
                 wire_dtype: str | None = None, uds: str | None = None, native: bool = True):
        self.host = host
        # the exchange itself (request document, framing, reply parsing) runs in libvodhip's client (`vodhip_client_*`, one call with the
        # GIL released) when the library can be loaded in this process and the address is plain http / a Unix socket; anything it cannot do
        # (subset filters, https, a transport hiccup) goes through the Python path below, which also produces the exceptions
        self.native = native
        self.port = port
        # a Unix-domain socket path the server also listens on (`--uds`): the searches go through it (same HTTP, no TCP stack);
        # `ping()` keeps using host:port, so a client on another host simply leaves it unset
        self.uds = uds
        self.binary = binary  # use the raw-bytes route (`/raw-search`) instead of base64-in-JSON (`/fast-search`)
        # The reference's faiss client drops `subset_ids`; set this to let the GPU index honour them (the server must
        # have been started with `--subset-ids-path`).
        self.forward_subset_ids = forward_subset_ids
        # "float16": send the queries as float16 (half the bytes on the wire).  Exact for a float16 store - the library rounds
        # float32 queries to the store dtype anyway (round to nearest even, like NumPy) - so only set it for such a store.
        self.wire_dtype = wire_dtype
        # Connections are kept open instead of a TCP handshake per batch - ONE PER THREAD and per process: the reference calls a
        # shared client from several executor threads at once (sharded_search.py:159-167, hybrid_search.py:103-119), and a socket
        # must not be shared across a fork / an unpickling
        self._local = threading.local()

    def __getstate__(self) -> dict:
        state = dict(self.__dict__)
        state.pop("_local", None)  # sockets do not pickle: a DataLoader worker opens its own on first use
        return state

    def __setstate__(self, state: dict) -> None:
        self.__dict__.update(state)
        self.__dict__.setdefault("wire_dtype", None)
        self.__dict__.setdefault("uds", None)
        self.__dict__.setdefault("native", True)
        self._local = threading.local()

    @property
    def session(self) -> requests.Session:
        loc = self._local
        if getattr(loc, "session", None) is None or loc.session_pid != os.getpid():  # a forked worker must not share its parent's socket
            loc.session = requests.Session()
            loc.session_pid = os.getpid()
        return loc.session

    def _post(self, path: str, body, content_type: str, timeout: float) -> tuple[int, "_Headers | http.client.HTTPMessage", bytes]:
        """POST on this thread's persistent connection.  Plain `http://` and Unix-socket connections use `_LeanConnection` (send, read
        Content-Length bytes; ~100 us less per exchange than `http.client`, which is ~half the cost of `requests` for multi-MB bodies
        in turn); `https://` and replies without a Content-Length go through `http.client`.  A connection the server closed while it
        was idle is re-opened once; failures surface as the `requests` exceptions callers of the reference client expect."""
        import socket

        for attempt in (0, 1):
            lean = None
            try:
                lean = self._lean_connection(timeout)
                if lean is not None:
                    return lean.post(path, body, content_type)
                conn = self._connection(timeout)
                conn.request("POST", path, body=body, headers={"content-type": content_type})
                resp = conn.getresponse()
                return resp.status, resp.headers, resp.read()
            except _UnframedReply:
                self._drop_connection()
                self._local.no_lean = True  # this server frames its replies differently: `http.client` from now on
                if attempt == 1:
                    raise requests.exceptions.ConnectionError(f"POST {self.url}{path}: the reply could not be framed") from None
            except socket.timeout as exc:
                self._drop_connection()
                raise requests.exceptions.ReadTimeout(f"POST {self.url}{path} timed out after {timeout} s") from exc
            except (http.client.HTTPException, OSError, ValueError) as exc:
                fresh = lean is not None and not lean.used
                self._drop_connection()
                if attempt == 1 or isinstance(exc, ConnectionRefusedError) or (fresh and isinstance(exc, (http.client.HTTPException, ValueError))):
                    raise requests.exceptions.ConnectionError(f"POST {self.url}{path}: {exc}") from exc
        raise AssertionError("unreachable")

    def _lean_connection(self, timeout: float) -> "_LeanConnection | None":
        import urllib.parse

        loc = self._local
        if getattr(loc, "no_lean", False):
            return None
        lean = getattr(loc, "lean", None)
        if lean is not None and loc.lean_pid == os.getpid() and lean.sock.fileno() >= 0:  # (a forked / unpickled worker opens its own socket)
            lean.settimeout(timeout)
            return lean
        u = urllib.parse.urlsplit(self.host if "://" in self.host else "http://" + self.host)
        if u.scheme != "http":
            loc.no_lean = True
            return None
        uds = self.uds if (self.uds and os.path.exists(self.uds)) else None  # (no socket file on THIS host: the TCP address still works)
        loc.lean = _LeanConnection(u.hostname, self.port, uds, timeout)
        loc.lean_pid = os.getpid()
        return loc.lean

    def _connection(self, timeout: float):
        import http.client
        import urllib.parse

        loc = self._local
        if getattr(loc, "conn", None) is None or loc.conn_pid != os.getpid():  # a forked / unpickled worker opens its own socket
            if self.uds and os.path.exists(self.uds):
                loc.conn = _UnixHTTPConnection(self.uds, timeout=timeout)
            else:  # (no socket file on THIS host - a client on another machine, a server without --uds: the TCP address still works)
                u = urllib.parse.urlsplit(self.host if "://" in self.host else "http://" + self.host)
                cls = http.client.HTTPSConnection if u.scheme == "https" else http.client.HTTPConnection
                loc.conn = cls(u.hostname, self.port, timeout=timeout)
            loc.conn_pid = os.getpid()
        elif loc.conn.sock is not None:
            loc.conn.sock.settimeout(timeout)
        return loc.conn

    def _drop_connection(self) -> None:
        lean = getattr(self._local, "lean", None)
        if lean is not None:
            lean.close()
            self._local.lean = None
        conn = getattr(self._local, "conn", None)
        if conn is not None:
            try:
                conn.close()
            finally:
                self._local.conn = None

    @staticmethod
    def _raise_for_status(status: int, data: bytes, url: str) -> None:
        if status < 400:
            return
        try:  # the reference client prints the server's trace before raising (client.py:81-88)
            import json

            print(json.loads(data)["detail"], file=sys.stderr)
        except Exception:
            print(data[:2000].decode("utf-8", "replace"), file=sys.stderr)
        kind = "Client Error" if status < 500 else "Server Error"
        raise requests.exceptions.HTTPError(f"{status} {kind} for url: {url}")

    def _wire(self, vector: np.ndarray) -> np.ndarray:
        v = np.asarray(vector)
        if self.wire_dtype is not None and v.dtype != np.dtype(self.wire_dtype):
            v = v.astype(self.wire_dtype)
        return np.ascontiguousarray(v)

    def __repr__(self) -> str:
        return f"{type(self).__name__}[{self.url}](requires_vectors={self.requires_vectors})"

    @property
    def url(self) -> str:
        return f"{self.host}:{self.port}"

    def ping(self, timeout: float = 120) -> bool:
        try:
            response = self.session.get(f"{self.url}/", timeout=timeout)
        except requests.exceptions.ConnectionError:
            self._local.session = None
            return False
        response.raise_for_status()
        return "OK" in response.text

    def search_py(self, query_vec: np.ndarray, top_k: int = 3, timeout: float = 120) -> vt.RetrievalBatch:
        """Legacy JSON-list route (`POST /search`, client.py:47-62)."""
        response = requests.post(f"{self.url}/search", json={"vectors": np.asarray(query_vec).tolist(), "top_k": top_k}, timeout=timeout)
        response.raise_for_status()
        data = response.json()
        return vt.RetrievalBatch.cast(indices=data["indices"], scores=_scores_from_json(data["scores"]))

    def search(
        self,
        *,
        vector: np.ndarray,
        text: None | list[str] = None,  # noqa: ARG002
        subset_ids: None | list[list[base.SubsetId]] = None,
        ids: None | list[list[base.SectionId]] = None,  # noqa: ARG002
        shard: None | list[base.ShardName] = None,  # noqa: ARG002
        top_k: int = 3,
        timeout: float = 120,
    ) -> vt.RetrievalBatch:
        start = time.time()
        filtered = self.forward_subset_ids and subset_ids is not None
        if self.native and not filtered:
            out = self._search_native(vector, top_k, timeout)
            if out is not None:
                return vt.RetrievalBatch(scores=out[0], indices=out[1], labels=None, meta={"time": time.time() - start})
        if self.binary and not filtered:
            return self._search_binary(np.asarray(vector), top_k, timeout, start)
        extra: dict = {"top_k": top_k}
        if self.forward_subset_ids and subset_ids is not None:
            extra["subset_ids"] = [list(map(str, s)) for s in subset_ids]
        # the same JSON document `requests.post(json=...)` would send: the base64 text is written straight into the body buffer
        body = io.json_body_with_arrays({"vectors": self._wire(vector)}, extra)
        status, _, content = self._post("/fast-search", body, "application/json", timeout)
        self._raise_for_status(status, content, f"{self.url}/fast-search")
        small, spans = io.find_payload_spans(content, ("scores", "indices"))
        take = lambda key: io.deserialize_np_array_span(content, *spans[key]) if key in spans else io.deserialize_np_array(small[key])  # noqa: E731
        return vt.RetrievalBatch.cast(indices=take("indices"), scores=take("scores"), labels=None, meta={"time": time.time() - start})


    def _native_handle(self):
        """This thread's `vodhip_client` handle (None: the library is not loadable here, or the address is not plain http / a socket)."""
        import ctypes
        import urllib.parse

        loc = self._local
        if getattr(loc, "native_off", False):
            return None
        h = getattr(loc, "native_h", None)
        if h is not None and h.pid == os.getpid():
            return h
        try:
            from vod_amd import _native

            lib = _native.load_library()
        except Exception:
            loc.native_off = True
            return None
        u = urllib.parse.urlsplit(self.host if "://" in self.host else "http://" + self.host)
        if u.scheme != "http" or not u.hostname:
            loc.native_off = True
            return None
        uds = self.uds if (self.uds and os.path.exists(self.uds)) else None
        handle = ctypes.c_void_p()
        if lib.vodhip_client_create(u.hostname.encode(), int(self.port), uds.encode() if uds else None, ctypes.byref(handle)) != 0:
            loc.native_off = True
            return None
        loc.native_h = _NativeClientHandle(lib, handle)
        return loc.native_h

    def _search_native(self, vector, top_k: int, timeout: float):
        """One call into libvodhip's client.  (scores, indices), or None when the Python path should take this request: a layout the
        native client does not send, or a transport failure - which the Python path then meets (and reports) itself."""
        v = self._wire(vector)
        if v.ndim != 2 or v.shape[0] < 1 or v.shape[1] < 1 or v.dtype not in (np.float32, np.float16) or not 1 <= int(top_k) <= 2048:  # VODHIP_MAX_K
            return None
        h = self._native_handle()
        if h is None:
            return None
        lib, handle = h.lib, h.handle
        nq, k = int(v.shape[0]), int(top_k)
        scores = np.empty((nq, k), dtype=np.float32)
        indices = np.empty((nq, k), dtype=np.int64)
        rc = lib.vodhip_client_search(handle, v.ctypes.data, 2 if v.dtype == np.float32 else 0, nq, int(v.shape[1]), k, 1 if self.binary else 0,
                                      float(timeout), scores.ctypes.data, indices.ctypes.data)
        if rc == 0:
            return scores, indices
        if rc > 0:  # an error reply of the server: the reference client's behaviour (print the trace, raise HTTPError)
            body = lib.vodhip_client_last_body(handle) or b""
            self._raise_for_status(int(rc), body, f"{self.url}/{'raw' if self.binary else 'fast'}-search")
        msg = (lib.vodhip_last_error() or b"").decode("utf-8", "replace")
        if "timed out" in msg:
            raise requests.exceptions.ReadTimeout(f"POST {self.url}: {msg} (timeout {timeout} s)")
        return None

    def _search_binary(self, vector: np.ndarray, top_k: int, timeout: float, start: float) -> vt.RetrievalBatch:
        v = self._wire(vector)
        head = io.npy_header(v)
        body = bytearray(len(head) + v.nbytes)  # the `.npy` bytes np.save would write, assembled with one copy of the data
        body[: len(head)] = head
        if v.nbytes:
            np.frombuffer(body, dtype=np.uint8, offset=len(head))[:] = v.reshape(-1).view(np.uint8)
        status, headers, raw = self._post(f"/raw-search?top_k={int(top_k)}", body, "application/octet-stream", timeout)
        self._raise_for_status(status, raw, f"{self.url}/raw-search")
        nq, k = int(headers["x-nq"]), int(headers["x-k"])
        scores = np.frombuffer(raw, dtype=np.float32, count=nq * k).reshape(nq, k).copy()
        indices = np.frombuffer(raw, dtype=np.int64, count=nq * k, offset=nq * k * 4).reshape(nq, k).copy()
        return vt.RetrievalBatch(scores=scores, indices=indices, labels=None, meta={"time": time.time() - start})


def _scores_from_json(rows: list) -> np.ndarray:
    # JSON has no -inf literal in strict mode; the server encodes pads as None on the legacy route
    return np.array([[(-np.inf if v is None else v) for v in r] for r in rows], dtype=np.float32)


class HipMipsMaster(base.SearchMaster[HipMipsClient]):
    """Spawns / terminates the HIP MIPS server.

    ```python
    with HipMipsMaster(vectors_path) as master:
        client = master.get_client()
        result = client.search(vector=queries, top_k=100)
    ```
    `vectors_path` is a `.npy` file ([N, D] float32 / float16) or a store directory written by
    `vod_amd.store.save_vectors`.  Counterpart of `FaissMaster(index_path, nprobe, logging_level, host, port,
    skip_setup, free_resources, serve_on_gpu)`; `nprobe` is accepted and ignored (the index is exact).
    `devices=[0, ..., 7]` serves ONE index row-sharded over several GPUs from the same address.
    """

    def __init__(  # noqa: PLR0913
        self,
        vectors_path: str | pathlib.Path,
        nprobe: int = 8,  # noqa: ARG002 - exact index: nothing to probe
        logging_level: str = "DEBUG",
        host: str = "http://localhost",
        port: int = 6637,
        skip_setup: bool = False,
        free_resources: bool = False,
        dtype: str = "float16",
        device: int = 0,
        devices: None | list[int] = None,
        serve_on_gpu: bool = True,  # noqa: ARG002 - accepted for call-site compatibility: this engine only exists on the GPU
        group_backend: str = "nccl",  # with `devices`: "nccl" (RCCL, one GPU per worker), "gloo" (host-staged; workers may share a GPU) or "node" (ONE server process drives every GPU: vodhip_node_index, no workers)
        micro_batch_wait_ms: float = 0.0,  # > 0: every batch additionally waits this long for company (concurrent requests are fused by default)
        http: str = "native",  # the server's HTTP shell: "native" (libvodhip's front, default) or "uvicorn" (FastAPI fallback)
        batcher_params: None | dict[str, int] = None,  # vodhip_batcher_set_param on the server (max_queries, grace_us, grace_pct, flat_queries, depth)
        uds: bool | str = False,  # also listen on a Unix-domain socket (True = a path in a private directory); `get_client()` then uses it
        exact_f32: bool = False,  # the server keeps the float32 rows and answers with the float32 brute-force result (HipFlatIndex)
    ):
        super().__init__(skip_setup=skip_setup, free_resources=free_resources)
        self.vectors_path = vectors_path if str(vectors_path).startswith("synthetic:") else pathlib.Path(vectors_path)
        self.logging_level = logging_level
        self.host = host
        self.port = find_available_port() if port < 0 else port
        self.dtype = dtype
        self.exact_f32 = bool(exact_f32)
        self.device = device
        # `devices=[...]`: the store is row-sharded over these GPUs behind the SAME host:port (one worker process per
        # GPU on an RCCL group; what `FaissMaster(serve_on_gpu=True)` gets from faiss's index_cpu_to_all_gpus,
        # /root/reference/src/vod_search/faiss_search/client.py:118-137, server.py:51-54)
        self.devices = None if devices is None else [int(d) for d in devices]
        self.group_backend = group_backend
        self.micro_batch_wait_ms = float(micro_batch_wait_ms)
        self.http = http
        self.batcher_params = dict(batcher_params or {})
        if uds is True:
            # named after the PORT only: a rank built with `skip_setup=True` (it only connects) must derive the same path as the
            # rank that spawned the server, whatever its pid - inside a directory only THIS user can write to
            uds = os.path.join(private_socket_dir(), f"vodhip-{self.port}.sock")
        self.uds = uds or None
        if self.uds and http == "uvicorn":
            raise ValueError("`uds` needs http='native': the uvicorn shell does not open the socket")

    def _make_env(self) -> dict[str, str]:
        env = copy(dict(os.environ))
        root = str(pathlib.Path(__file__).resolve().parents[2])
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        return env

    def _make_cmd(self) -> list[str]:
        return [
            sys.executable, "-m", "vod_amd.search.server",
            "--vectors-path", str(self.vectors_path) if str(self.vectors_path).startswith("synthetic:") else str(self.vectors_path.absolute()),
            "--host", str(self.host),
            "--port", str(self.port),
            "--logging-level", str(self.logging_level),
            "--dtype", self.dtype,
            *(["--exact-f32"] if self.exact_f32 else []),
            "--http", self.http,
            *(["--micro-batch-wait-ms", str(self.micro_batch_wait_ms)] if self.micro_batch_wait_ms > 0 else []),
            *(["--uds", str(self.uds)] if self.uds else []),
            *[a for k, v in self.batcher_params.items() for a in ("--batcher-param", f"{k}={int(v)}")],
            *(["--devices", ",".join(map(str, self.devices)), "--group-backend", self.group_backend]
              if self.devices is not None else ["--device", str(self.device)]),
        ]

    def get_client(self) -> HipMipsClient:
        return HipMipsClient(host=self.host, port=self.port, uds=self.uds)

    @property
    def url(self) -> str:
        return f"{self.host}:{self.port}"

    @property
    def service_info(self) -> str:
        return f"HipMipsServer[{self.url}]"

    @property
    def service_name(self) -> str:
        return super().service_name + f"-{self.port}"

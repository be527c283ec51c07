"""A small HTTP/1.1 server for the four routes of the search service, built on `asyncio.BufferedProtocol`.

Why: through uvicorn's pure-Python h11 parser a 4.2 MB `/fast-search` body (1024 x 768 float32 queries as base64-in-JSON, the
reference's wire format: /root/reference/src/vod_search/faiss_search/server.py:76-91, io.py:17-32) is received in 64 KB events,
joined, then copied again by starlette - several milliseconds before the handler runs, for a search that takes one.  Here the
socket reads land DIRECTLY in one preallocated body buffer (`get_buffer` / `buffer_updated`: no per-chunk objects), the
handler decodes the base64 text in place and writes the reply's base64 text straight into the response buffer
(`vod_amd.io.json_body_with_arrays`).  Handlers run on a thread pool (the codec releases the GIL inside libvodhip), so
concurrent clients are accepted - and fused by the library's batcher - while another request is being decoded.

Same contract as the FastAPI app (`vod_amd.search.server.create_app`, kept for ASGI hosting and tests): both are thin shells
around `server.Endpoints`, which owns routes, validation and error mapping (422 for a malformed document, 500 with the trace
for a failing search).  Connections are keep-alive; `Expect: 100-continue` is honoured; chunked REQUEST bodies are not supported (501: send Content-Length,
as every client of this service does).
"""
from __future__ import annotations

import asyncio
import concurrent.futures
import socket
import urllib.parse

_REASONS = {200: "OK", 100: "Continue", 400: "Bad Request", 404: "Not Found", 405: "Method Not Allowed", 411: "Length Required",
            413: "Content Too Large", 422: "Unprocessable Entity", 431: "Request Header Fields Too Large", 500: "Internal Server Error",
            501: "Not Implemented"}
MAX_HEAD = 64 * 1024
MAX_BODY = 512 << 20  # default cap on Content-Length (a 2048 x 4096 float32 batch is 45 MB of base64): the body buffer is allocated up front


class _Connection(asyncio.BufferedProtocol):
    def __init__(self, endpoints, pool: concurrent.futures.Executor, max_body: int = MAX_BODY):
        self.endpoints, self.pool, self.max_body = endpoints, pool, max_body
        self.transport = None
        self.head = bytearray(MAX_HEAD)   # request line + headers (and whatever of the body arrived with them)
        self.head_len = 0
        self.body: bytearray | None = None
        self.body_len = 0
        self.filled = 0
        self.request = None               # (method, path, query, keep_alive) while a body is being received / handled
        self.busy = False

    # -- transport events ---------------------------------------------------------------------------------------------
    def connection_made(self, transport) -> None:
        self.transport = transport
        sock = transport.get_extra_info("socket")
        if sock is not None:
            try:
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            except OSError:  # pragma: no cover - not a TCP socket
                pass

    def get_buffer(self, sizehint: int):
        if self.body is not None and self.filled < self.body_len:
            return memoryview(self.body)[self.filled : self.body_len]
        if self.head_len >= MAX_HEAD:
            self._fail(431, "request headers too large")
            return memoryview(bytearray(1))
        return memoryview(self.head)[self.head_len :]

    def buffer_updated(self, nbytes: int) -> None:
        if self.body is not None and self.filled < self.body_len:
            self.filled += nbytes
            if self.filled == self.body_len:
                self._dispatch()
            return
        self.head_len += nbytes
        self._parse_head()

    def eof_received(self):
        return False

    # -- request parsing ----------------------------------------------------------------------------------------------
    def _parse_head(self) -> None:
        if self.busy:
            return  # a pipelined request waits in `head` until the current reply is out
        end = self.head.find(b"\r\n\r\n", 0, self.head_len)
        if end < 0:
            return
        try:
            lines = bytes(self.head[:end]).decode("latin-1").split("\r\n")
            method, target, version = lines[0].split(" ", 2)
            headers = {}
            for ln in lines[1:]:
                k, _, v = ln.partition(":")
                headers[k.strip().lower()] = v.strip()
        except ValueError:
            return self._fail(400, "malformed request line")
        if "chunked" in headers.get("transfer-encoding", "").lower():
            return self._fail(501, "chunked request bodies are not supported: send Content-Length")
        try:
            n = int(headers.get("content-length", "0"))
        except ValueError:
            return self._fail(400, "malformed Content-Length")
        if n < 0 or n > self.max_body:
            return self._fail(413, "request body too large")
        if method in ("POST", "PUT") and "content-length" not in headers:
            return self._fail(411, "Content-Length required")
        url = urllib.parse.urlsplit(target)
        keep = (version == "HTTP/1.1" and headers.get("connection", "").lower() != "close") or headers.get("connection", "").lower() == "keep-alive"
        self.request = (method, url.path, dict(urllib.parse.parse_qsl(url.query)), keep)
        if headers.get("expect", "").lower() == "100-continue":
            self.transport.write(b"HTTP/1.1 100 Continue\r\n\r\n")
        # the bytes that arrived behind the headers are the beginning of the body (or of the next request)
        start = end + 4
        have = self.head_len - start
        self.body = bytearray(n)
        self.body_len = n
        take = min(have, n)
        self.body[:take] = self.head[start : start + take]
        self.filled = take
        rest = have - take
        self.head[:rest] = self.head[start + take : start + take + rest]
        self.head_len = rest
        if self.filled == n:
            self._dispatch()

    # -- handling -----------------------------------------------------------------------------------------------------
    def _dispatch(self) -> None:
        method, path, query, keep = self.request
        body, self.body, self.body_len, self.filled = self.body, None, 0, 0
        self.busy = True
        self.transport.pause_reading()  # a pipelined request stays in the socket (and in what `head` already holds) until this one is answered
        loop = asyncio.get_running_loop()
        fut = loop.run_in_executor(self.pool, self.endpoints.handle, method, path, query, body, id(self))  # the connection = the client tag
        fut.add_done_callback(lambda f: self._reply(f, keep))

    def _reply(self, fut, keep: bool) -> None:
        if self.transport is None or self.transport.is_closing():
            return
        try:
            status, ctype, payload, extra = fut.result()
        except Exception as exc:  # noqa: BLE001 - `Endpoints.handle` maps its own errors; this is a bug in the shell
            status, ctype, payload, extra = 500, "application/json", b'{"detail": "internal error: %s"}' % type(exc).__name__.encode(), {}
        self._write(status, ctype, payload, extra, keep)
        self.busy = False
        if not keep:
            self.transport.close()
            return
        self.transport.resume_reading()
        if self.head_len:
            self._parse_head()  # a pipelined request was already waiting

    def _write(self, status: int, ctype: str, payload, extra: dict, keep: bool) -> None:
        head = [f"HTTP/1.1 {status} {_REASONS.get(status, 'Status')}", f"content-type: {ctype}", f"content-length: {len(payload)}",
                "connection: " + ("keep-alive" if keep else "close")]
        head += [f"{k}: {v}" for k, v in extra.items()]
        self.transport.write(("\r\n".join(head) + "\r\n\r\n").encode("latin-1"))
        if len(payload):
            self.transport.write(payload)

    def _fail(self, status: int, detail: str) -> None:
        import json

        self._write(status, "application/json", json.dumps({"detail": detail}).encode(), {}, False)
        self.transport.close()

    def connection_lost(self, exc) -> None:
        self.transport = None
        batcher = getattr(self.endpoints, "batcher", None)
        if batcher is not None:
            batcher.forget_client(id(self))  # nobody should wait for this client's next request


async def serve(endpoints, host: str, port: int, workers: int = 64, ready: "asyncio.Event | None" = None, stop: "asyncio.Event | None" = None,
                max_body: int = MAX_BODY, uds: "str | None" = None) -> None:
    """Serve until `stop` is set (or forever).  `workers`: handler threads = requests that may be in flight at once (each DataLoader
    worker of each trainer rank holds one connection: src/vod_dataloaders/realm_dataloader.py:92-118)."""
    loop = asyncio.get_running_loop()
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="vodhip-http")
    server = await loop.create_server(lambda: _Connection(endpoints, pool, max_body), host=host, port=port, reuse_address=True, backlog=256)
    # `uds`: the same routes on a Unix-domain socket as well (clients on the server's host - the dataloader workers of a
    # single-node job - skip the TCP stack: no checksums, no loopback MTU segmentation of multi-megabyte bodies)
    unix_server = None
    if uds:
        import os

        try:
            os.unlink(uds)
        except FileNotFoundError:
            pass
        unix_server = await loop.create_unix_server(lambda: _Connection(endpoints, pool, max_body), path=uds, backlog=256)
    if ready is not None:
        ready.set()
    try:
        if stop is None:
            await server.serve_forever()
        else:
            await stop.wait()
    finally:
        server.close()
        await server.wait_closed()
        if unix_server is not None:
            unix_server.close()
            await unix_server.wait_closed()
            try:
                import os

                os.unlink(uds)
            except OSError:
                pass
        pool.shutdown(wait=False, cancel_futures=True)


def run(endpoints, host: str, port: int, workers: int = 64, max_body: int = MAX_BODY, uds: "str | None" = None) -> None:
    """Blocking entry point (the server process's main thread): SIGTERM / SIGINT stop it cleanly."""
    import signal

    async def main() -> None:
        stop = asyncio.Event()
        loop = asyncio.get_running_loop()
        for sig in (signal.SIGTERM, signal.SIGINT):
            try:
                loop.add_signal_handler(sig, stop.set)
            except (NotImplementedError, RuntimeError):  # pragma: no cover - not the main thread
                pass
        await serve(endpoints, host, port, workers=workers, stop=stop, max_body=max_body, uds=uds)

    asyncio.run(main())

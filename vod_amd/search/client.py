"""`HipMipsClient` / `HipMipsMaster`: drop-in for `FaissClient` / `FaissMaster`
(/root/reference/src/vod_search/faiss_search/client.py:18-186).

Same constructor fields (`host`, `port`), same keyword-only `search()` that ignores
`text / subset_ids / ids / shard` (client.py:64-74), same `/fast-search` request and response encoding,
same `RetrievalBatch` result with `meta["time"]`.  The master spawns `python -m vod_amd.search.server`,
which owns the GPU-resident index; clients stay plain HTTP so they can be pickled into DataLoader workers.
"""
from __future__ import annotations

import os
import pathlib
import sys
import time
from copy import copy

import numpy as np
import requests

from vod_amd import io
from vod_amd import types as vt
from vod_amd.search import base
from vod_amd.search.socket import find_available_port


class HipMipsClient(base.SearchClient):
    """HTTP client of the HIP MIPS server."""

    requires_vectors = True

    def __init__(self, host: str = "http://localhost", port: int = 7678, binary: bool = False, forward_subset_ids: bool = False):
        self.host = host
        self.port = port
        self.binary = binary  # use the raw-bytes route (`/raw-search`) instead of base64-in-JSON (`/fast-search`)
        # The reference's faiss client drops `subset_ids`; set this to let the GPU index honour them (the server must
        # have been started with `--subset-ids-path`).
        self.forward_subset_ids = forward_subset_ids

    def __repr__(self) -> str:
        return f"{type(self).__name__}[{self.url}](requires_vectors={self.requires_vectors})"

    @property
    def url(self) -> str:
        return f"{self.host}:{self.port}"

    def ping(self, timeout: float = 120) -> bool:
        try:
            response = requests.get(f"{self.url}/", timeout=timeout)
        except requests.exceptions.ConnectionError:
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
        if self.binary and not (self.forward_subset_ids and subset_ids is not None):
            return self._search_binary(np.asarray(vector), top_k, timeout, start)
        extra: dict = {"top_k": top_k}
        if self.forward_subset_ids and subset_ids is not None:
            extra["subset_ids"] = [list(map(str, s)) for s in subset_ids]
        # the same JSON document `requests.post(json=...)` would send, assembled without encoding the 4 MB field
        body = io.json_body({"vectors": io.serialize_np_array(np.asarray(vector))}, extra)
        response = requests.post(f"{self.url}/fast-search", data=body, headers={"content-type": "application/json"}, timeout=timeout)
        try:
            response.raise_for_status()
        except requests.exceptions.HTTPError:
            try:
                print(response.json()["detail"], file=sys.stderr)
            except Exception:
                print(response.text, file=sys.stderr)
            raise
        data = io.parse_json_body(response.content, ("scores", "indices"))
        return vt.RetrievalBatch.cast(
            indices=io.deserialize_np_array(data["indices"]),
            scores=io.deserialize_np_array(data["scores"]),
            labels=None,
            meta={"time": time.time() - start},
        )


    def _search_binary(self, vector: np.ndarray, top_k: int, timeout: float, start: float) -> vt.RetrievalBatch:
        import io as _io

        buf = _io.BytesIO()
        np.save(buf, vector, allow_pickle=False)
        response = requests.post(f"{self.url}/raw-search", params={"top_k": top_k}, data=buf.getvalue(),
                                 headers={"content-type": "application/octet-stream"}, timeout=timeout)
        response.raise_for_status()
        nq, k = int(response.headers["x-nq"]), int(response.headers["x-k"])
        raw = response.content
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
        group_backend: str = "nccl",  # with `devices`: "nccl" (RCCL, one GPU per worker) or "gloo" (host-staged; workers may share a GPU)
    ):
        super().__init__(skip_setup=skip_setup, free_resources=free_resources)
        self.vectors_path = pathlib.Path(vectors_path)
        self.logging_level = logging_level
        self.host = host
        self.port = find_available_port() if port < 0 else port
        self.dtype = dtype
        self.device = device
        # `devices=[...]`: the store is row-sharded over these GPUs behind the SAME host:port (one worker process per
        # GPU on an RCCL group; what `FaissMaster(serve_on_gpu=True)` gets from faiss's index_cpu_to_all_gpus,
        # /root/reference/src/vod_search/faiss_search/client.py:118-137, server.py:51-54)
        self.devices = None if devices is None else [int(d) for d in devices]
        self.group_backend = group_backend

    def _make_env(self) -> dict[str, str]:
        env = copy(dict(os.environ))
        root = str(pathlib.Path(__file__).resolve().parents[2])
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        return env

    def _make_cmd(self) -> list[str]:
        return [
            sys.executable, "-m", "vod_amd.search.server",
            "--vectors-path", str(self.vectors_path.absolute()),
            "--host", str(self.host),
            "--port", str(self.port),
            "--logging-level", str(self.logging_level),
            "--dtype", self.dtype,
            *(["--devices", ",".join(map(str, self.devices)), "--group-backend", self.group_backend]
              if self.devices is not None else ["--device", str(self.device)]),
        ]

    def get_client(self) -> HipMipsClient:
        return HipMipsClient(host=self.host, port=self.port)

    @property
    def url(self) -> str:
        return f"{self.host}:{self.port}"

    @property
    def service_info(self) -> str:
        return f"HipMipsServer[{self.url}]"

    @property
    def service_name(self) -> str:
        return super().service_name + f"-{self.port}"

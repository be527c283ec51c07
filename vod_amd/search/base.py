"""`SearchClient` / `SearchMaster` contracts (mirror of /root/reference/src/vod_search/base.py:32-200).

A *client* is a small picklable object that talks to a server (it is shipped into DataLoader workers);
a *master* owns the server process: context manager, spawns with `subprocess.Popen`, polls `ping()` every
0.1 s for at most `_timeout` = 300 s, redirects the server's output to
`<service-name>.std{out,err}.log`, terminates the process on exit, and refuses to be pickled.
"""
from __future__ import annotations

import abc
import copy
import logging
import os
import pathlib
import subprocess
import time
import typing as typ

import numpy as np

from vod_amd import types as vt

logger = logging.getLogger("vod_amd.search")

ShardName = str
SubsetId = str
SectionId = str


def _camel_to_snake(name: str) -> str:
    return "".join("_" + c.lower() if c.isupper() else c for c in name).lstrip("_")


class DoNotPickleError(Exception):
    def __init__(self, msg: str | None = None):
        super().__init__(msg or "This object cannot be pickled.")


class SearchClient(abc.ABC):
    """Talks to a search server.  `requires_vectors` tells the caller to pass `vector=`."""

    requires_vectors: bool = True

    def __repr__(self) -> str:
        return f"{type(self).__name__}(requires_vectors={self.requires_vectors})"

    @abc.abstractmethod
    def ping(self) -> bool:
        ...

    @abc.abstractmethod
    def search(
        self,
        *,
        text: list[str],
        vector: None | np.ndarray = None,
        subset_ids: None | list[list[SubsetId]] = None,
        ids: None | list[list[SectionId]] = None,
        shard: None | list[ShardName] = None,
        top_k: int = 3,
    ) -> vt.RetrievalBatch:
        ...

    async def async_search(self, *, text, vector=None, subset_ids=None, ids=None, shard=None, top_k: int = 3) -> vt.RetrievalBatch:
        """Default: delegate to the blocking `search` (base.py:59-77)."""
        return self.search(text=text, vector=vector, subset_ids=subset_ids, ids=ids, shard=shard, top_k=top_k)


Sc = typ.TypeVar("Sc", bound=SearchClient)


class SearchMaster(typ.Generic[Sc], abc.ABC):
    """Owns a search server process for the duration of a `with` block."""

    _timeout: float = 300
    _server_proc: None | subprocess.Popen = None
    _allow_existing_server: bool = False

    def __init__(self, skip_setup: bool = False, free_resources: bool = False) -> None:
        self.skip_setup = skip_setup
        self.free_resources = free_resources

    def __enter__(self):
        if self.free_resources:
            self._free_resources()
        if not self.skip_setup:
            self._setup()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        self._on_exit()
        if self._server_proc is not None:
            self._server_proc.terminate()
            try:
                self._server_proc.wait(timeout=30)
            except subprocess.TimeoutExpired:  # pragma: no cover
                self._server_proc.kill()
            self._server_proc = None

    def _setup(self) -> None:
        self._server_proc = self._start_server()
        self._on_init()

    def _free_resources(self) -> None:
        pass

    def _on_init(self) -> None:
        pass

    def _on_exit(self) -> None:
        pass

    @abc.abstractmethod
    def get_client(self) -> Sc:
        ...

    @abc.abstractmethod
    def _make_cmd(self) -> list[str]:
        ...

    def _make_env(self) -> dict[str, str]:
        return copy.copy(dict(os.environ))

    def _start_server(self) -> None | subprocess.Popen:
        client = self.get_client()
        if client.ping():
            if self._allow_existing_server:
                logger.debug("connecting to existing %s", self.service_info)
                return None
            raise RuntimeError(f"Server {self.service_name} is already running.")
        stdout_file = pathlib.Path(f"{self.service_name}.stdout.log")
        stderr_file = pathlib.Path(f"{self.service_name}.stderr.log")
        for f in (stdout_file, stderr_file):
            if f.exists():
                f.unlink()
        proc = subprocess.Popen(  # noqa: S603
            self._make_cmd(), env=self._make_env(), stdout=stdout_file.open("w"), stderr=stderr_file.open("w")
        )
        t0 = time.time()
        logger.info("spawning %s ...", self.service_info)
        while not client.ping():
            time.sleep(0.1)
            if proc.poll() is not None:
                tail = stderr_file.read_text()[-2000:] if stderr_file.exists() else ""
                raise RuntimeError(f"{self.service_info} exited with code {proc.returncode} during start-up:\n{tail}")
            if time.time() - t0 > self._timeout:
                proc.terminate()
                raise TimeoutError(f"Couldn't ping the server after {self._timeout:.0f}s.")
        logger.debug("spawned %s in %.1fs", self.service_info, time.time() - t0)
        return proc

    @property
    def service_name(self) -> str:
        return _camel_to_snake(type(self).__name__)

    @property
    def service_info(self) -> str:
        return self.service_name

    def __getstate__(self):
        raise DoNotPickleError(
            f"{type(self).__name__} is not pickleable. To use in multiprocessing, use a client instead (`master.get_client()`)."
        )

    def __setstate__(self, state):
        raise DoNotPickleError(f"{type(self).__name__} is not pickleable.")

"""One search over N processes, each owning a row shard of the store on its own GPU.

Counterpart of what the reference's server gets from `faiss.index_cpu_to_all_gpus(index, co)` with
`co.shard = True` (/root/reference/src/vod_search/faiss_search/server.py:51-54, src/vod_configs/search.py:80):
ONE server address answers for an index sharded over every GPU of the node.  faiss does it with threads inside one
process (`IndexShards`); here it is one process per GPU on an RCCL group (MI355X-native process model): rank 0
owns the HTTP endpoint, broadcasts every request (a 6-word header, the queries, optionally the subset labels), all
ranks run the same `ShardedFlatIndex.search` (local fused top-k -> one packed all-gather -> merge) and rank 0
answers.  Ranks > 0 sit in `worker_loop()` until rank 0 broadcasts the stop word.

The collective sequence is identical on every rank by construction (header, payload, search); rank 0 must
serialise its callers (the server's lock does).  `device` is where the broadcast tensors live: the rank's GPU
under RCCL, the CPU under gloo (tests, `--group-backend gloo`); `search_device` is where the shard lives.
"""
from __future__ import annotations

import logging

import numpy as np
import torch
import torch.distributed as dist

OP_STOP, OP_SEARCH = 0, 1
MAX_K = 2048              # VODHIP_MAX_K (include/vodhip.h)
MAX_SUBSET_LABELS = 64    # vodhip_index_set_query_labels


class GroupDispatcher:
    def __init__(self, sharded, rank: int, world: int, device: torch.device, group: dist.ProcessGroup | None = None,
                 search_device: torch.device | None = None, dim: int | None = None):
        self.dim = dim  # query dimension of the store, checked on rank 0 before a request is broadcast
        self.errors = 0
        self.sharded = sharded  # ShardedFlatIndex-like: .search(queries, k, subset=None) -> (scores, ids), collective
        self.rank, self.world, self.device, self.group = rank, world, device, group
        # where the shard lives when that is not where the broadcast tensors live (gloo group in front of GPU shards)
        self.search_device = device if search_device is None else search_device

    # -- rank 0 ------------------------------------------------------------------------------------
    def search(self, query_vec: np.ndarray, top_k: int, subset: np.ndarray | None = None) -> tuple[np.ndarray, np.ndarray]:
        assert self.rank == 0, "only rank 0 drives the group"
        # Everything that can fail is checked HERE, before the first broadcast: a request the library would refuse
        # (k outside [1, VODHIP_MAX_K], more than 64 subset labels per query, a malformed batch) raises on rank 0 alone and
        # becomes this request's HTTP 500 - the workers never see it and the collective sequence stays in step.
        query_vec = np.asarray(query_vec)
        if query_vec.ndim != 2:
            raise ValueError(f"Expected 2D array, got {query_vec.ndim}D array")
        top_k = int(top_k)
        if not 1 <= top_k <= MAX_K:
            raise ValueError(f"top_k={top_k} out of range [1, {MAX_K}]")
        if self.dim is not None and query_vec.shape[1] != self.dim:
            raise ValueError(f"query dimension {query_vec.shape[1]} != index dimension {self.dim}")
        if subset is not None:
            subset = np.asarray(subset)
            if subset.ndim != 2 or subset.shape[0] != query_vec.shape[0]:
                raise ValueError(f"expected subset labels of shape [{query_vec.shape[0]}, S], got {subset.shape}")
            if not 1 <= subset.shape[1] <= MAX_SUBSET_LABELS:
                raise ValueError(f"{subset.shape[1]} subset labels per query: the limit is {MAX_SUBSET_LABELS}")
        if query_vec.shape[0] == 0:  # nothing to search: answered locally, no collective
            return np.empty((0, top_k), dtype=np.float32), np.empty((0, top_k), dtype=np.int64)
        q = torch.from_numpy(np.ascontiguousarray(query_vec, dtype=np.float32)).to(self.device)
        sub = None
        if subset is not None:
            sub = torch.from_numpy(np.ascontiguousarray(subset, dtype=np.int32)).to(self.device)
        header = torch.tensor([OP_SEARCH, q.shape[0], q.shape[1], int(top_k), 0 if sub is None else sub.shape[1], 0],
                              dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.broadcast(header, 0, group=self.group)
            dist.broadcast(q, 0, group=self.group)
            if sub is not None:
                dist.broadcast(sub, 0, group=self.group)
        try:
            scores, ids = self._local(q, int(top_k), sub)
        except Exception as exc:
            if self.world > 1 and not self._same_on_every_rank(exc):
                self._fatal(exc)
            raise
        return scores.cpu().numpy(), ids.cpu().numpy()

    def _same_on_every_rank(self, exc: BaseException) -> bool:
        """An argument error raised BEFORE the search's collective is raised by every rank alike (same request, same checks): the
        ranks stay in step and the request becomes an HTTP 500.  Anything else - a HIP / RCCL failure, an out-of-memory on one shard,
        any error after the all-gather was entered - is one-sided: the other ranks are (or will be) blocked in a collective this rank
        skipped."""
        return isinstance(exc, (ValueError, TypeError)) and not getattr(self.sharded, "entered_collective", False)

    def _fatal(self, exc: BaseException) -> None:
        """Fail fast: leave non-zero so that the owner process sees a dead rank, terminates the group and reports it
        (`server.run_owner`) - instead of every later request hanging until the client's timeout."""
        import os

        logging.getLogger(__name__).critical("rank %d: one-sided failure inside a group search (%s: %s): terminating the group",
                                             self.rank, type(exc).__name__, exc, exc_info=exc)
        logging.shutdown()
        os._exit(3)

    def _local(self, q: torch.Tensor, k: int, sub: torch.Tensor | None) -> tuple[torch.Tensor, torch.Tensor]:
        if self.search_device != self.device:
            q = q.to(self.search_device)
            sub = None if sub is None else sub.to(self.search_device)
        return self.sharded.search(q, k, subset=sub)

    def stop(self) -> None:
        if self.rank == 0 and self.world > 1:
            dist.broadcast(torch.tensor([OP_STOP, 0, 0, 0, 0, 0], dtype=torch.int64, device=self.device), 0, group=self.group)

    # -- ranks > 0 ---------------------------------------------------------------------------------
    def worker_loop(self) -> int:
        """Serve requests until rank 0 says stop.  Returns the number of searches served."""
        assert self.rank != 0
        served = 0
        while True:
            header = torch.zeros(6, dtype=torch.int64, device=self.device)
            dist.broadcast(header, 0, group=self.group)
            op, nq, dim, k, n_sub, _ = (int(v) for v in header.cpu())
            if op == OP_STOP:
                return served
            q = torch.empty((nq, dim), dtype=torch.float32, device=self.device)
            dist.broadcast(q, 0, group=self.group)
            sub = None
            if n_sub:
                sub = torch.empty((nq, n_sub), dtype=torch.int32, device=self.device)
                dist.broadcast(sub, 0, group=self.group)
            try:
                self._local(q, k, sub)
            except Exception as exc:  # noqa: BLE001
                # rank 0 validated the request before broadcasting it, so what may be left are argument errors that every
                # rank raises alike BEFORE its collective (rank 0 turns its own into that request's HTTP 500): log and keep
                # serving - one bad request must not take the group down.  A one-sided failure must: see `_fatal`
                if not self._same_on_every_rank(exc):
                    self._fatal(exc)
                logging.getLogger(__name__).exception("rank %d: search failed; still serving", self.rank)
                self.errors += 1
            served += 1

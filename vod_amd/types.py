"""Result containers at the search boundary: `RetrievalBatch` / `RetrievalSample` / `RetrievalTuple`.

Field-, shape- and dtype-compatible with the reference's containers
(/root/reference/src/vod_types/retrieval.py:18-315): `scores` float [nq, k], `indices` int64 [nq, k],
optional `labels`, a free-form `meta` dict; padding convention index -1 / score -inf (:284-285).
Only what the hot path touches is provided: `cast`, indexing / iteration, `__mul__` (weighting, :222-233),
`sorted` (:208-220), `__add__` / `concatenate_batches` (:198-206, :240-249) and `stack_samples` (:235-238,
:259-287) which right-pads ragged rows.
"""
from __future__ import annotations

import copy
import math
import typing as typ
import warnings
from numbers import Number

import numpy as np

try:  # torch is optional for the containers themselves
    import torch
except Exception:  # pragma: no cover
    torch = None  # type: ignore[assignment]


def _to_numpy(x: typ.Any) -> np.ndarray:
    if torch is not None and isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _describe(x: np.ndarray) -> str:
    return f"{type(x).__name__}(shape={x.shape}, dtype={x.dtype})"


class RetrievalData:
    """Scores / indices / labels of one or many search results; `_ndim` fixes the expected rank of `scores`."""

    __slots__ = ("scores", "indices", "labels", "meta", "allow_unsafe")
    _ndim: int = -1
    _sep: str = ""

    def __init__(self, scores, indices, labels=None, meta: dict | None = None, allow_unsafe: bool = False):
        nd = len(indices.shape)
        # shapes are only compared up to the rank of `indices` (merged batches may carry extra score dims)
        if not allow_unsafe and scores.shape[:nd] != indices.shape[:nd]:
            raise ValueError(f"`scores` and `indices` shapes differ: {_describe(scores)} vs {_describe(indices)}")
        if labels is not None and scores.shape[:nd] != labels.shape[:nd]:
            raise ValueError(f"`scores` and `labels` shapes differ: {_describe(scores)} vs {_describe(labels)}")
        if len(scores.shape) != self._ndim:
            raise ValueError(f"{type(self).__name__} expects {self._ndim}-D scores, got {_describe(scores)}")
        self.scores = scores
        self.indices = indices
        self.labels = labels
        self.meta = meta or {}
        self.allow_unsafe = allow_unsafe

    @classmethod
    def cast(cls, scores, indices, labels=None, meta: dict | None = None, allow_unsafe: bool = False):
        """Build from array-likes / tensors (moved to host NumPy)."""
        return cls(
            scores=_to_numpy(scores),
            indices=_to_numpy(indices),
            labels=None if labels is None else _to_numpy(labels),
            meta=meta,
            allow_unsafe=allow_unsafe,
        )

    def __len__(self) -> int:
        return len(self.scores)

    @property
    def shape(self) -> tuple[int, ...]:
        return tuple(self.scores.shape)

    def __repr__(self) -> str:
        parts = [
            f"{type(self).__name__}[{type(self.scores).__name__}](",
            f"scores={self.scores!r}, ",
            f"indices={self.indices!r}, ",
            f"labels={self.labels!r}, ",
            f"meta={self.meta!r}",
        ]
        return self._sep.join(parts) + ")"

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, type(self)):
            raise NotImplementedError(f"cannot compare {type(self)} with {type(other)}")
        return bool(np.all(self.scores == other.scores) and np.all(self.indices == other.indices))

    __hash__ = None  # type: ignore[assignment]

    def to_dict(self) -> dict[str, typ.Any]:
        return {
            "scores": self.scores.tolist(),
            "indices": self.indices.tolist(),
            "labels": None if self.labels is None else self.labels.tolist(),
        }


class RetrievalTuple(RetrievalData):
    """One (score, index, label) hit."""

    _ndim = 0

    def __getitem__(self, item):
        raise NotImplementedError("RetrievalTuple is not indexable")

    def __iter__(self):
        raise NotImplementedError("RetrievalTuple is not iterable")


class RetrievalSample(RetrievalData):
    """The hits of one query."""

    _ndim = 1

    def __getitem__(self, item: int) -> RetrievalTuple:
        return RetrievalTuple(
            scores=self.scores[item],
            indices=self.indices[item],
            labels=None if self.labels is None else self.labels[item],
        )

    def __iter__(self) -> typ.Iterator[RetrievalTuple]:
        return (self[i] for i in range(len(self)))

    def __add__(self, other: "RetrievalSample") -> "RetrievalBatch":
        return RetrievalBatch.stack_samples([self, other])


class RetrievalBatch(RetrievalData):
    """The hits of a batch of queries: `scores[nq, k]`, `indices[nq, k]`."""

    _ndim = 2
    _sep = "\n"

    def __getitem__(self, item: int) -> RetrievalSample:
        return RetrievalSample(
            scores=self.scores[item],
            indices=self.indices[item],
            labels=None if self.labels is None else self.labels[item],
        )

    def __iter__(self) -> typ.Iterator[RetrievalSample]:
        return (self[i] for i in range(len(self)))

    def __add__(self, other: "RetrievalBatch") -> "RetrievalBatch":
        """Concatenate along the query dimension; a missing label array becomes -1."""
        la, lb = self.labels, other.labels
        if la is None and lb is None:
            labels = None
        else:
            la = np.full_like(lb, -1) if la is None else la
            lb = np.full_like(la, -1) if lb is None else lb
            labels = np.concatenate([la, lb])
        return RetrievalBatch(
            scores=np.concatenate([self.scores, other.scores]),
            indices=np.concatenate([self.indices, other.indices]),
            labels=labels,
        )

    def __mul__(self, value: float) -> "RetrievalBatch":
        """Weight the scores (0 * -inf -> NaN is expected and silenced, as in the reference)."""
        if not isinstance(value, Number):
            raise TypeError(f"expected a number, got {type(value)}")
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore", RuntimeWarning)
            return RetrievalBatch(scores=self.scores * value, indices=self.indices, labels=self.labels, meta=copy.copy(self.meta))

    def sorted(self) -> "RetrievalBatch":
        """Rows re-ordered by descending score (the flip of an ascending argsort, as the reference does)."""
        order = np.flip(np.argsort(self.scores, axis=-1), axis=-1)
        take = lambda a: np.take_along_axis(a, order, axis=-1)  # noqa: E731
        return RetrievalBatch(
            scores=take(self.scores),
            indices=take(self.indices),
            labels=None if self.labels is None else take(self.labels),
            meta=copy.copy(self.meta),
        )

    @classmethod
    def stack_samples(cls, samples: typ.Iterable[RetrievalSample]) -> "RetrievalBatch":
        """Stack ragged per-query results, right-padding with score -inf / index -1 / label -1."""
        samples = list(samples)
        labels = [s.labels for s in samples]
        return RetrievalBatch(
            scores=_stack_ragged([s.scores for s in samples], -math.inf),
            indices=_stack_ragged([s.indices for s in samples], -1),
            labels=None if any(lab is None for lab in labels) else _stack_ragged(labels, -1),
        )

    @classmethod
    def concatenate_batches(cls, batches: typ.Iterable["RetrievalBatch"]) -> "RetrievalBatch":
        out = None
        for b in batches:
            out = b if out is None else out + b
        if out is None:
            raise ValueError("cannot concatenate an empty list of batches")
        return out


def _stack_ragged(rows: list[np.ndarray], fill: typ.Any) -> np.ndarray:
    if not isinstance(rows, list) or not all(isinstance(r, np.ndarray) for r in rows):
        raise TypeError("expected a list of numpy arrays")
    width = max(len(r) for r in rows)
    out = np.full((len(rows), width), fill, dtype=rows[0].dtype)
    for j, r in enumerate(rows):
        out[j, : len(r)] = r
    return out

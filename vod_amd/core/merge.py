"""Hybrid score merge on the GPU (host wrapper over `vodhip_merge_hybrid`).

Replaces the numba path `_merge_search_results` -> `normalize_search_scores_` -> `merge_search_results`
-> `gather_values_by_indices` (/root/reference/src/vod_dataloaders/core/search.py:79-125,
normalize.py:6-20, merge.py:8-164, numpy_ops.py:24-143) with one kernel launch; output layout is the
reference's, bit for bit: ids in first-seen order (lookup, then each engine), scores
`sum_e w_e * (s_e - rowmin_e)` in float32, one trailing pad column, labels from the lookup (-1 when
absent), raw per-engine scores min-subtracted with NaN for "not returned by this engine".
"""
from __future__ import annotations

import ctypes
import dataclasses

import numpy as np
import torch

from vod_amd import _native


@dataclasses.dataclass
class MergedOnDevice:
    """`vodhip_merge_hybrid`'s outputs as they lie in HBM: full-stride rows + the per-stage cursor maxima.

    The reference cuts its buffer to `[: max_cursor + 1]` after every pairwise fold (merge.py:160-162); that width is a
    function of `stage_max` (device int32) and the static engine widths.  Consumers on the device
    (`vodhip_priority_sample_merged`) derive it there; `width()` reads it back (a host sync) for the NumPy wrappers."""

    indices: torch.Tensor                 # int64 [nq, stride]
    scores: torch.Tensor                  # float32 [nq, stride]
    labels: torch.Tensor | None           # int64 [nq, stride]
    raw: dict[str, torch.Tensor]          # float32 [nq, stride] per scored engine
    stage_max: torch.Tensor | None        # int32 [MAX_ENGINES] cursor maxima over rows, or None
    k_lookup: int
    engine_k: list[int]
    row_cursor: torch.Tensor | None = None  # int32 [nq, MAX_ENGINES] cursors per row (the consumer takes the maximum), or None

    @property
    def stride(self) -> int:
        return int(self.indices.shape[1])

    def width(self) -> int:
        if self.stage_max is None and self.row_cursor is None:
            return self.k_lookup
        maxima = (self.stage_max if self.stage_max is not None else self.row_cursor.amax(dim=0)).tolist()
        w = self.k_lookup
        for k_e, m in zip(self.engine_k, maxima):
            w = min(m + 1, w + k_e)
        return w

    def cut(self) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor | None, dict[str, torch.Tensor]]:
        w = self.width()
        return (self.indices[:, :w], self.scores[:, :w], None if self.labels is None else self.labels[:, :w],
                {n: t[:, :w] for n, t in self.raw.items()})


def merge_hybrid_device(
    lookup_idx: torch.Tensor,
    lookup_lbl: torch.Tensor | None,
    engines: dict[str, tuple[torch.Tensor, torch.Tensor]],
    weights: dict[str, float],
) -> MergedOnDevice:
    """Device-tensor API without any host synchronisation: ONE launch, outputs left at full stride.

    lookup_idx int64 [nq, kl]; engines[name] = (idx int64 [nq, k], scores f32 [nq, k])."""
    lib = _native.load_library()
    dev = lookup_idx.device
    if dev.type != "cuda":
        raise _native.NativeLibraryError("merge_hybrid_device needs device tensors (there is no CPU path)")
    names = list(engines)
    if len(names) > _native.MAX_ENGINES:
        raise ValueError(f"at most {_native.MAX_ENGINES} scored engines are supported, got {len(names)}")
    nq, kl = lookup_idx.shape
    if not names:
        # a single result set is returned as is, weighted (merge.py:18-22): no union, no pad column
        zeros = torch.zeros((nq, kl), dtype=torch.float32, device=dev)
        return MergedOnDevice(lookup_idx, zeros, lookup_lbl, {}, None, int(kl), [])
    lookup_idx = lookup_idx.contiguous().long()
    if lookup_lbl is not None:
        lookup_lbl = lookup_lbl.contiguous().long()
    e_idx = [engines[n][0].contiguous().long() for n in names]
    e_scr = [engines[n][1].contiguous().float() for n in names]
    for n, i, s in zip(names, e_idx, e_scr):
        if i.shape != s.shape or i.shape[0] != nq:
            raise ValueError(f"engine `{n}`: indices {tuple(i.shape)} / scores {tuple(s.shape)} do not match nq={nq}")
    ks = [int(i.shape[1]) for i in e_idx]
    stride = kl + sum(ks) + 1
    out_idx = torch.empty((nq, stride), dtype=torch.int64, device=dev)
    out_scr = torch.empty((nq, stride), dtype=torch.float32, device=dev)
    out_lbl = torch.empty((nq, stride), dtype=torch.int64, device=dev) if lookup_lbl is not None else None
    out_raw = [torch.empty((nq, stride), dtype=torch.float32, device=dev) for _ in names]
    # the per-stage cursors that fix the reference's cut width: per row (plain stores, ONE launch; the sampling kernel takes
    # the maximum) - or, for very tall batches, the maxima themselves (cleared by the library on the stream + atomics)
    per_row = nq <= 4096
    cursors = torch.empty((nq, _native.MAX_ENGINES) if per_row else (_native.MAX_ENGINES,), dtype=torch.int32, device=dev)

    n_e = len(names)
    VP = ctypes.c_void_p
    arr_idx = (VP * max(n_e, 1))(*[t.data_ptr() for t in e_idx])
    arr_scr = (VP * max(n_e, 1))(*[t.data_ptr() for t in e_scr])
    arr_raw = (VP * max(n_e, 1))(*[t.data_ptr() for t in out_raw])
    arr_k = (ctypes.c_int * max(n_e, 1))(*ks)
    arr_w = (ctypes.c_float * max(n_e, 1))(*[float(weights[n]) for n in names])
    with torch.cuda.device(dev):
        _native.check(
            lib.vodhip_merge_hybrid(
                lookup_idx.data_ptr(), lookup_lbl.data_ptr() if lookup_lbl is not None else None, kl, n_e,
                arr_idx, arr_scr, arr_k, arr_w, nq,
                out_idx.data_ptr(), out_scr.data_ptr(), out_lbl.data_ptr() if out_lbl is not None else None,
                arr_raw, stride, None if per_row else cursors.data_ptr(), cursors.data_ptr() if per_row else None,
                _native.current_stream_ptr(dev),
            )
        )
    return MergedOnDevice(out_idx, out_scr, out_lbl, dict(zip(names, out_raw)), None if per_row else cursors, int(kl), ks,
                          cursors if per_row else None)


def merge_hybrid_tensors(
    lookup_idx: torch.Tensor,
    lookup_lbl: torch.Tensor | None,
    engines: dict[str, tuple[torch.Tensor, torch.Tensor]],
    weights: dict[str, float],
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor | None, dict[str, torch.Tensor]]:
    """Device-tensor API.  Returns (indices, scores, labels, raw_scores) already cut to the reference's width
    (`[: max_cursor + 1]` after every pairwise fold, merge.py:160-162): reading the width back is one host sync -
    `merge_hybrid_device` / `vod_amd.core.collate.collate_on_device` avoid it."""
    return merge_hybrid_device(lookup_idx, lookup_lbl, engines, weights).cut()


def merge_hybrid(
    lookup_idx: np.ndarray,
    lookup_lbl: np.ndarray | None,
    engines: dict[str, tuple[np.ndarray, np.ndarray]],
    weights: dict[str, float],
    device: int | torch.device = 0,
) -> tuple[np.ndarray, np.ndarray, np.ndarray | None, dict[str, np.ndarray]]:
    """NumPy API (what the collate sees): copies the small inputs to the GPU, merges, copies back."""
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    lbl_dtype = None if lookup_lbl is None else lookup_lbl.dtype
    idx, scr, lbl, raw = merge_hybrid_tensors(
        t(lookup_idx.astype(np.int64, copy=False)),
        None if lookup_lbl is None else t(lookup_lbl.astype(np.int64)),
        {n: (t(i.astype(np.int64, copy=False)), t(s.astype(np.float32, copy=False))) for n, (i, s) in engines.items()},
        weights,
    )
    out_lbl = None if lbl is None else lbl.cpu().numpy().astype(lbl_dtype, copy=False)
    return idx.cpu().numpy(), scr.cpu().numpy(), out_lbl, {n: r.cpu().numpy() for n, r in raw.items()}

"""The collate-side chain merge -> priority sampling -> (optional) in-batch flattening, resident on the GPU.

Reference flow (/root/reference/src/vod_dataloaders/realm_collate.py:110-139): `_merge_search_results`
(core/search.py:79-125) -> `sample_search_results` (core/sample.py:22-84) -> `flatten_samples`
(core/in_batch_negatives.py:10-52) -> `_samples_to_dict` (realm_collate.py:247-278), every stage a NumPy array on the
host.  SURVEY 8(f) row 2 asks for the `section__*` tensors "directly as torch device tensors": here the three stages are
three kernel launches on the caller's stream with NO host synchronisation in between -

  1. `vodhip_merge_hybrid`            full-stride union rows + per-stage cursor maxima (the reference's cut width stays
                                      on the device),
  2. `vodhip_priority_sample_merged`  derives the width on the device, samples, and gathers ids / scores / raw engine
                                      scores + the rank diagnostic in its epilogue,
  3. `vodhip_flatten_inbatch`         (in-batch negatives) sorted distinct ids padded with the reference's 1s + the
                                      per-row gather of every value array.

The NumPy drop-in functions (`vod_amd.core.{merge,sample,in_batch_negatives}`) run the same kernels and cut / copy on the
host; this module is for callers that keep the batch on the device (e.g. a trainer that feeds `RetrievalGradients`).
"""
from __future__ import annotations

import ctypes
import dataclasses

import torch

from vod_amd import _native
from vod_amd.core.merge import MergedOnDevice, merge_hybrid_device
from vod_amd.core.sample import support_flag


@dataclasses.dataclass
class DeviceSampledSections:
    """`PrioritySampledSections` (sample.py:12-20) with device tensors.  After flattening `indices` is 1-D [B * n]."""

    indices: torch.Tensor            # int64 [B, n] | [U]
    scores: torch.Tensor             # float32 [B, n] | [B, U]
    labels: torch.Tensor             # bool, same shape as scores
    log_weights: torch.Tensor        # float32, same shape as scores
    lse_pos: torch.Tensor            # float32 [B]
    lse_neg: torch.Tensor            # float32 [B]
    max_sampling_id: torch.Tensor    # float32 [B]
    raw_scores: dict[str, torch.Tensor]
    local_ids: torch.Tensor | None = None   # int64 [B, n] column of the merged row each sample was taken from (-1 = pad)
    n_unique: torch.Tensor | None = None    # int32 [1] distinct ids of the batch (flattened only)

    def to_dict(self, prefix: str = "", relevances: torch.Tensor | None = None) -> dict[str, torch.Tensor]:
        """The `section__*` fields of `_samples_to_dict` (realm_collate.py:247-278), as device tensors."""
        out = {
            f"{prefix}idx": self.indices,
            f"{prefix}score": self.scores,
            f"{prefix}label": self.labels,
            f"{prefix}log_weight": self.log_weights,
            f"{prefix}lse_pos": self.lse_pos,
            f"{prefix}lse_neg": self.lse_neg,
            **{f"{prefix}{k}": v for k, v in self.raw_scores.items()},
        }
        if relevances is not None:
            out[f"{prefix}relevance"] = relevances
        return out


_WORKSPACES: dict[tuple, tuple[torch.Tensor, torch.Tensor]] = {}  # (device, stream, nq, stride, engines) -> merged-row buffers


def _ptr_array(tensors: list[torch.Tensor]):
    return (ctypes.c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])


def sample_merged_on_device(
    merged: MergedOnDevice,
    noise: torch.Tensor,
    *,
    total: int | None,
    max_pos_sections: int | None,
    temperature: float = 1.0,
    max_support_size: int | None = None,
    width: int | None = None,
    support: str = "reference",
) -> DeviceSampledSections:
    """`sample_search_results` (sample.py:22-84) on the merge's device outputs: ONE launch, no host sync.

    `noise`: float32 Exp(1) draws, one row per query, at least as wide as the merged rows (the reference draws
    `np.random.exponential(size=scores.shape)` on the host, sample.py:398; a device pipeline draws
    `torch.empty(nq, merged.stride, device=...).exponential_()`).  `width`: columns in use if already known on the host
    (None: derived on the device from the merge's cursor maxima - the reference's `[: max_cursor + 1]` cut)."""
    lib = _native.load_library()
    dev = merged.indices.device
    nq, stride = merged.indices.shape
    if width is None and merged.stage_max is None and merged.row_cursor is None:
        width = merged.k_lookup
    if total is None:
        # the reference defaults to the merged width (sample.py:38), which would need the device value here
        raise ValueError("`total` (the number of sections to sample) must be given for the device-resident pipeline")
    total = int(total)
    k_pos = int(max_pos_sections or total)
    max_support = int(max_support_size or -1)
    if max_support >= 0:
        max_support = max(max_support, total)  # sample.py:126-128
    labels = merged.labels if merged.labels is not None else torch.zeros((nq, stride), dtype=torch.int64, device=dev)
    names = list(merged.raw)
    raws = [merged.raw[n] for n in names]
    nz = noise if (noise.dtype == torch.float32 and noise.stride(-1) == 1) else noise.float().contiguous()
    if nz.shape[0] != nq or nz.shape[1] < (stride if width is None else width):
        raise ValueError(f"noise of shape {tuple(nz.shape)} does not cover the merged rows [{nq}, {stride if width is None else width}]")
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    samples = torch.empty((nq, total), dtype=torch.int64, device=dev)
    out_ids = torch.empty((nq, total), dtype=torch.int64, device=dev)
    out_scores, out_logw, lse, max_id = f32(nq, total), f32(nq, total), f32(nq, 2), f32(nq)
    out_lab = torch.empty((nq, total), dtype=torch.uint8, device=dev)
    out_raw = [f32(nq, total) for _ in names]
    eng_k = (ctypes.c_int * 4)(*(merged.engine_k + [0] * (4 - len(merged.engine_k))))
    with torch.cuda.device(dev):
        _native.check(
            lib.vodhip_priority_sample_merged(
                merged.indices.data_ptr(), merged.scores.data_ptr(), labels.data_ptr(), len(names), _ptr_array(raws),
                nz.data_ptr(), int(nz.stride(0)), nq, stride,
                -1 if width is None else int(width), None if merged.stage_max is None else merged.stage_max.data_ptr(),
                None if merged.row_cursor is None else merged.row_cursor.data_ptr(), merged.k_lookup, len(merged.engine_k), eng_k,
                k_pos, total, float(temperature), max_support, 1 | support_flag(support),
                samples.data_ptr(), out_ids.data_ptr(), out_scores.data_ptr(), out_logw.data_ptr(), out_lab.data_ptr(),
                _ptr_array(out_raw), lse.data_ptr(), max_id.data_ptr(), _native.current_stream_ptr(dev),
            )
        )
    return DeviceSampledSections(
        indices=out_ids, scores=out_scores, labels=out_lab.view(torch.bool), log_weights=out_logw, lse_pos=lse[:, 0], lse_neg=lse[:, 1],
        max_sampling_id=max_id, raw_scores=dict(zip(names, out_raw)), local_ids=samples,
    )


def flatten_on_device(samples: DeviceSampledSections) -> DeviceSampledSections:
    """`flatten_samples(samples, padding=True)` (in_batch_negatives.py:10-52) in ONE launch, no host sync: the batch's sorted
    distinct ids padded to B * n entries with the reference's constant 1, every value array gathered onto that list."""
    lib = _native.load_library()
    ids = samples.indices.contiguous()
    dev = ids.device
    B, n = ids.shape
    U = B * n
    names = ["scores", "log_weights", *samples.raw_scores]
    f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()  # noqa: E731
    values = [f32(samples.scores), f32(samples.log_weights), *[f32(v) for v in samples.raw_scores.values()]]
    fills = [float("nan")] * len(values)
    if len(values) > 8:
        raise ValueError("at most 6 raw score arrays can be flattened in one launch")
    lab = samples.labels
    lab = lab.view(torch.uint8) if lab.dtype == torch.bool else (lab > 0).view(torch.uint8)  # bool labels are gathered as they lie
    lab = lab if lab.is_contiguous() else lab.contiguous()
    outs = [torch.empty((B, U), dtype=torch.float32, device=dev) for _ in values]
    out_lab = torch.empty((B, U), dtype=torch.uint8, device=dev)
    unique = torch.empty((U,), dtype=torch.int64, device=dev)
    n_unique = torch.empty((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _native.check(
            lib.vodhip_flatten_inbatch(ids.data_ptr(), B, n, len(values), _ptr_array(values), (ctypes.c_float * len(values))(*fills),
                                       _ptr_array(outs), lab.data_ptr(), out_lab.data_ptr(), unique.data_ptr(), n_unique.data_ptr(),
                                       _native.current_stream_ptr(dev))
        )
    got = dict(zip(names, outs))
    return DeviceSampledSections(
        indices=unique, scores=got["scores"], labels=out_lab.view(torch.bool), log_weights=got["log_weights"], lse_pos=samples.lse_pos,
        lse_neg=samples.lse_neg, max_sampling_id=samples.max_sampling_id, raw_scores={k: got[k] for k in samples.raw_scores},
        local_ids=None, n_unique=n_unique,
    )


def collate_on_device(
    lookup_idx: torch.Tensor,
    lookup_lbl: torch.Tensor | None,
    engines: dict[str, tuple[torch.Tensor, torch.Tensor]],
    weights: dict[str, float],
    noise: torch.Tensor | None = None,
    *,
    total: int,
    max_pos_sections: int | None = None,
    temperature: float = 1.0,
    max_support_size: int | None = None,
    in_batch_negatives: bool = False,
    generator: torch.Generator | None = None,
    support: str = "reference",
) -> DeviceSampledSections:
    """merge -> sample -> (flatten) without leaving the GPU: ONE call into libvodhip (`vodhip_collate`) that enqueues the 2 (3)
    launches back to back on the current stream, zero host syncs.

    lookup_idx int64 [B, k_lookup] and lookup_lbl int64 [B, k_lookup] (the lookup engine's hits and labels; its scores are
    discarded, search.py:92); engines[name] = (ids int64 [B, k], scores float32 [B, k]); weights[name]; `noise` float32
    [B, >= k_lookup + sum(k) + 1] Exp(1) draws (None: drawn on the device with `generator`).
    Returns device tensors; `.to_dict("section__")` gives the fields `_samples_to_dict` hands to the model."""
    names = list(engines)
    n_e = len(names)
    if not 1 <= n_e <= _native.MAX_ENGINES:  # lookup only: nothing to merge - the staged path handles it
        merged = merge_hybrid_device(lookup_idx, lookup_lbl, engines, weights)
        if noise is None:
            noise = torch.empty((merged.indices.shape[0], merged.stride), dtype=torch.float32, device=merged.indices.device).exponential_(generator=generator)
        out = sample_merged_on_device(merged, noise, total=total, max_pos_sections=max_pos_sections, temperature=temperature,
                                      max_support_size=max_support_size, support=support)
        return flatten_on_device(out) if in_batch_negatives else out
    lib = _native.load_library()
    dev = lookup_idx.device
    if dev.type != "cuda":
        raise _native.NativeLibraryError("collate_on_device needs device tensors (there is no CPU path)")
    i64 = lambda t: t if (t.dtype is torch.int64 and t.is_contiguous()) else t.contiguous().long()  # noqa: E731
    f32 = lambda t: t if (t.dtype is torch.float32 and t.is_contiguous()) else t.contiguous().float()  # noqa: E731
    lookup_idx = i64(lookup_idx)
    lookup_lbl = None if lookup_lbl is None else i64(lookup_lbl)
    nq, kl = lookup_idx.shape
    e_idx = [i64(engines[n][0]) for n in names]
    e_scr = [f32(engines[n][1]) for n in names]
    ks = [int(t.shape[1]) for t in e_idx]
    for n, i, sc in zip(names, e_idx, e_scr):
        if i.shape != sc.shape or i.shape[0] != nq:
            raise ValueError(f"engine `{n}`: indices {tuple(i.shape)} / scores {tuple(sc.shape)} do not match nq={nq}")
    stride = kl + sum(ks) + 1
    total = int(total)
    U = nq * total
    flat = bool(in_batch_negatives)
    if flat and (U > 8192 or total > 1024):
        raise ValueError(f"{U} sampled ids in the batch ({total} per row): the one-launch flattening holds at most 8192 (1024 per row)")
    if noise is None:
        noise = torch.empty((nq, stride), dtype=torch.float32, device=dev).exponential_(generator=generator)
    elif noise.dtype is not torch.float32 or noise.stride(-1) != 1:
        noise = noise.float().contiguous()
    if noise.shape[0] != nq or noise.shape[1] < stride:
        raise ValueError(f"noise of shape {tuple(noise.shape)} does not cover the merged rows [{nq}, {stride}]")
    max_support = int(max_support_size or -1)
    if max_support >= 0:
        max_support = max(max_support, total)  # sample.py:126-128
    # few allocations, carved by pointer arithmetic: merged rows (workspace), sampled sections, flattened batch.
    # The merged rows never leave this function: they live in a workspace that is reused by the next call on the same stream
    # (stream order keeps the reuse safe; a different stream gets its own)
    stream = _native.current_stream_ptr(dev)
    ws_key = (dev.index, stream, nq, stride, n_e)
    ws = _WORKSPACES.get(ws_key)
    if ws is None:
        if len(_WORKSPACES) >= 8:
            _WORKSPACES.pop(next(iter(_WORKSPACES)))
        ws = _WORKSPACES[ws_key] = (
            torch.empty((2, nq, stride), dtype=torch.int64, device=dev),            # merged ids | labels
            torch.empty((1 + n_e, nq, stride), dtype=torch.float32, device=dev),    # merged scores | raw scores per engine
        )
    m_i64, m_f32 = ws
    s_i64 = torch.empty((2, nq, total), dtype=torch.int64, device=dev)            # sampled columns | ids
    s_f32 = torch.empty((2 + n_e, nq, total), dtype=torch.float32, device=dev)    # sampled scores | log-weights | raw scores
    s_row = torch.empty((3, nq), dtype=torch.float32, device=dev)                 # lse_pos | lse_neg | max_sampling_id
    s_lab = torch.empty((nq, total), dtype=torch.uint8, device=dev)
    cursors = torch.empty((nq * _native.MAX_ENGINES + 1,), dtype=torch.int32, device=dev)  # per-row cursors | n_unique
    a = _native.CollateArgs()
    a.lookup_idx, a.lookup_lbl = lookup_idx.data_ptr(), (None if lookup_lbl is None else lookup_lbl.data_ptr())
    a.k_lookup, a.n_engines, a.nq = kl, n_e, nq
    a.noise, a.noise_stride = noise.data_ptr(), noise.stride(0)
    a.k_positive, a.k_total, a.max_support_size, a.in_batch_negatives = int(max_pos_sections or total), total, max_support, int(flat)
    a.temperature = float(temperature)
    a.flags = support_flag(support)
    p_mi, p_mf, p_si, p_sf, p_row = m_i64.data_ptr(), m_f32.data_ptr(), s_i64.data_ptr(), s_f32.data_ptr(), s_row.data_ptr()
    a.merged_idx, a.merged_lbl, a.merged_scr = p_mi, p_mi + nq * stride * 8, p_mf
    a.row_cursor = cursors.data_ptr()
    a.out_local, a.out_ids = p_si, p_si + U * 8
    a.out_scores, a.out_log_weights, a.out_labels = p_sf, p_sf + U * 4, s_lab.data_ptr()
    a.out_lse_pos, a.out_lse_neg, a.out_max_sampling_id = p_row, p_row + nq * 4, p_row + nq * 8
    for e in range(n_e):
        a.engine_idx[e], a.engine_scr[e], a.engine_k[e], a.engine_weight[e] = e_idx[e].data_ptr(), e_scr[e].data_ptr(), ks[e], float(weights[names[e]])
        a.merged_raw[e] = p_mf + (1 + e) * nq * stride * 4
        a.out_raw[e] = p_sf + (2 + e) * U * 4
    if flat:
        f_f32 = torch.empty((2 + n_e, nq, U), dtype=torch.float32, device=dev)
        f_lab = torch.empty((nq, U), dtype=torch.uint8, device=dev)
        f_ids = torch.empty((U,), dtype=torch.int64, device=dev)
        p_ff = f_f32.data_ptr()
        a.flat_ids, a.flat_scores, a.flat_log_weights, a.flat_labels = f_ids.data_ptr(), p_ff, p_ff + nq * U * 4, f_lab.data_ptr()
        a.flat_n_unique = cursors.data_ptr() + nq * _native.MAX_ENGINES * 4
        for e in range(n_e):
            a.flat_raw[e] = p_ff + (2 + e) * nq * U * 4
    if torch.cuda.current_device() != dev.index:
        with torch.cuda.device(dev):
            _native.check(lib.vodhip_collate(ctypes.byref(a), stream))
    else:
        _native.check(lib.vodhip_collate(ctypes.byref(a), stream))
    lse_pos, lse_neg, max_id = s_row.unbind(0)
    if flat:
        fv = f_f32.unbind(0)
        return DeviceSampledSections(
            indices=f_ids, scores=fv[0], labels=f_lab.view(torch.bool), log_weights=fv[1], lse_pos=lse_pos, lse_neg=lse_neg, max_sampling_id=max_id,
            raw_scores=dict(zip(names, fv[2:])), local_ids=None, n_unique=cursors[-1:],
        )
    local, ids = s_i64.unbind(0)
    sv = s_f32.unbind(0)
    return DeviceSampledSections(
        indices=ids, scores=sv[0], labels=s_lab.view(torch.bool), log_weights=sv[1], lse_pos=lse_pos, lse_neg=lse_neg, max_sampling_id=max_id,
        raw_scores=dict(zip(names, sv[2:])), local_ids=local,
    )

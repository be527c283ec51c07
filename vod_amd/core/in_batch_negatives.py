"""Flatten the sampled sections of a batch into one in-batch section set on the GPU.

Mirror of `flatten_samples` (/root/reference/src/vod_dataloaders/core/in_batch_negatives.py:10-52): the sorted unique
section ids of the whole batch become ONE id list shared by every row (padded to B * n with the reference's constant
`1`), and each row's score / label / log-weight / raw engine scores are gathered onto that list
(`gather_values_by_indices`, numpy_ops.py:24-143: first occurrence, NaN - or 0 for labels - where the row does not
hold the id).  The gather of all value arrays is one launch of `vodhip_gather_by_id`.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from vod_amd import _native
from vod_amd import types as vt
from vod_amd.core.sample import PrioritySampledSections


def gather_by_id_tensors(queries: torch.Tensor, keys: torch.Tensor, values: list[torch.Tensor], fills: list[float]) -> list[torch.Tensor]:
    """Device-tensor API: `queries` int64 [U], `keys` int64 [B, n], each value float32 [B, n] -> float32 [B, U]."""
    lib = _native.load_library()
    if not keys.is_cuda:
        raise _native.NativeLibraryError("gather_by_id_tensors needs device tensors (there is no CPU path)")
    if not 1 <= len(values) <= 8 or len(fills) != len(values):
        raise ValueError("between 1 and 8 value arrays, one fill value each")
    dev = keys.device
    q = queries.to(dev, torch.int64).contiguous()
    k = keys.to(torch.int64).contiguous()
    if k.ndim != 2 or q.ndim != 1:
        raise ValueError("expected queries [U] and keys [B, n]")
    vals = [v.to(dev, torch.float32).contiguous() for v in values]
    for v in vals:
        if v.shape != k.shape:
            raise ValueError(f"value array of shape {tuple(v.shape)} does not match the keys {tuple(k.shape)}")
    outs = [torch.empty((k.shape[0], q.shape[0]), dtype=torch.float32, device=dev) for _ in vals]
    n = len(vals)
    v_ptrs = (ctypes.c_void_p * n)(*[v.data_ptr() for v in vals])
    o_ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    f_arr = (ctypes.c_float * n)(*[float(f) for f in fills])
    with torch.cuda.device(dev):
        _native.check(
            lib.vodhip_gather_by_id(q.data_ptr(), q.shape[0], k.data_ptr(), k.shape[0], k.shape[1], n, v_ptrs, f_arr,
                                    o_ptrs, _native.current_stream_ptr(dev))
        )
    return outs


def flatten_samples(samples: PrioritySampledSections, padding: bool = True, device: int = 0) -> PrioritySampledSections:
    """Merge all sections (positive and negative) as a flat batch (in_batch_negatives.py:10-52)."""
    if samples.batch.labels is None:
        raise ValueError("The `search_results` must have labels.")
    dev = torch.device("cuda", device)
    if (0 < samples.batch.indices.size <= 8192 and len(samples.raw_scores) <= 5 and samples.batch.indices.ndim == 2
            and samples.batch.indices.shape[1] <= 1024):
        return _flatten_one_launch(samples, padding, dev)
    indices = torch.from_numpy(np.ascontiguousarray(samples.batch.indices)).to(dev)
    unique = torch.unique(indices)  # sorted, like np.unique
    if padding:
        n_pad = indices.numel() - unique.shape[0]
        unique = torch.cat([unique, torch.ones((n_pad,), dtype=torch.int64, device=dev)])
    names = ["scores", "labels", "log_weights", *samples.raw_scores]
    arrays = [samples.batch.scores, samples.batch.labels, samples.log_weights, *samples.raw_scores.values()]
    fills = [float("nan"), 0.0, float("nan")] + [float("nan")] * len(samples.raw_scores)
    values = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev) for a in arrays]
    outs: dict[str, torch.Tensor] = {}
    for lo in range(0, len(values), 8):  # the kernel takes 8 value arrays per launch
        got = gather_by_id_tensors(unique, indices, values[lo : lo + 8], fills[lo : lo + 8])
        outs.update(dict(zip(names[lo : lo + 8], got)))
    host = {k: v.cpu().numpy() for k, v in outs.items()}
    return PrioritySampledSections(
        batch=vt.RetrievalBatch(
            indices=unique.cpu().numpy(),
            scores=host["scores"].astype(samples.batch.scores.dtype, copy=False),
            labels=host["labels"].astype(samples.batch.labels.dtype),
            allow_unsafe=True,
        ),
        max_sampling_id=samples.max_sampling_id,
        raw_scores={k: host[k].astype(np.asarray(samples.raw_scores[k]).dtype, copy=False) for k in samples.raw_scores},
        log_weights=host["log_weights"].astype(np.asarray(samples.log_weights).dtype, copy=False),
        lse_pos=samples.lse_pos,
        lse_neg=samples.lse_neg,
    )


def _flatten_one_launch(samples: PrioritySampledSections, padding: bool, dev: torch.device) -> PrioritySampledSections:
    """The whole flattening (distinct ids + padding + every gather) in one launch of `vodhip_flatten_inbatch`."""
    from vod_amd.core.collate import DeviceSampledSections, flatten_on_device

    up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)  # noqa: E731
    B = samples.batch.indices.shape[0]
    zero = torch.zeros((B,), dtype=torch.float32, device=dev)
    # label VALUES (the reference gathers whatever dtype the labels have, fill 0) travel as one more float array
    flat = flatten_on_device(DeviceSampledSections(
        indices=up(samples.batch.indices, np.int64), scores=up(samples.batch.scores, np.float32), labels=up(samples.batch.labels != 0, np.bool_),
        log_weights=up(samples.log_weights, np.float32), lse_pos=zero, lse_neg=zero, max_sampling_id=zero,
        raw_scores={**{k: up(v, np.float32) for k, v in samples.raw_scores.items()}, "__label_values__": up(samples.batch.labels, np.float32)},
    ))
    n = None if padding else int(flat.n_unique.item())
    host = lambda t: t[..., :n].cpu().numpy()  # noqa: E731
    label_values = np.nan_to_num(host(flat.raw_scores.pop("__label_values__")), nan=0.0)
    return PrioritySampledSections(
        batch=vt.RetrievalBatch(
            indices=host(flat.indices).astype(samples.batch.indices.dtype, copy=False),
            scores=host(flat.scores).astype(samples.batch.scores.dtype, copy=False),
            labels=label_values.astype(samples.batch.labels.dtype),
            allow_unsafe=True,
        ),
        max_sampling_id=samples.max_sampling_id,
        raw_scores={k: host(flat.raw_scores[k]).astype(np.asarray(samples.raw_scores[k]).dtype, copy=False) for k in samples.raw_scores},
        log_weights=host(flat.log_weights).astype(np.asarray(samples.log_weights).dtype, copy=False),
        lse_pos=samples.lse_pos,
        lse_neg=samples.lse_neg,
    )

"""Labeled priority sampling of the merged candidates on the GPU (host wrapper over `vodhip_priority_sample`).

Mirror of `sample_search_results` / `labeled_priority_sampling`
(/root/reference/src/vod_dataloaders/core/sample.py:22-157): same arguments, same
`PrioritySampledSections` result (batch of sampled indices / scores / labels, log-weights, lse_pos / lse_neg,
max_sampling_id, sampled raw scores).  The Exp(1) noise is drawn on the host with NumPy exactly where the
reference draws it (`np.random.exponential(size=scores.shape).astype(dtype)`, sample.py:398), so a seeded
`np.random` gives the same draws; the per-row selection runs in one kernel launch.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import torch

from vod_amd import _native
from vod_amd import types as vt


def support_flag(support: str) -> int:
    """`support="reference" | "keep_top"` -> the flag bit of include/vodhip.h (VODHIP_SAMPLE_KEEP_TOP_SUPPORT)."""
    if support not in ("reference", "keep_top"):
        raise ValueError(f"support must be 'reference' or 'keep_top', got {support!r}")
    return 2 if support == "keep_top" else 0


@dataclasses.dataclass(frozen=True)
class PrioritySampledSections:
    batch: vt.RetrievalBatch
    log_weights: np.ndarray
    max_sampling_id: np.ndarray
    lse_pos: np.ndarray
    lse_neg: np.ndarray
    raw_scores: dict[str, np.ndarray]


def labeled_priority_sampling_tensors(
    scores: torch.Tensor,
    labels: torch.Tensor,
    noise: torch.Tensor,
    k_positive: int,
    k_total: int,
    normalized: bool = True,
    temperature: float = 1.0,
    max_support_size: int | None = None,
    support: str = "reference",
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Device-tensor API: returns (samples i64 [nq,k_total], log_weights f32, labels bool, lse f32 [nq,2]).

    `support`: what `max_support_size` does to each class's candidates - "reference" removes the `max_support_size` best ones (the
    reference's behaviour, SURVEY 9 Q8: sample.py:176-178 masks `>= threshold`), "keep_top" keeps them and masks the rest (the
    corrected mode).  The default is bit parity with the reference."""
    lib = _native.load_library()
    if not scores.is_cuda:
        raise _native.NativeLibraryError("labeled_priority_sampling_tensors needs device tensors (there is no CPU path)")
    max_support_size = max_support_size or -1
    if max_support_size >= 0:
        max_support_size = max(max_support_size, k_total)  # sample.py:126-128
    nq, width = scores.shape
    dev = scores.device
    sc = scores.contiguous().float()
    lb = (labels > 0).to(torch.uint8).contiguous()
    nz = noise.contiguous().float()
    samples = torch.empty((nq, k_total), dtype=torch.int64, device=dev)
    logw = torch.empty((nq, k_total), dtype=torch.float32, device=dev)
    olab = torch.empty((nq, k_total), dtype=torch.uint8, device=dev)
    lse = torch.empty((nq, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(
            lib.vodhip_priority_sample(
                sc.data_ptr(), lb.data_ptr(), nz.data_ptr(), nq, width, int(k_positive), int(k_total), float(temperature),
                int(max_support_size), int(bool(normalized)) | support_flag(support), samples.data_ptr(), logw.data_ptr(), olab.data_ptr(),
                lse.data_ptr(), _native.current_stream_ptr(dev),
            )
        )
    return samples, logw, olab.bool(), lse


def sample_search_results(
    *,
    search_results: vt.RetrievalBatch,
    raw_scores: dict[str, np.ndarray],
    total: None | int,
    max_pos_sections: None | int,
    temperature: float = 1.0,
    max_support_size: None | int = None,
    device: int = 0,
    support: str = "reference",
) -> PrioritySampledSections:
    """Sample positive and negative sections with per-label priority sampling (sample.py:22-84)."""
    from vod_amd.core.collate import sample_merged_on_device
    from vod_amd.core.merge import MergedOnDevice

    total = total or search_results.shape[-1]
    max_pos_sections = max_pos_sections or total
    dev = torch.device("cuda", device)
    scores_ref = np.ascontiguousarray(search_results.scores, dtype=np.float32)
    nq, width = scores_ref.shape
    if width > 4096 or total > 4096:
        raise ValueError(f"{width} candidates / {total} samples per row: the sampling kernel holds at most 4096")
    if nq == 0 or width == 0:
        raise ValueError("cannot sample from an empty candidate pool")
    labels_ref = np.zeros(scores_ref.shape, dtype=np.int64) if search_results.labels is None else (search_results.labels > 0).astype(np.int64)
    noise = np.random.exponential(size=scores_ref.shape).astype(scores_ref.dtype)  # same draw as the reference (sample.py:398)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    # ONE launch: selection + take_along_axis of ids / scores / raw scores + the rank diagnostic (sample.py:52-70)
    merged = MergedOnDevice(
        indices=up(search_results.indices.astype(np.int64, copy=False)), scores=up(scores_ref), labels=up(labels_ref),
        raw={k: up(np.asarray(v, dtype=np.float32)) for k, v in raw_scores.items()}, stage_max=None, k_lookup=width, engine_k=[],
    )
    out = sample_merged_on_device(merged, up(noise), total=total, max_pos_sections=max_pos_sections, temperature=temperature,
                                  max_support_size=max_support_size, width=width, support=support)
    host = lambda t: t.cpu().numpy()  # noqa: E731
    return PrioritySampledSections(
        batch=vt.RetrievalBatch(indices=host(out.indices).astype(search_results.indices.dtype, copy=False), scores=host(out.scores),
                                labels=host(out.labels)),
        max_sampling_id=host(out.max_sampling_id),
        lse_pos=host(out.lse_pos),
        lse_neg=host(out.lse_neg),
        log_weights=host(out.log_weights),
        raw_scores={k: host(v).astype(np.asarray(raw_scores[k]).dtype, copy=False) for k, v in out.raw_scores.items()},
    )


def samples_to_dict(samples: PrioritySampledSections, relevances, prefix: str = "", as_torch: bool = False) -> dict:
    """The `section__*` fields the collate hands to the model, from the sampled (or flattened) sections.

    Mirror of `_samples_to_dict` (/root/reference/src/vod_dataloaders/realm_collate.py:247-278): keys
    `{prefix}idx | score | label | relevance | log_weight | lse_pos | lse_neg` plus one per raw engine score; with
    `as_torch` the score is float32, the label bool, the relevance a float32 tensor, the rest `torch.from_numpy`.
    """
    if samples.batch.labels is None:
        raise ValueError("The sections must have labels.")
    out = {
        f"{prefix}idx": samples.batch.indices,
        f"{prefix}score": samples.batch.scores,
        f"{prefix}label": samples.batch.labels > 0,
        f"{prefix}relevance": relevances,
        f"{prefix}log_weight": samples.log_weights,
        f"{prefix}lse_pos": samples.lse_pos,
        f"{prefix}lse_neg": samples.lse_neg,
        **{f"{prefix}{k}": v for k, v in samples.raw_scores.items()},
    }
    if as_torch:
        special = {
            f"{prefix}score": lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(torch.float32),
            f"{prefix}label": lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(torch.bool),
            f"{prefix}relevance": lambda x: torch.tensor(x, dtype=torch.float32),
        }
        out = {k: special.get(k, lambda x: torch.from_numpy(np.ascontiguousarray(x)))(v) for k, v in out.items()}
    return out

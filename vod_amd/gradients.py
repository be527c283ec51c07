"""In-batch retrieval scoring + loss on the GPU (host wrapper over `vodhip_retrieval_forward/backward`).

Mirror of `RetrievalGradients` (/root/reference/src/vod_models/vod_gradients/retrieval.py:14-92): same
constructor, same keyword-only call `(batch, query_encoding, section_encoding)`, same outputs (`loss`,
`retriever_scores`, `diagnostics` with kl_score / kl_sparse / kl_dense).  The ~15 torch kernels of the
reference's forward (einsum, masked_fill, log_softmax, targets, loss, three KLs) and the autograd backward
become one fused forward launch (+ finalize) and two backward launches, wrapped in a
`torch.autograd.Function` so `loss.backward()` keeps working.  The auxiliary losses (guidance,
self-supervision, score decay: retrieval.py:94-150; all weight 0 in the shipped config) are extra terms of the same
row kernel: values in `diagnostics`, gradients folded into the same dLoss/dScores.
"""
from __future__ import annotations

import dataclasses
import typing as typ

import torch

from vod_amd import _native


@dataclasses.dataclass
class RealmOutput:
    """`loss`, `retriever_scores [B, D]`, `diagnostics` -- fields of the reference's RealmOutput (vod_types/batch.py:106-114)."""

    loss: torch.Tensor
    retriever_scores: torch.Tensor
    diagnostics: dict[str, typ.Any] = dataclasses.field(default_factory=dict)


class _RetrievalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, s, score, relevance, sparse, dense, aux_cfg=(0, 0.0, 0.0, 0.0)):  # noqa: ANN001
        lib = _native.load_library()
        if not q.is_cuda:
            raise _native.NativeLibraryError("RetrievalGradients needs device tensors (there is no CPU path)")
        enc = q.dtype if q.dtype in (torch.float16, torch.bfloat16, torch.float32) else torch.float32
        as_ = lambda t, dt: t.detach() if (t.dtype is dt and t.is_contiguous()) else t.detach().to(dt).contiguous()  # noqa: E731
        qc = as_(q, enc)
        sc = as_(s, enc)
        three_d = sc.dim() == 3
        if sc.dim() not in (2, 3):
            raise ValueError(f"Invalid dimension for `section_encoding`: {tuple(sc.shape)}")
        B, H = qc.shape
        D = sc.shape[1] if three_d else sc.shape[0]
        score_c = as_(score, torch.float32)
        rel_c = as_(relevance, torch.int64)
        if score_c.shape != (B, D) or rel_c.shape != (B, D):
            raise ValueError(f"section__score / section__relevance must be [{B}, {D}]")
        sparse_c = None if sparse is None else as_(sparse, torch.float32)
        dense_c = None if dense is None else as_(dense, torch.float32)
        dev = q.device
        both = torch.empty((2, B, D), dtype=torch.float32, device=dev)  # retriever scores | dLoss/dScores
        scores, d_scores = both.unbind(0)
        # loss [1] | kl [3] | auxiliary terms [3] (the library writes all three: NaN where the weight is 0) | pad | workspace
        n_work = 16 * B + (4 * B * D if not three_d else 0)  # + room for the split-K slabs of the in-batch contraction
        small = torch.empty((8 + n_work,), dtype=torch.float32, device=dev)
        loss, kl, aux, work = small[0], small[1:4], small[4:7], small[8:]
        g_type, w_g, w_ss, w_sd = aux_cfg
        any_aux = w_g > 0 or w_ss > 0 or w_sd > 0
        aux_grad = torch.empty((3, B, D), dtype=torch.float32, device=dev) if any_aux else None
        with torch.cuda.device(dev):
            _native.check(
                lib.vodhip_retrieval_forward_aux(
                    qc.data_ptr(), sc.data_ptr(), _native.torch_dtype_code(enc), int(three_d), B, D, H,
                    score_c.data_ptr(), rel_c.data_ptr(),
                    None if sparse_c is None else sparse_c.data_ptr(), None if dense_c is None else dense_c.data_ptr(),
                    int(g_type), float(w_g), float(w_ss), float(w_sd),
                    scores.data_ptr(), d_scores.data_ptr(), loss.data_ptr(), kl.data_ptr(), aux.data_ptr(),
                    None if aux_grad is None else aux_grad.data_ptr(), work.data_ptr(), n_work, _native.current_stream_ptr(dev),
                )
            )
        ctx.save_for_backward(qc, sc, d_scores)
        ctx.meta = (enc, three_d, B, D, H, q.dtype, s.dtype)
        ctx.mark_non_differentiable(scores, kl, aux)
        return loss, scores, kl, aux

    @staticmethod
    def backward(ctx, g_loss, _g_scores, _g_kl, _g_aux):  # noqa: ANN001
        lib = _native.load_library()
        qc, sc, d_scores = ctx.saved_tensors
        enc, three_d, B, D, H, q_dt, s_dt = ctx.meta
        dev = qc.device
        go = g_loss.detach()
        if go.dtype is not torch.float32 or not go.is_contiguous():
            go = go.float().contiguous()
        dq = torch.empty((B, H), dtype=torch.float32, device=dev)
        ds = torch.empty(tuple(sc.shape), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _native.check(
                lib.vodhip_retrieval_backward(
                    qc.data_ptr(), sc.data_ptr(), _native.torch_dtype_code(enc), int(three_d), B, D, H,
                    d_scores.data_ptr(), go.data_ptr(), dq.data_ptr(), ds.data_ptr(), _native.current_stream_ptr(dev),
                )
            )
        return (dq if q_dt is torch.float32 else dq.to(q_dt)), (ds if s_dt is torch.float32 else ds.to(s_dt)), None, None, None, None, None


class RetrievalGradients:
    """KL-style retrieval objective with in-batch scoring, fused on the GPU."""

    def __init__(self, guidance: str = "zero", guidance_weight: float = 0.0, self_supervision_weight: float = 0.0,
                 score_decay: float = 0.0):
        if guidance not in ("zero", "sparse"):
            raise ValueError(f"guidance must be 'zero' or 'sparse', got {guidance!r}")  # the reference's GuidanceType (:11)
        self.guidance = guidance
        self.guidance_weight = guidance_weight
        self.self_supervision_weight = self_supervision_weight
        self.score_decay = score_decay

    def __call__(self, *, batch: typ.Any, query_encoding: torch.Tensor, section_encoding: torch.Tensor,
                 lm_logits: None | torch.Tensor = None) -> RealmOutput:  # noqa: ARG002
        get = (lambda k: batch.get(k)) if isinstance(batch, dict) else (lambda k: getattr(batch, k, None))
        loss, scores, kl, aux = _RetrievalLoss.apply(
            query_encoding, section_encoding, get("section__score"), get("section__relevance"),
            get("section__sparse"), get("section__dense"),
            (1 if self.guidance == "sparse" else 0, float(self.guidance_weight), float(self.self_supervision_weight), float(self.score_decay)),
        )
        diagnostics = {}
        if self.guidance_weight > 0:  # insertion order and keys of the reference's `_auxiliary_losses` (:104-118)
            diagnostics[f"{self.guidance}_guidance"] = aux[0]
        if self.self_supervision_weight > 0:
            diagnostics["self_supervision"] = aux[1]
        if self.score_decay > 0:
            diagnostics["score_decay"] = aux[2]
        diagnostics["kl_score"] = kl[0]
        if get("section__sparse") is not None:
            diagnostics["kl_sparse"] = kl[1]
        if get("section__dense") is not None:
            diagnostics["kl_dense"] = kl[2]
        return RealmOutput(loss=loss, retriever_scores=scores, diagnostics=diagnostics)

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


_ENC = (torch.float16, torch.bfloat16, torch.float32)
_scratch: dict = {}  # (device index, stream, floats) -> the forward's device scratch (row words + split-K slabs), consumed in stream order


def _as(t: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    """`t` itself when it already has the dtype and layout the kernels read (no new tensor object: this sits on a ~100 us host path)."""
    return t if (t.dtype is dt and t.is_contiguous()) else t.to(dt).contiguous()


class _on_device:
    """`torch.cuda.device(dev)` only when `dev` is not current already (the context manager costs ~5 us per entry)."""

    __slots__ = ("ctx",)

    def __init__(self, dev: torch.device):
        self.ctx = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


class _RetrievalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, s, score, relevance, sparse, dense, aux_cfg=(0, 0.0, 0.0, 0.0)):  # noqa: ANN001
        lib = _native.load_library()
        if not q.is_cuda:
            raise _native.NativeLibraryError("RetrievalGradients needs device tensors (there is no CPU path)")
        enc = q.dtype if q.dtype in _ENC else torch.float32
        qc, sc = _as(q, enc), _as(s, enc)  # (autograd is off inside `forward`: no detach needed)
        three_d = sc.dim() == 3
        if sc.dim() not in (2, 3):
            raise ValueError(f"Invalid dimension for `section_encoding`: {tuple(sc.shape)}")
        B, H = qc.shape
        D = sc.shape[1] if three_d else sc.shape[0]
        score_c, rel_c = _as(score, torch.float32), _as(relevance, torch.int64)
        if score_c.shape != (B, D) or rel_c.shape != (B, D):
            raise ValueError(f"section__score / section__relevance must be [{B}, {D}]")
        sparse_c = None if sparse is None else _as(sparse, torch.float32)
        dense_c = None if dense is None else _as(dense, torch.float32)
        dev = q.device
        stream = _native.current_stream_ptr(dev)
        # outputs: retriever scores | dLoss/dScores, and loss [1] | kl [3] | auxiliary terms [3] (NaN where the weight is 0) - fresh per
        # call (the caller keeps them); the kernels' scratch (16 row words per query + the split-K slabs of the in-batch contraction)
        # is cached per (device, stream, size): the next forward on the stream runs after this one has consumed it
        both = torch.empty((2, B, D), dtype=torch.float32, device=dev)
        small = torch.empty((8,), dtype=torch.float32, device=dev)
        n_work = 16 * B + (4 * B * D if not three_d else 0)
        if torch.cuda.is_current_stream_capturing():
            # under hipGraph capture the scratch must come from the graph's own pool (and must not leak into the cache)
            work = torch.empty((n_work,), dtype=torch.float32, device=dev)
        else:
            key = (dev.index, stream, n_work)
            work = _scratch.get(key)
            if work is None:
                if len(_scratch) > 64:
                    _scratch.clear()
                work = _scratch[key] = torch.empty((n_work,), dtype=torch.float32, device=dev)
        g_type, w_g, w_ss, w_sd = aux_cfg
        aux_grad = torch.empty((3, B, D), dtype=torch.float32, device=dev) if (w_g > 0 or w_ss > 0 or w_sd > 0) else None
        p_both, p_small = both.data_ptr(), small.data_ptr()
        with _on_device(dev):
            _native.check(
                lib.vodhip_retrieval_forward_aux(
                    qc.data_ptr(), sc.data_ptr(), _native.torch_dtype_code(enc), int(three_d), B, D, H,
                    score_c.data_ptr(), rel_c.data_ptr(),
                    None if sparse_c is None else sparse_c.data_ptr(), None if dense_c is None else dense_c.data_ptr(),
                    int(g_type), float(w_g), float(w_ss), float(w_sd),
                    p_both, p_both + 4 * B * D, p_small, p_small + 4, p_small + 16,
                    None if aux_grad is None else aux_grad.data_ptr(), work.data_ptr(), n_work, stream,
                )
            )
        scores, d_scores = both[0], both[1]
        ctx.save_for_backward(qc, sc, d_scores)
        ctx.meta = (enc, three_d, B, D, H, q.dtype, s.dtype)
        loss, kl, aux = small[0], small[1:4], small[4:7]
        ctx.mark_non_differentiable(scores, kl, aux)
        return loss, scores, kl, aux

    @staticmethod
    def backward(ctx, g_loss, _g_scores, _g_kl, _g_aux):  # noqa: ANN001
        lib = _native.load_library()
        qc, sc, d_scores = ctx.saved_tensors
        enc, three_d, B, D, H, q_dt, s_dt = ctx.meta
        dev = qc.device
        go = g_loss if (g_loss.dtype is torch.float32 and g_loss.is_contiguous()) else g_loss.float().contiguous()
        dq = torch.empty((B, H), dtype=torch.float32, device=dev)
        ds = torch.empty(sc.shape, dtype=torch.float32, device=dev)
        with _on_device(dev):
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


class GraphedRetrievalStep:
    """The fused loss, forward AND autograd backward, captured once as ONE hipGraph and replayed with one host call per step.

    Eager, a step of `RetrievalGradients` costs ~170-190 us of host time (two ctypes calls, four small allocations and ~80 us of torch
    autograd machinery around a custom Function) for ~90 us of kernels; a replay of the captured step is launch-free on the host:
    81 us (3-D, 64 x 32 x 768) / 118 us (in-batch, 64 x 2048 x 768) wall including the device synchronisation (tools/probe_h5_graph.py),
    bit-identical to the eager step.  The captured step owns static buffers: `query_encoding`, `section_encoding`, the `section__*` fields
    of `batch` - write the step's inputs into them (`load(...)` copies, or produce them there), `replay()`, then read `output.loss`,
    `output.retriever_scores`, `output.diagnostics` and the gradients `dq` / `ds` (to continue into the encoders:
    `torch.autograd.backward([q_enc, s_enc], [step.dq, step.ds])`).  Shapes, dtypes and the set of optional fields are fixed at capture.
    """

    def __init__(self, gradients: RetrievalGradients, *, batch_size: int, n_sections: int, hidden: int, sections_3d: bool = False,
                 dtype: torch.dtype = torch.float32, device: torch.device | int = 0, sparse: bool = True, dense: bool = True):
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        B, D, H = int(batch_size), int(n_sections), int(hidden)
        self.gradients = gradients
        self.query_encoding = torch.zeros((B, H), dtype=dtype, device=dev, requires_grad=True)
        self.section_encoding = torch.zeros(((B, D, H) if sections_3d else (D, H)), dtype=dtype, device=dev, requires_grad=True)
        self.batch = {"section__score": torch.zeros((B, D), device=dev), "section__relevance": torch.zeros((B, D), dtype=torch.int64, device=dev),
                      "section__sparse": torch.zeros((B, D), device=dev) if sparse else None,
                      "section__dense": torch.zeros((B, D), device=dev) if dense else None}
        self.batch["section__relevance"][:, 0] = 1  # a well-formed batch for the warm-up steps
        self.query_encoding.grad = torch.zeros_like(self.query_encoding)
        self.section_encoding.grad = torch.zeros_like(self.section_encoding)
        # warm-up on a side stream (allocator pools, one-time driver calls such as the LDS attribute), then the capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.output = self._step()

    def _step(self) -> RealmOutput:
        self.query_encoding.grad.zero_()
        self.section_encoding.grad.zero_()
        out = self.gradients(batch=self.batch, query_encoding=self.query_encoding, section_encoding=self.section_encoding)
        out.loss.backward()
        return out

    @property
    def dq(self) -> torch.Tensor:
        return self.query_encoding.grad

    @property
    def ds(self) -> torch.Tensor:
        return self.section_encoding.grad

    def load(self, *, batch: typ.Any, query_encoding: torch.Tensor, section_encoding: torch.Tensor) -> None:
        """Copy one step's inputs into the static buffers (device-to-device, no synchronisation)."""
        get = (lambda k: batch.get(k)) if isinstance(batch, dict) else (lambda k: getattr(batch, k, None))
        with torch.no_grad():
            self.query_encoding.copy_(query_encoding)
            self.section_encoding.copy_(section_encoding)
            for key, dst in self.batch.items():
                if dst is None:
                    if get(key) is not None:
                        raise ValueError(f"the step was captured without `{key}`: the set of optional fields is fixed at capture")
                    continue
                src = get(key)
                if src is None:
                    raise ValueError(f"the captured step expects `{key}`")
                dst.copy_(src)

    def replay(self) -> RealmOutput:
        self.graph.replay()
        return self.output

    def __call__(self, *, batch: typ.Any, query_encoding: torch.Tensor, section_encoding: torch.Tensor) -> tuple[RealmOutput, torch.Tensor, torch.Tensor]:
        self.load(batch=batch, query_encoding=query_encoding, section_encoding=section_encoding)
        return self.replay(), self.dq, self.ds

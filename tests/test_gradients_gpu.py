"""GPU parity tests for the fused in-batch retrieval loss (H5), through the C-ABI.

Floating point: tolerance rtol 2e-4 / atol 2e-5 against the golden vectors (reference ran fp32 torch on CPU)
and against the fp64 oracle; the contraction is an fp32 dot product in a different summation order."""
import numpy as np
import pytest

from conftest import GOLDEN

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL = dict(rtol=2e-4, atol=2e-5)
NAMES = ["retrieval_grad_2d", "retrieval_grad_3d", "retrieval_grad_nopos", "retrieval_grad_padded", "retrieval_grad_inbatch"]


def _run(q, s, score, rel, sparse, dense, dtype=torch.float32, upstream=1.0):
    from vod_amd.gradients import RetrievalGradients

    qt = torch.tensor(q, device="cuda", dtype=dtype, requires_grad=True)
    st = torch.tensor(s, device="cuda", dtype=dtype, requires_grad=True)
    batch = {
        "section__score": torch.tensor(score, device="cuda"),
        "section__relevance": torch.tensor(rel, device="cuda"),
        "section__sparse": None if sparse is None else torch.tensor(sparse, device="cuda"),
        "section__dense": None if dense is None else torch.tensor(dense, device="cuda"),
    }
    out = RetrievalGradients()(batch=batch, query_encoding=qt, section_encoding=st)
    (out.loss * upstream).backward()
    return out, qt.grad, st.grad


@pytest.mark.parametrize("name", NAMES)
def test_matches_reference_golden(name):
    g = np.load(GOLDEN / f"{name}.npz")
    out, dq, ds = _run(g["q"], g["s"], g["score"], g["relevance"], g["sparse"], g["dense"])
    np.testing.assert_allclose(out.loss.item(), g["loss"], **TOL)
    np.testing.assert_allclose(out.retriever_scores.cpu().numpy(), g["retriever_scores"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(dq.cpu().numpy(), g["dq"], **TOL)
    np.testing.assert_allclose(ds.cpu().numpy(), g["ds"], **TOL)
    for k in ("kl_score", "kl_sparse", "kl_dense"):
        np.testing.assert_allclose(out.diagnostics[k].item(), g[k], **TOL)
    assert out.retriever_scores.requires_grad is False and out.loss.dtype == torch.float32


@pytest.mark.parametrize("three_d", [False, True])
@pytest.mark.parametrize("B,D,H", [(64, 32, 768), (64, 2048, 768), (3, 5, 17), (16, 300, 1024)])
def test_matches_fp64_oracle(B, D, H, three_d):
    from oracle.gradients import retrieval_gradients

    if three_d and D > 512:
        pytest.skip("3-D section encodings are per-query sets (D = n_sections), never the flattened in-batch set")
    rng = np.random.default_rng(B + D + H)
    q = (rng.normal(size=(B, H)) / np.sqrt(H) * 3).astype(np.float32)
    s = rng.normal(size=((B, D, H) if three_d else (D, H))).astype(np.float32)
    score = rng.normal(size=(B, D)).astype(np.float32)
    pad = rng.uniform(size=(B, D)) < 0.1
    pad[:, 0] = False
    score[pad] = -np.inf
    rel = (rng.uniform(size=(B, D)) < 0.05).astype(np.int64)
    rel[:, 0] = 1
    rel[B // 2] = 0  # a row without positives -> n_positives falls back to the non-pad count
    sparse = rng.normal(size=(B, D)).astype(np.float32)
    sparse[rng.uniform(size=(B, D)) < 0.2] = np.nan
    out, dq, ds = _run(q, s, score, rel, sparse, None, upstream=2.5)
    ref = retrieval_gradients(q, s, score, rel, sparse, None)
    np.testing.assert_allclose(out.loss.item(), ref["loss"], **TOL)
    np.testing.assert_allclose(out.retriever_scores.cpu().numpy(), ref["retriever_scores"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(dq.cpu().numpy(), 2.5 * ref["dq"], **TOL)
    np.testing.assert_allclose(ds.cpu().numpy(), 2.5 * ref["ds"], **TOL)
    np.testing.assert_allclose(out.diagnostics["kl_score"].item(), ref["kl_score"], **TOL)
    np.testing.assert_allclose(out.diagnostics["kl_sparse"].item(), ref["kl_sparse"], **TOL)
    assert "kl_dense" not in out.diagnostics


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_half_precision_encodings(dtype):
    from oracle.gradients import retrieval_gradients

    rng = np.random.default_rng(3)
    B, D, H = 8, 64, 256
    q = torch.tensor(rng.normal(size=(B, H)) / np.sqrt(H) * 3).to(dtype)
    s = torch.tensor(rng.normal(size=(D, H))).to(dtype)
    score = rng.normal(size=(B, D)).astype(np.float32)
    rel = (rng.uniform(size=(B, D)) < 0.2).astype(np.int64)
    rel[:, 0] = 1
    out, dq, ds = _run(q.float().numpy(), s.float().numpy(), score, rel, None, None, dtype=dtype)
    ref = retrieval_gradients(q.float().numpy(), s.float().numpy(), score, rel)  # oracle on the rounded inputs
    np.testing.assert_allclose(out.loss.item(), ref["loss"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dq.float().cpu().numpy(), ref["dq"], rtol=2e-2, atol=2e-3)  # grads are cast back to 16 bit
    np.testing.assert_allclose(ds.float().cpu().numpy(), ref["ds"], rtol=2e-2, atol=2e-3)
    assert dq.dtype == dtype and ds.dtype == dtype


AUX_NAMES = ["retrieval_aux_guidance_sparse", "retrieval_aux_guidance_zero", "retrieval_aux_self_supervision",
             "retrieval_aux_score_decay", "retrieval_aux_all", "retrieval_aux_all_nopos"]


@pytest.mark.parametrize("name", AUX_NAMES)
def test_auxiliary_losses_match_reference_golden(name):
    """The reference's guidance / self-supervision / score-decay terms (retrieval.py:94-150) as extra terms of the fused
    row kernel: loss, every diagnostic (same keys, same order) and both gradients vs reference-generated vectors."""
    import json

    from vod_amd.gradients import RetrievalGradients

    g = np.load(GOLDEN / f"{name}.npz")
    params = json.loads((GOLDEN / "manifest.json").read_text())[name]["params"]
    qt = torch.tensor(g["q"], device="cuda", requires_grad=True)
    st = torch.tensor(g["s"], device="cuda", requires_grad=True)
    batch = {k: torch.tensor(g[v], device="cuda") for k, v in
             [("section__score", "score"), ("section__relevance", "relevance"), ("section__sparse", "sparse"), ("section__dense", "dense")]}
    out = RetrievalGradients(**params["config"])(batch=batch, query_encoding=qt, section_encoding=st)
    out.loss.backward()
    np.testing.assert_allclose(out.loss.item(), g["loss"], **TOL)
    assert list(out.diagnostics) == params["diagnostic_keys"]
    for key in params["diagnostic_keys"]:
        np.testing.assert_allclose(out.diagnostics[key].item(), g[f"diag_{key}"], **TOL)
    np.testing.assert_allclose(qt.grad.cpu().numpy(), g["dq"], **TOL)
    np.testing.assert_allclose(st.grad.cpu().numpy(), g["ds"], **TOL)


@pytest.mark.parametrize("three_d", [False, True])
def test_auxiliary_losses_match_fp64_oracle_at_c5_shape(three_d):
    from oracle.gradients import retrieval_gradients
    from vod_amd.gradients import RetrievalGradients

    rng = np.random.default_rng(9)
    B, D, H = (64, 32, 768) if three_d else (64, 2048, 256)
    q = (rng.normal(size=(B, H)) / np.sqrt(H)).astype(np.float32)
    s = rng.normal(size=(B, D, H) if three_d else (D, H)).astype(np.float32)
    score = rng.normal(size=(B, D)).astype(np.float32)
    score[rng.uniform(size=(B, D)) < 0.1] = -np.inf
    score[:, 0] = 0.5
    rel = (rng.uniform(size=(B, D)) < 0.05).astype(np.int64)
    rel[:, 0] = 1
    sparse = (rng.gamma(2.0, 4.0, size=(B, D)) - 12).astype(np.float32)
    sparse[rng.uniform(size=(B, D)) < 0.3] = np.nan
    cfg = dict(guidance="sparse", guidance_weight=0.2, self_supervision_weight=0.3, score_decay=0.01)
    ref = retrieval_gradients(q, s, score, rel, sparse, None, **cfg)
    qt = torch.tensor(q, device="cuda", requires_grad=True)
    st = torch.tensor(s, device="cuda", requires_grad=True)
    batch = {"section__score": torch.tensor(score, device="cuda"), "section__relevance": torch.tensor(rel, device="cuda"),
             "section__sparse": torch.tensor(sparse, device="cuda"), "section__dense": None}
    out = RetrievalGradients(**cfg)(batch=batch, query_encoding=qt, section_encoding=st)
    out.loss.backward()
    np.testing.assert_allclose(out.loss.item(), ref["loss"], **TOL)
    for key in ("sparse_guidance", "self_supervision", "score_decay", "kl_score", "kl_sparse"):
        np.testing.assert_allclose(out.diagnostics[key].item(), ref[key], **TOL)
    assert "kl_dense" not in out.diagnostics
    np.testing.assert_allclose(qt.grad.cpu().numpy(), ref["dq"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(st.grad.cpu().numpy(), ref["ds"], rtol=2e-4, atol=2e-5)


def test_invalid_guidance_type_is_rejected():
    from vod_amd.gradients import RetrievalGradients

    with pytest.raises(ValueError, match="guidance"):
        RetrievalGradients(guidance="dense", guidance_weight=0.1)


@pytest.mark.parametrize("adjacent", [True, False])
def test_plain_entry_point_matches_the_autograd_wrapper_in_batch(adjacent):
    """`vodhip_retrieval_forward` (no auxiliary terms, 8 B workspace words) on the flattened in-batch section set: with the two
    [B, D] outputs adjacent in memory the K-split contraction uses them as its two slabs (each row's workgroup reads its slab rows
    before it writes), otherwise it runs unsplit - both must give the wrapper's numbers (which splits 4 ways in its workspace)."""
    from vod_amd import _native

    lib = _native.load_library()
    rng = np.random.default_rng(11)
    B, D, H = 64, 2048, 768
    q = rng.standard_normal((B, H)).astype(np.float32)
    s = rng.standard_normal((D, H)).astype(np.float32)
    score = rng.standard_normal((B, D)).astype(np.float32)
    score[rng.random((B, D)) < 0.1] = -np.inf
    rel = (rng.random((B, D)) < 0.02).astype(np.int64)
    out, _, _ = _run(q, s, score, rel, None, None)
    qt, st = torch.tensor(q, device="cuda"), torch.tensor(s, device="cuda")
    sc, rl = torch.tensor(score, device="cuda"), torch.tensor(rel, device="cuda")
    both = torch.zeros((3, B, D), device="cuda")
    scores, d_scores = (both[0], both[1]) if adjacent else (both[0], both[2])
    small = torch.zeros(4 + 16 * B, device="cuda")
    _native.check(lib.vodhip_retrieval_forward(qt.data_ptr(), st.data_ptr(), _native.torch_dtype_code(torch.float32), 0, B, D, H,
                                               sc.data_ptr(), rl.data_ptr(), None, None, scores.data_ptr(), d_scores.data_ptr(),
                                               small[0:1].data_ptr(), small[1:4].data_ptr(), small[4:].data_ptr(),
                                               _native.current_stream_ptr(qt.device)))
    torch.cuda.synchronize()
    np.testing.assert_allclose(scores.cpu().numpy(), out.retriever_scores.cpu().numpy(), rtol=2e-4, atol=2e-4)  # fp32 sums of 768 products in 1 / 2 / 4 slabs
    np.testing.assert_allclose(small[0].item(), out.loss.item(), rtol=1e-5)
    np.testing.assert_allclose(small[1].item(), out.diagnostics["kl_score"].item(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("three_d,D", [(True, 32), (False, 2048)])
def test_graphed_step_replays_the_eager_step_bit_for_bit(three_d, D):
    """`GraphedRetrievalStep`: forward + autograd backward captured as ONE hipGraph (the kernels are enqueued on torch's capturing stream
    through the C-ABI).  Replays on fresh inputs - copied into the static buffers - must equal the eager step bit for bit: loss, scores,
    diagnostics, dq, ds; with the auxiliary losses on; and the padded / no-positive rows of the fixtures' kind."""
    from vod_amd.gradients import GraphedRetrievalStep, RetrievalGradients

    B, H = 64, 768
    rng = torch.Generator(device="cuda").manual_seed(11)
    for cfg in ({}, {"guidance": "sparse", "guidance_weight": 0.3, "self_supervision_weight": 0.2, "score_decay": 0.01}):
        grad = RetrievalGradients(**cfg)
        step = GraphedRetrievalStep(grad, batch_size=B, n_sections=D, hidden=H, sections_3d=three_d, device=0)
        for trial in range(3):
            q = torch.randn((B, H), device="cuda", generator=rng) * (3.0 / H ** 0.5)
            s = torch.randn(((B, D, H) if three_d else (D, H)), device="cuda", generator=rng)
            score = torch.randn((B, D), device="cuda", generator=rng)
            score[torch.rand((B, D), device="cuda", generator=rng) < 0.1] = float("-inf")
            score[:, 0] = 0.5
            rel = (torch.rand((B, D), device="cuda", generator=rng) < 0.05).long()
            rel[:, 0] = 1
            if not cfg:
                rel[B // 2] = 0  # a row without positives (with the self-supervision term on, the reference's loss is NaN for such a row)
            batch = {"section__score": score, "section__relevance": rel, "section__sparse": torch.randn((B, D), device="cuda", generator=rng),
                     "section__dense": torch.randn((B, D), device="cuda", generator=rng)}
            out, dq, ds = step(batch=batch, query_encoding=q, section_encoding=s)
            qe, se = q.clone().requires_grad_(), s.clone().requires_grad_()
            ref = grad(batch=batch, query_encoding=qe, section_encoding=se)
            ref.loss.backward()
            same = lambda a, b: bool(torch.allclose(a, b, rtol=0, atol=0, equal_nan=True))  # noqa: E731  (bit for bit, NaN == NaN)
            assert torch.isfinite(ref.loss), "the test data must give a finite loss"
            assert same(out.loss, ref.loss) and same(out.retriever_scores, ref.retriever_scores)
            assert list(out.diagnostics) == list(ref.diagnostics)
            for key in ref.diagnostics:
                assert same(out.diagnostics[key], ref.diagnostics[key]), key
            assert same(dq, qe.grad) and same(ds, se.grad), trial

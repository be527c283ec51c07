import sys, pathlib
import torch
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.c5_baseline import eager_reference_loss
from vod_amd.gradients import RetrievalGradients
B, H, D = 64, 768, 32
rng = torch.Generator(device="cuda").manual_seed(11)
for three_d in (True, False):
    q = torch.randn((B, H), device="cuda", generator=rng) * (3.0 / H ** 0.5)
    s = torch.randn(((B, D, H) if three_d else (D, H)), device="cuda", generator=rng)
    score = torch.randn((B, D), device="cuda", generator=rng)
    score[torch.rand((B, D), device="cuda", generator=rng) < 0.1] = float("-inf")
    score[:, 0] = 0.5
    rel = (torch.rand((B, D), device="cuda", generator=rng) < 0.05).long()
    rel[:, 0] = 1
    rel[B // 2] = 0
    base = {"section__score": score, "section__relevance": rel}
    for name, extra in {"plain": {}, "+sparse": {"section__sparse": torch.randn((B, D), device="cuda", generator=rng)},
                        "+dense": {"section__dense": torch.randn((B, D), device="cuda", generator=rng)}}.items():
        batch = dict(base, **extra)
        for k in ("section__sparse", "section__dense"):
            batch.setdefault(k, None)
        o = RetrievalGradients()(batch=batch, query_encoding=q.clone().requires_grad_(), section_encoding=s.clone().requires_grad_())
        l_ref, _, _ = eager_reference_loss(torch, q, s, {k: v for k, v in batch.items() if v is not None} | {"section__sparse": batch["section__sparse"], "section__dense": batch["section__dense"]})
        print(three_d, name, "fused", float(o.loss), "reference ops", float(l_ref), {k: float(v) for k, v in o.diagnostics.items()})
    # without the no-positive row
    rel2 = rel.clone(); rel2[B // 2, 0] = 1
    o = RetrievalGradients()(batch={"section__score": score, "section__relevance": rel2, "section__sparse": None, "section__dense": None}, query_encoding=q, section_encoding=s)
    print(three_d, "all rows have a positive: fused", float(o.loss))
    score2 = score.clone(); score2[torch.isinf(score2)] = 0.0
    o = RetrievalGradients()(batch={"section__score": score2, "section__relevance": rel, "section__sparse": None, "section__dense": None}, query_encoding=q, section_encoding=s)
    print(three_d, "no pads: fused", float(o.loss))

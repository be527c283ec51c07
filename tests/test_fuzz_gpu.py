"""A short, seeded slice of the randomised campaigns (tests/fuzz/fuzz_search.py, tests/fuzz/fuzz_collate.py) on every GPU test run: random
store sizes / dims / batch sizes / k / dtypes / kernel variants / planner knobs / row orders / subset filters for the search,
random shapes and pad / NaN / duplicate patterns for the collate-side kernels - all against the CPU oracle."""
import importlib.util

import numpy as np
import pytest

from conftest import ROOT

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _load(name):
    spec = importlib.util.spec_from_file_location(name, ROOT / "tests" / "fuzz" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [11, 12])
def test_search_campaign_slice(seed):
    fz = _load("fuzz_search")
    rng = np.random.default_rng(seed)
    for t in range(60):
        c = fz.draw(rng)
        try:
            fz.run_trial(c)
        except AssertionError as e:
            raise AssertionError(f"trial {t}: {c}: {e}") from e
    # and configurations whose (tiny, 256-slot) candidate lists overflow on duplicate-heavy / clustered rows: the recovery path must be taken
    # and stay exact.  (Whether a list overflows depends on the stage sizes the planner picks - the 60 % headroom of its capacity bound is
    # beaten by the data, not by construction - so several stores are tried and the first that overflows is used.)
    c = fz.draw(rng)
    c.update(n=70000, d=64, nq=300, k=100, dtype="f16", tile=0, data="duplicates", cand_cap=256, dense_rows=0, sample_div=0, growth=0,
             small_chunk_tiles=-1, subset=False, id_base=0, build="once", node_shards=0, exact=False, lossy="none", exact_expand=0)
    reruns = 0
    for upd in ({}, {"n": 200000}, {"data": "clustered"}, {"n": 200000, "data": "clustered"}, {"n": 400000}):
        trial = dict(c, **upd)
        reruns = fz.run_trial(trial)["last_safe_reruns"]
        if reruns >= 1:
            c = trial
            break
    assert reruns >= 1
    # the same store with float32 rows kept (exact-f32), the scan listing only k rows, wide integer rows the fp16 scan cannot represent: the
    # scan recovers its own overflow, no list proves complete, the band pass overflows the 256-slot lists too - and the answer is still exact
    c.update(exact=True, lossy="rows", exact_expand=1)
    st = fz.run_trial(c)
    assert st["last_exact_band_queries"] > 0


@pytest.mark.parametrize("seed", [21])
def test_collate_campaign_slice(seed):
    fz = _load("fuzz_collate")
    rng = np.random.default_rng(seed)
    fns = [fz.fuzz_merge, fz.fuzz_sampling, fz.fuzz_gradients, fz.fuzz_merge_topk, fz.fuzz_flatten, fz.fuzz_chain]
    for t in range(180):
        fn = fns[t % len(fns)]
        try:
            fn(rng)
        except AssertionError as e:
            raise AssertionError(f"trial {t} ({fn.__name__}) {fz.LAST}: {e}") from e

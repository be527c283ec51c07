"""Which small configurations still overflow their candidate lists (tests/test_fuzz_gpu.py wants one that takes the recovery path)."""
import importlib.util, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[2]
spec = importlib.util.spec_from_file_location("fuzz_search", ROOT / "tests" / "fuzz" / "fuzz_search.py")
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
rng = np.random.default_rng(11)
base = fz.draw(rng)
base.update(n=70000, d=64, nq=300, k=100, dtype="f16", tile=0, data="duplicates", cand_cap=256, dense_rows=0, sample_div=0, growth=0,
            small_chunk_tiles=-1, subset=False, id_base=0, build="once", node_shards=0, exact=False, lossy="none", exact_expand=0)
for upd in [dict(), dict(k=200), dict(k=240), dict(n=200000), dict(n=200000, k=200), dict(sample_div=500), dict(sample_div=500, k=200), dict(growth=25600),
            dict(nq=1100), dict(nq=1100, k=200), dict(data="clustered") if "clustered" in getattr(fz, "DATA", ["clustered"]) else dict()]:
    c = dict(base); c.update(upd)
    try:
        st = fz.run_trial(c)
        print(upd, "reruns", st.get("last_safe_reruns"), "overflow", st.get("last_overflow"), "recovered", st.get("last_recovered_queries"))
    except Exception as e:  # noqa
        print(upd, "ERR", type(e).__name__, str(e)[:200])

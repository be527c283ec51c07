#!/usr/bin/env python3
"""Re-run one drawn configuration of tests/fuzz/fuzz_search.py (printed by a failing campaign) with single knobs toggled, to isolate a failure.
usage: python experiments/tools/repro_fuzz_trial.py "<dict literal>" """
import ast
import importlib.util
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
spec = importlib.util.spec_from_file_location("fuzz_search", ROOT / "tests" / "fuzz" / "fuzz_search.py")
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
base = ast.literal_eval(sys.argv[1])
variants = {"as drawn": {}, "plain index": {"node_shards": 0}, "2 shards": {"node_shards": 2}, "no subset": {"subset": False},
            "plain, no subset": {"node_shards": 0, "subset": False}, "cand_cap default": {"cand_cap": 0}, "sample_div default": {"sample_div": 0},
            "build once": {"build": "once"}, "k 100": {"k": 100}, "nq 256": {"nq": 256}, "nq 64": {"nq": 64}}
for name, delta in variants.items():
    c = dict(base)
    c.update(delta)
    try:
        out = fz.run_trial(c)
        print(f"{name:>20}: ok {out}")
    except AssertionError as e:
        print(f"{name:>20}: FAIL {str(e)[:200]}")

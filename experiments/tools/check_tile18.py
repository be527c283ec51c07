"""Experiment builds only (VODHIP_LIB=vod_amd/csrc/libvodhip_ablation.so): tile 18 (K-split wave pair, query fragments resident) must return what tile 8 returns, bit for bit -
repeated (race screen) over shapes that cover one / many K-tiles per corpus tile, partly filled last tiles, 1-4 query tiles, both dtypes, the
subset filter, both stage orders and candidate-list overflow."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
from vod_amd.index import HipFlatIndex  # noqa: E402


def one(n, d, nq, k, dt, data, order, subset, reps, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if data == "int":
        x = torch.randint(-8, 9, (n, d), generator=g, device="cuda").float()
        q = torch.randint(-8, 9, (nq, d), generator=g, device="cuda").float()
    else:
        x = torch.randn((n, d), generator=g, device="cuda")
        q = torch.randn((nq, d), generator=g, device="cuda")
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    lab = torch.randint(0, 4, (n,), generator=g, device="cuda", dtype=torch.int32)
    res = {}
    for tile in (8, 18):
        with HipFlatIndex(d, n, dtype=tdt, device=0) as ix:
            ix.add(x.to(tdt))
            ix.set_param("tile", tile)
            ix.set_param("tile_order", order)
            sub = None
            if subset:
                ix.set_row_labels(lab)
                sub = torch.full((nq, 1), -1, dtype=torch.int32)
                sub[::2, 0] = 1
            outs = []
            for _ in range(reps if tile == 18 else 1):
                s, i = ix.search(q.to(tdt), k, subset=sub)
                outs.append((s.clone(), i.clone()))
            res[tile] = outs
    s8, i8 = res[8][0]
    bad = 0
    for r, (s, i) in enumerate(res[18]):
        if not (torch.equal(i, i8) and torch.equal(s, s8)):
            bad += 1
    return bad


def main():
    t0 = time.time()
    cases = []
    # (tile 18 takes dim_pad 384 / 768 without a subset filter; the other shapes check that it falls back to the production kernel)
    for n, d, nq in ((70_001, 768, 300), (200_000, 768, 1024), (131_072, 384, 256), (90_000, 1024, 512), (33_000, 740, 700), (500_000, 384, 257), (1_000_000, 768, 256), (255, 768, 200), (8_193, 350, 129)):
        for dt in ("f16", "bf16"):
            for data in ("int", "gauss"):
                cases.append((n, d, nq, 100 if d != 1024 else 200, dt, data, 0, False))
    cases += [(150_000, 256, 512, 50, "f16", "int", 1, False), (150_000, 256, 512, 50, "f16", "int", 0, True), (60_000, 96, 290, 33, "bf16", "int", 1, True)]
    fails = 0
    for c in cases:
        bad = one(*c, reps=6, seed=hash(c) & 0xFFFF)
        fails += bad > 0
        print(("FAIL" if bad else "ok  "), c, f"{bad}/6 runs differ", flush=True)
    print(f"tile 18 vs tile 8: {len(cases)} cases, {fails} failing, {time.time() - t0:.0f} s")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()

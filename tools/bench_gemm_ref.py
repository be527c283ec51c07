"""Yardstick, not product code: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this chip for the
headline shape WITHOUT any top-k - scores[nq, rows] = Q[nq, dim] @ X[rows, dim]^T on random data, written to HBM.

    python3 tools/bench_gemm_ref.py [--rows 10000000] [--dim 768] [--nq 1024] [--dtype f16] [--chunk 1000000]

Prints one JSON line: ms per full pass over `rows`, TFLOP/s, fraction of the 2.5 PF dense peak.  The fused search kernel
does the same contraction and keeps the top-k instead of writing the scores.
"""
import argparse
import json

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--chunk", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.empty(a.rows, a.dim, device=dev, dtype=dt)
    for lo in range(0, a.rows, 1 << 20):
        hi = min(a.rows, lo + (1 << 20))
        x[lo:hi] = torch.randn(hi - lo, a.dim, device=dev, generator=g).to(dt)
    q = torch.randn(a.nq, a.dim, device=dev, generator=g).to(dt)
    out = torch.empty(a.nq, a.chunk, device=dev, dtype=dt)
    out_t = torch.empty(a.chunk, a.nq, device=dev, dtype=dt)

    def one_pass(transposed):
        for lo in range(0, a.rows, a.chunk):
            hi = min(a.rows, lo + a.chunk)
            if transposed:  # X @ Q^T : [rows, nq]
                torch.matmul(x[lo:hi], q.t(), out=out_t[: hi - lo])
            else:           # Q @ X^T : [nq, rows]
                torch.matmul(q, x[lo:hi].t(), out=out[:, : hi - lo])

    res = {}
    for name, tr in (("q_xT", False), ("x_qT", True)):
        for _ in range(a.warmup):
            one_pass(tr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            one_pass(tr)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        tf = 2.0 * a.nq * a.rows * a.dim / (ms * 1e-3) / 1e12
        res[name] = {"ms_per_pass": round(ms, 3), "tflops": round(tf, 1), "frac_of_2.5PF": round(tf / 2500.0, 4)}
    print(json.dumps({"what": "torch.matmul (vendor GEMM), scores written to HBM, no top-k", "rows": a.rows, "dim": a.dim,
                      "nq": a.nq, "dtype": a.dtype, "chunk": a.chunk, "data": "random normal", **res}))


if __name__ == "__main__":
    main()

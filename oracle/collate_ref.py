"""Oracle (test infrastructure): ctypes front of oracle/collate_ref.c - the reference's numba loops of the collate-side chain in
plain C (gcc -O3 -fopenmp), used (a) as a second restatement checked against the reference-generated fixtures and (b) as the CPU
figure reported beside the C5 kernels (`cpu_baseline` of bench.py's C5 side entry).  Never imported by the product."""
from __future__ import annotations

import ctypes

import numpy as np

_lib = None
_vp = ctypes.c_void_p


def _load():
    global _lib
    if _lib is None:
        from vod_amd.build import build_oracle

        build_oracle()
        from vod_amd.build import ORACLE

        lib = ctypes.CDLL(str(ORACLE / "_build" / "liboracle_collate.so"))
        lib.vodref_merge_hybrid.restype = ctypes.c_int
        lib.vodref_merge_hybrid.argtypes = [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp, ctypes.c_int, _vp]
        lib.vodref_sample_search_results.restype = ctypes.c_int
        lib.vodref_sample_search_results.argtypes = [_vp, _vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_float, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        lib.vodref_flatten_samples.restype = ctypes.c_int
        lib.vodref_flatten_samples.argtypes = [_vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]
        lib.vodref_set_threads.restype = None
        lib.vodref_set_threads.argtypes = [ctypes.c_int]
        _lib = lib
    return _lib


def _ptrs(arrays):
    return (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])


def merge_hybrid(lookup: tuple, engines: dict[str, tuple], weights: dict[str, float]):
    """Same contract as `oracle.hybrid.merge_hybrid`: lookup = (indices, scores, labels), engines[name] = (indices, scores);
    returns (indices, scores, labels, raw{name}) cut to the reference's width."""
    lib = _load()
    l_idx = np.ascontiguousarray(lookup[0], dtype=np.int64)
    l_lbl = None if lookup[2] is None else np.ascontiguousarray(lookup[2], dtype=np.int64)
    names = list(engines)
    e_idx = [np.ascontiguousarray(engines[n][0], dtype=np.int64) for n in names]
    e_scr = [np.ascontiguousarray(engines[n][1], dtype=np.float32) for n in names]
    nq, k_lookup = l_idx.shape
    e_k = np.array([a.shape[1] for a in e_idx], dtype=np.int32)
    e_w = np.array([weights[n] for n in names], dtype=np.float32)
    stride = k_lookup + int(e_k.sum()) + 1
    out_idx = np.empty((nq, stride), np.int64)
    out_scr = np.empty((nq, stride), np.float32)
    out_lbl = np.empty((nq, stride), np.int64)
    out_raw = [np.empty((nq, stride), np.float32) for _ in names]
    width = ctypes.c_int()
    rc = lib.vodref_merge_hybrid(l_idx.ctypes.data, None if l_lbl is None else l_lbl.ctypes.data, k_lookup, len(names), _ptrs(e_idx), _ptrs(e_scr),
                                 e_k.ctypes.data, e_w.ctypes.data, nq, out_idx.ctypes.data, out_scr.ctypes.data, out_lbl.ctypes.data, _ptrs(out_raw),
                                 stride, ctypes.byref(width))
    assert rc == 0
    w = width.value
    return out_idx[:, :w], out_scr[:, :w], (None if l_lbl is None else out_lbl[:, :w]), {n: r[:, :w] for n, r in zip(names, out_raw)}


def sample_search_results(indices, scores, labels, raw_scores: dict, noise, total, max_pos_sections, temperature=1.0, max_support_size=None):
    """Same contract as `oracle.sampling.sample_search_results`."""
    lib = _load()
    ids = np.ascontiguousarray(indices, dtype=np.int64)
    scr = np.ascontiguousarray(scores, dtype=np.float32)
    nq, width = scr.shape
    lab = np.zeros((nq, width), np.int64) if labels is None else np.ascontiguousarray(labels, dtype=np.int64)
    names = list(raw_scores)
    raw = [np.ascontiguousarray(raw_scores[n], dtype=np.float32) for n in names]
    nz = np.ascontiguousarray(noise, dtype=np.float32)
    total = total or width
    max_pos_sections = max_pos_sections or total
    o_local, o_ids = np.empty((nq, total), np.int64), np.empty((nq, total), np.int64)
    o_scr, o_logw = np.empty((nq, total), np.float32), np.empty((nq, total), np.float32)
    o_lab = np.empty((nq, total), np.uint8)
    o_raw = [np.empty((nq, total), np.float32) for _ in names]
    o_lse, o_max = np.empty((nq, 2), np.float32), np.empty((nq,), np.float32)
    rc = lib.vodref_sample_search_results(ids.ctypes.data, scr.ctypes.data, lab.ctypes.data, len(names), _ptrs(raw), nz.ctypes.data, nq, width,
                                          int(max_pos_sections), int(total), float(temperature), int(max_support_size or -1), o_local.ctypes.data,
                                          o_ids.ctypes.data, o_scr.ctypes.data, o_logw.ctypes.data, o_lab.ctypes.data, _ptrs(o_raw), o_lse.ctypes.data,
                                          o_max.ctypes.data)
    assert rc == 0
    return {"local": o_local, "indices": o_ids, "scores": o_scr, "labels": o_lab.astype(bool), "log_weights": o_logw, "lse_pos": o_lse[:, 0],
            "lse_neg": o_lse[:, 1], "max_sampling_id": o_max, "raw": dict(zip(names, o_raw))}


def flatten_samples(indices, scores, labels, log_weights, raw_scores: dict):
    """Same contract as `oracle.sampling.flatten_samples` (padding on)."""
    lib = _load()
    ids = np.ascontiguousarray(indices, dtype=np.int64)
    n_rows, n_keys = ids.shape
    names = list(raw_scores)
    vals = [np.ascontiguousarray(scores, dtype=np.float32), np.ascontiguousarray(log_weights, dtype=np.float32)] + \
           [np.ascontiguousarray(raw_scores[n], dtype=np.float32) for n in names]
    lab = np.ascontiguousarray(labels).astype(np.uint8)
    U = n_rows * n_keys
    outs = [np.empty((n_rows, U), np.float32) for _ in vals]
    o_lab = np.empty((n_rows, U), np.uint8)
    uniq = np.empty((U,), np.int64)
    n_unique = lib.vodref_flatten_samples(ids.ctypes.data, n_rows, n_keys, len(vals), _ptrs(vals), _ptrs(outs), lab.ctypes.data, o_lab.ctypes.data,
                                          uniq.ctypes.data)
    return {"indices": uniq, "scores": outs[0], "labels": o_lab.astype(bool), "log_weights": outs[1], "raw": dict(zip(names, outs[2:])),
            "n_unique": n_unique}

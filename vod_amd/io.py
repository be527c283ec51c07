"""Wire codec of the search server: NumPy array <-> urlsafe-base64 of its `.npy` bytes.

String-compatible with the reference (/root/reference/src/vod_search/io.py:17-32): the payload is exactly
`base64.urlsafe_b64encode(np.save(...))`.  Decoding refuses pickled object arrays (the reference passes
`allow_pickle=True`; nothing on this path needs it and it would execute untrusted bytes).
"""
from __future__ import annotations

import base64
import io

import numpy as np


def serialize_np_array(array: np.ndarray) -> str:
    buf = io.BytesIO()
    np.save(buf, np.asarray(array), allow_pickle=False)
    return base64.urlsafe_b64encode(buf.getvalue()).decode("utf-8")


def deserialize_np_array(encoded: str, *, dtype=None) -> np.ndarray:
    raw = base64.urlsafe_b64decode(encoded)
    arr = np.load(io.BytesIO(raw), allow_pickle=False)
    if dtype is not None:
        arr = arr.astype(dtype)
    return arr

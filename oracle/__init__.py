"""CPU oracle for the dense-retrieval hot path -- TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this package, and only as the checker / the reported CPU baseline.  Nothing under
`vod_amd/` imports it: the product path fails loudly when the HIP library is missing.

Pinning status per function is stated in each module header:
  * flat_ip    -- PARITY UNPINNED against faiss (faiss-cpu 1.7.4 is a pip dependency of the
                  reference, `requirements.txt:42`, not vendored, not installable here; the
                  reference holds no test or golden vector for `faiss_index.search`).
  * hybrid     -- pinned by golden fixtures produced by running the reference's own modules.
  * sampling   -- pinned likewise.
  * gradients  -- pinned likewise.
"""

"""Build the in-tree native pieces: libvodhip.so (hipcc, gfx950) and the oracle's C restatement (gcc)."""
from __future__ import annotations

import os
import pathlib
import shutil
import subprocess

ROOT = pathlib.Path(__file__).resolve().parent.parent
CSRC = ROOT / "vod_amd" / "csrc"
ORACLE = ROOT / "oracle"


def build_native(force: bool = False, verbose: bool = False) -> pathlib.Path:
    """hipcc --offload-arch=gfx950 over vod_amd/csrc/*.hip -> vod_amd/csrc/libvodhip.so."""
    target = CSRC / "libvodhip.so"
    srcs = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")) + list(CSRC.glob("*.h")) + [ROOT / "include" / "vodhip.h"]
    if target.exists() and not force and all(target.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return target
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        raise RuntimeError("hipcc not found: cannot build libvodhip.so")
    env = dict(os.environ)
    env["PATH"] = env.get("PATH", "") + ":/opt/rocm/bin"
    cmd = ["make", "-C", str(CSRC), "-j", str(min(4, os.cpu_count() or 1))]
    if force:
        subprocess.run(["make", "-C", str(CSRC), "clean"], check=True, env=env, capture_output=not verbose)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"building libvodhip.so failed:\n{res.stdout[-4000:]}\n{res.stderr[-4000:]}")
    if verbose:
        print(res.stdout[-2000:])
    return target


def build_oracle(force: bool = False) -> pathlib.Path:
    """gcc the oracle's plain-C restatement (test infrastructure; never linked into the product)."""
    out_dir = ORACLE / "_build"
    out_dir.mkdir(exist_ok=True)
    first = None
    # flat_ip_ref.c: the H2 restatement (fp32 + heap); collate_ref.c: the reference's numba loops of the collate-side chain
    for src_name, lib_name in (("flat_ip_ref.c", "liboracle_flat_ip.so"), ("collate_ref.c", "liboracle_collate.so")):
        target = out_dir / lib_name
        src = ORACLE / src_name
        first = first or target
        if target.exists() and not force and target.stat().st_mtime >= src.stat().st_mtime:
            continue
        # (no -ffast-math: the restatement keeps IEEE NaN / inf semantics like NumPy)
        cmd = ["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", str(src), "-o", str(target), "-lm"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"building the C oracle ({src_name}) failed:\n{res.stderr[-4000:]}")
    return first


if __name__ == "__main__":
    print(build_native(verbose=True))
    print(build_oracle())

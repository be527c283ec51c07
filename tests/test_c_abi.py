"""The drop-in boundary is a C ABI: `include/vodhip.h` must be a valid C99 header, and a plain C program (no Python,
no torch, no C++) must be able to build an index, search it and get the oracle-exact answer through it."""
import os
import pathlib
import shutil
import subprocess

import pytest

from conftest import ROOT

SRC = ROOT / "tests" / "c_abi" / "abi_smoke.c"


def test_header_is_valid_c99():
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", str(ROOT / "include" / "vodhip.h")], check=True)


def _build(tmp_path: pathlib.Path) -> pathlib.Path:
    from vod_amd.build import build_native

    build_native()
    exe = tmp_path / "abi_smoke"
    libdir = ROOT / "vod_amd" / "csrc"
    subprocess.run(
        ["gcc", "-std=c99", "-O1", "-Wall", str(SRC), "-I", str(ROOT / "include"), "-I", "/opt/rocm/include", "-L", str(libdir),
         "-L", "/opt/rocm/lib", "-lvodhip", "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)],
        check=True,
    )
    return exe


def test_c_consumer_compiles_and_links(tmp_path):
    if shutil.which("gcc") is None or not pathlib.Path("/opt/rocm/include/hip/hip_runtime_api.h").exists():
        pytest.skip("gcc / ROCm headers not available")
    assert _build(tmp_path).exists()


@pytest.mark.gpu
def test_c_consumer_gets_the_exact_answer(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=dict(os.environ))
    assert out.returncode == 0, out.stderr + out.stdout
    assert "C ABI smoke ok" in out.stdout


def test_collate_args_struct_layout_matches_the_ctypes_mirror(tmp_path):
    """`vodhip_collate_args_t` crosses the boundary by pointer: the ctypes mirror (`_native.CollateArgs`) must have the C
    compiler's size and field offsets."""
    import ctypes

    from vod_amd import _native

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    fields = [name for name, _ in _native.CollateArgs._fields_]
    prog = tmp_path / "layout.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "vodhip.h"\nint main(void) {\n'
        '  printf("%zu\\n", sizeof(vodhip_collate_args_t));\n'
        + "".join(f'  printf("%zu\\n", offsetof(vodhip_collate_args_t, {f}));\n' for f in fields)
        + "  return 0;\n}\n"
    )
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", str(prog), "-I", str(ROOT / "include"), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got[0] == ctypes.sizeof(_native.CollateArgs)
    assert got[1:] == [getattr(_native.CollateArgs, f).offset for f in fields]

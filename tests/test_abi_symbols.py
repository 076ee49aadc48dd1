"""CPU-side checks of the C-ABI shared library: it builds for gfx950, loads, exports every
symbol include/smoqy_hip.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L


def declared_symbols():
    txt = open(L.HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smoqy_[A-Za-z_0-9]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    path = L.build()
    assert os.path.exists(path)
    lib = C.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in smoqy_hip.h but not exported"
    # and the Python binding covers the same set
    assert set(names) == set(L.SIGNATURES) | {"smoqy_last_error", "smoqy_team_last_error", "smoqy_member_last_error"}  # the three that return a string


def test_only_the_c_abi_leaves_the_libraries():
    """the internal C++ symbols (smoqy::launch_*, …) must not be exported next to the C entry points — a clash risk
    inside a host process that loads other HIP libraries.  Both libraries are linked with an export map (csrc/smoqy.map)."""
    import subprocess

    L.build()
    for path in (L.LIB_PATH, L.MEMBER_LIB_PATH):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
        assert names and all(n.startswith("smoqy_") for n in names), [n for n in names if not n.startswith("smoqy_")][:5]
    declared = set(declared_symbols())
    exported = {ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout.splitlines() if ln.strip()}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_member_library_has_no_gpu_dependency():
    """libsmoqy_member.so (member side of a published walker team) is what a GPU-less rank loads: every smoqy_member_* entry point, and
    no dependency on the HIP runtime or rocFFT."""
    import subprocess

    L.build()
    lib = C.CDLL(L.MEMBER_LIB_PATH)
    members = [n for n in declared_symbols() if n.startswith("smoqy_member_")]
    assert len(members) >= 12
    for n in members:
        assert hasattr(lib, n), n
    needed = subprocess.run(["readelf", "-d", L.MEMBER_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "amdhip" not in needed and "rocfft" not in needed and "hsa" not in needed, needed
    ml = L.load_member()
    m = C.c_void_p()
    assert ml.smoqy_member_attach(C.byref(m), b"/smoqy-no-such-team", 0, C.c_double(0.05)) == 9
    assert b"no team published" in ml.smoqy_member_last_error(None)


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = sq.lattice.bssh_chain(4, 4)
    nt, perm, colors = sq.lattice.checkerboard_decomposition(m.fpi.neighbor_table)
    with pytest.raises(L.SmoqyError) as e:
        L.Handle(4, 4, nt, colors)
    assert "no CPU path" in str(e.value) or "HIP" in str(e.value)


def test_product_package_does_not_import_the_oracle():
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); import smoqyelphqmc_amd; bad=[m for m in sys.modules if m.startswith('oracle')]; assert not bad, bad" % root
    subprocess.run([sys.executable, "-c", code], check=True)
    pkg = os.path.join(root, "smoqyelphqmc.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "smoqy_oracle" not in src, f


def test_header_and_c_example_compile_as_plain_c(tmp_path):
    """include/smoqy_hip.h is a C header (no C++ or torch types) and examples/c_abi_demo.c — the ccall sequence of INTEGRATION.md
    written in C — builds against libsmoqy_hip.so with gcc -std=c99 (the run itself needs a GPU: tests/test_gpu_c_abi.py)."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(root, "include", "smoqy_hip.h")], check=True)
    lib_dir = os.path.join(root, "smoqyelphqmc.jl_amd", "csrc")
    out = tmp_path / "c_abi_demo"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_demo.c"), "-L" + lib_dir, "-lsmoqy_hip", "-lm",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)], check=True)
    assert out.exists()
    # the member rank of a served walker team (no GPU, no handle): examples/team_member_demo.c
    out2 = tmp_path / "team_member_demo"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "team_member_demo.c"), "-L" + lib_dir, "-lsmoqy_member", "-lm",
                    "-Wl,-rpath," + lib_dir, "-o", str(out2)], check=True)   # the member library alone: no ROCm on this rank's link line
    assert out2.exists()


def test_member_attach_fails_cleanly_without_a_team():
    """the member side of a cross-process team needs no GPU: joining a team nobody serves times out with code 9 and a message"""
    lib = L.load()
    m = C.c_void_p()
    assert lib.smoqy_member_attach(C.byref(m), b"/smoqy-no-such-team", 0, C.c_double(0.05)) == 9
    assert b"no team published" in lib.smoqy_member_last_error(None)

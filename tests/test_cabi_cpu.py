"""CPU: the C-ABI library loads and exports every symbol declared in include/protosam_hip.h (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "protosam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"\bint\s+(psam_\w+)\s*\(", src)


def test_header_symbols_exported():
    from protosam_amd import _lib
    names = _declared()
    assert len(names) >= 25
    L = _lib.lib()
    for n in names:
        assert getattr(L, n) is not None
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_header_compiles_as_c(tmp_path):
    import subprocess
    c = tmp_path / "t.c"
    c.write_text('#include "protosam_hip.h"\nint main(void){return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o",
                           str(tmp_path / "t.o")])


def test_arity_matches_header():
    """ctypes argtypes in protosam_amd/_lib.py have the same arity as the C prototypes."""
    from protosam_amd import _lib
    src = open(os.path.join(ROOT, "include", "protosam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name, args in re.findall(r"\bint\s+(psam_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        n = len([a for a in args.split(",") if a.strip()])
        assert n == len(_lib.SIGNATURES[name]), (name, n, len(_lib.SIGNATURES[name]))


def test_missing_library_is_loud(monkeypatch):
    import importlib
    from protosam_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libprotosam_hip.so")
    try:
        _lib.lib()
        raise AssertionError("expected HipExtensionMissing")
    except _lib.HipExtensionMissing:
        pass


def test_product_never_imports_oracle():
    import glob
    for f in glob.glob(os.path.join(ROOT, "protosam_amd", "**", "*.py"), recursive=True):
        txt = open(f).read()
        assert "import oracle" not in txt and "from oracle" not in txt, f


def test_cpu_tensor_is_rejected():
    import pytest
    import torch
    from protosam_amd import ops
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros((128, 64), dtype=torch.float16), torch.zeros((128, 64), dtype=torch.float16))

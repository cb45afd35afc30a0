"""The C-ABI shared library loads and exports every symbol include/hmx.h declares (no compute calls)."""
import ctypes
import os
import re

from htool_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hmx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hmx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 30
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libhmx.so does not export %s" % n
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(names) == bound, "python binding and header disagree: %s" % (set(names) ^ bound)


def test_no_torch_types_in_header():
    text = open(os.path.join(ROOT, "include", "hmx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "torch" not in text and "at::" not in text and "std::" not in text


def test_product_does_not_reference_oracle():
    """The product path must never import, link or call anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "htool_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), "%s mentions the oracle" % f


def test_header_is_self_contained_c_and_cxx(tmp_path):
    """include/hmx.h must compile on its own as C99 and as C++14 (it is the contract a foreign-language binding reads)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "t.c").write_text('#include "hmx.h"\nint main(void) { return (int)sizeof(hmx_leaf) - (int)sizeof(hmx_leaf); }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I" + os.path.join(root, "include"), str(tmp_path / "t.c")])
    (tmp_path / "t.cpp").write_text('#include "hmx.h"\nint main() { return 0; }\n')
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(root, "include"), str(tmp_path / "t.cpp")])

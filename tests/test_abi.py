"""CPU: the C-ABI library builds, loads, and exports exactly what include/amdarrow.h declares
(no compute calls -- there is no GPU here)."""
import os
import re
import subprocess

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "amdarrow.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(amdarrow_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    if not os.path.exists(p._abi.LIB_PATH):
        ge.build()
    return p


def test_header_and_binding_agree(pkg):
    decl = declared_symbols()
    assert len(decl) >= 25
    assert sorted(pkg._abi.SYMBOLS) == decl


def test_library_exports_every_declared_symbol(pkg):
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg._abi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (amdarrow_[a-z0-9_]+)", out))
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, missing


def test_library_loads_and_reports_abi(pkg):
    lib = pkg.load_library()
    assert lib.amdarrow_abi_version() == 1
    assert lib.amdarrow_last_error() is not None


def test_code_objects_are_gfx950_only(pkg):
    """No dual paths: the fat binary carries gfx950 code objects and nothing else."""
    data = open(pkg._abi.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx90a", b"gfx942", b"gfx1100", b"sm_80", b"sm_90"):
        assert other not in data


def test_missing_library_fails_loudly(pkg, tmp_path):
    with pytest.raises(ImportError):
        pkg._abi.load_library(str(tmp_path / "nope.so"))


def test_no_product_code_touches_the_oracle():
    """The product path may not import, link or execute anything under oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "arrow-1_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"\boracle\b|\borc\b|liboracle|libref_driver|orc_", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad

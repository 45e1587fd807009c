"""The C-ABI library loads and exports every symbol include/elfihip.h declares (no GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "elfihip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(elfihip_[a-z0-9_]+)\s*\(", text)))


def test_header_functions_exported_and_bound():
    import elfi_amd
    from elfi_amd import _lib
    names = _declared_functions()
    assert len(names) >= 10
    lib = elfi_amd.load_library()
    for n in names:
        assert hasattr(lib, n), "libelfihip.so does not export %s" % n
        assert n in _lib.PROTOTYPES, "elfi_amd/_lib.py has no prototype for %s" % n
    for n in _lib.PROTOTYPES:
        assert n in names, "prototype %s is not declared in include/elfihip.h" % n


def test_version_and_status_without_gpu():
    import elfi_amd
    lib = elfi_amd.load_library()
    assert lib.elfihip_version() == 100
    # NULL-context calls must fail with a status, never crash
    assert lib.elfihip_ctx_synchronize(None) != 0
    assert lib.elfihip_last_error(None)


def test_no_cpu_fallback_symbols():
    """The product library must not link the oracle or any BLAS/LAPACK CPU path."""
    import subprocess
    from elfi_amd import LIB_PATH
    out = subprocess.run(["readelf", "-d", LIB_PATH], capture_output=True, text=True).stdout
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    for lib in needed:
        assert not re.search(r"blas|lapack|oracle", lib, re.I), needed
    assert any("amdhip64" in l for l in needed)


def test_python_mirror_does_not_import_oracle():
    import glob
    for f in glob.glob(os.path.join(ROOT, "elfi_amd", "**", "*.py"), recursive=True):
        src = open(f).read()
        assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f

"""The C-ABI libraries load on a machine without a GPU and export every function that
include/qaray_hip.h and include/qaray_host.h declare; no compute call is made here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from qaray_amd import hip, host


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qa_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,path", [("qaray_hip.h", hip.HIP_LIB_PATH), ("qaray_host.h", host.HOST_LIB_PATH)])
def test_library_exports_every_declared_symbol(header, path):
    lib = C.CDLL(path)
    names = declared_functions(header)
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"{os.path.basename(path)} lacks {missing}"


def test_hip_library_is_in_tree_and_has_gfx950_code():
    assert hip.HIP_LIB_PATH.startswith(ROOT)
    blob = open(hip.HIP_LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"qa_integrate" in blob


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.HipError) as e:
        hip.Context(0)
    assert "no CPU fallback" in str(e.value) or e.value.code == -4


def test_product_package_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under qaray_amd/ (python or C++) may reference it
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "qaray_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip", "Makefile")):
                t = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"qa_oracle|libqa_oracle|from oracle|import oracle|oracle/", t):
                    # comments that merely cite the oracle file name are allowed in headers
                    hits = [l for l in t.splitlines() if re.search(r"qa_oracle|libqa_oracle|from oracle|import oracle|oracle/", l)
                            and not l.strip().startswith(("//", "*", "/*", "#"))]
                    if hits:
                        bad.append((f, hits[:2]))
    assert not bad, bad

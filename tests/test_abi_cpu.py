"""CPU-side checks of the drop-in boundary (no compute calls): libsvc_hip.so loads, exports every entry point that
include/svc_hip.h declares (parsed from the header, so header and library cannot drift apart), the ctypes binding's
EXPORTS list agrees with the header, and the product path refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "svc_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return sorted(set(re.findall(r"^\s*(?:const\s+char\s*\*|int|long\s+long|void)\s+(svc_[A-Za-z0-9_]+)\s*\(", src, flags=re.M)))


def test_header_declares_entry_points():
    names = _declared()
    assert len(names) >= 59 and "svc_conv1d_f32" in names and "svc_last_error" in names


def test_library_loads_and_exports_every_declared_symbol():
    import svc_hip as S
    assert os.path.exists(S.LIB_PATH), "libsvc_hip.so not built: run `python __graft_entry__.py`"
    lib = ctypes.CDLL(S.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.svc_abi_version.restype = ctypes.c_int
    assert lib.svc_abi_version() >= 1
    lib.svc_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.svc_last_error(), bytes)


def test_binding_export_list_matches_header():
    import svc_hip as S
    assert sorted(set(S.EXPORTS)) == _declared()


def test_product_path_has_no_cpu_fallback():
    import svc_hip as S
    x = torch.zeros(1, 4, 8)
    with pytest.raises(S.SvcError):
        S.require_gpu(x)
    with pytest.raises(S.SvcError):
        S.copy_bct(x)


def test_product_modules_never_import_the_oracle():
    pkg = os.path.join(ROOT, "so-vits-svc_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad

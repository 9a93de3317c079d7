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
    hdr_version = int(re.search(r"#define\s+SVC_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.svc_abi_version() == hdr_version == S.ABI_VERSION
    lib.svc_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.svc_last_error(), bytes)


def test_argument_struct_layouts_match_the_header(tmp_path):
    """The argument structs cross the boundary by pointer and are read in full, so the ctypes mirrors must have the C compiler's
    layout of include/svc_hip.h: sizeof and the offset of every field, taken from a program gcc builds from the header itself.  (A
    struct that grows without SVC_ABI_VERSION moving is what ADVICE r5 flagged: svc_attention_args.ws, svc_gemm_args.split_k_atomic.)"""
    import shutil
    import subprocess
    import svc_hip as S
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler on this host")
    pairs = {"svc_conv1d_args": S.Conv1dArgs, "svc_convt1d_args": S.ConvT1dArgs, "svc_conv1d_direct_args": S.Conv1dDirectArgs,
             "svc_resblock_pair_args": S.ResblockPairArgs, "svc_attention_args": S.AttentionArgs, "svc_conv1d_h_args": S.Conv1dHArgs,
             "svc_conv_weight_args": S.ConvWeightArgs, "svc_wgrad_args": S.WgradArgs, "svc_gemm_args": S.GemmArgs,
             "svc_coupling_args": S.CouplingArgs, "svc_resblock16_args": S.Resblock16Args}
    src = open(HEADER).read()
    declared = set(re.findall(r"^}\s*(svc_[a-z0-9_]+_args)\s*;", src, flags=re.M))
    assert declared == set(pairs), (declared ^ set(pairs))
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{HEADER}"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(c)], check=True, capture_output=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), (cname, got[cname], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


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


def test_bench_and_entry_use_the_oracle_only_as_checker():
    """bench.py may touch oracle/ only inside its cpu_baseline legs, __graft_entry__.py only inside smoke(): everything
    they MEASURE runs on the HIP engine with data from synthetic_data.py (a generator, no reference algorithm)."""
    import ast
    allowed = {"bench.py": ("cpu_baseline", "cpu_baseline_train"), "__graft_entry__.py": ("smoke",)}
    for fname, funcs in allowed.items():
        tree = ast.parse(open(os.path.join(ROOT, fname)).read())
        parents = {}
        for node in ast.walk(tree):
            for ch in ast.iter_child_nodes(node):
                parents[ch] = node
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or ""]
            if not any(m == "oracle" or m.startswith("oracle.") for m in mods):
                continue
            fn = node
            while fn in parents and not isinstance(fn, ast.FunctionDef):
                fn = parents[fn]
            assert isinstance(fn, ast.FunctionDef) and fn.name in funcs, (fname, getattr(fn, "name", None), mods)
    s = open(os.path.join(ROOT, "synthetic_data.py")).read()
    assert not re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M)

"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/dslsph.h declares; the parameter block layouts match; creating an engine without
a GPU fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dslsph.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsl_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from dieselfluid_amd import _lib
    _lib.build_library()
    L = C.CDLL(_lib.library_path())
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/dslsph.h but not exported"
    assert set(declared) == set(_lib.EXPORTS), "ctypes binding table drifted from the header"


def test_params_struct_layout_matches_header():
    from dieselfluid_amd import _lib, engine
    p = engine.reference_params(16)
    assert p.struct_size == C.sizeof(_lib.Params)
    assert p.abi_version == 1 and p.n_particles == 4096 and p.lsh_bucket_size == 24
    assert p.ref_density == 512.0 and abs(p.mu - 1.3059) < 1e-6


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dieselfluid_amd import DslError, SPHEngine, engine
    with pytest.raises(DslError, match="no HIP device|hip"):
        SPHEngine(engine.reference_params(4))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under dieselfluid_amd/ may reference it."""
    pkg = os.path.join(ROOT, "dieselfluid_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".go")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in src and "dsloracle" not in src and "dsl_oracle" not in src, f


def test_go_binding_names_every_export():
    """bindings/go/dslsph (the cgo stub a dieselfluid maintainer adds; not compilable here, no Go toolchain)
    calls every function include/dslsph.h declares."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "dslsph.h")).read()
    declared = set(re.findall(r"\b(dsl_[a-z_0-9]+)\s*\(", header))
    go = ""
    gdir = os.path.join(root, "bindings", "go", "dslsph")
    for f in os.listdir(gdir):
        if f.endswith(".go"):
            go += open(os.path.join(gdir, f)).read()
    bound = set(re.findall(r"C\.(dsl_[a-z_0-9]+)\b", go))
    assert declared - bound == set(), sorted(declared - bound)


def _split_top_level(args: str):
    """comma-separated pieces of an argument list, commas inside brackets of any kind not counted"""
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _call_args(text: str, open_paren: int) -> str:
    """the text between the parenthesis at `open_paren` and its partner"""
    depth = 0
    for k in range(open_paren, len(text)):
        if text[k] == "(":
            depth += 1
        elif text[k] == ")":
            depth -= 1
            if depth == 0:
                return text[open_paren + 1:k]
    raise AssertionError("unbalanced call")


def test_go_binding_calls_have_the_header_arity():
    """names alone do not catch a stale call.  Every `C.dsl_*( ... )` call of the Go stub passes
    exactly as many arguments as include/dslsph.h declares for that function, and the ctypes table used by
    every test agrees with the header as well (a third, independent count)."""
    import re
    from dieselfluid_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "dslsph.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    header = re.sub(r"//[^\n]*", "", header)
    arity = {}
    for m in re.finditer(r"\b(dsl_[a-z_0-9]+)\s*\(", header):
        args = _call_args(header, m.end() - 1).strip()
        n = 0 if args in ("", "void") else len(_split_top_level(args))
        arity[m.group(1)] = n
    assert len(arity) >= 60
    for name, (_res, argtypes) in _lib.EXPORTS.items():
        assert arity[name] == len(argtypes), (name, arity[name], len(argtypes))
    gdir = os.path.join(root, "bindings", "go", "dslsph")
    calls = 0
    for f in sorted(os.listdir(gdir)):
        if not f.endswith(".go"):
            continue
        go = open(os.path.join(gdir, f)).read()
        go = re.sub(r"/\*.*?\*/", lambda mm: " " * len(mm.group(0)), go, flags=re.S)  # (the cgo preamble is a comment)
        go = re.sub(r"//[^\n]*", "", go)
        for m in re.finditer(r"C\.(dsl_[a-z_0-9]+)\s*\(", go):
            args = _call_args(go, m.end() - 1)
            n = len(_split_top_level(args))
            assert n == arity[m.group(1)], f"{f}: C.{m.group(1)} called with {n} arguments, header declares {arity[m.group(1)]}"
            calls += 1
    assert calls >= len(arity)


def test_custom_transport_communicator_needs_every_callback():
    """dsl_comm_create_custom (the host's own transport instead of RCCL) touches no device: a table with a
    missing callback is refused, a complete one yields a communicator of the stated shape."""
    import ctypes as C
    from dieselfluid_amd import _lib
    L = _lib.load_library()
    ok_group = _lib.TR_GROUP(lambda ctx: 0)
    ok_xfer = _lib.TR_XFER(lambda ctx, buf, n, peer, st: 0)
    ok_red = _lib.TR_REDUCE(lambda ctx, buf, n, st: 0)
    out = C.c_void_p()
    bad = _lib.Transport(None, ok_group, ok_group, ok_xfer, _lib.TR_XFER(), ok_red)
    assert L.dsl_comm_create_custom(2, 0, 0, C.cast(C.byref(bad), C.c_void_p), C.byref(out)) != 0
    assert b"callback" in L.dsl_comm_last_error()
    good = _lib.Transport(None, ok_group, ok_group, ok_xfer, ok_xfer, ok_red)
    assert L.dsl_comm_create_custom(2, 2, 0, C.cast(C.byref(good), C.c_void_p), C.byref(out)) != 0  # rank outside
    assert L.dsl_comm_create_custom(2, 1, 0, C.cast(C.byref(good), C.c_void_p), C.byref(out)) == 0 and out.value
    assert L.dsl_comm_destroy(out) == 0

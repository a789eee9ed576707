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

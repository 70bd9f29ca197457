"""CPU tests of the boundary: libzethprover.so loads and exports every symbol the header declares
(no compute call -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "zeth_prover.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(zp_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_expected_surface():
    syms = header_symbols()
    for must in ("zp_create", "zp_destroy", "zp_ntt", "zp_intt", "zp_lde", "zp_poseidon_perm", "zp_merkle_commit",
                 "zp_merkle_open", "zp_fri_fold", "zp_set_constants", "zp_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from eigen_zeth_amd import native
    assert os.path.exists(native.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(native.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), "missing export " + s
    # the binding table covers the header one-to-one
    assert sorted(native.SIGNATURES) == header_symbols()


def test_binding_loads_and_reports_version():
    from eigen_zeth_amd import native
    lib = native.load_library()
    assert b"gfx950" in lib.zp_version()


def test_product_never_imports_oracle():
    # the product package must not reference oracle/ in any form
    for dp, _, fns in os.walk(os.path.join(ROOT, "eigen_zeth_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, fn


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from eigen_zeth_amd import native
    with pytest.raises(native.ZpError):
        native.Prover(0)


def test_integration_doc_lists_every_entry_point():
    """INTEGRATION.md's Rust extern block binds exactly the functions include/zeth_prover.h declares"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "zeth_prover.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    declared = set(re.findall(r"^(?:int32_t|size_t|void|const char \*)\s*\*?(zp_[a-z0-9_]+)\(", hdr, re.M))
    bound = set(re.findall(r"pub fn (zp_[a-z0-9_]+)", doc))
    assert declared == bound, (declared - bound, bound - declared)


def test_compiled_host_builds_against_the_header_alone():
    """host/prove_chunk.cpp uses include/zeth_prover.h from a plain g++ translation unit and links the in-tree library"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "host"), "-s", "-B"])
    assert os.path.exists(os.path.join(root, "host", "prove_chunk")) and os.path.exists(os.path.join(root, "host", "aggregate"))
    agg = open(os.path.join(root, "host", "aggregate.cpp")).read()
    assert "#include \"../include/zeth_prover.h\"" in agg and "Python.h" not in agg and "torch" not in agg.replace("no torch", "")
    src = open(os.path.join(root, "host", "prove_chunk.cpp")).read()
    assert "#include \"../include/zeth_prover.h\"" in src and "import" not in src and "torch" not in src.replace("no torch", "")

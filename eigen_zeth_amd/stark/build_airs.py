"""Generates and compiles the constraint kernels of the built-in AIRs for gfx950:
generated/<air>.hip -> generated/libzpair_<air>.so (AIR plug-in ABI, include/zeth_prover.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "generated")
CSRC = os.path.join(os.path.dirname(HERE), "csrc")


def _stem(air):
    """built-in AIRs: their name.  AIRs made per shape (the verifier AIRs of the recursion layers: one name, many statements): name + digest"""
    return air.name + ("_" + air.digest() if air.fixed_cols else "")


def lib_path(air):
    return os.path.join(GEN, "libzpair_%s.so" % _stem(air))


def build_air(air, force=False):
    from .air import emit_quotient_source
    os.makedirs(GEN, exist_ok=True)
    src = os.path.join(GEN, _stem(air) + ".hip")
    code = emit_quotient_source(air, "hip")
    if force or not os.path.exists(src) or open(src).read() != code:
        with open(src, "w") as f:
            f.write(code)
    out = lib_path(air)
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, "gl.hpp"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
                               "-I", CSRC, "-o", out, src])
    return out


def build_all(force=False):
    from .air import BUILTIN_AIRS
    return [build_air(f(), force) for f in BUILTIN_AIRS.values()]


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from eigen_zeth_amd.stark.build_airs import build_all as _b
    print("\n".join(_b("--force" in sys.argv)))

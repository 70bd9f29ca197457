"""The limb form of the NTT butterflies (eigen_zeth_amd/csrc/gl_limb.hpp): host build against big-integer arithmetic (CPU), and the pass
kernels built on it against the oracle and against the canonical kernels (GPU, knob ntt_limb)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_limb_arithmetic_host_build(tmp_path):
    exe = str(tmp_path / "gl_limb_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "eigen_zeth_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "gl_limb_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatches" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("logn", [13, 14, 16, 17, 20, 22])
def test_limb_form_passes_match_oracle(prover, logn):
    from oracle import oracle as O
    W = 3 if logn >= 20 else 5
    x = O.random_field((W, 1 << logn), 0x11B0 + logn)
    d_in, d_out = prover.upload(x), prover.alloc(W << logn)
    prover.set_tuning("ntt_limb", 1)
    try:
        prover.ntt(d_in, d_out, logn, W)
        y = prover.download(d_out, (W, 1 << logn))
        prover.intt(d_out, d_in, logn, W)
        back = prover.download(d_in, (W, 1 << logn))
        d_l = prover.alloc(W << (logn + 1))
        prover.lde(d_in, d_l, logn, 1, W)
        e = prover.download(d_l, (W, 1 << (logn + 1)))
    finally:
        prover.set_tuning("ntt_limb", 0)
    assert (y == O.ntt(x)).all()
    assert (back == x).all()
    assert (e == O.lde(x, 1)).all()


@pytest.mark.gpu
def test_limb_form_equals_canonical_form_at_the_bench_plan(prover):
    logn, W = 24, 8
    rng = np.random.default_rng(7)
    x = rng.integers(0, 0xFFFFFFFF00000001, size=(W, 1 << logn), dtype=np.uint64)
    x[0, :8] = [0, 1, 0xFFFFFFFF00000000, 0xFFFFFFFF, 1 << 32, 0x7FFFFFFF00000000, 0x8000000000000000, 0xFFFFFFFEFFFFFFFF]
    d_in, d_a, d_b = prover.upload(x), prover.alloc(W << logn), prover.alloc(W << logn)
    prover.ntt(d_in, d_a, logn, W)
    prover.set_tuning("ntt_limb", 1)
    try:
        prover.ntt(d_in, d_b, logn, W)
    finally:
        prover.set_tuning("ntt_limb", 0)
    assert (prover.download(d_a, (W, 1 << logn)) == prover.download(d_b, (W, 1 << logn))).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seam", [0, 2])
def test_lde_seam_kernel_and_the_two_launch_form_match_the_oracle(prover, seam):
    """blow-up 2 at a size whose plans meet in radix-256 passes (2^16): the fused seam kernel (knob lde_seam = 2: also with the coefficient store)
    and the two-launch form (0), extension and scaled coefficients against the oracle"""
    from oracle import oracle as O
    logn, W = 16, 5
    x = O.random_field((W, 1 << logn), 0x5EA0 + seam)
    d_in, d_out, d_coef = prover.upload(x), prover.alloc(W << (logn + 1)), prover.alloc(W << logn)
    prover.set_tuning("lde_seam", seam)
    try:
        prover.lde(d_in, d_out, logn, 1, W)
        a = prover.download(d_out, (W, 1 << (logn + 1)))
        prover.lde(d_in, d_out, logn, 1, W, d_coef=d_coef)
        b = prover.download(d_out, (W, 1 << (logn + 1)))
        c = prover.download(d_coef, (W, 1 << logn))
    finally:
        prover.set_tuning("lde_seam", 1)
    want = O.lde(x, 1)
    assert (a == want).all() and (b == want).all()
    coef = O.intt(x)
    from eigen_zeth_amd import native
    s = int(prover.get_constants(native.ZP_CONST_COSET_SHIFT, 1)[0])
    P = 0xFFFFFFFF00000001
    pw = np.ones(1 << logn, dtype=object)
    for i in range(1, 1 << logn):
        pw[i] = pw[i - 1] * s % P
    assert all(int(c[0, i]) == int(coef[0, i]) * int(pw[i]) % P for i in range(0, 1 << logn, 257))

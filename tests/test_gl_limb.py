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


@pytest.mark.gpu
@pytest.mark.parametrize("logn", [16, 17, 18, 19, 20, 21, 22, 23])
def test_lde_seam_plans_at_the_provers_sizes(prover, logn):
    """round 5: the fused extension where the DEFAULT plans do not meet in radix-256 passes -- an inverse plan that ENDS in one and a forward
    plan that STARTS with one (csrc/ntt.hip zpi_get_plan_role; knob lde_seam_plans).  The extension a prover issues (no coefficient store)
    against the oracle, with the seam plans on (default) and off, with and without the coefficient store; zp_ntt_plan_json says which path
    a size takes: by default fused at 2^16 and from 2^21 on (2^22 = the chunk size of BASELINE configs[2] / [4]); forced (knob 2) also at
    2^19 / 2^20, where it measured slower; never where a seam plan would cost a pass more (2^17, 2^18)."""
    from oracle import oracle as O
    W = 3 if logn >= 21 else 6
    x = O.random_field((W, 1 << logn), 0x5EA5 + logn)
    want = O.lde(x, 1)
    d_in, d_out, d_coef = prover.upload(x), prover.alloc(W << (logn + 1)), prover.alloc(W << logn)
    plan = prover.ntt_plan(logn)["lde"]
    assert plan["seam_fused"] == (logn == 16 or logn >= 21), plan     # default: from 2^21 on (below, the seam plans measured slower: profiles/r5_lde_seam_plans_ab.txt)
    prover.set_tuning("lde_seam_plans", 2)                            # forced: wherever a seam plan costs no extra pass
    forced = prover.ntt_plan(logn)["lde"]
    prover.set_tuning("lde_seam_plans", 1)
    assert forced["seam_fused"] == (logn not in (17, 18)), forced
    if forced["seam_fused"]:
        assert forced["inverse_radix_logs"][-1] == 8 and forced["forward_radix_logs"][0] == 8
        assert sum(forced["inverse_radix_logs"]) == logn and sum(forced["forward_radix_logs"]) == logn + 1
    try:
        for knob in (2, 1, 0):
            prover.set_tuning("lde_seam_plans", knob)
            prover.memset(d_out, 0, (W << (logn + 1)) * 8)
            prover.lde(d_in, d_out, logn, 1, W)
            assert (prover.download(d_out, (W, 1 << (logn + 1))) == want).all(), "lde_seam_plans=%d" % knob
        prover.set_tuning("lde_seam_plans", 2)
        prover.set_tuning("lde_seam", 2)              # the fused kernel WITH the coefficient store, on the seam plans
        prover.lde(d_in, d_out, logn, 1, W, d_coef=d_coef)
        assert (prover.download(d_out, (W, 1 << (logn + 1))) == want).all()
        prover.set_tuning("lde_seam", 0)
        c0 = prover.alloc(W << logn)
        prover.lde(d_in, d_out, logn, 1, W, d_coef=c0)
        assert (prover.download(d_coef, (W, 1 << logn)) == prover.download(c0, (W, 1 << logn))).all()
        c0.free()
    finally:
        prover.set_tuning("lde_seam", 1)
        prover.set_tuning("lde_seam_plans", 1)
    for d in (d_in, d_out, d_coef):
        d.free()

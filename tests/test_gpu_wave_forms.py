"""The in-wave forms behind knobs (csrc/wave_xchg.hpp: DPP quad permutes, ds_swizzle, ds_bpermute) are the same functions as the product's
LDS / register forms: small transforms with their last six stages as lane exchanges (ntt_small_wave) against the oracle at every size the small
kernel serves, the FRI fold by 16 over DPP rows (fri_fold_lanes) against the oracle's fold.  The A/B that decides where they run by default (knob 0: transforms of <= 64 points, folds of <= 2^16 inputs; 1: always, 2: never):
profiles/r5_dpp_ab.txt."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("logn", list(range(0, 13)))
def test_small_transforms_with_in_wave_stages_equal_the_oracle(prover, logn):
    W = 5
    x = O.random_field((W, 1 << logn), 700 + logn)
    d, o = prover.upload(x), prover.alloc(W << logn)
    prover.set_tuning("ntt_small_wave", 1)
    try:
        prover.ntt(d, o, logn, W)
        assert (prover.download(o, x.shape) == O.ntt(x)).all()
        prover.intt(d, o, logn, W)
        assert (prover.download(o, x.shape) == O.intt(x)).all()
        if logn >= 1:                       # the extension's zero-padded, coset-scaled forward transform runs through the same kernel
            e = prover.alloc(W << (logn + 1)) if logn < 12 else None
            if e is not None:
                prover.lde(d, e, logn, 1, W, 7)
                assert (prover.download(e, (W, 2 << logn)) == O.lde(x, 1, 7)).all()
                e.free()
    finally:
        prover.set_tuning("ntt_small_wave", 0)
        d.free(); o.free()


@pytest.mark.parametrize("logn", [4, 5, 9, 14, 18])
def test_fold_by_16_over_dpp_rows_equals_the_oracle(prover, logn):
    x = O.random_field((3, 1 << logn), 800 + logn)
    beta = [int(v) for v in O.random_field((3,), 5)]
    d, o = prover.upload(x), prover.alloc(3 << (logn - 4))
    prover.set_tuning("fri_fold_lanes", 1)
    try:
        prover.fri_fold(d, o, logn, 4, beta, 7)
        got = prover.download(o, (3, 1 << (logn - 4)))
    finally:
        prover.set_tuning("fri_fold_lanes", 0)
    want = O.fri_fold(x, 4, beta, 7)
    assert (got == want).all()
    prover.set_tuning("fri_fold_lanes", 2)               # ... and the register form
    try:
        prover.fri_fold(d, o, logn, 4, beta, 7)
        assert (prover.download(o, (3, 1 << (logn - 4))) == want).all()
    finally:
        prover.set_tuning("fri_fold_lanes", 0)
    d.free(); o.free()

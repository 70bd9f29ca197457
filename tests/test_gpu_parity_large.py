"""Parity cases the round-2 review found uncovered (VERDICT.md "Next round" item 1): every kernel instantiation that the
product can reach is compared with the CPU oracle directly, not only through sampled openings of whole proofs.

  * merkle_level_kernel<true/false> (csrc/poseidon.hip): the lane-per-node kernel of every tree level above 2^15 nodes --
    whole trees at M = 2^17 and at BASELINE configs[1]'s 2^21 x 32 against O.merkle_commit, with the default MDS (inline
    constants) and an injected one (LDS table), and at small sizes with the cooperative threshold lowered;
  * the two-pass NTT plans (ntt_maxl 10 / 11 / 12: 1024-thread workgroups, radix 2^10..2^12) against the oracle;
  * the 64-bit-offset (BIG) instantiation: one 2^29-row column through size-independent properties.
Bar: bit-exact.  Serves GenChunkProof (reference: src/prover/provider.rs:358-390); parity with the external prover is
unpinned."""
import numpy as np
import pytest

from eigen_zeth_amd import native
from oracle import oracle as O

pytestmark = pytest.mark.gpu
P = O.P


def _commit(prover, cols):
    W, M = cols.shape
    d_cols = prover.upload(cols)
    d_tree = prover.alloc((2 * M - 1) * 4)
    prover.merkle_commit(d_cols, M, W, d_tree)
    got = prover.download(d_tree, (2 * M - 1, 4))
    d_cols.free()
    d_tree.free()
    return got


@pytest.mark.parametrize("M,W", [(1 << 17, 16), (1 << 16, 9)])
def test_merkle_level_kernel_whole_tree_default_mds(prover, tables, M, W):
    """levels of 2^17 and 2^16 nodes run merkle_level_kernel<true>; every node of the tree is compared"""
    rc, mds = tables
    cols = O.random_field((W, M), 3100 + W)
    assert (_commit(prover, cols) == O.merkle_commit(cols, rc, mds)).all()


def test_merkle_level_kernel_whole_tree_injected_mds(prover, tables):
    """an injected MDS takes the generic path: merkle_level_kernel<false>"""
    rc, mds = tables
    rc2 = O.random_field((360,), 19)
    mds2 = (O.random_field((144,), 20) % np.uint64(1 << 20)).astype(np.uint64)
    M, W = 1 << 17, 12
    cols = O.random_field((W, M), 3200)
    try:
        prover.set_constants(native.ZP_CONST_POSEIDON_RC, rc2)
        prover.set_constants(native.ZP_CONST_POSEIDON_MDS, mds2)
        assert (_commit(prover, cols) == O.merkle_commit(cols, rc2, mds2)).all()
    finally:
        prover.set_constants(native.ZP_CONST_POSEIDON_RC, rc)
        prover.set_constants(native.ZP_CONST_POSEIDON_MDS, mds)


@pytest.mark.parametrize("M,W", [(1 << 10, 8), (1 << 13, 5), (1 << 9, 64)])
def test_merkle_cooperative_threshold_lowered(prover, tables, M, W):
    """merkle_coop_log = 8: levels above 256 nodes go to the lane-per-node kernel, so it is compared at small sizes too;
    the default (2^15) sends the same levels to merkle_subtree_kernel -- both give the oracle's tree"""
    rc, mds = tables
    cols = O.random_field((W, M), 3300 + W)
    ref = O.merkle_commit(cols, rc, mds)
    try:
        prover.set_tuning("merkle_coop_log", 8)
        assert (_commit(prover, cols) == ref).all()
    finally:
        prover.set_tuning("merkle_coop_log", 0)
    assert (_commit(prover, cols) == ref).all()


@pytest.mark.parametrize("M,W", [(1 << 15, 8), (1 << 10, 5), (1 << 7, 12), (64, 9), (4, 8), (2, 3)])
def test_merkle_top_levels_by_wave_shuffles(prover, tables, M, W):
    """knob merkle_top_wave (round 5): the subtree kernel of the small levels with the permutation state exchanged by 64-bit wave shuffles
    (12 lanes of a 16-lane row per node, one workgroup barrier per LEVEL) instead of LDS and two barriers per round -- every node of the tree
    against the oracle, for the default and for an injected MDS matrix, next to the LDS form"""
    rc, mds = tables
    cols = O.random_field((W, M), 3400 + W)
    ref = O.merkle_commit(cols, rc, mds)
    try:
        for knob in (1, 0):
            prover.set_tuning("merkle_top_wave", knob)
            assert (_commit(prover, cols) == ref).all(), knob
        mds2 = (O.random_field((144,), 21) % np.uint64(1 << 20)).astype(np.uint64)
        prover.set_constants(native.ZP_CONST_POSEIDON_MDS, mds2)
        prover.set_tuning("merkle_top_wave", 1)
        assert (_commit(prover, cols) == O.merkle_commit(cols, rc, mds2)).all()
    finally:
        prover.set_constants(native.ZP_CONST_POSEIDON_MDS, mds)
        prover.set_tuning("merkle_top_wave", 0)


def test_config1_ntt_and_merkle_2p20_rows_bit_exact_vs_cpu(prover, tables):
    """BASELINE.json configs[1]: "2^20-row Goldilocks NTT + Poseidon Merkle on 1 MI355X, bit-exact vs CPU" -- the whole
    NTT output, the whole LDE (blow-up 2) and EVERY node of the 2^21-leaf x 32-column tree against the CPU oracle"""
    rc, mds = tables
    logn, W = 20, 32
    x = O.random_field((W, 1 << logn), 0xE16E2E70 + 1)
    d_in = prover.upload(x)
    d_f = prover.alloc(W << logn)
    prover.ntt(d_in, d_f, logn, W)
    assert (prover.download(d_f, (W, 1 << logn)) == O.ntt(x)).all()
    d_f.free()
    M = 2 << logn
    d_ext = prover.alloc(W * M)
    prover.lde(d_in, d_ext, logn, 1, W)
    ext = prover.download(d_ext, (W, M))
    assert (ext == O.lde(x, 1)).all()
    d_tree = prover.alloc((2 * M - 1) * 4)
    prover.merkle_commit(d_ext, M, W, d_tree)
    got = prover.download(d_tree, (2 * M - 1, 4))
    ref = O.merkle_commit(ext, rc, mds)
    assert (got == ref).all()
    for idx in (0, M - 1, 123457):
        assert (prover.merkle_open(d_tree, M, idx) == O.merkle_path(ref, idx)).all()
    for b in (d_in, d_ext, d_tree):
        b.free()


@pytest.mark.parametrize("maxl,logn,W", [(10, 20, 3), (11, 22, 2), (12, 24, 1), (12, 23, 1), (11, 21, 2), (10, 19, 2)])
def test_two_pass_ntt_plans_match_oracle(prover, maxl, logn, W):
    """ntt_maxl = 10 / 11 / 12: the radix-2^10 .. 2^12 passes (1024-thread workgroups, twiddles from L2, XCD-aware tile
    order) -- forward, inverse and a zero-padded LDE against the oracle"""
    x = O.random_field((W, 1 << logn), 4100 + maxl + logn)
    x[0, :4] = np.array([0, P - 1, 1, 2 ** 32], dtype=np.uint64)
    ref = O.ntt(x)
    try:
        prover.set_tuning("ntt_maxl", maxl)
        plan = prover.ntt_plan(logn)
        assert len(plan["passes"]) == 2 and max(q["radix_log"] for q in plan["passes"]) == maxl, plan
        d_in = prover.upload(x)
        d_out = prover.alloc(W << logn)
        prover.ntt(d_in, d_out, logn, W)
        assert (prover.download(d_out, (W, 1 << logn)) == ref).all()
        prover.intt(d_out, d_out, logn, W)
        assert (prover.download(d_out, (W, 1 << logn)) == x).all()
        if logn <= 20:
            d_ext = prover.alloc(W << (logn + 1))
            prover.lde(d_in, d_ext, logn, 1, W)
            assert (prover.download(d_ext, (W, 2 << logn)) == O.lde(x, 1)).all()
            d_ext.free()
        d_in.free()
        d_out.free()
    finally:
        prover.set_tuning("ntt_maxl", 0)


@pytest.mark.parametrize("logn,W", [(16, 3), (18, 2), (19, 9), (20, 8), (21, 5)])
def test_first_pass_table_and_chain_forms_match_oracle(prover, logn, W):
    """the transposing first pass in both forms -- MODE 3 (default: the N twiddles from one table shared by the columns, a
    one-dimensional XCD-ordered grid, any column count) and the per-lane chain (knob ntt_tw1 = 0) -- forward, inverse and a
    zero-padded LDE against the oracle"""
    x = O.random_field((W, 1 << logn), 4300 + logn)
    x[0, :4] = np.array([0, P - 1, 1, 2 ** 32], dtype=np.uint64)
    ref = O.ntt(x)
    ext = O.lde(x, 1) if logn <= 20 else None
    d_in, d_out = prover.upload(x), prover.alloc(W << logn)
    d_ext = prover.alloc(W << (logn + 1)) if ext is not None else None
    try:
        for tw1 in (26, 0):
            prover.set_tuning("ntt_tw1", tw1)
            # radix-2^7 / 2^8 first passes take the table; 2^17 and 2^18 rows are two radix-512 passes (three rounds): chain only
            assert prover.ntt_plan(logn)["first_pass_table"] == (tw1 > 0 and logn != 18)
            prover.ntt(d_in, d_out, logn, W)
            assert (prover.download(d_out, (W, 1 << logn)) == ref).all()
            prover.intt(d_out, d_out, logn, W)
            assert (prover.download(d_out, (W, 1 << logn)) == x).all()
            if ext is not None:
                prover.lde(d_in, d_ext, logn, 1, W)
                assert (prover.download(d_ext, (W, 2 << logn)) == ext).all()
    finally:
        prover.set_tuning("ntt_tw1", 26)
        d_in.free()
        d_out.free()
        if d_ext is not None:
            d_ext.free()


def test_ntt_2p29_rows_uses_64bit_offsets(prover):
    """logn = 29 > 28: the BIG instantiation (64-bit lane offsets).  One 4 GiB column, size-independent properties:
    iNTT(NTT(x)) = x, X[0] = sum x, X[N/2] = alternating sum, and the transform of a delta is the powers of the root"""
    logn = 29
    n = 1 << logn
    g = np.random.default_rng(29)
    x = g.integers(0, P, size=(1, n), dtype=np.uint64)
    d = prover.upload(x)
    d2 = prover.alloc(n)
    prover.ntt(d, d2, logn, 1)
    fx = prover.download(d2, (1, n))[0]
    step = 1 << 22
    tot = alt = 0
    for i in range(0, n, step):
        blk = x[0, i:i + step]
        tot += int(np.sum(blk.astype(object)))
        alt += int(np.sum(blk[0::2].astype(object))) - int(np.sum(blk[1::2].astype(object)))
    assert int(fx[0]) == tot % P
    assert int(fx[n // 2]) == alt % P
    prover.intt(d2, d2, logn, 1)
    back = prover.download(d2, (1, n))
    assert (back == x).all()
    del back, fx
    x[:] = 0
    x[0, 1] = 1
    prover.h2d(d, x)
    prover.ntt(d, d2, logn, 1)
    got = prover.download(d2, (1, n))[0]
    w = O.lib().orc_root(O.ROOT32_DEFAULT, logn)
    for k in [0, 1, 2, 3, n // 2, n - 1, 12345, (1 << 28) + 7, (1 << 28) - 1]:
        assert int(got[k]) == pow(w, k, P)
    d.free()
    d2.free()

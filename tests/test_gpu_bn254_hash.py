"""GPU tests of the BN128-hash mode (Poseidon over the BN254 scalar field, 16-ary Merkle tree) against the definition-level
Python in oracle/naive.py, anchored on the published t = 3 vector and (round 6) on the published 16-input vector of the width-17 instance --
the width the final STARK's trees and the wrap circuit's gadget use."""
import random

import numpy as np
import pytest

from eigen_zeth_amd import poseidon_constants as PC
from oracle import naive as NV
from oracle import oracle as O

pytestmark = pytest.mark.gpu
R = NV.BN254_R


@pytest.fixture(scope="module")
def p254(prover):
    prover.install_poseidon_bn254(3)
    prover.install_poseidon_bn254(17)
    return prover


def test_published_vector_on_the_gpu(p254):
    out = p254.poseidon_bn254_perm([[0, 1, 2]])
    assert out[0][0] == 0x115CC0F5E7D690413DF64C6B9662E9CF2A3617F2743245519E19607A4417189A


PUBLISHED_T17_HASH_1_TO_16 = 9989051620750914585850546081941653841776809718687451684622678807385399211877


def test_published_width17_vector_on_the_gpu(p254):
    """the public JavaScript / circuit library of this hash family publishes poseidon([1, 2, ..., 16]) for its 16-input (t = 17, R_P = 68)
    instance; written down from memory BEFORE it was computed here (tests/test_poseidon_constants.py has the CPU side).  Through the C-ABI: the
    cooperative kernel (one state), the lane-per-permutation kernel with its partial rounds in blocks (2^14 + 3 copies of the state), and the
    round-by-round form of that kernel (knob p254_block = 2) all return it in element 0"""
    st = [0] + list(range(1, 17))
    assert p254.poseidon_bn254_perm([st])[0][0] == PUBLISHED_T17_HASH_1_TO_16
    count = (1 << 14) + 3
    words = np.zeros((count, 17, 4), dtype=np.uint64)
    words[:, :, 0] = np.arange(17, dtype=np.uint64)
    try:
        for knob in (0, 2):
            p254.set_tuning("p254_block", knob)
            d = p254.upload(words.reshape(-1))
            p254._chk(p254.lib.zp_poseidon_bn254_perm(p254.ctx, d.ptr, count, 17))
            got = p254.download(d, words.shape)
            d.free()
            assert (got == got[0]).all()
            assert sum(int(got[0, 0, k]) << (64 * k) for k in range(4)) == PUBLISHED_T17_HASH_1_TO_16
    finally:
        p254.set_tuning("p254_block", 0)


@pytest.mark.parametrize("t,count", [(3, 1), (3, 50), (17, 1), (17, 7)])
def test_permutation_matches_the_definition(p254, t, count):
    rc, mds, rp = PC.bn254_poseidon_params(t)
    rnd = random.Random(t * 100 + count)
    states = [[rnd.randrange(R) for _ in range(t)] for _ in range(count)]
    states[0][:3] = [0, R - 1, 1]
    got = p254.poseidon_bn254_perm(states)
    for st, g in zip(states, got):
        assert g == NV.poseidon_bn254_perm(st, rc, mds, rp)


@pytest.mark.parametrize("M,W", [(1, 5), (16, 3), (20, 50), (64, 49), (300, 7), (33, 52), (17, 56), (40, 57), (5, 120)])
def test_merkle16_tree_and_openings(p254, M, W):
    rc, mds, rp = PC.bn254_poseidon_params(17)
    cols = O.random_field((W, M), 5000 + M)
    d_cols = p254.upload(cols)
    nodes = p254.merkle16_nodes(M)
    d_tree = p254.alloc(nodes * 4)
    p254.merkle16_commit_bn254(d_cols, M, W, d_tree)
    tree = p254._fr_ints(p254.download(d_tree, (nodes * 4,)))
    levels = NV.merkle16_tree([[int(v) for v in cols[:, i]] for i in range(M)], rc, mds, rp)
    assert tree == [v for lvl in levels for v in lvl]
    for idx in {0, M - 1, M // 2}:
        path = p254.merkle16_open_bn254(d_tree, M, idx)
        pos, digest = idx, levels[0][idx]
        for l, grp in enumerate(path):       # re-hash upwards from the opened groups
            assert grp[pos % 16] == digest
            digest = NV.poseidon_bn254_perm([0] + grp, rc, mds, rp)[0]
            pos //= 16
        assert digest == levels[-1][0]


def test_tables_must_be_installed_and_reduced(prover):
    from eigen_zeth_amd import native
    rc, mds, rp = PC.bn254_poseidon_params(3)
    bad = prover._fr_words([R] + rc[1:])
    mw = prover._fr_words([v for row in mds for v in row])
    with pytest.raises(native.ZpError):
        prover._chk(prover.lib.zp_set_poseidon_bn254(prover.ctx, 3, rp, bad.ctypes.data, mw.ctypes.data))
    with pytest.raises(native.ZpError):
        prover._chk(prover.lib.zp_set_poseidon_bn254(prover.ctx, 5, rp, mw.ctypes.data, mw.ctypes.data))


def test_merkle16_batch_openings_equal_single_openings(prover):
    import numpy as np
    prover.install_poseidon_bn254(17)
    rng = np.random.default_rng(3)
    for M, W in ((1, 4), (16, 3), (17, 5), (300, 7), (4096, 9)):
        cols = rng.integers(0, 1 << 63, size=(W, M), dtype=np.uint64)
        d = prover.upload(cols)
        tree = prover.alloc(prover.merkle16_nodes(M) * 4)
        prover.merkle16_commit_bn254(d, M, W, tree)
        idx = sorted(set([0, M - 1, M // 2] + [int(v) for v in rng.integers(0, M, size=5)]))
        batch = prover.merkle16_open_batch_bn254(tree, M, idx)
        assert batch == [prover.merkle16_open_bn254(tree, M, i) for i in idx]


def _commit16(p, cols):
    W, M = cols.shape
    d_cols = p.upload(cols)
    nodes = p.merkle16_nodes(M)
    d_tree = p.alloc(nodes * 4)
    p.merkle16_commit_bn254(d_cols, M, W, d_tree)
    out = p.download(d_tree, (nodes, 4))
    d_cols.free()
    d_tree.free()
    return out


def test_lane_per_permutation_kernel_equals_the_cooperative_one_and_the_oracle(p254):
    """launches of >= 2^14 permutations (the commitments of the final STARK) take p254_bulk_kernel: one lane per permutation,
    state in LDS, dot products of six terms per Montgomery reduction.  It must write the words of the cooperative kernel
    (p254_bulk_log = 31 switches it off) and of the CPU checker (oracle/bn254_hash.c, textbook schedule)."""
    rc, mds, rp = PC.bn254_poseidon_params(17)
    O.p254_set(17, rp, rc, mds)
    count = (1 << 14) + 37
    g = np.random.default_rng(254)
    st = g.integers(0, 1 << 62, size=(count, 17, 4), dtype=np.uint64)
    st[:, :, 3] >>= 2                                   # < 2^252 < r
    st[0, :3] = 0
    st[0, 1, 0] = 1
    d = p254.upload(st.reshape(-1))
    p254._chk(p254.lib.zp_poseidon_bn254_perm(p254.ctx, d.ptr, count, 17))
    bulk = p254.download(d, st.shape)
    try:
        p254.set_tuning("p254_bulk_log", 31)
        p254.h2d(d, st.reshape(-1))
        p254._chk(p254.lib.zp_poseidon_bn254_perm(p254.ctx, d.ptr, count, 17))
        coop = p254.download(d, st.shape)
    finally:
        p254.set_tuning("p254_bulk_log", 0)
    assert (bulk == coop).all()
    sample = [0, 1, 63, 64, 8191, count - 1]
    ints = lambda a: [sum(int(a[e, k]) << (64 * k) for k in range(4)) for e in range(17)]
    want = O.p254_perm([ints(st[i]) for i in sample], 17)
    assert [ints(bulk[i]) for i in sample] == want


@pytest.mark.parametrize("M,W", [(1 << 14, 26), (1 << 14, 9), ((1 << 14) + 48, 50), (1 << 18, 4), (1 << 14, 52), ((1 << 14) + 5, 60)])
def test_merkle16_bulk_kernels_match_cooperative_and_oracle(p254, M, W):
    """leaves (one and two sponge blocks per leaf, a leaf count that is no multiple of 64) and a tree level of 2^14 nodes
    through the lane-per-permutation kernel: the whole tree equals the cooperative kernels' tree; up to 2^14 leaves it also
    equals the CPU checker's tree node by node"""
    rc, mds, rp = PC.bn254_poseidon_params(17)
    O.p254_set(17, rp, rc, mds)
    cols = O.random_field((W, M), 7000 + W)
    bulk = _commit16(p254, cols)
    try:
        p254.set_tuning("p254_bulk_log", 31)
        coop = _commit16(p254, cols)
    finally:
        p254.set_tuning("p254_bulk_log", 0)
    assert (bulk == coop).all()
    if M <= (1 << 14) + 64:
        assert (bulk == O.merkle16_tree(cols)).all()


def test_partial_rounds_in_blocks_equal_round_by_round(p254):
    """round 6: the lane-per-permutation kernel walks the partial rounds in blocks of four (cross terms made on the host when the tables are
    installed; the last round of a leaf block / tree node computes element 0 only).  Knob p254_block = 2 walks them one by one, as rounds 3-5
    did: whole permutations (17 outputs, generic entry point) and a tree with two sponge blocks per leaf must come out word for word the
    same -- and both already equal the cooperative kernels and the CPU checker (the two tests above run on the blocked form)"""
    count = (1 << 14) + 5
    g = np.random.default_rng(2540)
    st = g.integers(0, 1 << 62, size=(count, 17, 4), dtype=np.uint64)
    st[:, :, 3] >>= 2
    cols = O.random_field((60, (1 << 14) + 64), 7777)
    res = {}
    try:
        for knob in (0, 2):
            p254.set_tuning("p254_block", knob)
            d = p254.upload(st.reshape(-1))
            p254._chk(p254.lib.zp_poseidon_bn254_perm(p254.ctx, d.ptr, count, 17))
            res[knob] = (p254.download(d, st.shape), _commit16(p254, cols))
    finally:
        p254.set_tuning("p254_block", 0)
    assert (res[0][0] == res[2][0]).all() and (res[0][1] == res[2][1]).all()

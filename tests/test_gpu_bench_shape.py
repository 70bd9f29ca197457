"""The kernels bench.py times, at the shape it times them, element by element against the CPU oracle -- and configs[3] at its full
size on the one-GPU box.

* BASELINE.json configs[3]'s one-GPU leg (what `python bench.py` reports): zp_ntt / zp_intt / zp_lde (blow-up 2) of 2^24 rows x 64
  columns with the DEFAULT plan (three radix-256 passes, 16-column XCD-ordered launches (2^28 elements per launch): the workgroup -> (XCD, column, tile) map of
  csrc/ntt.hip depends on the column count of a launch, so one column of a W = 2 launch does not exercise it).  EVERY column is
  compared with oracle.ntt / intt / lde.  Host memory stays at a few GiB: slices of 8 columns are generated from their seed, uploaded
  into the resident matrix, and generated again when their transform is compared.
* one whole column each at 2^23, 2^25 and 2^26 against the oracle (round 3 checked 2^25 / 2^26 through properties only).
* configs[3] as stated -- 2^24-row trace, blow-up 2, 64 columns sharded 8-way -- with 8 THREAD-ranks on the one GPU (in-process
  communicator: RCCL refuses two ranks per device): zp_merkle_commit_sharded over M = 2^25 rows x 64 columns (8 per rank) against
  zp_merkle_commit of the whole matrix, and zp_ntt_sharded of one 2^28-element column against zp_ntt of the whole column.
"""
import threading

import numpy as np
import pytest

from eigen_zeth_amd import native
from oracle import oracle as O

pytestmark = pytest.mark.gpu

LOGN, W, SLICE = 24, 64, 8


def _slice(k):
    return O.random_field((SLICE, 1 << LOGN), 0xE16E2E70 + 3 + 1000 * k)      # seed = 0xE16E2E70 + config id


def _fill(prover, d_in):
    for k in range(W // SLICE):
        x = np.ascontiguousarray(_slice(k))
        prover._chk(prover.lib.zp_h2d(prover.ctx, d_in.offset(k * SLICE << LOGN), x.ctypes.data, x.nbytes))


def _download_slice(prover, d, k, rows):
    out = np.empty((SLICE, rows), dtype=np.uint64)
    prover._chk(prover.lib.zp_d2h(prover.ctx, out.ctypes.data, d.offset(k * SLICE * rows), out.nbytes))
    return out


def test_bench_shape_ntt_and_intt_every_column_against_the_oracle(prover):
    N = 1 << LOGN
    d_in, d_out = prover.alloc(W * N), prover.alloc(W * N)
    try:
        _fill(prover, d_in)
        prover.ntt(d_in, d_out, LOGN, W)            # ONE call over the 64 resident columns, as bench.py's timed step
        for k in range(W // SLICE):
            assert (_download_slice(prover, d_out, k, N) == O.ntt(_slice(k))).all(), "forward, columns %d.." % (k * SLICE)
        prover.intt(d_in, d_out, LOGN, W)
        for k in (0, 2, 5, 7):      # (round 6: the inverse on half of the slices -- one from every launch of the call; the suite's wall time)
            assert (_download_slice(prover, d_out, k, N) == O.intt(_slice(k))).all(), "inverse, columns %d.." % (k * SLICE)
        # in place, as the bench's timed loop runs it (d -> d)
        prover.ntt(d_in, d_in, LOGN, W)
        for k in (0, 3, 7):
            assert (_download_slice(prover, d_in, k, N) == O.ntt(_slice(k))).all(), "in place, columns %d.." % (k * SLICE)
    finally:
        d_in.free()
        d_out.free()


def test_bench_shape_lde_every_column_against_the_oracle(prover):
    """the shape bench.py's pipeline.lde_ms times: 2^24 rows x 32 columns, blow-up 2 (round 6: 32 columns, as the bench line has it, instead of 64: the
    CPU checker's extensions are what this test waits for)"""
    N, M = 1 << LOGN, 1 << (LOGN + 1)
    W = 32
    d_in, d_out, d_coef = prover.alloc(W * N), prover.alloc(W * M), prover.alloc(W * N)
    try:
        for k in range(W // SLICE):
            x = np.ascontiguousarray(_slice(k))
            prover._chk(prover.lib.zp_h2d(prover.ctx, d_in.offset(k * SLICE << LOGN), x.ctypes.data, x.nbytes))
        prover.lde(d_in, d_out, LOGN, 1, W)
        for k in range(W // SLICE):
            assert (_download_slice(prover, d_out, k, M) == O.lde(_slice(k), 1)).all(), "lde, columns %d.." % (k * SLICE)
        # the form the prover uses: coefficients kept (c_i shift^i)
        prover.lde(d_in, d_out, LOGN, 1, W, d_coef=d_coef)
        for k in (1, 3):
            x = _slice(k)
            assert (_download_slice(prover, d_out, k, M) == O.lde(x, 1)).all()
            assert (_download_slice(prover, d_coef, k, N) == O.coset_scaled_coefficients(x)).all()
    finally:
        d_in.free()
        d_out.free()
        d_coef.free()


@pytest.mark.parametrize("logn", [23, 25, 26])
def test_one_whole_column_against_the_oracle(prover, logn):
    """(8,8,7), (9,8,8) and (9,9,8): the radix-512 passes, element by element"""
    n = 1 << logn
    x = O.random_field((1, n), 4321 + logn)
    d = prover.upload(x)
    o = prover.alloc(n)
    try:
        prover.ntt(d, o, logn, 1)
        assert (prover.download(o, (1, n)) == O.ntt(x)).all()
        prover.intt(d, o, logn, 1)
        assert (prover.download(o, (1, n)) == O.intt(x)).all()
    finally:
        d.free()
        o.free()


def _run_ranks(G, fn):
    group = native.CommGroup(G)
    out, err = [None] * G, [None] * G

    def body(r):
        p = None
        try:
            p = native.Prover(0)
            c = native.Comm(p, r, G, group=group)
            out[r] = fn(r, p, c)
            c.close()
        except BaseException as e:      # noqa
            err[r] = e
        finally:
            if p is not None:
                p.close()
    ts = [threading.Thread(target=body, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    group.close()
    assert not any(t.is_alive() for t in ts), "a rank is stuck in a collective"
    for e in err:
        if e is not None:
            raise e
    return out


def test_config3_sharded_commitment_at_full_size_with_8_thread_ranks(prover):
    """M = 2^25 LDE rows x 64 columns, 8 columns per rank: pack + all-to-all (256 MiB per peer) + local subtree over 2^22 rows of
    all 64 columns + all-gather of 8 sub-roots + tree top -- equal to zp_merkle_commit of the whole matrix, root AND every rank's
    local subtree (leaves and the levels above them are slices of the whole tree's levels)."""
    G, logm = 8, LOGN + 1
    M, wl = 1 << logm, W // G
    Ml = M // G
    d_mat = prover.alloc(W * M)                     # 16 GiB, generated on the device: column 0 random, column k+1 = NTT(column k)
    d_tree = prover.alloc((2 * M - 1) * 4)
    try:
        x0 = O.random_field((1, M), 0xC3)
        prover._chk(prover.lib.zp_h2d(prover.ctx, d_mat.ptr, x0.ctypes.data, x0.nbytes))
        for k in range(1, W):
            prover._chk(prover.lib.zp_ntt(prover.ctx, d_mat.offset((k - 1) * M), d_mat.offset(k * M), logm, 1))
        prover.sync()
        # the generator is checked against the oracle on the columns a mistake would show in: the first hop and the last
        c1 = np.empty((1, M), dtype=np.uint64)
        prover._chk(prover.lib.zp_d2h(prover.ctx, c1.ctypes.data, d_mat.offset(M), c1.nbytes))
        assert (c1 == O.ntt(x0)).all()
        prover.merkle_commit(d_mat, M, W, d_tree)
        prover.sync()
        root = np.empty((1, 4), dtype=np.uint64)
        prover._chk(prover.lib.zp_d2h(prover.ctx, root.ctypes.data, d_tree.offset((2 * M - 2) * 4), 32))
        want = [int(v) for v in root[0]]

        def fn(r, p, c):
            # thread-ranks share the device: rank r's columns are a window of the resident matrix (no copy)
            d_loc = p.alloc((2 * Ml - 1) * 4)
            got = c.merkle_commit_sharded(d_mat.offset(r * wl * M), M, wl, d_loc)
            leaves = p.download(d_loc, (2 * Ml - 1, 4))[:Ml] if r in (0, 5) else None
            top = p.download(d_loc, (2 * Ml - 1, 4))[-3:] if r in (0, 5) else None
            d_loc.free()
            return got, leaves, top
        res = _run_ranks(G, fn)
        assert all(got == want for got, _, _ in res), "sharded root differs from the single-GPU root"
        for r in (0, 5):
            ref = np.empty((Ml, 4), dtype=np.uint64)
            prover._chk(prover.lib.zp_d2h(prover.ctx, ref.ctypes.data, d_tree.offset(r * Ml * 4), ref.nbytes))
            assert (res[r][1] == ref).all(), "rank %d: leaf digests" % r
            # the rank's sub-root = node r of the whole tree's level log2(G) from the top
            lvl_start = (2 * M - 1) - (2 * G - 1)           # first node of the level with G nodes
            sub = np.empty((1, 4), dtype=np.uint64)
            prover._chk(prover.lib.zp_d2h(prover.ctx, sub.ctypes.data, d_tree.offset((lvl_start + r) * 4), 32))
            assert (res[r][2][-1] == sub[0]).all(), "rank %d: sub-root" % r
    finally:
        d_mat.free()
        d_tree.free()


def test_config3_four_step_ntt_of_one_2p28_column_with_8_thread_ranks(prover):
    """the shape `bench.py --gpus 8` runs (pipeline.four_step_single_column): N = 2^28 = 2^14 x 2^14, three all-to-all transposes"""
    logn, G = 28, 8
    N = 1 << logn
    x = O.random_field((1, N), 0xC4)[0]
    d, o = prover.upload(x), prover.alloc(N)
    prover.ntt(d, o, logn, 1)
    want = prover.download(o, (N,))
    prover.intt(o, d, logn, 1)
    assert (prover.download(d, (N,)) == x).all()      # the single-GPU transform it is compared with inverts (64-bit-offset-free limit: 2^28)
    d.free()
    o.free()
    # anchor of the reference value itself: X[0] = sum x, X[N/2] = alternating sum
    P = (1 << 64) - (1 << 32) + 1
    xo = x.astype(object)
    assert int(want[0]) == int(xo.sum() % P) and int(want[N // 2]) == int((xo[0::2].sum() - xo[1::2].sum()) % P)

    def fn(r, p, c):
        blk = np.ascontiguousarray(x[r * (N // G):(r + 1) * (N // G)])
        dd, t = p.upload(blk), p.alloc(2 * (N // G))
        c.ntt_sharded(dd, t, logn)
        got = p.download(dd, (N // G,))
        c.ntt_sharded(dd, t, logn, inverse=True)
        back = p.download(dd, (N // G,))
        dd.free()
        t.free()
        return got, bool((back == blk).all())
    res = _run_ranks(G, fn)
    assert (np.concatenate([g for g, _ in res]) == want).all()
    assert all(ok for _, ok in res)

"""GPU tests of the STARK stage (-m gpu): per-kernel parity against the oracle, then whole proofs:
the MI355X proof must (a) be bit-identical to the proof the CPU restatement produces from the same
witness and (b) pass the independent verifier (oracle/stark_verify.py)."""
import copy

import numpy as np
import pytest

from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark import prover as PR
from oracle import naive as NV
from oracle import oracle as O
from oracle import stark_verify as V
from oracle.stark_cpu import CpuBackend

pytestmark = pytest.mark.gpu
P = O.P


@pytest.fixture(scope="module")
def hip_backend(prover):
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    return HipBackend(prover=prover)


@pytest.fixture(scope="module")
def cpu_backend(tables):
    return CpuBackend(*tables)


@pytest.mark.parametrize("logn,W", [(4, 1), (12, 3), (13, 5), (16, 2)])
def test_poly_eval_ext_matches_oracle(prover, logn, W):
    coef = O.random_field((W, 1 << logn), 40 + logn)
    z = O.random_field((3,), 41).tolist()
    got = prover.poly_eval_ext(prover.upload(coef), logn, W, z)
    assert (got == O.poly_eval_e3_cols(coef, z)).all()
    # base-field point embedded: must equal plain Horner
    got = prover.poly_eval_ext(prover.upload(coef), logn, W, [7, 0, 0])
    for c in range(W):
        assert got[c].tolist() == [O.poly_eval(coef[c], 7), 0, 0]


@pytest.mark.parametrize("logn,logb,W", [(0, 0, 1), (1, 1, 2), (4, 1, 1), (9, 0, 9), (12, 1, 3), (13, 2, 5), (16, 1, 6), (20, 1, 2)])
def test_ood_eval_from_values_matches_the_coefficient_definition(prover, logn, logb, W):
    """zp_ood_eval (barycentric form over the resident extension, round 5) against the oracle's definition: interpolate, then evaluate the
    coefficient form -- at zeta and at zeta w, from the 2^logn-point sub-coset of an extension with blow-up 2^logb (row stride 2^logb),
    on the coset (shift 49) and on the trace domain itself (shift 1), for a full ext point and a base-field point"""
    n, M = 1 << logn, 1 << (logn + logb)
    x = O.random_field((W, n), 0x00D0 + logn)
    coef = O.intt(x)                                                   # p_c(w^i) = x[c][i]
    wn = pow(O.ROOT32_DEFAULT, 1 << (32 - logn), P) if logn else 1
    for shift in (49, 1):
        # the columns as an extension would hold them: values of p on shift <w_M>, so that rows 0, 2^logb, ... are p(shift w_n^i)
        sc = np.array([[int(c) * pow(shift, i, P) % P for i, c in enumerate(row)] for row in coef.tolist()], dtype=np.uint64) if logn <= 13 else None
        if sc is None:      # large sizes: the library's own LDE lays the extension down (it is compared with the oracle elsewhere)
            if shift == 1:
                continue
            d_ext = prover.alloc(W * M)
            prover.lde(prover.upload(x), d_ext, logn, logb, W, shift)
        else:
            pad = np.zeros((W, M), dtype=np.uint64)
            pad[:, :n] = sc
            d_ext = prover.upload(O.ntt(pad))
        for z in (O.random_field((3,), 0x00D1 + logn).tolist(), [12345, 0, 0]):
            zw = [v * wn % P for v in z]
            got_z, got_zw = prover.ood_eval(d_ext, M, 1 << logb, W, logn, shift, z, want_next=True)
            assert (got_z == O.poly_eval_e3_cols(coef, z)).all()
            assert (got_zw == O.poly_eval_e3_cols(coef, zw)).all()
            assert (prover.ood_eval(d_ext, M, 1 << logb, W, logn, shift, z) == got_z).all()
        d_ext.free()
    # a point ON the domain is refused (the protocol excludes it), not answered with garbage
    if logn >= 1:
        d = prover.upload(x)
        with pytest.raises(native.ZpError):
            prover.ood_eval(d, n, 1, W, logn, 1, [pow(wn, 3, P), 0, 0])
        d.free()


@pytest.mark.parametrize("logm,Wa,Wb,nn", [(6, 3, 0, 0), (10, 5, 3, 5), (12, 4, 3, 2)])
def test_deep_quotient_matches_oracle(prover, logm, Wa, Wb, nn):
    a = O.random_field((Wa, 1 << logm), 50)
    b = O.random_field((max(Wb, 1), 1 << logm), 51)
    z, zw, g = (O.random_field((3,), s).tolist() for s in (52, 53, 54))
    ez = O.random_field((Wa + Wb, 3), 55)
    ezw = O.random_field((max(nn, 1), 3), 56)
    ref = O.deep_quotient(a, b[:Wb] if Wb else None, nn, z, zw, g, ez, ezw[:nn] if nn else None)
    assert (O.deep_quotient(a, b[:Wb] if Wb else None, nn, z, zw, g, ez, ezw[:nn] if nn else None, fast=True) == ref).all()
    d_out = prover.alloc(3 << logm)
    prover.deep_quotient(prover.upload(a), Wa, prover.upload(b) if Wb else None, Wb, logm, nn, z, zw, g, ez, ezw, 49, d_out)
    assert (prover.download(d_out, (3, 1 << logm)) == ref).all()


def test_gather_and_batch_open(prover, tables):
    rc, mds = tables
    M, W = 1 << 10, 7
    cols = O.random_field((W, M), 60)
    d = prover.upload(cols)
    idx = [0, M - 1, 5, 5, 777]
    assert (prover.gather_rows(d, M, W, idx) == cols[:, idx].T).all()
    tree = O.merkle_commit(cols, rc, mds)
    d_tree = prover.alloc((2 * M - 1) * 4)
    prover.merkle_commit(d, M, W, d_tree)
    paths = prover.merkle_open_batch(d_tree, M, idx)
    for i, j in enumerate(idx):
        assert (paths[i] == O.merkle_path(tree, j)).all()
    with pytest.raises(native.ZpError):
        prover.gather_rows(d, M, W, [M])


@pytest.mark.parametrize("n", [1, 2, 17, 4096, 4097, (1 << 16) + 5, 1 << 20])
def test_grand_product_matches_oracle(prover, n):
    a = O.random_field((n,), 80)
    b = a[np.random.default_rng(81).permutation(n)]
    g = O.random_field((3,), 82).tolist()
    d_out = prover.alloc(3 * n)
    prover.grand_product(prover.upload(a), prover.upload(b), n, g, d_out)
    got = prover.download(d_out, (3, n))
    if n <= (1 << 16) + 5:
        assert (got == O.grand_product(a, b, g)).all()
    else:   # size-independent properties: Z[0] = 1 and the recurrence at sampled rows
        assert got[:, 0].tolist() == [1, 0, 0]
        for i in [0, 1, 4095, 4096, n // 2, n - 2]:
            zi = [int(got[c, i]) for c in range(3)]
            zn = [int(got[c, i + 1]) for c in range(3)]
            lhs = NV.e3_mul(zn, [(int(b[i]) + g[0]) % P, g[1], g[2]])
            assert lhs == NV.e3_mul(zi, [(int(a[i]) + g[0]) % P, g[1], g[2]])


@pytest.mark.parametrize("n", [1, 2, 17, 4096, 4097, (1 << 16) + 5, 1 << 20])
def test_logup_columns_match_oracle(prover, n):
    rng = np.random.default_rng(90)
    k = max(1, min(12, n.bit_length() - 1))
    a = rng.integers(0, 1 << k, size=n, dtype=np.uint64)
    t_ = (np.arange(n, dtype=np.uint64) % np.uint64(1 << k))
    m = rng.integers(0, 5, size=n, dtype=np.uint64)     # the kernel does not care whether the lookup holds
    g = O.random_field((3,), 91).tolist()
    d_out = prover.alloc(9 * n)
    prover.logup_columns(prover.upload(a), prover.upload(t_), prover.upload(m), n, g, d_out)
    got = prover.download(d_out, (9, n))
    if n <= (1 << 16) + 5:
        assert (got == O.logup_columns(a, t_, m, g)).all()
    else:   # definitions at sampled rows: h1 (a+g) = 1, h2 (t+g) = m, S' = S + h1 - h2, S[0] = 0
        assert got[6:9, 0].tolist() == [0, 0, 0]
        for i in [0, 1, 4095, 4096, n // 2, n - 2]:
            h1 = [int(got[c, i]) for c in range(3)]
            h2 = [int(got[3 + c, i]) for c in range(3)]
            assert NV.e3_mul(h1, [(int(a[i]) + g[0]) % P, g[1], g[2]]) == [1, 0, 0]
            assert NV.e3_mul(h2, [(int(t_[i]) + g[0]) % P, g[1], g[2]]) == [int(m[i]), 0, 0]
            for c in range(3):
                assert int(got[6 + c, i + 1]) == (int(got[6 + c, i]) + h1[c] - h2[c]) % P


@pytest.mark.parametrize("name,logn", [("fib", 5), ("wide8", 8), ("wide32", 10)])
def test_constraint_kernel_matches_the_checkers_interpreter(hip_backend, cpu_backend, name, logn):
    """generated gfx950 kernel (product code generator) vs the checker's own interpreter of the constraint program blob
    (oracle/gl_oracle.c: orc_quotient_program) -- two lowerings that share no code"""
    air = AIR.get_air(name)
    tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 99)
    from eigen_zeth_amd.stark import field as F
    apow = [O.random_field((3,), 70 + k).tolist() for k in range(len(air.constraints))]
    zhinv = [3, 5]
    wl = F.inv(F.root(logn, hip_backend.root32))
    q_cpu = cpu_backend.quotient(air, cpu_backend.commit_trace(tr, logn, 1), cpu_backend.fixed_ext(logn, 1), pub, apow, zhinv, logn, 1, wl)
    c1 = hip_backend.commit_trace(tr, logn, 1)
    d_q = hip_backend.quotient(air, c1, hip_backend.fixed_ext(logn, 1), pub, apow, zhinv, logn, 1, wl)
    assert (hip_backend.download(d_q, q_cpu.shape) == q_cpu).all()


@pytest.mark.parametrize("name,logn,queries", [("fib", 6, 5), ("fib", 12, 8), ("perm", 7, 6), ("perm", 13, 8), ("wide8", 10, 8), ("wide32", 13, 12), ("wide64", 14, 8), ("chunk16", 8, 6), ("chunk64", 13, 8)])
def test_gpu_proof_is_bit_identical_to_cpu_and_verifies(hip_backend, cpu_backend, tables, name, logn, queries):
    rc, mds = tables
    air = AIR.get_air(name)
    tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 2024 + logn)
    params = PR.StarkParams(logn, logb=1, fri_logf=3, fri_final_log=4, n_queries=queries)
    gpu = PR.prove(air, tr, pub, params, hip_backend)
    cpu = PR.prove(air, tr, pub, params, cpu_backend)
    assert PR.proof_to_json(gpu) == PR.proof_to_json(cpu)
    assert V.verify(gpu, air.program(), rc, mds, V.expectation(params.to_dict()))
    bad = copy.deepcopy(gpu)
    bad["queries"][1]["fri"][0]["values"][0] ^= 1
    with pytest.raises(V.Reject):
        V.verify(bad, air.program(), rc, mds, V.expectation(params.to_dict()))


def test_gpu_rejects_bad_witness_downstream(hip_backend, tables):
    rc, mds = tables
    air = AIR.get_air("wide8")
    tr, pub = native.synth_trace(1, 9, 8, 3)
    tr[3, 100] = (int(tr[3, 100]) + 1) % P
    params = PR.StarkParams(9, 1, 3, 4, 6)
    proof = PR.prove(air, tr, pub, params, hip_backend)
    with pytest.raises(V.Reject):
        V.verify(proof, air.program(), rc, mds, V.expectation(params.to_dict()))


def test_blowup_four(hip_backend, cpu_backend, tables):
    rc, mds = tables
    air = AIR.get_air("fib")
    tr, pub = native.synth_trace(0, 9, 2, 8)
    params = PR.StarkParams(9, logb=2, fri_logf=2, fri_final_log=3, n_queries=6)
    gpu = PR.prove(air, tr, pub, params, hip_backend)
    assert PR.proof_to_json(gpu) == PR.proof_to_json(PR.prove(air, tr, pub, params, cpu_backend))
    assert V.verify(gpu, air.program(), rc, mds, V.expectation(params.to_dict()))


@pytest.mark.parametrize("name,logn", [("wide64", 22), ("wide8", 24), ("perm", 22), ("chunk64", 22)])
def test_full_size_proofs_pass_the_independent_verifier(hip_backend, tables, name, logn):
    """BASELINE configs[2]/[3] sizes: the CPU prover is too slow to compare against, but the verifier's
    cost does not depend on the trace length -- a 2^22..2^24-row proof from the MI355X must verify."""
    rc, mds = tables
    air = AIR.get_air(name)
    tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 31337)
    params = PR.StarkParams(logn, logb=1, fri_logf=3, fri_final_log=5, n_queries=16)
    proof = PR.prove(air, tr, pub, params, hip_backend)
    del tr
    assert V.verify(proof, air.program(), rc, mds, V.expectation(params.to_dict()))
    bad = copy.deepcopy(proof)
    bad["evals"]["zw"][1][2] ^= 1
    with pytest.raises(V.Reject):
        V.verify(bad, air.program(), rc, mds, V.expectation(params.to_dict()))


@pytest.mark.parametrize("name,logn", [("fib", 6), ("perm", 8), ("wide32", 11), ("chunk16", 9), ("chunk64", 12)])
def test_constraint_program_interpreter_gives_the_same_proof(prover, hip_backend, cpu_backend, tables, name, logn):
    """zp_eval_quotient (the AIR as a data blob, interpreted on the GPU) == the generated kernel == the checker's CPU interpreter:
    whole proofs are byte-identical whichever evaluates the constraints"""
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    rc, mds = tables
    air = AIR.get_air(name)
    tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 555 + logn)
    params = PR.StarkParams(logn, logb=1, fri_logf=3, fri_final_log=4, n_queries=5, pow_bits=6)
    be_prog = HipBackend(prover=prover, quotient="program")
    a = PR.proof_to_json(PR.prove(air, tr, pub, params, be_prog))
    assert a == PR.proof_to_json(PR.prove(air, tr, pub, params, hip_backend))
    assert a == PR.proof_to_json(PR.prove(air, tr, pub, params, cpu_backend))
    import json
    assert V.verify(json.loads(a), air.program(), rc, mds, V.expectation(params.to_dict()))


def test_eval_quotient_rejects_malformed_programs(prover):
    air = AIR.get_air("fib")
    good = air.program()
    M = 1 << 6
    d_cols, d_fixed, d_out = prover.alloc(2 * M), prover.alloc(2 * M), prover.alloc(3 * M)
    apow = [[1, 0, 0]] * len(air.constraints)
    args = lambda blob, pubs=(1, 2, 3): (blob, d_cols, d_fixed, 6, 1, list(pubs), apow, [1, 1], 49, 5, d_out)
    prover.eval_quotient(*args(good))
    for mutate in (lambda b: b.__setitem__(0, 7), lambda b: b.__setitem__(7, int(b[7]) + 1), lambda b: b.__setitem__(9, 99),
                   lambda b: b.__setitem__(12 + int(b[6]), 9), lambda b: b.__setitem__(12 + int(b[6]), int(b[12 + int(b[6])]) | (0xFFFF << 28))):
        bad = good.copy()
        mutate(bad)
        with pytest.raises(native.ZpError):
            prover.eval_quotient(*args(bad))
    with pytest.raises(native.ZpError):
        prover.eval_quotient(*args(good, pubs=(1, 2)))


def test_row_window_entry_points_match_the_whole_domain(prover):
    """zp_deep_quotient_rows / zp_eval_quotient_rows on a window (with its blow-up halo) == the same rows of the full call"""
    logm, logb, Wt = 10, 1, 8
    M, b = 1 << logm, 2
    air = AIR.get_air("wide8")
    cols = O.random_field((Wt, M), 901)
    q = O.random_field((3, M), 902)
    fixed = O.random_field((2, M), 903)
    z, zw, gamma = [O.random_field((3,), 904 + i).tolist() for i in range(3)]
    ev_z = O.random_field((Wt + 3, 3), 910)
    ev_zw = O.random_field((Wt, 3), 911)
    d_cols, d_q, d_fixed = prover.upload(cols), prover.upload(q), prover.upload(fixed)
    d_full = prover.alloc(3 * M)
    prover.deep_quotient(d_cols, Wt, d_q, 3, logm, Wt, z, zw, gamma, ev_z, ev_zw, 49, d_full)
    full = prover.download(d_full, (3, M))
    apow = [O.random_field((3,), 920 + k).tolist() for k in range(len(air.constraints))]
    pubs = [5, 6, 7, 8]
    d_qf = prover.alloc(3 * M)
    prover.eval_quotient(air.program(), d_cols, d_fixed, logm, logb, pubs, apow, [3, 9], 49, 12345, d_qf)
    qfull = prover.download(d_qf, (3, M))
    for row0, nrows in ((0, M // 4), (M // 4, M // 2), (3 * M // 4, M // 4)):
        d_a = prover.upload(cols[:, row0:row0 + nrows])
        d_b = prover.upload(q[:, row0:row0 + nrows])
        d_o = prover.alloc(3 * nrows)
        prover.deep_quotient_rows(d_a, Wt, nrows, d_b, 3, nrows, logm, row0, nrows, Wt, z, zw, gamma, ev_z, ev_zw, 49, d_o, nrows)
        assert (prover.download(d_o, (3, nrows)) == full[:, row0:row0 + nrows]).all()
        halo = np.concatenate([cols[:, row0:row0 + nrows], cols[:, [(row0 + nrows + k) % M for k in range(b)]]], axis=1)
        d_h = prover.upload(halo)
        d_f = prover.upload(fixed[:, row0:row0 + nrows])
        prover.eval_quotient_rows(air.program(), d_h, nrows + b, d_f, nrows, logm, logb, row0, nrows, pubs, apow, [3, 9], 49, 12345, d_o, nrows)
        assert (prover.download(d_o, (3, nrows)) == qfull[:, row0:row0 + nrows]).all()
    with pytest.raises(native.ZpError):      # a partial window without room for its halo
        prover.eval_quotient_rows(air.program(), d_cols, M // 2, d_fixed, M // 2, logm, logb, 0, M // 2, pubs, apow, [3, 9], 49, 12345, d_qf, M // 2)


def test_torch_side_paths_in_a_fresh_process():
    """four-step NTT through the layout kernels and the multi-GPU proof orchestration (stark/sharded.py, HIP ops, world size
    1) need torch CUDA tensors beside the library: torch has to initialise the GPU first, so they run as their own process"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_torch_checks.py")], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] and res["four_step_matches_plain_ntt"] and res["four_step_inverse_round_trip"]
    assert res["sharded_backend_world1_matches_plain_backend"]


@pytest.mark.parametrize("mode", ["kernel", "program"])
def test_degree_three_air_quotient_in_two_pieces(prover, cpu_backend, tables, mode):
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    rc, mds = tables
    air = AIR.get_air("cubic")
    tr, pub = AIR.cubic_witness(11, 9)
    params = PR.StarkParams(11, logb=1, fri_logf=3, fri_final_log=4, n_queries=6, pow_bits=6)
    gpu = PR.prove(air, tr, pub, params, HipBackend(prover=prover, quotient=mode))
    assert PR.proof_to_json(gpu) == PR.proof_to_json(PR.prove(air, tr, pub, params, cpu_backend))
    assert V.verify(gpu, air.program(), rc, mds, V.expectation(params.to_dict()))


@pytest.mark.parametrize("logn,logb", [(6, 1), (9, 2), (13, 1)])
def test_periodic_fixed_columns_gpu_matches_cpu_and_verifies(hip_backend, cpu_backend, tables, logn, logb):
    """sparse periodic fixed columns (constants with period 4, one entry per 8 rows, a public input at one row): the GPU
    interpreter reads ONE extended period per column (zp_fixed_columns), the CPU checker materialises whole columns; the
    quotients, the proofs (Python orchestration and zp_stark_prove) and the verdict of the independent verifier must agree"""
    rc, mds = tables
    air = AIR.periodic_air(logn)
    tr, pub = AIR.periodic_witness(logn, 5 + logn)
    params = PR.StarkParams(logn, logb, 2, 3, 6, pow_bits=4)
    p_cpu = PR.prove(air, tr, pub, params, cpu_backend)
    p_gpu = PR.prove(air, tr, pub, params, hip_backend)
    assert PR.proof_to_json(p_gpu) == PR.proof_to_json(p_cpu)
    assert V.verify(p_gpu, air.program(), rc, mds, V.expectation(params.to_dict()))
    d_tr = hip_backend.p.upload(tr)
    text = hip_backend.p.stark_prove(air.name, air.program(), d_tr, [int(v) for v in pub], logn, logb, 2, 3, 6, 4)
    assert text == PR.proof_to_json(p_cpu)
    bad = tr.copy()
    bad[2, 15] = (int(bad[2, 15]) + 1) % AIR.P                      # breaks S (a - b) = S c at one checkpoint row
    with pytest.raises(V.Reject):
        V.verify(PR.prove(air, bad, pub, params, hip_backend), air.program(), rc, mds, V.expectation(params.to_dict()))

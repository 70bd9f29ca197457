"""oracle/r1cs_blob.py -- the checker's own reader of a circuit blob (TEST INFRASTRUCTURE; imports nothing from the product package).

The Groth16 wrap's R1CS reaches the library as a u64 blob ("PZR1CS01" / "PZR1CS02"; the layout is documented in the product's service/r1cs.py and
service/arith.py).  This module decodes that layout on its own and checks a COMPLETE witness against every constraint -- the instances of the
Poseidon template, the explicit constraints, and (round 6) the instances of the arithmetic templates -- so that a witness completed by the
library (host or GPU) is judged by code that shares nothing with the builder, the host evaluator or the kernels.  It does not complete
witnesses: whoever claims a wire value must bring it.

    blob: [0] magic [1] n_wires [2] n_constraints [3] logm [4] t [5] n_local [6] tc [7] n_inst [8] n_extra [9] n_pub [10] n_waves [11] n_arith
          template: def[tc] | A, B, C as CSR (ptr[tc + 1], idx[nnz], val[nnz][4])            instances: n_inst x (t inputs, base, first row)
          waves[n_waves + 1]      extras: def[n_extra] | A, B, C as CSR over global wires
          per arithmetic template: [n_in, n_int, n_coef, n_lc, nnz, n_rows, n_ops, n_inst, first_row, 0, 0, 0] | coef[n_coef][4] | lc_ptr[n_lc + 1]
          | lc_ent[nnz] (coef id << 32 | local wire) | rows[n_rows][2] (a | b << 32, c) | ops[n_ops][4] | inst[n_inst][n_in + 1]
          (an LC id with bit 31 set = the unit combination of local wire id & 0x7fffffff; local wire 0 = the constant, 1..n_in inputs, then internals)
PARITY UNPINNED w.r.t. the external prover."""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MAGIC1 = int.from_bytes(b"PZR1CS01", "little")
MAGIC2 = int.from_bytes(b"PZR1CS02", "little")


class BadBlob(Exception):
    pass


def _val(d, at):
    return d[at] | (d[at + 1] << 64) | (d[at + 2] << 128) | (d[at + 3] << 192)


def _csr(d, at, rows):
    ptr = d[at:at + rows + 1]
    at += rows + 1
    nnz = ptr[rows]
    idx = d[at:at + nnz]
    at += nnz
    vals = [_val(d, at + 4 * k) for k in range(nnz)]
    at += 4 * nnz
    return (ptr, idx, vals), at


def rows_of(blob):
    """yields (row number, A, B, C) with A, B, C lists of (global wire, coefficient) for every constraint of the blob, in constraint order"""
    d = [int(v) for v in blob]
    if len(d) < 16 or d[0] not in (MAGIC1, MAGIC2):
        raise BadBlob("not a circuit blob")
    n_wires, n_cons, t, n_local, tc, n_inst, n_extra, n_waves = d[1], d[2], d[4], d[5], d[6], d[7], d[8], d[10]
    n_arith = d[11] if d[0] == MAGIC2 else 0
    at = 16 + tc                                            # (the def table: which wire a template row defines -- irrelevant to a checker)
    T = []
    for _ in range(3):
        m, at = _csr(d, at, tc)
        T.append(m)
    inst = [d[at + i * (t + 2):at + (i + 1) * (t + 2)] for i in range(n_inst)]
    at += n_inst * (t + 2) + n_waves + 1
    at += n_extra                                           # def table of the extras
    E = []
    for _ in range(3):
        m, at = _csr(d, at, n_extra)
        E.append(m)
    row = 0
    for ins in inst:
        g = lambda lw: 0 if lw == 0 else ins[lw - 1] if lw <= t else ins[t] + (lw - 1 - t)
        for q in range(tc):
            yield (row,) + tuple([(g(m[1][e]), m[2][e]) for e in range(m[0][q], m[0][q + 1])] for m in T)
            row += 1
    for q in range(n_extra):
        yield (row,) + tuple([(m[1][e], m[2][e]) for e in range(m[0][q], m[0][q + 1])] for m in E)
        row += 1
    for _ in range(n_arith):
        n_in, n_int, n_coef, n_lc, nnz, n_rows, n_ops, n_i, first_row = d[at:at + 9]
        at += 12
        if first_row != row:
            raise BadBlob("arithmetic template out of sequence")
        coef = [_val(d, at + 4 * k) for k in range(n_coef)]
        at += 4 * n_coef
        lc_ptr = d[at:at + n_lc + 1]
        at += n_lc + 1
        ent = d[at:at + nnz]
        at += nnz
        rws = d[at:at + 2 * n_rows]
        at += 2 * n_rows + 4 * n_ops                        # the witness program is the prover's business
        insts = [d[at + i * (n_in + 1):at + (i + 1) * (n_in + 1)] for i in range(n_i)]
        at += n_i * (n_in + 1)
        for ins in insts:
            g = lambda lw: 0 if lw == 0 else ins[lw - 1] if lw <= n_in else ins[n_in] + (lw - 1 - n_in)

            def lc(i):
                if i & 0x80000000:
                    return [(g(i & 0x7FFFFFFF), 1)]
                return [(g(ent[e] & 0xFFFFFFFF), coef[ent[e] >> 32]) for e in range(lc_ptr[i], lc_ptr[i + 1])]
            for q in range(n_rows):
                yield row, lc(rws[2 * q] & 0xFFFFFFFF), lc(rws[2 * q] >> 32), lc(rws[2 * q + 1])
                row += 1
    if at != len(d) or row != n_cons:
        raise BadBlob("length or constraint count does not match the header")
    del n_wires, n_local


def first_violated(blob, witness):
    """witness: list of n_wires ints (wire 0 = 1).  Index of the first constraint with (A w)(B w) != C w, or -1."""
    if witness[0] != 1 or len(witness) != int(blob[1]):
        raise BadBlob("witness does not fit the circuit")
    for row, A, B, C in rows_of(blob):
        dot = lambda M: sum(c * witness[k] for k, c in M) % R
        if dot(A) * dot(B) % R != dot(C):
            return row
    return -1

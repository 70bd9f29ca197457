"""oracle/naive.py -- pure-Python big-int definitions (TEST INFRASTRUCTURE, never product).

These are the *definitions* the C restatement (gl_oracle.c) and the HIP kernels are checked
against on small sizes: O(n^2) DFT, direct polynomial evaluation, textbook Poseidon.  They use
only Python ints, so no shared code (and no shared bug) with either implementation.

PARITY UNPINNED at the arithmetic level: the reference (eigen-zeth) holds no arithmetic for this
path; reference call sites are src/prover/provider.rs:358-503.
"""
P = 0xFFFFFFFF00000001
ROOT32_DEFAULT = 1753635133440165772   # 7^((p-1)/2^32)   (candidate 1)
ROOT32_ALT = 7277203076849721926       # SURVEY.md 8a-N1, candidate 2
SHIFT_DEFAULT = 49


def root(logn, root32=ROOT32_DEFAULT):
    return pow(root32, 1 << (32 - logn), P)


def dft(x, w):
    """X[k] = sum_n x[n] w^(nk)   -- O(n^2)"""
    n = len(x)
    pw = [pow(w, i, P) for i in range(n)]
    return [sum(x[j] * pw[(j * k) % n] for j in range(n)) % P for k in range(n)]


def ntt(x, root32=ROOT32_DEFAULT):
    n = len(x)
    logn = n.bit_length() - 1
    return dft(x, root(logn, root32))


def intt(x, root32=ROOT32_DEFAULT):
    n = len(x)
    logn = n.bit_length() - 1
    winv = pow(root(logn, root32), P - 2, P)
    ninv = pow(n, P - 2, P)
    return [(v * ninv) % P for v in dft(x, winv)]


def poly_eval(coef, x):
    acc = 0
    for c in reversed(coef):
        acc = (acc * x + c) % P
    return acc


def lde(x, logb, shift=SHIFT_DEFAULT, root32=ROOT32_DEFAULT):
    """evaluate the interpolant of x (on <w_n>) on the coset shift*<w_{bn}>, natural order"""
    n = len(x)
    coef = intt(x, root32)
    m = n << logb
    wm = root(m.bit_length() - 1, root32)
    return [poly_eval(coef, shift * pow(wm, i, P) % P) for i in range(m)]


# ---------------------------------------------------------------- Poseidon (textbook)
def poseidon_perm(state, rc, mds):
    """width 12, x^7, 4 full + 22 partial + 4 full;  ARK -> S-box -> MDS each round.
    rc: flat list of 360;  mds: flat row-major 144, out[r] = sum_j mds[r*12+j]*in[j]"""
    st = list(state)
    for r in range(30):
        st = [(st[i] + rc[r * 12 + i]) % P for i in range(12)]
        if r < 4 or r >= 26:
            st = [pow(v, 7, P) for v in st]
        else:
            st[0] = pow(st[0], 7, P)
        st = [sum(mds[i * 12 + j] * st[j] for j in range(12)) % P for i in range(12)]
    return st


def linear_hash(row, rc, mds):
    if len(row) <= 4:
        return list(row) + [0] * (4 - len(row))
    cap = [0, 0, 0, 0]
    for off in range(0, len(row), 8):
        blk = list(row[off:off + 8])
        blk += [0] * (8 - len(blk))
        st = poseidon_perm(blk + cap, rc, mds)
        cap = st[:4]
    return cap


def hash_pair(l, r, rc, mds):
    return poseidon_perm(list(l) + list(r) + [0, 0, 0, 0], rc, mds)[:4]


def merkle_root(rows, rc, mds):
    lvl = [linear_hash(r, rc, mds) for r in rows]
    while len(lvl) > 1:
        lvl = [hash_pair(lvl[2 * i], lvl[2 * i + 1], rc, mds) for i in range(len(lvl) // 2)]
    return lvl[0]


# ---------------------------------------------------------------- F_{p^3} = F_p[x]/(x^3 - x - 1)
def e3_mul(a, b):
    d = [0] * 5
    for i in range(3):
        for j in range(3):
            d[i + j] += a[i] * b[j]
    # x^3 = x + 1 ; x^4 = x^2 + x
    return [(d[0] + d[3]) % P, (d[1] + d[3] + d[4]) % P, (d[2] + d[4]) % P]


def e3_add(a, b):
    return [(a[i] + b[i]) % P for i in range(3)]


def e3_pow(a, e):
    r = [1, 0, 0]
    while e:
        if e & 1:
            r = e3_mul(r, a)
        a = e3_mul(a, a)
        e >>= 1
    return r


def e3_inv(a):
    return e3_pow(a, P ** 3 - 2)


def poly_eval_e3(coef3, x):
    """coef3: list of F_{p^3} coefficients (ascending); x base-field or ext point (as 3-list)"""
    acc = [0, 0, 0]
    for c in reversed(coef3):
        acc = e3_add(e3_mul(acc, x), c)
    return acc


def fri_fold(vals3, logf, beta, shift=SHIFT_DEFAULT, root32=ROOT32_DEFAULT):
    """vals3: list of n ext values = f on shift*<w_n> natural order.  Returns n>>logf ext values of
    sum_j beta^j g_j(y) on shift^(2^logf)*<w_{n>>logf}>, where f(x)=sum_j x^j g_j(x^(2^logf)).
    Definition-level: interpolate f (ext coefficients), split, re-evaluate."""
    n = len(vals3)
    logn = n.bit_length() - 1
    f = 1 << logf
    m = n >> logf
    # interpolate: coefficients of f(shift*X) via inverse DFT per component, then unscale
    planes = [intt([v[c] for v in vals3], root32) for c in range(3)]
    sinv = pow(shift, P - 2, P)
    coef = [[planes[c][i] * pow(sinv, i, P) % P for c in range(3)] for i in range(n)]
    # g_j coefficients: coef[j + f*t]
    bp = [1, 0, 0]
    folded = [[0, 0, 0] for _ in range(m)]
    for j in range(f):
        for t in range(m):
            folded[t] = e3_add(folded[t], e3_mul(coef[j + f * t], bp))
        bp = e3_mul(bp, beta)
    wm = root(logn - logf, root32) if logn - logf > 0 else 1
    s2 = pow(shift, f, P)
    return [poly_eval_e3(folded, [s2 * pow(wm, i, P) % P, 0, 0]) for i in range(m)]


# ---------------------------------------------------------------- stage-2 arguments, OOD evaluation, DEEP quotient
def _shift3(v, g):
    """v + g for a base value v and an F_{p^3} challenge g"""
    return [(v + g[0]) % P, g[1] % P, g[2] % P]


def grand_product(a, b, g):
    """Z[0] = 1, Z[i+1] = Z[i] * (a[i] + g) / (b[i] + g) in F_{p^3}; returns the three planes [Z.c0, Z.c1, Z.c2]"""
    z, out = [1, 0, 0], []
    for ai, bi in zip(a, b):
        out.append(z)
        z = e3_mul(z, e3_mul(_shift3(ai, g), e3_inv(_shift3(bi, g))))
    return [[v[c] for v in out] for c in range(3)]


def logup_columns(a, t, m, g):
    """LogUp lookup of the values a in the table t with multiplicities m: h1 = 1/(a+g), h2 = m/(t+g),
    S[0] = 0, S[i+1] = S[i] + h1[i] - h2[i]; returns nine planes h1.c0..c2, h2.c0..c2, S.c0..c2"""
    s, rows = [0, 0, 0], []
    for ai, ti, mi in zip(a, t, m):
        h1 = e3_inv(_shift3(ai, g))
        h2 = [v * mi % P for v in e3_inv(_shift3(ti, g))]
        rows.append(h1 + h2 + s)
        s = [(s[c] + h1[c] - h2[c]) % P for c in range(3)]
    return [[r[k] for r in rows] for k in range(9)]


def ood_eval(coef, z):
    """sum_i coef[i] z^i for base-field coefficients and an F_{p^3} point (Horner)"""
    acc = [0, 0, 0]
    for c in reversed(coef):
        acc = e3_mul(acc, z)
        acc[0] = (acc[0] + c) % P
    return acc


def deep_quotient(cols, n_next, z, zw, gamma, ev_z, ev_zw, shift=SHIFT_DEFAULT, root32=ROOT32_DEFAULT):
    """cols: W columns of M values on shift*<w_M>.  F(x) = sum_{k<W} gamma^k (p_k(x) - ev_z[k]) / (x - z)
                                                        + sum_{k<n_next} gamma^(W+k) (p_k(x) - ev_zw[k]) / (x - zw)"""
    W, M = len(cols), len(cols[0])
    wm = root(M.bit_length() - 1, root32)
    gp, cur = [], [1, 0, 0]
    for _ in range(W + n_next):
        gp.append(cur)
        cur = e3_mul(cur, gamma)
    out = []
    for r in range(M):
        x = shift * pow(wm, r, P) % P
        i1 = e3_inv([(x - z[0]) % P, -z[1] % P, -z[2] % P])
        i2 = e3_inv([(x - zw[0]) % P, -zw[1] % P, -zw[2] % P])
        acc = [0, 0, 0]
        for k in range(W):
            num = [(cols[k][r] - ev_z[k][0]) % P, -ev_z[k][1] % P, -ev_z[k][2] % P]
            acc = e3_add(acc, e3_mul(gp[k], e3_mul(num, i1)))
        for k in range(n_next):
            num = [(cols[k][r] - ev_zw[k][0]) % P, -ev_zw[k][1] % P, -ev_zw[k][2] % P]
            acc = e3_add(acc, e3_mul(gp[W + k], e3_mul(num, i2)))
        out.append(acc)
    return [[v[c] for v in out] for c in range(3)]


# ---------------------------------------------------------------- BN128-hash mode: Poseidon over the BN254 scalar field
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def poseidon_bn254_perm(state, rc, mds, rp):
    """textbook schedule: ARK -> x^5 (all elements in the 4 + 4 full rounds, element 0 in the rp partial rounds) -> MDS"""
    t = len(state)
    st = [v % BN254_R for v in state]
    for r in range(8 + rp):
        st = [(st[i] + rc[r * t + i]) % BN254_R for i in range(t)]
        if r < 4 or r >= 4 + rp:
            st = [pow(v, 5, BN254_R) for v in st]
        else:
            st[0] = pow(st[0], 5, BN254_R)
        st = [sum(mds[i][j] * st[j] for j in range(t)) % BN254_R for i in range(t)]
    return st


def poseidon_bn254_hash(inputs, rc, mds, rp):
    """the published convention: state = [0, inputs...], digest = state[0] after one permutation (len(inputs) = t - 1)"""
    return poseidon_bn254_perm([0] + list(inputs), rc, mds, rp)[0]


def pack3(vals):
    """three Goldilocks values per BN254 element: a + b 2^64 + c 2^128 (missing values = 0)"""
    vals = list(vals) + [0] * (-len(vals) % 3)
    return [vals[i] + (vals[i + 1] << 64) + (vals[i + 2] << 128) for i in range(0, len(vals), 3)]


LEAF_BLOCK = 56      # Goldilocks values one permutation of a leaf sponge absorbs


def pack_leaf_block(vals):
    """one sponge block of a 16-ary tree's leaf: up to 56 Goldilocks values in 16 field elements -- element e holds values 3e, 3e+1,
    3e+2 in bits 0..191 (as pack3) and, in bits 192..223, 32-bit half number e of values 48..55 (half 2i = low word of value 48+i,
    half 2i+1 = its high word).  Below 2^224 < r; rows of at most 48 values pack exactly as pack3."""
    vals = list(vals) + [0] * (LEAF_BLOCK - len(vals))
    out = []
    for e in range(16):
        x = vals[48 + (e >> 1)]
        half = (x >> 32) if (e & 1) else (x & 0xFFFFFFFF)
        out.append(vals[3 * e] + (vals[3 * e + 1] << 64) + (vals[3 * e + 2] << 128) + (half << 192))
    return out


def merkle16_leaf(row, rc, mds, rp):
    """sponge over the row in blocks of 56 values (16 elements per permutation), the digest being the capacity of the next block"""
    row = list(row)
    cap = 0
    for off in range(0, max(len(row), 1), LEAF_BLOCK):
        cap = poseidon_bn254_perm([cap] + pack_leaf_block(row[off:off + LEAF_BLOCK]), rc, mds, rp)[0]
    return cap


def merkle16_tree(rows, rc, mds, rp):
    """all levels, leaves first; a node hashes its (up to) 16 children, missing ones as 0"""
    levels = [[merkle16_leaf(r, rc, mds, rp) for r in rows]]
    while len(levels[-1]) > 1:
        prev = levels[-1]
        levels.append([poseidon_bn254_perm([0] + prev[i:i + 16] + [0] * (16 - len(prev[i:i + 16])), rc, mds, rp)[0]
                       for i in range(0, len(prev), 16)])
    return levels


# ---- F_r transforms of the Groth16 QAP step (definition level: O(n^2) sums, schoolbook products) ----
FR = BN254_R
def fr_root(logn):
    """5^((r-1)/2^logn): 5 generates F_r^* (r - 1 = 2^28 * odd); the convention of the snarkjs / circom family"""
    assert 0 <= logn <= 28
    return pow(5, (FR - 1) >> logn, FR)


def fr_ntt(x, inverse=False, coset=1):
    """forward: y_k = sum_i x_i (g w^k)^i ;  inverse: x_i = g^-i / n * sum_k y_k w^(-i k)"""
    n = len(x)
    logn = n.bit_length() - 1
    assert 1 << logn == n
    w = fr_root(logn)
    if not inverse:
        return [poly_eval_mod(x, coset * pow(w, k, FR) % FR, FR) for k in range(n)]
    winv, ninv, ginv = pow(w, FR - 2, FR), pow(n, FR - 2, FR), pow(coset, FR - 2, FR)
    return [sum(x[k] * pow(winv, i * k, FR) for k in range(n)) % FR * ninv % FR * pow(ginv, i, FR) % FR for i in range(n)]


def poly_eval_mod(coef, z, mod):
    acc = 0
    for c in reversed(coef):
        acc = (acc * z + c) % mod
    return acc


def qap_quotient(a_ev, b_ev, c_ev):
    """evaluations of A, B, C on <w> -> coefficients of H with A B - C = H (x^m - 1), by interpolation, a schoolbook
    product and the division identity  P_(j+m) = H_j,  P_j = -H_j  (j < m)"""
    m = len(a_ev)
    A, B, Cc = fr_ntt(a_ev, inverse=True), fr_ntt(b_ev, inverse=True), fr_ntt(c_ev, inverse=True)
    prod = [0] * (2 * m)
    for i, ai in enumerate(A):
        if ai:
            for j, bj in enumerate(B):
                prod[i + j] = (prod[i + j] + ai * bj) % FR
    for i, ci in enumerate(Cc):
        prod[i] = (prod[i] - ci) % FR
    H = prod[m:]
    assert all((prod[j] + H[j]) % FR == 0 for j in range(m)), "A B - C is not divisible by x^m - 1"
    return H


def fr_ntt_fast(x, inverse=False, coset=1):
    """the same map as fr_ntt by recursive even/odd splitting (for sizes the O(n^2) sums cannot reach); pinned to fr_ntt
    on small sizes by tests/test_oracle.py"""
    n = len(x)
    logn = n.bit_length() - 1
    assert 1 << logn == n
    w = fr_root(logn)

    def rec(a, ww):
        if len(a) == 1:
            return a
        ev, od = rec(a[0::2], ww * ww % FR), rec(a[1::2], ww * ww % FR)
        h, t, out = len(a) // 2, 1, [0] * len(a)
        for i in range(h):
            v = t * od[i] % FR
            out[i], out[i + h] = (ev[i] + v) % FR, (ev[i] - v) % FR
            t = t * ww % FR
        return out

    if not inverse:
        gi, xs = 1, []
        for v in x:
            xs.append(v * gi % FR)
            gi = gi * coset % FR
        return rec(xs, w)
    y = rec(list(x), pow(w, FR - 2, FR))
    ninv, ginv, gi, out = pow(n, FR - 2, FR), pow(coset, FR - 2, FR), 1, []
    for v in y:
        out.append(v * ninv % FR * gi % FR)
        gi = gi * ginv % FR
    return out
